"""--dcfilter by the exact parallel-in-time kernels (msd_dc_kernels.hip, round 6) against the oracle's in-order
convert_*_generic (convert.c:113-213, 374-423): magnitudes and both means bit for bit, the filter state carried from
call to call, for contents that put the filter state in each of its regimes (a state near zero that changes sign and
binade all the time; a DC offset; constant and alternating input, where the rounding of z * dc_b is systematic), and the
two other ways through the same entry (the in-order kernel alone; one parallel pass, so that the in-order kernel behind
it has to do the batch) as cross-checks of the switch."""
import zlib

import numpy as np
import pytest

from helpers import fmt_ids

CHUNK = 131072
pytestmark = pytest.mark.gpu


def content(kind, fmt, n, seed):
    rng = np.random.default_rng(seed)
    full = {"uc8": 127.5, "sc16": 32768.0, "sc16q11": 2048.0}[fmt]
    if kind == "noise":          # no offset: z hovers around zero
        v = rng.standard_normal((n, 2)) * 0.05
    elif kind == "offset":       # an RTL dongle's half LSB and more
        v = rng.standard_normal((n, 2)) * 0.05 + np.array([0.004, -0.02])
    elif kind == "strong":
        v = rng.standard_normal((n, 2)) * 0.4 + np.array([-0.1, 0.3])
    elif kind == "constant":
        v = np.tile(np.array([0.37, -1.0]), (n, 1))
    elif kind == "alternating":
        v = np.where((np.arange(n) & 1)[:, None] == 1, 1.0, -1.0) * np.array([1.0, 0.5])
    elif kind == "random":
        v = rng.uniform(-1, 1, (n, 2))
    elif kind == "step":         # the offset jumps in the middle of the stream
        v = rng.standard_normal((n, 2)) * 0.05 + np.where(np.arange(n)[:, None] < n // 2, 0.01, -0.3)
    else:
        raise ValueError(kind)
    if fmt == "uc8":
        return np.clip(np.rint(127.5 + 127.5 * v), 0, 255).astype(np.uint8).reshape(-1)
    return np.clip(np.rint(v * full), -full, full - 1).astype("<i2").reshape(-1).view(np.uint8)


@pytest.mark.parametrize("fmt", ["uc8", "sc16", "sc16q11"])
@pytest.mark.parametrize("kind", ["noise", "offset", "strong", "constant", "alternating", "random", "step"])
def test_parallel_dc_block_equals_the_in_order_converter(pkg, oracle, torch_cuda, fmt, kind):
    f, of = fmt_ids(pkg, oracle, fmt)
    bps = 2 if fmt == "uc8" else 4
    sizes = (CHUNK, 1, 4097, 8 * CHUNK + 777, 63, 64, 65, 3 * CHUNK, 1024, 2 * CHUNK - 1)
    n = sum(sizes)
    iq = content(kind, fmt, n, seed=zlib.crc32((fmt + kind).encode()) & 0xffff)
    dem = pkg.Demodulator(fmt=f, nfix_crc=1, max_batch_samples=16 * CHUNK, dc_filter=True, flags=0)
    orc = oracle.Oracle(of, 58, 1, 0, dc_filter=True)
    off = 0
    for m in sizes:
        blk = iq[off * bps:(off + m) * bps]
        gm, gl, gp = dem.convert(blk, m)
        wm, wl, wp = orc.convert(blk, m)
        exact, passes, guessed, blocks = dem.dc_filter_status()
        assert exact == 1, (fmt, kind, off, m, passes, guessed, blocks)   # the parallel kernels did it, nothing fell through
        assert np.array_equal(gm[:m], wm), (fmt, kind, off, m, int(np.flatnonzero(gm[:m] != wm)[0]))
        assert np.array_equal(np.float64(gl), np.float64(wl), equal_nan=True), (fmt, kind, off, m)
        assert np.array_equal(np.float64(gp), np.float64(wp), equal_nan=True), (fmt, kind, off, m)
        off += m


@pytest.mark.parametrize("fmt", ["uc8", "sc16"])
@pytest.mark.parametrize("way", ["sequential", "one_pass"])
def test_the_in_order_kernel_behind_the_passes(pkg, oracle, torch_cuda, fmt, way):
    """MSD_CFG_DC_SEQUENTIAL: the in-order kernel alone.  MSD_CFG_DC_ONE_PASS: one parallel pass is queued, the batch is not
    exact by then (status 0), the in-order kernel behind the passes does it from the untouched converter state -- the path a
    batch takes whose passes ever run out."""
    f, of = fmt_ids(pkg, oracle, fmt)
    bps = 2 if fmt == "uc8" else 4
    sizes = (2 * CHUNK, 4097, CHUNK)
    iq = content("offset", fmt, sum(sizes), seed=77)
    flag = pkg.capi.CFG_DC_SEQUENTIAL if way == "sequential" else pkg.capi.CFG_DC_ONE_PASS
    dem = pkg.Demodulator(fmt=f, nfix_crc=1, max_batch_samples=2 * CHUNK, dc_filter=True, flags=flag)
    orc = oracle.Oracle(of, 58, 1, 0, dc_filter=True)
    off = 0
    for m in sizes:
        blk = iq[off * bps:(off + m) * bps]
        gm, gl, gp = dem.convert(blk, m)
        wm, wl, wp = orc.convert(blk, m)
        assert dem.dc_filter_status()[0] == 0
        assert np.array_equal(gm[:m], wm) and gl == wl and gp == wp, (fmt, way, off, m)
        off += m


@pytest.mark.parametrize("fmt", ["uc8", "sc16"])
@pytest.mark.parametrize("kind", ["noise", "offset", "constant"])
def test_the_passes_in_one_cooperative_launch(pkg, oracle, torch_cuda, fmt, kind):
    """MSD_CFG_DC_FUSED_LAUNCH: all passes in one cooperative launch, the evaluation following the walk block by block through
    bounded waits (msd_dcp_fused_kernel; measured slower than two launches per pass, kept as a switch) -- the same magnitudes,
    the same means, exact as well."""
    f, of = fmt_ids(pkg, oracle, fmt)
    bps = 2 if fmt == "uc8" else 4
    sizes = (CHUNK, 4097, 8 * CHUNK + 777, 3 * CHUNK)
    iq = content(kind, fmt, sum(sizes), seed=zlib.crc32((kind + fmt).encode()) & 0xffff)
    dem = pkg.Demodulator(fmt=f, nfix_crc=1, max_batch_samples=16 * CHUNK, dc_filter=True, flags=pkg.capi.CFG_DC_FUSED_LAUNCH)
    orc = oracle.Oracle(of, 58, 1, 0, dc_filter=True)
    off = 0
    for m in sizes:
        blk = iq[off * bps:(off + m) * bps]
        gm, gl, gp = dem.convert(blk, m)
        wm, wl, wp = orc.convert(blk, m)
        assert dem.dc_filter_status()[0] == 1, (fmt, kind, off, m, dem.dc_filter_status())
        assert np.array_equal(gm[:m], wm) and gl == wl and gp == wp, (fmt, kind, off, m)
        off += m


def test_two_receivers_with_the_dc_block_on_one_gpu(pkg, oracle, torch_cuda):
    """Two contexts converting alternately, one of them with the cooperative launch: whatever way a batch goes
    (msd_dc_filter_status may say 0 when a bounded wait ran out), the magnitudes are the oracle's."""
    n = 4 * CHUNK
    a = content("offset", "uc8", 3 * n, seed=11)
    b = content("noise", "uc8", 3 * n, seed=12)
    da = pkg.Demodulator(max_batch_samples=4 * CHUNK, dc_filter=True, flags=0)
    db = pkg.Demodulator(max_batch_samples=4 * CHUNK, dc_filter=True, flags=pkg.capi.CFG_DC_FUSED_LAUNCH)
    oa = oracle.Oracle(oracle.FMT_UC8, 58, 1, 0, dc_filter=True)
    ob = oracle.Oracle(oracle.FMT_UC8, 58, 1, 0, dc_filter=True)
    for k in range(3):
        for dem, orc, iq in ((da, oa, a), (db, ob, b)):
            blk = iq[2 * k * n:2 * (k + 1) * n]
            gm = dem.convert(blk, n)[0]
            assert np.array_equal(gm[:n], orc.convert(blk, n)[0]), k


def test_parallel_dc_block_in_the_stream_interface(pkg, oracle, torch_cuda):
    """The same kernels in front of the scan (msd_launch_device of a MSD_CFG_DC_FILTER context): three batches of a capture
    with a DC offset, pipelined; the message list and the per-buffer means are the oracle's, every batch came out exact."""
    n = 20 * CHUNK + 4321
    iq = pkg.siggen.generate(pkg.siggen.make_cfg(seed=909, msgs_per_sec=5000, n_aircraft=40), n)
    iq = np.clip(iq.reshape(-1, 2).astype(np.int32) + np.array([7, -4]), 0, 255).astype(np.uint8).reshape(-1)
    d_iq = torch_cuda.from_numpy(iq).to("cuda:0")
    dem = pkg.Demodulator(nfix_crc=1, max_batch_samples=8 * CHUNK, message_capacity=1 << 17, dc_filter=True)
    got, off = [], 0
    for m, last in ((8 * CHUNK, False), (8 * CHUNK, False), (n - 16 * CHUNK, True)):
        dem.launch_device(d_iq.data_ptr() + 2 * off, m, last=last)
        assert dem.dc_filter_status()[0] == 1
        off += m
    for _ in range(3):
        got.append(dem.collect())
    got = np.concatenate(got)
    want, wstats = oracle.Oracle(oracle.FMT_UC8, 58, 1, 0, dc_filter=True).replay(iq, cap=1 << 17)
    assert len(want) > 1000
    assert got.tobytes() == want.tobytes()


@pytest.mark.parametrize("fmt", ["uc8", "sc16"])
@pytest.mark.parametrize("kind", ["noise", "offset", "random"])
def test_every_block_table_is_monotone(pkg, torch_cuda, fmt, kind):
    """What the walk's second exactness rule rests on, checked on the hardware: the map of a block is monotone non-decreasing
    (every step z -> fl(t + fl(z b)) is), so the table values ascend with the candidates -- every block's table after two passes,
    read back from the workspace of msd_launch_dcfilter_parallel (white box: the layout of msd_dc_kernels.hip's dcp_launch).  Also
    the brackets tile the line (a lane's upper end is the next lane's candidate) and a flat mark sits on equal table values only."""
    import ctypes as C
    L = C.CDLL(pkg.capi.LIB_PATH)
    L.msd_dcp_work_bytes.restype = C.c_size_t
    L.msd_dcp_work_bytes.argtypes = [C.c_uint64, C.c_uint32]
    L.msd_launch_dcfilter_parallel.restype = C.c_int
    L.msd_launch_dcfilter_parallel.argtypes = [C.c_int, C.c_void_p, C.c_uint64, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p,
                                               C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_void_p]
    f = pkg.FMT_UC8 if fmt == "uc8" else pkg.FMT_SC16
    bps = 2 if fmt == "uc8" else 4
    n, blk = 1 << 19, 1024
    nb = n // blk
    iq = content(kind, fmt, n, seed=21)
    d_iq = torch_cuda.from_numpy(iq.copy()).cuda()
    work = torch_cuda.zeros(L.msd_dcp_work_bytes(n, blk), dtype=torch_cuda.uint8, device="cuda")
    mag = torch_cuda.zeros(n, dtype=torch_cuda.int16, device="cuda")
    sq = torch_cuda.zeros(n, dtype=torch_cuda.float32, device="cuda")
    state = torch_cuda.tensor([0.003, -0.0007], dtype=torch_cuda.float32, device="cuda")
    b = np.float32(np.exp(-2 * np.pi / 2.4e6))
    a = np.float32(1.0 - float(b))
    assert len(iq) == n * bps
    assert L.msd_launch_dcfilter_parallel(f, d_iq.data_ptr(), n, float(a), float(b), state.data_ptr(), mag.data_ptr(), sq.data_ptr(),
                                          work.data_ptr(), blk, 2, 0, None) == 0
    torch_cuda.cuda.synchronize()
    e0 = 256 + ((nb * 24 + 255) & ~255)
    T = work[e0:e0 + 2 * nb * 64 * 16].cpu().numpy().view(np.float32).reshape(2 * nb, 64, 4)
    x0, x_up, y0, w = T[:, :, 0], T[:, :, 1], T[:, :, 2], T[:, :, 3]
    assert not np.isnan(T[:, :63, :]).any()
    assert (np.diff(x0, axis=1) >= 0).all()                       # the candidates ascend with the lanes
    assert (np.diff(y0, axis=1) >= 0).all()                       # ... and so do the block's values at them: F is monotone
    assert np.array_equal(x_up[:, :62], x0[:, 1:63])              # lane j's bracket ends where lane j + 1's begins
    flat = w[:, :63].view(np.uint32) == 0x80000000
    assert flat.any() and (y0[:, :63][flat] == y0[:, 1:][flat]).all()
    assert (w[:, :63][~flat] >= 0).all()                          # a secant of a monotone map is not negative

