"""The two literal drop-in entries of the C-ABI, directly: the iq_convert_fn-shaped converter
(msd_convert, convert.h:33-38) against the oracle's converters on exhaustive / lattice inputs, and the
demodulate2400-shaped entry (msd_demodulate_magbuf, demod_2400.h:37-38) fed buffer by buffer from
Python for the float formats and Mode A/C."""
import ctypes as C

import numpy as np
import pytest

from helpers import assert_same_messages, fmt_ids

pytestmark = pytest.mark.gpu

CHUNK = 131072


def convert_both(pkg, oracle, fmt, iq_bytes, nsamples):
    f, of = fmt_ids(pkg, oracle, fmt)
    dem = pkg.Demodulator(fmt=f, nfix_crc=1, max_batch_samples=CHUNK)
    orc = oracle.Oracle(of, 58, 1, 0)
    out_g, out_o = [], []
    for off in range(0, max(nsamples, 1), CHUNK):          # one block of MODES_MAG_BUF_SAMPLES per call, sdr_ifile.c:214
        n = min(CHUNK, nsamples - off)
        bps = 2 if fmt == "uc8" else 4
        part = iq_bytes[off * bps:(off + n) * bps]
        out_g.append(dem.convert(part if n else np.zeros(16, np.uint8), n))
        out_o.append(orc.convert(part if n else np.zeros(16, np.uint8), n))
    return out_g, out_o


def assert_blocks_equal(out_g, out_o):
    for (mg, lg, pg), (mo, lo, po) in zip(out_g, out_o):
        assert np.array_equal(mg, mo)
        assert np.array_equal(np.float64(lg), np.float64(lo), equal_nan=True), (lg, lo)
        assert np.array_equal(np.float64(pg), np.float64(po), equal_nan=True), (pg, po)


def test_uc8_converter_on_all_65536_byte_pairs(pkg, oracle, torch_cuda):
    """Every (I, Q) byte pair once (convert.c:35-61's whole table), in an order that spreads the pairs over
    the blocks, plus a second pass in natural order: magnitudes and both means bit for bit."""
    pairs = np.arange(65536, dtype=np.uint32)
    shuffled = np.random.default_rng(1).permutation(pairs)
    both = np.concatenate([shuffled, pairs])
    iq = np.empty(2 * both.size, dtype=np.uint8)
    iq[0::2] = both & 0xFF          # I
    iq[1::2] = both >> 8            # Q
    out_g, out_o = convert_both(pkg, oracle, "uc8", iq, both.size)
    assert_blocks_equal(out_g, out_o)
    mags = np.concatenate([m for m, _, _ in out_g])[65536:]
    assert np.array_equal(mags, oracle.uc8_table())   # the reference's table itself, slot = I + 256 Q


def test_scan_kernel_on_all_65536_byte_pairs(pkg, oracle, torch_cuda):
    """The scan kernel reads its own copy of the UC8 table (256-entry pitch, swizzled columns, msd_internal.h) through
    an index of its own: every (I, Q) byte pair through msd_scan_kernel -- eight buffers whose halves are differently
    shuffled permutations of all 65536 pairs and the natural order -- and the buffers' exact level / power sums, the
    counters and whatever messages the permutations happen to hold against the oracle.  One wrong table slot moves the
    sums of the eight buffers that hold its pair."""
    pairs = np.arange(65536, dtype=np.uint32)
    parts = [np.random.default_rng(k).permutation(pairs) for k in range(15)] + [pairs]
    both = np.concatenate(parts)                       # 16 x 65536 samples = 8 buffers
    iq = np.empty(2 * both.size, dtype=np.uint8)
    iq[0::2] = both & 0xFF
    iq[1::2] = both >> 8
    n = both.size
    from test_gpu_parity import assert_same
    d = torch_cuda.from_numpy(iq).to("cuda:0")
    dem = pkg.Demodulator(nfix_crc=1, max_batch_samples=8 * 131072, message_capacity=1 << 14)
    got = dem.submit_device(d.data_ptr(), n, last=True)
    want, wstats, wmeans = oracle.Oracle(oracle.FMT_UC8, 58, 1, 0).replay(iq, cap=1 << 14, want_means=True)
    assert_same(got, dem.stats(), want, wstats)
    gm = dem.buffer_means()
    assert len(gm) == wstats["buffers"] == 9 and np.array_equal(gm, wmeans[: len(gm)], equal_nan=True)   # (8 full buffers + the empty one an exact multiple ends with, sdr_ifile.c:199-216)


@pytest.mark.parametrize("fmt", ["sc16", "sc16q11"])
def test_s16_converters_on_a_lattice_of_2_24_pairs(pkg, oracle, torch_cuda, fmt):
    """convert.c:215-253 / :332-370 on 4096 x 4096 (I, Q) pairs: every value the 12-bit ADCs behind SC16Q11
    produce (-2048..2047) and, for SC16, a lattice over the whole int16 range that holds both ends, zero,
    the values around the clamp at magsq > 1 and odd steps in between."""
    if fmt == "sc16q11":
        axis = np.arange(-2048, 2048, dtype=np.int32)
    else:
        axis = np.arange(-32768, 32768, 16, dtype=np.int32)       # 4096 values, holds -32768 and 0
        special = [-32767, -23171, -23170, -1, 1, 15, 23170, 23171, 32766, 32767]
        axis[np.arange(len(special)) * 97 + 33] = special
    i_vals, q_vals = np.meshgrid(axis, axis, indexing="ij")
    order = np.random.default_rng(2).permutation(i_vals.size)   # sequential float sums see a mixed stream
    iq16 = np.empty(2 * i_vals.size, dtype=np.int16)
    iq16[0::2] = i_vals.reshape(-1)[order]
    iq16[1::2] = q_vals.reshape(-1)[order]
    out_g, out_o = convert_both(pkg, oracle, fmt, iq16.view(np.uint8), i_vals.size)
    assert_blocks_equal(out_g, out_o)
    mags = np.concatenate([m for m, _, _ in out_g])
    assert mags.max() == 65535 and mags.min() == 0


@pytest.mark.parametrize("fmt", ["uc8", "sc16", "sc16q11"])
@pytest.mark.parametrize("n", [0, 1, 7, 8, 9, 1000, 131071])
def test_converter_ragged_lengths_and_null_out_pointers(pkg, oracle, torch_cuda, fmt, n):
    f, of = fmt_ids(pkg, oracle, fmt)
    bps = 2 if fmt == "uc8" else 4
    rng = np.random.default_rng(n + 5)
    iq = rng.integers(0, 256, size=max(n, 4) * bps, dtype=np.uint8)
    out_g, out_o = convert_both(pkg, oracle, fmt, iq, n)
    assert_blocks_equal(out_g, out_o)          # n = 0: both means are 0/0 = NaN (convert.c:105-109)
    # either out pointer may be NULL (convert.c:104-110)
    dem = pkg.Demodulator(fmt=f, nfix_crc=1, max_batch_samples=CHUNK)
    mag = np.zeros(max(n, 1), dtype=np.uint16)
    rc = pkg.capi.lib().msd_convert(dem._h, iq.ctypes.data, mag.ctypes.data, n, None, None)
    assert rc == 0 and np.array_equal(mag[:n], out_o[0][0])
    lvl = C.c_double()
    assert pkg.capi.lib().msd_convert(dem._h, iq.ctypes.data, mag.ctypes.data, n, C.byref(lvl), None) == 0
    assert np.array_equal(np.float64(lvl.value), np.float64(out_o[0][1]), equal_nan=True)


def magbuf_feed(pkg, oracle, fmt, iq, n, nfix, mode_ac):
    """What the reference's reader thread + consumer loop do with the two entries (sdr_ifile.c:187-216,
    fifo.c:179-188, readsb.c:826-833): convert a block, put the previous tail in front, demodulate."""
    f, of = fmt_ids(pkg, oracle, fmt)
    bps = 2 if fmt == "uc8" else 4
    conv = pkg.Demodulator(fmt=f, nfix_crc=nfix, max_batch_samples=CHUNK)
    dem = pkg.Demodulator(fmt=f, nfix_crc=nfix, mode_ac=mode_ac, max_batch_samples=CHUNK, message_capacity=1 << 15)
    overlap = pkg.capi.OVERLAP
    carry = np.zeros(overlap, dtype=np.uint16)
    out, counter = [], 0
    nblocks = n // CHUNK + 1                     # a capture ends with a short (possibly empty) block
    for b in range(nblocks):
        m = min(CHUNK, n - b * CHUNK)
        part = iq[b * CHUNK * bps:(b * CHUNK + m) * bps]
        mag, level, power = conv.convert(part if m else np.zeros(16, np.uint8), m)
        data = np.concatenate([carry, mag])
        ts = counter * 5                          # sampleCounter * 12e6 / 2.4e6
        out.append(dem.demodulate_magbuf(data, overlap + m, overlap, ts, ts // 12000, level, power))
        carry = data[-overlap:] if data.size >= overlap else carry
        counter += m
    return np.concatenate(out), dem.stats()


@pytest.mark.parametrize("fmt,mode_ac,nfix", [("sc16q11", 1, 1), ("sc16", 0, 0), ("uc8", 1, 1)])
def test_magbuf_entry_from_python(pkg, oracle, torch_cuda, fmt, mode_ac, nfix):
    f, of = fmt_ids(pkg, oracle, fmt)
    n = 5 * CHUNK + 4321
    cfg = pkg.siggen.make_cfg(seed=77, fmt=f, msgs_per_sec=3000, n_aircraft=40, ac_per_sec=1500 if mode_ac else 0)
    iq = pkg.siggen.generate(cfg, n)
    got, gstats = magbuf_feed(pkg, oracle, fmt, iq, n, nfix, mode_ac)
    want, wstats = oracle.Oracle(of, 58, nfix, mode_ac).replay(iq, cap=1 << 16)
    assert len(want) > 100 and (not mode_ac or (want["msgtype"] == 32).sum() > 10)
    assert_same_messages(got, want)
    for k in ("demod_preambles", "demod_rejected_bad", "demod_rejected_unknown_icao", "demod_accepted", "demod_modeac",
              "demod_preamblePhase", "demod_bestPhase"):
        assert gstats[k] == wstats[k], (k, gstats[k], wstats[k])


@pytest.mark.parametrize("fmt,mode_ac,nfix,group", [("uc8", 1, 1, 3), ("sc16", 0, 1, 8), ("uc8", 0, 0, 5)])
def test_several_magbufs_in_one_call(pkg, oracle, torch_cuda, fmt, mode_ac, nfix, group):
    """msd_demodulate_magbufs: what a consumer that finds several buffers queued hands over at once (the ifile handler's
    mag_buf mode does, readsb's own loop takes them one by one): groups of `group` consecutive buffers -- the last group
    short, the last buffer ragged -- with the caller's clocks and means per buffer, against the oracle and against the
    one-buffer-per-call feed; the signal levels come from the caller's host magnitudes, across buffer boundaries too."""
    f, of = fmt_ids(pkg, oracle, fmt)
    bps = 2 if fmt == "uc8" else 4
    n = 11 * CHUNK + 2345
    cfg = pkg.siggen.make_cfg(seed=91, fmt=f, msgs_per_sec=4000, n_aircraft=40, ac_per_sec=1500 if mode_ac else 0)
    iq = pkg.siggen.generate(cfg, n)
    conv = pkg.Demodulator(fmt=f, nfix_crc=nfix, max_batch_samples=CHUNK)
    dem = pkg.Demodulator(fmt=f, nfix_crc=nfix, mode_ac=mode_ac, max_batch_samples=group * CHUNK, message_capacity=1 << 16)
    overlap = pkg.capi.OVERLAP
    carry = np.zeros(overlap, dtype=np.uint16)
    bufs, counter = [], 0
    for b in range(n // CHUNK + 1):
        m = min(CHUNK, n - b * CHUNK)
        mag, level, power = conv.convert(iq[b * CHUNK * bps:(b * CHUNK + m) * bps] if m else np.zeros(16, np.uint8), m)
        data = np.concatenate([carry, mag])
        bufs.append((data, overlap + m, overlap, counter * 5, counter * 5 // 12000, level, power))
        carry = data[-overlap:]
        counter += m
    got = np.concatenate([pkg.capi.demodulate_magbufs(dem, bufs[i:i + group]) for i in range(0, len(bufs), group)])
    want, wstats = oracle.Oracle(of, 58, nfix, mode_ac).replay(iq, cap=1 << 16)
    assert len(want) > 300
    assert_same_messages(got, want)
    gstats = dem.stats()
    for k in ("demod_preambles", "demod_rejected_bad", "demod_rejected_unknown_icao", "demod_accepted", "demod_modeac",
              "demod_preamblePhase", "demod_bestPhase"):
        assert gstats[k] == wstats[k], (k, gstats[k], wstats[k])
    one_by_one, _ = magbuf_feed(pkg, oracle, fmt, iq, n, nfix, mode_ac)
    assert np.array_equal(got, one_by_one)
    with pytest.raises(pkg.MsdError):           # more buffers than the context was made for
        pkg.capi.demodulate_magbufs(dem, bufs[:group + 1])


@pytest.mark.parametrize("fmt", ["sc16", "sc16q11"])
def test_sequential_float_sums_on_structured_inputs(pkg, oracle, torch_cuda, fmt):
    """convert.c:228-252's sums are float and sequential; the kernel evaluates them block-parallel with
    predicted binades and per-block functions.  Inputs chosen to sit on what that machinery has to get
    right: constant blocks (every element the same, rounding ties in long runs), values that are exact
    halves of the sum's unit, blocks of zeros between bursts, full scale (the sum races through the
    binades), a ramp, and lengths that end inside a block."""
    full = 32767 if fmt == "sc16" else 2047
    rng = np.random.default_rng(11)
    n = 131072
    cases = {
        "constant_small": np.full((n, 2), 3, dtype=np.int16),
        "constant_mid": np.full((n, 2), full // 16, dtype=np.int16),
        "full_scale": np.full((n, 2), full, dtype=np.int16),
        "powers_of_two": np.stack([np.tile(np.array([1, 2, 4, 8, 16, 32, 64, 128], dtype=np.int16) * (full // 2048 + 1), n // 8),
                                   np.zeros(n, dtype=np.int16)], axis=1),
        "bursts": np.where((np.arange(n) // 3000 % 3 == 0)[:, None], rng.integers(-full, full, size=(n, 2)), 0).astype(np.int16),
        "ramp": np.stack([(np.arange(n) % (2 * full) - full).astype(np.int16), np.zeros(n, dtype=np.int16)], axis=1),
        "noise": rng.normal(0, full * 0.02, size=(n, 2)).round().astype(np.int16),
    }
    for name, iq16 in cases.items():
        for length in (n, n - 1, 70000 + 513, 1025, 1023, 17):
            iq = np.ascontiguousarray(iq16[:length]).view(np.uint8).reshape(-1)
            out_g, out_o = convert_both(pkg, oracle, fmt, iq, length)
            try:
                assert_blocks_equal(out_g, out_o)
            except AssertionError as e:
                raise AssertionError(f"{name} length {length}: {e}")


@pytest.mark.parametrize("fmt", ["sc16", "sc16q11"])
def test_stream_float_sums_on_structured_buffers(pkg, oracle, torch_cuda, fmt):
    """The same structured inputs through the stream path, one pattern per 131072-sample buffer: there the
    float-sum kernel predicts the binades of its sums from the per-tile sums the scan kernel leaves (16-bit
    magnitudes, not the floats themselves), so what is tested is that a prediction which is off -- tiny sums,
    block totals on a power of two, bursts after silence -- only ever costs time.  The per-buffer means reach
    the statistics (noise power) that the comparison with the oracle covers."""
    from test_gpu_parity import assert_same
    full = 32767 if fmt == "sc16" else 2047
    rng = np.random.default_rng(12)
    n = 131072
    bufs = [
        np.full((n, 2), 3, dtype=np.int16),
        np.full((n, 2), full // 16, dtype=np.int16),
        np.full((n, 2), full, dtype=np.int16),
        np.stack([np.tile(np.array([1, 2, 4, 8, 16, 32, 64, 128], dtype=np.int16) * (full // 2048 + 1), n // 8),
                  np.zeros(n, dtype=np.int16)], axis=1),
        np.where((np.arange(n) // 3000 % 3 == 0)[:, None], rng.integers(-full, full, size=(n, 2)), 0).astype(np.int16),
        np.stack([(np.arange(n) % (2 * full) - full).astype(np.int16), np.zeros(n, dtype=np.int16)], axis=1),
        rng.normal(0, full * 0.02, size=(n, 2)).round().astype(np.int16),
        np.zeros((n, 2), dtype=np.int16),
        np.where((np.arange(n) % 1024 == 1023)[:, None], np.full((n, 2), full), 0).astype(np.int16),  # one full-scale sample per block
        rng.normal(0, full * 0.3, size=(n // 2 + 77, 2)).round().astype(np.int16),   # a ragged last buffer
    ]
    iq = np.ascontiguousarray(np.concatenate(bufs)).view(np.uint8).reshape(-1)
    nsamples = iq.size // 4
    pfmt, ofmt = (pkg.FMT_SC16, oracle.FMT_SC16) if fmt == "sc16" else (pkg.FMT_SC16Q11, oracle.FMT_SC16Q11)
    d = torch_cuda.from_numpy(iq).to("cuda:0")
    dem = pkg.Demodulator(fmt=pfmt, nfix_crc=1, max_batch_samples=4 * pkg.CHUNK, message_capacity=1 << 16)
    got = pkg.replay_device(dem, d.data_ptr(), nsamples, 4 * pkg.CHUNK)
    want, wstats = oracle.Oracle(ofmt, 58, 1, 0).replay(iq, cap=1 << 16)
    assert_same(got, dem.stats(), want, wstats)


class MagBuf(C.Structure):
    pass


MagBuf._fields_ = [("data", C.POINTER(C.c_uint16)), ("totalLength", C.c_uint), ("validLength", C.c_uint), ("overlap", C.c_uint),
                   ("sampleTimestamp", C.c_uint64), ("sysTimestamp", C.c_uint64), ("flags", C.c_int), ("mean_level", C.c_double),
                   ("mean_power", C.c_double), ("dropped", C.c_uint), ("next", C.POINTER(MagBuf))]


@pytest.mark.parametrize("fmt,mode_ac", [("uc8", 1), ("sc16q11", 1), ("uc8", 0)])
def test_bound_void_demodulators_deliver_the_oracles_order(pkg, oracle, torch_cuda, fmt, mode_ac):
    """msd_demodulate2400(struct mag_buf *) / msd_demodulate2400AC(struct mag_buf *) with a bound context of their own
    (msd_demod_bind), called like readsb.c:826-829 calls the reference's pair: all Mode S messages of a buffer, then its
    Mode A/C replies, buffer after buffer -- message for message the oracle's list; msd_demod_error() speaks for the
    latest buffer only (an error of an earlier call does not linger)."""
    import os
    f, of = fmt_ids(pkg, oracle, fmt)
    bps = 2 if fmt == "uc8" else 4
    H = C.CDLL(os.path.join(os.path.dirname(pkg.capi.LIB_PATH), "libmsd_host.so"))
    sink_t = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p)
    H.msd_demod_bind.argtypes = [C.c_void_p, C.c_int, sink_t, C.c_void_p]
    H.msd_demodulate2400.argtypes = H.msd_demodulate2400AC.argtypes = [C.POINTER(MagBuf)]
    H.msd_demodulate2400.restype = H.msd_demodulate2400AC.restype = None
    H.msd_demod_error.restype = C.c_char_p
    n = 4 * CHUNK + 1234
    iq = pkg.siggen.generate(pkg.siggen.make_cfg(seed=78, fmt=f, msgs_per_sec=3000, n_aircraft=40,
                                                 ac_per_sec=1500 if mode_ac else 0), n)
    conv = pkg.Demodulator(fmt=f, nfix_crc=1, max_batch_samples=CHUNK)
    dem = pkg.Demodulator(fmt=f, nfix_crc=1, mode_ac=mode_ac, max_batch_samples=CHUNK, message_capacity=1 << 15)
    got, order = [], []

    def sink(mm, user):
        rec = np.frombuffer(C.string_at(mm, pkg.capi.MESSAGE_DTYPE.itemsize), dtype=pkg.capi.MESSAGE_DTYPE)[0].copy()
        got.append(rec)
        order.append(int(rec["msgtype"]) == 32)

    cb = sink_t(sink)
    # unbound: nothing is delivered and the error says so; binding clears it
    H.msd_demod_bind(None, 0, cb, None)
    dummy = MagBuf()
    H.msd_demodulate2400(C.byref(dummy))
    assert b"no receiver bound" in H.msd_demod_error()
    assert H.msd_demod_bind(dem._h, mode_ac, cb, None) == 0
    overlap = pkg.capi.OVERLAP
    carry = np.zeros(overlap, dtype=np.uint16)
    counter, per_buffer = 0, []
    for b in range(n // CHUNK + 1):
        m = min(CHUNK, n - b * CHUNK)
        part = iq[b * CHUNK * bps:(b * CHUNK + m) * bps]
        mag, level, power = conv.convert(part if m else np.zeros(16, np.uint8), m)
        data = np.ascontiguousarray(np.concatenate([carry, mag]))
        buf = MagBuf(data=data.ctypes.data_as(C.POINTER(C.c_uint16)), totalLength=data.size, validLength=overlap + m,
                     overlap=overlap, sampleTimestamp=counter * 5, sysTimestamp=counter * 5 // 12000, flags=0,
                     mean_level=level, mean_power=power, dropped=0)
        k0 = len(got)
        H.msd_demodulate2400(C.byref(buf))
        assert H.msd_demod_error() == b""          # the unbound call's error did not linger
        k1 = len(got)
        assert not any(order[k0:k1])                # Mode S only from the first call
        if mode_ac:
            H.msd_demodulate2400AC(C.byref(buf))
            assert all(order[k1:])                  # then the buffer's replies
        per_buffer.append((k1 - k0, len(got) - k1))
        carry = data[-overlap:]
        counter += m
    H.msd_demod_bind(None, 0, cb, None)
    want, _ = oracle.Oracle(of, 58, 1, mode_ac).replay(iq, cap=1 << 16)
    assert len(want) > 100 and (not mode_ac or sum(k for _, k in per_buffer) > 10)
    assert_same_messages(np.array(got, dtype=pkg.capi.MESSAGE_DTYPE), want)


@pytest.mark.parametrize("fmt", ["uc8", "sc16", "sc16q11"])
def test_dc_filter_converter_carries_its_state_from_call_to_call(pkg, oracle, torch_cuda, fmt):
    """init_converter(format, 2.4 MHz, filter_dc = 1): convert_uc8_generic / convert_sc16_generic / convert_sc16q11_generic
    (convert.c:113-213, 374-423).  z1_I / z1_Q live in the converter state and run on through every call
    (convert.c:28-33): five blocks of different lengths, one context each side, magnitudes and both means bit for bit --
    and a second converter that starts in the middle of the stream must NOT agree (its state starts at zero)."""
    f, of = fmt_ids(pkg, oracle, fmt)
    bps = 2 if fmt == "uc8" else 4
    n = 3 * CHUNK
    iq = pkg.siggen.generate(pkg.siggen.make_cfg(seed=515, fmt=f, msgs_per_sec=3000), n)
    if fmt == "uc8":     # a DC offset for the block to find
        iq = np.clip(iq.astype(np.int32) + np.tile(np.array([9, -6]), n), 0, 255).astype(np.uint8)
    dem = pkg.Demodulator(fmt=f, nfix_crc=1, max_batch_samples=CHUNK, dc_filter=True)
    orc = oracle.Oracle(of, 58, 1, 0, dc_filter=True)
    off, first_mags = 0, None
    for m in (CHUNK, 1, 4097, 0, CHUNK - 5000):
        blk = iq[off * bps:(off + m) * bps]
        gm, gl, gp = dem.convert(blk if m else iq[:8], m)
        wm, wl, wp = orc.convert(blk if m else iq[:8], m) if m else (np.zeros(0, np.uint16), np.nan, np.nan)
        assert np.array_equal(gm[:m], wm), (fmt, off, m)
        assert np.array_equal(np.float64(gl), np.float64(wl), equal_nan=True) and np.array_equal(np.float64(gp), np.float64(wp), equal_nan=True)
        if first_mags is None:
            first_mags = gm.copy()
        off += m
    fresh = pkg.Demodulator(fmt=f, nfix_crc=1, max_batch_samples=CHUNK, dc_filter=True)
    m2, _, _ = fresh.convert(iq[CHUNK * bps:(CHUNK + 4097) * bps], 4097)
    cont = orc.convert(iq[off * bps:(off + 16) * bps], 16)[0]   # the running oracle is further along: only used to keep it honest
    assert cont.size == 16
    if fmt == "uc8":
        nodc = pkg.Demodulator(fmt=f, nfix_crc=1, max_batch_samples=CHUNK).convert(iq[:CHUNK * bps], CHUNK)[0]
        assert not np.array_equal(nodc, first_mags)   # the block did find the offset


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", ["uc8", "sc16"])
def test_dc_filter_converter_at_another_sample_rate(pkg, torch_cuda, fmt):
    """init_converter derives dc_b = exp(-2 pi / sample_rate) from whatever rate it is given (convert.c:479-482), not only
    from Modes.sample_rate: msd_config.sample_rate = 2.0 MHz against the second reading's converter at that rate (numpy
    float32, tests/indep_demod.py) -- magnitudes bit for bit -- and the 2.4 MHz converter must differ from it."""
    import indep_demod
    f = pkg.FMT_UC8 if fmt == "uc8" else pkg.FMT_SC16
    bps = 2 if fmt == "uc8" else 4
    n = 20000
    iq = pkg.siggen.generate(pkg.siggen.make_cfg(seed=616, fmt=f, msgs_per_sec=3000), n)
    if fmt == "uc8":
        iq = np.clip(iq.astype(np.int32) + np.tile(np.array([11, -7]), n), 0, 255).astype(np.uint8)
    got = pkg.Demodulator(fmt=f, max_batch_samples=CHUNK, dc_filter=True, sample_rate=2.0e6).convert(iq[:n * bps], n)[0][:n]
    want = indep_demod.convert(fmt, iq[:n * bps].tobytes(), dc=True, sample_rate=2.0e6)[0]
    assert np.array_equal(got, want)
    at_2400 = pkg.Demodulator(fmt=f, max_batch_samples=CHUNK, dc_filter=True).convert(iq[:n * bps], n)[0][:n]
    assert np.array_equal(at_2400, indep_demod.convert(fmt, iq[:n * bps].tobytes(), dc=True)[0]) and not np.array_equal(at_2400, got)


@pytest.mark.gpu
def test_convert_in_two_halves(pkg, oracle, torch_cuda):
    """msd_convert_begin / msd_convert_end (what the ifile handler's reader uses to read its next block while the GPU
    converts): the same magnitudes and means as msd_convert; one conversion in flight per context; _end without _begin and a
    second _begin are refused."""
    import errno
    L = pkg.capi.lib()
    f, of = fmt_ids(pkg, oracle, "sc16")
    n = 50000
    iq = pkg.siggen.generate(pkg.siggen.make_cfg(seed=5, fmt=f, msgs_per_sec=3000), n)
    dem = pkg.Demodulator(fmt=f, max_batch_samples=CHUNK)
    want_mag, want_level, want_power = dem.convert(iq, n)
    mag = np.zeros(n, dtype=np.uint16)
    level, power = C.c_double(), C.c_double()
    L.msd_convert_begin.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint]
    L.msd_convert_end.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.msd_thread_attach.argtypes = [C.c_void_p]
    assert L.msd_convert_end(dem._h, C.byref(level), C.byref(power)) == -errno.EINVAL
    assert L.msd_convert_begin(dem._h, iq.ctypes.data, mag.ctypes.data, n) == 0
    assert L.msd_convert_begin(dem._h, iq.ctypes.data, mag.ctypes.data, n) == -errno.EBUSY
    assert L.msd_convert_end(dem._h, C.byref(level), C.byref(power)) == 0
    assert np.array_equal(mag, want_mag[:n]) and level.value == want_level and power.value == want_power
    assert L.msd_thread_attach(dem._h) == 0
