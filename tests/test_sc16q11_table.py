"""convert_sc16q11_table (convert.c:264-328): what a reference built with -DSC16Q11_TABLE_BITS=n does with SC16Q11
samples (debian/rules:19 sets n = 8 on armhf).  CPU part: the oracle's restatement against an independent numpy one, and
the product's table builder against the oracle's.  GPU part: the whole path against the oracle."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

CHUNK = 131072


def numpy_table(bits):
    """init_sc16q11_lookup, convert.c:271-295, in numpy float32 (FLT_EVAL_METHOD 0: every operation rounds to float)."""
    lose = 11 - bits
    axis = np.arange(0, 2048, 1 << lose, dtype=np.int32)
    f = (axis / 2048.0).astype(np.float32)
    sq = f * f
    magsq = sq[:, None] + sq[None, :]
    magsq = np.where(magsq > np.float32(1), np.float32(1), magsq).astype(np.float32)
    mag = np.sqrt(magsq).astype(np.float32)
    return (mag * np.float32(65535.0) + np.float32(0.5)).astype(np.uint16).reshape(-1)   # index = (i >> lose) << bits | q >> lose


def numpy_convert(iq16, bits):
    """convert_sc16q11_table, convert.c:297-328"""
    lose = 11 - bits
    tab = numpy_table(bits)
    a = np.abs(iq16.astype(np.int32)) & 2047       # abs((int16_t)x) is an int: -32768 -> 32768 -> 0
    idx = ((a[0::2] >> lose) << bits) | (a[1::2] >> lose)
    mag = tab[idx]
    n = mag.size
    level = float(mag.astype(np.uint64).sum()) / 65536.0 / n
    power = float((mag.astype(np.uint64) ** 2).sum()) / 65535.0 / 65535.0 / n
    return mag, level, power


@pytest.mark.parametrize("bits", [1, 7, 8, 9, 11])
def test_oracle_table_equals_the_numpy_restatement(oracle, bits):
    o = oracle.Oracle(oracle.FMT_SC16Q11, 58, 1, 0, sc16q11_table_bits=bits)
    tab = np.ctypeslib.as_array(oracle.lib().orc_sc16q11_table(o._h), shape=(1 << (2 * bits),)).copy()
    assert np.array_equal(tab, numpy_table(bits))
    assert tab[0] == 0 and tab[-1] == 65535 if bits > 1 else True


@pytest.mark.parametrize("bits", [7, 8, 11])
def test_oracle_table_converter_known_answers(oracle, bits):
    rng = np.random.default_rng(bits)
    iq16 = rng.integers(-32768, 32768, size=2 * 5000, dtype=np.int64).astype(np.int16)
    iq16[:8] = [-32768, 0, 2047, 0, -2047, 2047, 2048, -1]   # abs(-32768) & 2047 = 0; 2048 & 2047 = 0
    o = oracle.Oracle(oracle.FMT_SC16Q11, 58, 1, 0, sc16q11_table_bits=bits)
    mag, lvl, pwr = o.convert(iq16.view(np.uint8), iq16.size // 2)
    wm, wl, wp = numpy_convert(iq16, bits)
    assert np.array_equal(mag, wm) and lvl == wl and pwr == wp
    lose = 11 - bits
    assert mag[0] == 0 and mag[3] == numpy_table(bits)[1 >> lose] and mag[1] == numpy_table(bits)[(2047 >> lose) << bits]
    assert mag[2] == numpy_table(bits)[((2047 >> lose) << bits) | (2047 >> lose)] == 65535
    # without the define the same samples go the float way (convert.c:439-441)
    f = oracle.Oracle(oracle.FMT_SC16Q11, 58, 1, 0)
    fm, _, _ = f.convert(iq16.view(np.uint8), iq16.size // 2)
    assert not np.array_equal(fm, mag)


@pytest.mark.parametrize("bits", [1, 5, 8, 11])
def test_product_table_builder_equals_the_oracles(pkg, oracle, bits):
    out = np.zeros(1 << (2 * bits), dtype=np.uint16)
    L = pkg.capi.lib()
    L.msd_sc16q11_table_build.restype = None
    L.msd_sc16q11_table_build.argtypes = [C.c_int, C.c_void_p]
    L.msd_sc16q11_table_build(bits, out.ctypes.data)
    assert np.array_equal(out, numpy_table(bits))


def test_create_rejects_table_bits_on_other_formats(pkg):
    for fmt, bits in ((pkg.FMT_UC8, 8), (pkg.FMT_SC16, 8), (pkg.FMT_SC16Q11, 12), (pkg.FMT_SC16Q11, -1)):
        cfg = pkg.capi.Config(device=0, format=fmt, preamble_threshold=58, nfix_crc=1, max_batch_samples=CHUNK,
                              sc16q11_table_bits=bits)
        h = C.c_void_p()
        assert pkg.capi.lib().msd_create(C.byref(cfg), C.byref(h)) == -22   # before any device is touched


# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("bits", [7, 8, 11])
@pytest.mark.parametrize("n", [0, 1, 9, 4096 * 4096])
def test_converter_entry_with_the_table(pkg, oracle, torch_cuda, bits, n):
    """iq_convert_fn of a -DSC16Q11_TABLE_BITS build: every (|I| & 2047, |Q| & 2047) pair, sign and wrap-around included"""
    if n == 4096 * 4096:
        axis = np.concatenate([np.arange(-2048, 2048, dtype=np.int32)[:4090], [-32768, 32767, 2048, -2049, 4095, -4096]])
        i_vals, q_vals = np.meshgrid(axis, axis, indexing="ij")
        order = np.random.default_rng(3).permutation(i_vals.size)
        iq16 = np.empty(2 * i_vals.size, dtype=np.int16)
        iq16[0::2] = i_vals.reshape(-1)[order]
        iq16[1::2] = q_vals.reshape(-1)[order]
    else:
        iq16 = np.random.default_rng(n).integers(-32768, 32768, size=2 * max(n, 4), dtype=np.int64).astype(np.int16)
    dem = pkg.Demodulator(fmt=pkg.FMT_SC16Q11, nfix_crc=1, max_batch_samples=max(n, CHUNK), sc16q11_table_bits=bits)
    mag, lvl, pwr = dem.convert(iq16.view(np.uint8), n)
    o = oracle.Oracle(oracle.FMT_SC16Q11, 58, 1, 0, sc16q11_table_bits=bits)
    wm, wl, wp = o.convert(iq16.view(np.uint8), n) if n else (np.zeros(0, np.uint16), np.nan, np.nan)
    assert np.array_equal(mag[:n], wm)
    assert np.array_equal(np.float64(lvl), np.float64(wl), equal_nan=True)
    assert np.array_equal(np.float64(pwr), np.float64(wp), equal_nan=True)


@pytest.mark.gpu
@pytest.mark.parametrize("resolve", ["gpu-resolve", "host-resolve"])
@pytest.mark.parametrize("bits,mode_ac,nfix,batch", [(8, 0, 1, None), (8, 1, 1, 4), (7, 0, 0, 2), (11, 1, 2, 8)])
def test_table_build_demodulates_like_the_oracle(pkg, oracle, torch_cuda, monkeypatch, bits, mode_ac, nfix, batch, resolve):
    """messages, every counter and the per-buffer means (integer sums here: convert.c:318-326) of a capture that goes
    through the table converter, single batch and pipelined, with and without Mode A/C"""
    from tests.test_gpu_parity import assert_same
    monkeypatch.setenv("MSD_GPU_RESOLVE", "1" if resolve == "gpu-resolve" else "0")
    n = 9 * CHUNK + 1237
    iq = pkg.siggen.generate(pkg.siggen.make_cfg(seed=7100 + bits, fmt=pkg.FMT_SC16Q11, msgs_per_sec=4000,
                                                  ac_per_sec=600 if mode_ac else 0), n)
    d = torch_cuda.from_numpy(iq).to("cuda:0")
    max_batch = (batch or 10) * CHUNK
    dem = pkg.Demodulator(fmt=pkg.FMT_SC16Q11, nfix_crc=nfix, mode_ac=mode_ac, max_batch_samples=max_batch,
                          message_capacity=1 << 17, sc16q11_table_bits=bits)
    got = pkg.replay_device(dem, d.data_ptr(), n, max_batch) if batch else dem.submit_device(d.data_ptr(), n, last=True)
    o = oracle.Oracle(oracle.FMT_SC16Q11, 58, nfix, mode_ac, sc16q11_table_bits=bits)
    want, wstats, wmeans = o.replay(iq, cap=1 << 17, want_means=True)
    assert len(want) > 100
    assert_same(got, dem.stats(), want, wstats)
    if not batch:
        gm = dem.buffer_means()
        assert np.array_equal(gm, wmeans[: len(gm)], equal_nan=True) and len(gm) == wstats["buffers"]
    # and it is not what the float path decodes
    f, fstats = oracle.Oracle(oracle.FMT_SC16Q11, 58, nfix, mode_ac).replay(iq, cap=1 << 17)
    assert fstats["noise_power_sum"] != wstats["noise_power_sum"]


@pytest.mark.gpu
def test_replay_cli_of_a_table_build(pkg, oracle, torch_cuda, tmp_path):
    n = 5 * CHUNK + 77
    iq = pkg.siggen.generate(pkg.siggen.make_cfg(seed=7177, fmt=pkg.FMT_SC16Q11, msgs_per_sec=3000), n)
    f = tmp_path / "capture.sc16q11"
    iq.tofile(f)
    exe = os.path.join(os.path.dirname(pkg.capi.LIB_PATH), "msd_replay")
    want, _ = oracle.Oracle(oracle.FMT_SC16Q11, 58, 1, 0, sc16q11_table_bits=8).replay(iq, cap=1 << 16)
    for path in ("fused", "magbuf"):
        out = subprocess.run([exe, "--ifile", str(f), "--iformat", "sc16q11", "--sc16q11-table-bits", "8", "--raw", "--path", path,
                              "--batch-buffers", "2"], capture_output=True, text=True, check=True)
        lines = out.stdout.split()
        assert len(lines) == len(want) > 50, path
        for line, m in zip(lines, want):
            assert line == "*%s;" % bytes(m["msg"][: m["msgbits"] // 8]).hex(), path
