"""Golden fixtures (tests/golden/*.npz, made by make_golden.py): seeded capture -> ordered message
list + counters.  CPU part: the generator still produces the pinned bytes and the oracle still
produces the pinned list.  GPU part: the HIP path, through the C-ABI, produces the same list."""
import hashlib

import numpy as np
import pytest

from helpers import (FIELDS, assert_same_messages, assert_same_stats, fmt_ids, golden_names, load_golden)


def _capture(pkg, meta):
    fmt = {"uc8": pkg.FMT_UC8, "sc16": pkg.FMT_SC16, "sc16q11": pkg.FMT_SC16Q11}[meta["format"]]
    cfg = pkg.siggen.make_cfg(seed=meta["seed"], fmt=fmt, **meta["gen"])
    return pkg.siggen.generate(cfg, meta["nsamples"], nthreads=4)


@pytest.mark.parametrize("name", golden_names())
def test_oracle_reproduces_golden(pkg, oracle, name):
    meta, z = load_golden(name)
    iq = _capture(pkg, meta)
    assert hashlib.sha256(iq.tobytes()).hexdigest() == meta["iq_sha256"], "generator output changed"
    _, of = fmt_ids(pkg, oracle, meta["format"])
    msgs, stats, means = oracle.Oracle(of, 58, meta["nfix_crc"], meta["mode_ac"]).replay(iq, want_means=True)
    assert_same_messages(msgs, z)
    assert_same_stats(stats, meta["stats"])
    assert np.array_equal(means, z["means"], equal_nan=True)
    # structure of every fixture: ordered per buffer, sane fields
    assert set(np.unique(msgs["msgbits"])) <= {16, 56, 112}
    assert (msgs["correctedbits"] <= meta["nfix_crc"]).all()


@pytest.mark.parametrize("name", golden_names())
def test_second_reading_reproduces_golden(pkg, name):
    """The committed vectors were written by the oracle; the second reading of the reference (tests/indep_demod.py, no
    oracle involved) produces the same lists and counters from the same bytes -- the fixtures are two readings' answer."""
    import indep_demod
    from test_indep_demod import FLOAT_STATS, INT_STATS
    meta, z = load_golden(name)
    iq = _capture(pkg, meta)
    msgs, stats = indep_demod.Receiver(meta["format"], 58, meta["nfix_crc"], bool(meta["mode_ac"])).replay(iq.tobytes())
    assert len(msgs) == len(z["timestampMsg"])
    for i, m in enumerate(msgs):
        for k in ("timestampMsg", "sysTimestampMsg", "addr", "msgtype", "msgbits", "correctedbits"):
            assert m[k] == int(z[k][i]), (i, k)
        nb = m["msgbits"] // 8
        assert m["msg"][:nb] == bytes(z["msg"][i][:nb]), i
        if m["msgtype"] != 32:
            for k in ("crc", "score", "bestphase", "iid"):
                assert m[k] == int(z[k][i]), (i, k)
            assert m["signalLevel"] == float(z["signalLevel"][i]), i
    for k in INT_STATS:
        assert stats[k] == meta["stats"][k], k
    for k in FLOAT_STATS:
        assert np.array_equal(np.float64(stats[k]), np.float64(meta["stats"][k]), equal_nan=True), k


def test_exact_multiple_capture_has_trailing_empty_buffer(oracle):
    """N = k * 131072 samples are k+1 buffers, the last empty with NaN means."""
    meta, z = load_golden("uc8_fix_exact_multiple")
    assert meta["stats"]["buffers"] == meta["nsamples"] // 131072 + 1
    assert np.isnan(z["means"][-1]).all() and np.isnan(meta["stats"]["noise_power_sum"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", golden_names())
def test_gpu_reproduces_golden(pkg, oracle, torch_cuda, name):
    meta, z = load_golden(name)
    iq = _capture(pkg, meta)
    pf, _ = fmt_ids(pkg, oracle, meta["format"])
    d = torch_cuda.from_numpy(iq).to("cuda:0")
    dem = pkg.Demodulator(fmt=pf, nfix_crc=meta["nfix_crc"], mode_ac=meta["mode_ac"], max_batch_samples=8 * pkg.CHUNK)
    got = dem.submit_device(d.data_ptr(), meta["nsamples"], last=True)
    assert_same_messages(got, z)
    assert_same_stats(dem.stats(), meta["stats"])
    assert np.array_equal(dem.buffer_means(), z["means"], equal_nan=True)
