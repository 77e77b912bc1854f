"""Shared helpers of the test-suite."""
import json
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FIELDS = ("timestampMsg", "sysTimestampMsg", "signalLevel", "addr", "crc", "score", "msgtype", "msgbits",
          "correctedbits", "bestphase", "iid")
COUNTERS = ("demod_preambles", "demod_rejected_bad", "demod_rejected_unknown_icao", "demod_accepted",
            "demod_preamblePhase", "demod_bestPhase", "demod_modeac", "strong_signal_count",
            "samples_processed", "noise_power_count", "signal_power_count", "buffers")
FLOAT_COUNTERS = ("noise_power_sum", "signal_power_sum", "peak_signal_power")


def golden_names():
    return sorted(f[:-4] for f in os.listdir(GOLDEN_DIR) if f.endswith(".npz"))


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    meta = json.loads(str(z["meta"]))
    return meta, z


def assert_same_messages(got, want, fields=FIELDS):
    assert len(got) == len(want["timestampMsg"]), (len(got), len(want["timestampMsg"]))
    for f in fields:
        assert np.array_equal(got[f], want[f]), f
    assert np.array_equal(got["msg"], want["msg"])


def assert_same_stats(gstats, wstats):
    for k in COUNTERS:
        assert gstats[k] == wstats[k], (k, gstats[k], wstats[k])
    for k in FLOAT_COUNTERS:
        assert np.array_equal(np.float64(gstats[k]), np.float64(wstats[k]), equal_nan=True), (k, gstats[k], wstats[k])


def fmt_ids(pkg, oracle, fmt):
    return {"uc8": (pkg.FMT_UC8, oracle.FMT_UC8), "sc16": (pkg.FMT_SC16, oracle.FMT_SC16),
            "sc16q11": (pkg.FMT_SC16Q11, oracle.FMT_SC16Q11)}[fmt]


def oracle_live_feed(orc, segments, drops, bytes_per_sample=2, chunk=131072, overlap=326, cap=1 << 16):
    """The oracle fed the way a live receiver feeds the reference: one mag_buf per 131072 samples with
    the FIFO's overlap handling (fifo.c:176-184: zeros in front of a MAGBUF_DISCONTINUOUS buffer, the
    previous tail otherwise), rtlsdrCallback's sample clock, which keeps counting over dropped
    samples (sdr_rtlsdr.c:281-300), and --ifile's system clock (sdr_ifile.c:190, startup_time 0).
    segments: IQ byte arrays, all but the last a whole number of buffers; drops[i]: samples lost in
    front of segment i.  The last segment ends like a file (a final short or empty buffer)."""
    counter, carry, out = 0, np.zeros(overlap, dtype=np.uint16), []
    for si, (seg, drop) in enumerate(zip(segments, drops)):
        seg = np.ascontiguousarray(seg).view(np.uint8).reshape(-1)
        n = seg.size // bytes_per_sample
        last = si == len(segments) - 1
        assert last or n % chunk == 0
        counter += drop
        discontinuous = drop > 0
        for b in range(n // chunk + (1 if last else 0)):
            part = seg[b * chunk * bytes_per_sample: (b + 1) * chunk * bytes_per_sample]
            m = part.size // bytes_per_sample
            mag, mean_level, mean_power = orc.convert(part, m) if m else (np.zeros(0, np.uint16), np.nan, np.nan)
            data = np.concatenate([np.zeros(overlap, np.uint16) if discontinuous else carry, mag])
            discontinuous = False
            sample_ts = int(counter * 12e6 / 2400000.0)
            out.append(orc.demod_buffer(data, sample_ts, sample_ts // 12000, mean_level, mean_power, cap=cap))
            carry = data[data.size - overlap:].copy()
            counter += m
    return np.concatenate(out) if out else np.zeros(0, dtype=out_dtype(orc)), orc.stats()


def out_dtype(orc):
    return orc.demod_buffer(np.zeros(326, np.uint16), cap=1).dtype


def assert_same(got, gstats, want, wstats):
    assert len(got) == len(want), (len(got), len(want))
    for f in FIELDS:
        assert np.array_equal(got[f], want[f]), f
    assert np.array_equal(got["msg"], want["msg"])
    assert_same_stats(gstats, wstats)
