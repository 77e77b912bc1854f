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
