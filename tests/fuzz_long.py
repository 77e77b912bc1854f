"""Long random captures (not collected by pytest): python tests/fuzz_long.py [cases] [first_seed]
tests/fuzz_parity.py draws at most forty buffers, so the ICAO filter's 60 s flips (icao_filter.c:150-164: 1100 buffers) and
batches of hundreds of buffers only meet in the handful of fixed tests.  Here every case is 600 to 2500 buffers of random
traffic -- up to 30 000 aircraft, so that addresses are dropped by a flip and come back -- in batches of 64 to 1024 buffers,
GPU resolve, against the oracle message for message and counter for counter."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402
from tests.test_gpu_parity import assert_same  # noqa: E402

pkg = g.load_package()
orc = g.load_oracle()
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
first = int(sys.argv[2]) if len(sys.argv) > 2 else 7000
os.environ["MSD_GPU_RESOLVE"] = "1"
bad = 0
for case in range(first, first + ncases):
    rng = np.random.default_rng(case)
    nbuf = int(rng.integers(600, 2500))
    n = nbuf * 131072 + int(rng.choice([0, 1234, 131071]))
    kw = dict(msgs_per_sec=int(rng.choice([500, 2000, 6000, 12000])), n_aircraft=int(rng.choice([50, 800, 5000, 30000])),
              overlap_permille=int(rng.choice([0, 10, 200])), flip_permille=int(rng.choice([0, 20, 200])),
              noise_fs=float(rng.choice([0.005, 0.02, 0.06])), ac_per_sec=int(rng.choice([0, 0, 500])))
    nfix = int(rng.integers(0, 3))
    mode_ac = int(kw["ac_per_sec"] > 0 and rng.integers(0, 2))
    batch = int(rng.choice([64, 256, 1024])) * 131072
    thr = int(rng.choice([58, 58, 75]))
    dc = bool(rng.integers(0, 4) == 0)
    fmt_name = str(rng.choice(["uc8", "uc8", "sc16", "sc16q11"]))  # (drawn last: the seeds of the first campaigns keep their traffic)
    fmt, ofmt = {"uc8": (pkg.FMT_UC8, orc.FMT_UC8), "sc16": (pkg.FMT_SC16, orc.FMT_SC16), "sc16q11": (pkg.FMT_SC16Q11, orc.FMT_SC16Q11)}[fmt_name]
    if fmt_name != "uc8":
        nbuf = min(nbuf, 1400)  # 4 bytes a sample
        n = min(n, nbuf * 131072)
    iq = pkg.siggen.generate(pkg.siggen.make_cfg(seed=case, fmt=fmt, **kw), n)
    d = torch.from_numpy(iq).to("cuda:0")
    dem = pkg.Demodulator(fmt=fmt, preamble_threshold=thr, nfix_crc=nfix, mode_ac=mode_ac, max_batch_samples=batch, message_capacity=1 << 22, dc_filter=dc)
    got = pkg.replay_device(dem, d.data_ptr(), n, batch)
    want, wstats = orc.Oracle(ofmt, thr, nfix, mode_ac, dc_filter=dc).replay(iq, cap=1 << 22)
    desc = f"case {case}: {fmt_name} buffers={nbuf} batch={batch // 131072} nfix={nfix} ac={mode_ac} dc={int(dc)} thr={thr} {kw}"
    try:
        assert_same(got, dem.stats(), want, wstats)
        print("ok  ", desc, "msgs", len(want), "passes", dem.timing()["resolve_passes"], "fallback", dem.timing()["resolve_fallback"], flush=True)
    except AssertionError as e:
        bad += 1
        print("FAIL", desc, str(e)[:200], flush=True)
    del dem, d
print("failures:", bad)
sys.exit(1 if bad else 0)
