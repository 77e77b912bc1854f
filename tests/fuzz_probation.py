"""Targeted stress of the resolve kernel's probation table (not collected by pytest): python tests/fuzz_probation.py [cases] [first_seed]
Every case is a fresh receiver (an empty ICAO filter) on two to four buffers of dense, overlapping, bit-flipped traffic from a few
dozen aircraft: the first clean squitter of every address is expected to add it, many of those are hidden by the message in front
of them or garbled, and corrected messages of the same address follow within the buffer -- the interleavings in which fuzz case
702780 found a stale confirmation (LABLOG R6.4).  GPU resolve against the oracle, message for message and counter for counter."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402
from tests.test_gpu_parity import assert_same  # noqa: E402

pkg = g.load_package()
orc = g.load_oracle()
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
first = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
os.environ["MSD_GPU_RESOLVE"] = "1"
bad = 0
for case in range(first, first + ncases):
    rng = np.random.default_rng(case)
    nbuf = int(rng.integers(2, 5))
    n = nbuf * 131072 + int(rng.choice([0, 777]))
    kw = dict(msgs_per_sec=int(rng.choice([6000, 12000, 24000, 40000])), n_aircraft=int(rng.choice([10, 30, 100, 400])),
              overlap_permille=int(rng.choice([100, 400, 800])), flip_permille=int(rng.choice([20, 100, 300])),
              noise_fs=float(rng.choice([0.005, 0.02])), ac_per_sec=0)
    nfix = int(rng.integers(1, 3))
    batch = int(rng.choice([1, 2, 4])) * 131072
    thr = int(rng.choice([58, 58, 40]))
    iq = pkg.siggen.generate(pkg.siggen.make_cfg(seed=case, **kw), n)
    d = torch.from_numpy(iq).to("cuda:0")
    dem = pkg.Demodulator(preamble_threshold=thr, nfix_crc=nfix, max_batch_samples=batch, message_capacity=1 << 19)
    got = pkg.replay_device(dem, d.data_ptr(), n, batch)
    want, wstats = orc.Oracle(orc.FMT_UC8, thr, nfix, 0).replay(iq, cap=1 << 19)
    desc = f"case {case}: n={n} batch={batch // 131072} nfix={nfix} thr={thr} {kw}"
    try:
        assert_same(got, dem.stats(), want, wstats)
        print("ok  ", desc, "msgs", len(want), "corrected", int((want["correctedbits"] > 0).sum()), "passes", dem.timing()["resolve_passes"], "fallback", dem.timing()["resolve_fallback"])
    except AssertionError as e:
        bad += 1
        print("FAIL", desc, str(e)[:200])
    del dem
print("failures:", bad)
sys.exit(1 if bad else 0)
