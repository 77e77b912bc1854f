"""Whole-job rate on the adversarial pulse train of tests/test_gpu_adversarial.py (36 % of the positions are preamble
hits, five trial phases each): python scripts/adversarial_rate.py [buffers]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g  # noqa: E402
from test_gpu_adversarial import pulse_train  # noqa: E402

pkg = g.load_package()
nbuf = int(sys.argv[1]) if len(sys.argv) > 1 else 512
n = nbuf * 131072
quiet = pkg.siggen.generate(pkg.siggen.make_cfg(seed=77, msgs_per_sec=0, noise_fs=0.005), n)
train = pulse_train(n, 11)
batch = 512 * 131072
print("pulse train in a share of every buffer (the rest quiet noise), %d buffers, one batch of 512 buffers at a time:" % nbuf)
for duty in (0.0, 0.05, 0.1, 0.15, 0.2, 0.25, 0.3, 0.5, 1.0):
    iq = quiet.copy().reshape(nbuf, 131072, 2)
    k = int(duty * 131072)
    iq[:, :k, :] = train.reshape(nbuf, 131072, 2)[:, :k, :]
    d = torch.from_numpy(iq.reshape(-1)).to("cuda:0")
    dem = pkg.Demodulator(fmt=pkg.FMT_UC8, preamble_threshold=58, nfix_crc=0, mode_ac=0, max_batch_samples=batch, message_capacity=1 << 21)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pkg.replay_device(dem, d.data_ptr(), n, batch)  # the first pass over the capture: the slots' region slices as msd_create made them
    dt_first = time.perf_counter() - t0
    tm_first = dem.timing()
    dem.reset()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    got = pkg.replay_device(dem, d.data_ptr(), n, batch)
    dt = time.perf_counter() - t0
    st, tm = dem.stats(), dem.timing()
    print("  share %4.2f: %8.2f GS/s (first pass, slices still growing: %6.2f GS/s, %d rescans, %d host-resolved)  hits/sample %.3f  tries/sample %.3f  "
          "rescans %d  host-resolved batches %d  messages %d" %
          (duty, n / dt / 1e9, n / dt_first / 1e9, tm_first["reruns"], tm_first["resolve_fallback"], st["demod_preambles"] / n,
           sum(st["demod_preamblePhase"]) / n, tm["reruns"], tm["resolve_fallback"], len(got)))
    del dem, d
