"""One of several processes sharing a GPU (tests/test_gpu_two_processes.py): its own context, its own capture, the
pipelined path, compared with the oracle message for message.  Usage: two_proc_worker.py <seed> <buffers> <start file>"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import __graft_entry__ as graft  # noqa: E402

seed, nbuf, start = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
pkg, oracle = graft.load_package(), graft.load_oracle()
n = nbuf * pkg.CHUNK + 1234
iq = pkg.siggen.generate(pkg.siggen.make_cfg(seed=seed), n)
want, wstats = oracle.Oracle(oracle.FMT_UC8, 58, 1, 0).replay(iq, cap=1 << 18)
d = torch.from_numpy(iq).to("cuda:0")
dem = pkg.Demodulator(fmt=pkg.FMT_UC8, nfix_crc=1, max_batch_samples=8 * pkg.CHUNK, message_capacity=1 << 18)
open(start + f".{seed}.ready", "w").close()
t0 = time.time()
while not os.path.exists(start) and time.time() - t0 < 120:  # both processes start their passes together
    time.sleep(0.01)
for rep in range(6):  # several passes, so that the two processes' kernels really interleave on the device
    dem.reset()
    got = pkg.replay_device(dem, d.data_ptr(), n, 8 * pkg.CHUNK)
    assert len(got) == len(want) and len(want) > 300, (len(got), len(want))
    for f in ("timestampMsg", "sysTimestampMsg", "signalLevel", "addr", "msgtype", "correctedbits", "score", "crc", "bestphase"):
        assert np.array_equal(got[f], want[f]), (rep, f)
    assert np.array_equal(got["msg"], want["msg"])
    st = dem.stats()
    for k in ("demod_preambles", "demod_rejected_bad", "demod_rejected_unknown_icao", "demod_accepted"):
        assert st[k] == wstats[k], (rep, k, st[k], wstats[k])
print(f"worker {seed}: {len(want)} messages x 6 passes identical to the oracle")
