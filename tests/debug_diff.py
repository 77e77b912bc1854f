"""Debug helper (not a test): run one synthetic case through the GPU-resolve path, the host-resolve
path and the oracle, and show the first differing messages.  Run on the GPU box:
python tests/debug_diff.py <seed> <overlap_permille> <n_aircraft>"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402

pkg = g.load_package()
seed, overlap, aircraft = (int(x) for x in sys.argv[1:4])
n = 48 * 131072 + 1234
cfg = pkg.siggen.make_cfg(seed=seed, n_aircraft=aircraft, msgs_per_sec=6000, overlap_permille=overlap, flip_permille=50)
iq = pkg.siggen.generate(cfg, n)
d = torch.from_numpy(iq).to("cuda:0")
out = {}
for mode in ("1", "0"):
    os.environ["MSD_GPU_RESOLVE"] = mode
    dem = pkg.Demodulator(fmt=pkg.FMT_UC8, nfix_crc=1, max_batch_samples=16 * 131072, message_capacity=1 << 18)
    out[mode] = pkg.replay_device(dem, d.data_ptr(), n, 16 * 131072).copy()
    print(mode, len(out[mode]), dem.timing())
orc = g.load_oracle()
want, wstats = orc.Oracle(orc.FMT_UC8, 58, 1, 0).replay(iq, cap=1 << 18)
print("oracle", len(want))
a, b = out["1"], want
i = 0
while i < min(len(a), len(b)) and a[i]["timestampMsg"] == b[i]["timestampMsg"] and a[i]["score"] == b[i]["score"]:
    i += 1
print("first difference at message", i)
for name, arr in (("gpu-resolve", a), ("oracle", b)):
    for m in arr[max(0, i - 2): i + 3]:
        ts = int(m["timestampMsg"])
        print(name, "buffer", ts // (131072 * 5), "j~", (ts % (131072 * 5)) // 5, "df", m["msgtype"], "addr %06x" % m["addr"],
              "score", m["score"], "fix", m["correctedbits"], "phase", m["bestphase"])
x = a[i]["addr"] if i < len(a) else 0
for name, arr in (("gpu-resolve", a), ("oracle", b)):
    sel = arr[arr["addr"] == x]
    print(name, "all messages of %06x:" % x, [(int(t) // (131072 * 5), int(s), int(df)) for t, s, df in zip(sel["timestampMsg"], sel["score"], sel["msgtype"])][:12])
