"""Reproduce one case of tests/fuzz_parity.py with overrides and show where the lists part:
python scripts/experiments/r5_fuzz_case.py <case> [KEY=VALUE ...]   (mode_ac=0, arena=200, growth=1, nfix=0, batch=8, gpu_resolve=0)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

pkg, orc = g.load_package(), g.load_oracle()
case = int(sys.argv[1])
ov = dict(a.split("=") for a in sys.argv[2:])
rng = np.random.default_rng(case)
fmt_name = rng.choice(["uc8", "uc8", "uc8", "sc16", "sc16q11"])
fmt, ofmt = {"uc8": (pkg.FMT_UC8, orc.FMT_UC8), "sc16": (pkg.FMT_SC16, orc.FMT_SC16), "sc16q11": (pkg.FMT_SC16Q11, orc.FMT_SC16Q11)}[fmt_name]
nbuf = int(rng.integers(1, 40))
n = nbuf * 131072 + int(rng.choice([0, 1, 7, 8, 1234, 65536, 131071]))
batch = int(rng.choice([1, 2, 4, 8, 16, 64])) * 131072
kw = dict(msgs_per_sec=int(rng.choice([200, 2000, 6000, 12000])), n_aircraft=int(rng.choice([3, 50, 800, 5000, 30000])),
          overlap_permille=int(rng.choice([0, 10, 200, 700])), flip_permille=int(rng.choice([0, 20, 200])),
          noise_fs=float(rng.choice([0.005, 0.02, 0.06])), ac_per_sec=int(rng.choice([0, 0, 500, 4000])))
nfix = int(rng.integers(0, 3))
mode_ac = int(kw["ac_per_sec"] > 0 and rng.integers(0, 2))
gpu_resolve = int(rng.integers(0, 2))
thr = int(rng.choice([58, 58, 58, 40, 75, 400]))
threads = int(rng.choice([1, 4, 16]))
cfg = pkg.siggen.make_cfg(seed=case, fmt=fmt, **kw)
iq = pkg.siggen.generate(cfg, n)
rng.integers(0, 2); rng.integers(0, 8)
arena = int(rng.choice([0, 0, 0, 50, 200, 1000])) if fmt_name != "sc16q11" else 0
growth = 0
mode_ac = int(ov.get("mode_ac", mode_ac)); arena = int(ov.get("arena", 50)); growth = int(ov.get("growth", 0)); nfix = int(ov.get("nfix", nfix))
batch = int(ov.get("batch", batch // 131072)) * 131072; gpu_resolve = int(ov.get("gpu_resolve", gpu_resolve))
os.environ.update(MSD_GPU_RESOLVE=str(gpu_resolve), MSD_RESOLVE_THREADS=str(threads), MSD_ARENA_SCALE_PERMILLE=str(arena), MSD_ARENA_GROWTH=str(growth))
d = torch.from_numpy(iq).to("cuda:0")
dem = pkg.Demodulator(fmt=fmt, preamble_threshold=thr, nfix_crc=nfix, mode_ac=mode_ac, max_batch_samples=batch, message_capacity=1 << 19)
got = pkg.replay_device(dem, d.data_ptr(), n, batch)
want, wstats = orc.Oracle(ofmt, thr, nfix, mode_ac).replay(iq, cap=1 << 19)
t = dem.timing()
print(f"case {case} {fmt_name} n={n} batch={batch // 131072} nfix={nfix} ac={mode_ac} gpu_resolve={gpu_resolve} thr={thr} arena={arena} growth={growth}: "
      f"got {len(got)} want {len(want)} reruns {t['reruns']} fallback {t['resolve_fallback']}")
i = 0
while i < min(len(got), len(want)) and got[i]["timestampMsg"] == want[i]["timestampMsg"] and got[i]["msgtype"] == want[i]["msgtype"]:
    i += 1
print("first difference at message", i, "of", len(got), len(want))
for name, arr in (("gpu", got), ("oracle", want)):
    for m in arr[max(0, i - 2): i + 3]:
        ts = int(m["timestampMsg"])
        print("  ", name, "buffer", ts // (131072 * 5), "j~", (ts % (131072 * 5)) // 5, "df", m["msgtype"], "addr %06x" % m["addr"], "score", m["score"])
key = lambda a: set(zip(a["timestampMsg"].tolist(), a["msgtype"].tolist(), [bytes(m).hex() for m in a["msg"]]))
kg, kw_ = key(got), key(want)
print("only in gpu:", sorted(kg - kw_)[:10])
print("only in oracle:", sorted(kw_ - kg)[:10])
ga, wa = got[got["msgtype"] == 32], want[want["msgtype"] == 32]
print("mode a/c replies: gpu", len(ga), "oracle", len(wa), "same order among themselves:", np.array_equal(ga["timestampMsg"], wa["timestampMsg"]))
gs, ws = got[got["msgtype"] != 32], want[want["msgtype"] != 32]
print("mode s: same list:", len(gs) == len(ws) and np.array_equal(gs["timestampMsg"], ws["timestampMsg"]))
gst, wst = dem.stats(), wstats
print({k: (gst[k], wst[k]) for k in ("demod_preambles", "demod_accepted", "demod_modeac") })
