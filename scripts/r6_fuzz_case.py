"""Reproduce one case of tests/fuzz_parity.py (round 6's draw order) with overrides and show where the lists part:
python scripts/r6_fuzz_case.py <case> [KEY=VALUE ...]   (dc=0, fields=0, arena=0, growth=1, gpu_resolve=0, batch=8, nfix=1, layout=inline)"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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
with_fields = int(rng.integers(0, 2))
dc = bool(rng.integers(0, 8) < 2)
q11 = int(rng.choice([0, 0, 7, 8, 11])) if fmt_name == "sc16q11" and not dc else 0
arena = int(rng.choice([0, 0, 0, 50, 200, 1000]))
growth = int(rng.integers(0, 4) != 0)
dc = bool(int(ov.get("dc", dc))); with_fields = int(ov.get("fields", with_fields)); arena = int(ov.get("arena", arena)); growth = int(ov.get("growth", growth))
nfix = int(ov.get("nfix", nfix)); batch = int(ov.get("batch", batch // 131072)) * 131072; gpu_resolve = int(ov.get("gpu_resolve", gpu_resolve))
mode_ac = int(ov.get("mode_ac", mode_ac)); nmax = int(ov.get("buffers", 0))
if nmax:
    n = min(n, nmax * 131072); iq = iq[: n * (2 if fmt_name == "uc8" else 4)]
os.environ.update(MSD_GPU_RESOLVE=str(gpu_resolve), MSD_RESOLVE_THREADS=str(threads), MSD_ARENA_SCALE_PERMILLE=str(arena), MSD_ARENA_GROWTH=str(growth))
for k, v in ov.items():
    if k.startswith("MSD_"):
        os.environ[k] = v
print("case", case, fmt_name, "n", n, "batch", batch // 131072, "nfix", nfix, "ac", mode_ac, "gpu_resolve", gpu_resolve, "fields", with_fields, "dc", int(dc), "thr", thr, "arena", arena, "growth", growth, kw)
d = torch.from_numpy(iq).to("cuda:0")
dem = pkg.Demodulator(fmt=fmt, preamble_threshold=thr, nfix_crc=nfix, mode_ac=mode_ac, max_batch_samples=batch, message_capacity=1 << 19,
                      decode_fields=bool(with_fields), dc_filter=dc, **({"sc16q11_table_bits": q11} if q11 else {}))
if with_fields:
    parts, bps = [], dem.bytes_per_sample
    for off in list(range(0, n, batch)) or [0]:
        m = min(batch, n - off)
        dem.launch_device(d.data_ptr() + off * bps, m, off + m >= n)
        parts.append(dem.collect_fields()[0])
    got = np.concatenate(parts)
else:
    got = pkg.replay_device(dem, d.data_ptr(), n, batch)
want, wstats = orc.Oracle(ofmt, thr, nfix, mode_ac, dc_filter=dc, sc16q11_table_bits=q11).replay(iq, cap=1 << 19)
print("messages", len(got), len(want), "timing", {k: dem.timing()[k] for k in ("resolve_passes", "resolve_fallback", "reruns")})
nn = min(len(got), len(want))
bad = [i for i in range(nn) if got[i].tobytes() != want[i].tobytes()]
print("differing records:", len(bad), "first", bad[:5])
for i in bad[:3]:
    print(" got ", {k: got[i][k] for k in ("timestampMsg", "addr", "msgtype", "score", "correctedbits", "crc")})
    print(" want", {k: want[i][k] for k in ("timestampMsg", "addr", "msgtype", "score", "correctedbits", "crc")})
    print("  buffer", int(want[i]["timestampMsg"]) // (131072 * 5), "of batch", (int(want[i]["timestampMsg"]) // (131072 * 5)) // (batch // 131072))
gs = dem.stats()
for k in ("demod_preambles", "demod_rejected_bad", "demod_rejected_unknown_icao", "demod_accepted"):
    if gs[k] != wstats[k]:
        print("counter", k, gs[k], wstats[k])
if bad and "neighbours" in ov:
    a = int(want[bad[0]]["addr"])
    t0 = int(want[bad[0]]["timestampMsg"])
    print("every message of address %06x within 3 buffers of the first difference (want | got score):" % a)
    for i in range(nn):
        if int(want[i]["addr"]) == a and abs(int(want[i]["timestampMsg"]) - t0) < 3 * 131072 * 5:
            print("  #%d ts %d (buffer %d, sample %d) DF%d corrected %d crc %06x score want %d got %d" % (i, int(want[i]["timestampMsg"]), int(want[i]["timestampMsg"]) // 655360,
                  (int(want[i]["timestampMsg"]) % 655360) // 5, int(want[i]["msgtype"]), int(want[i]["correctedbits"]), int(want[i]["crc"]), int(want[i]["score"]), int(got[i]["score"])))
