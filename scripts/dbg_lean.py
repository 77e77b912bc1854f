import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import __graft_entry__ as g
pkg = g.load_package(); oracle = g.load_oracle()
fmt, ofmt, fix, ac = pkg.FMT_UC8, oracle.FMT_UC8, 0, 0
n, batch = 36 * pkg.CHUNK + 4321, 8 * pkg.CHUNK
dem = pkg.Demodulator(fmt=fmt, nfix_crc=fix, mode_ac=ac, max_batch_samples=batch, message_capacity=1 << 18)
caps, dev = [], []
for seed in (4711, 4712):
    iq = pkg.siggen.generate(pkg.siggen.make_cfg(seed=seed, fmt=fmt), n)
    caps.append(iq); dev.append(torch.from_numpy(iq).to("cuda:0"))
got, inflight = {0: [], 1: []}, []
def collect_one():
    cap, last = inflight.pop(0)
    got[cap].append(dem.collect())
for cap in (0, 1):
    if cap == 1:
        dem.restart()
    off = 0
    while off < n:
        if len(inflight) == pkg.capi.PIPELINE_DEPTH:
            collect_one()
        m = min(batch, n - off)
        dem.launch_device(dev[cap].data_ptr() + off * 2, m, off + m >= n)
        inflight.append((cap, off + m >= n)); off += m
while inflight:
    collect_one()
for cap in (0, 1):
    want, wstats = oracle.Oracle(ofmt, 58, fix, ac).replay(caps[cap], cap=1 << 18)
    g_ = np.concatenate(got[cap])
    print("capture", cap, len(g_), len(want))
    for f in ("timestampMsg", "addr", "score", "crc", "msgtype", "signalLevel"):
        if len(g_) == len(want):
            bad = np.nonzero(g_[f] != want[f])[0]
            print(" ", f, "diffs", len(bad), bad[:10], [ (int(g_["timestampMsg"][i]) // (5*131072), ) for i in bad[:10]])
            for i in bad[:3]:
                print("    got", g_[f][i], "want", want[f][i], "ts", g_["timestampMsg"][i], want["timestampMsg"][i])
