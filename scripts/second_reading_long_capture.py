"""Oracle against the second reading of the reference (tests/indep_demod.py) on 158 s of signal: 2900 buffers, 30 000
aircraft (the ICAO filter's tables fill up), the filter flips at 0, 60 and 120 s.  CPU only, about two minutes.
Output kept as profiles/r04_second_reading_long_capture.txt."""
import sys, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np
import __graft_entry__ as g
pkg=g.load_package(); orc=g.load_oracle()
import indep_demod
from test_indep_demod import assert_second_reading_agrees
nbuf = 2900
n = nbuf*131072 + 4096
cfg = pkg.siggen.make_cfg(seed=4242, msgs_per_sec=600, n_aircraft=30000, noise_fs=0.005, flip_permille=20, overlap_permille=10)
t=time.time(); iq = pkg.siggen.generate(cfg, n); print("generated", n, "samples = %.1f s of signal in %.1fs" % (n/2.4e6, time.time()-t), flush=True)
t=time.time(); want, wstats = orc.Oracle(orc.FMT_UC8, 58, 1, 0).replay(iq, cap=1<<20); print("oracle", len(want), "msgs %.1fs" % (time.time()-t), flush=True)
t=time.time(); r = indep_demod.Receiver("uc8", 58, 1, False); s, ss = r.replay(iq); print("second reading", len(s), "msgs %.1fs" % (time.time()-t), flush=True)
assert_second_reading_agrees(s, ss, want, wstats)
# how much the flips mattered: messages of address-parity formats whose address had been announced more than 60 s earlier
last = {}
old = 0
for m in s:
    a, ts = m["addr"], m["timestampMsg"]
    if m["msgtype"] in (17, 11) and m["correctedbits"] == 0:
        last[a] = ts
    elif m["msgtype"] in (0, 4, 5, 16, 20, 21) and a in last and ts - last[a] > 60 * 12_000_000:
        old += 1
print("agree: %d messages, counters %s" % (len(s), {k: ss[k] for k in ("demod_preambles", "demod_rejected_unknown_icao", "demod_accepted")}))
print("filter flips at 0, 60, 120 s of signal; accepted address-parity messages whose last clean squitter was > 60 s earlier:", old)
