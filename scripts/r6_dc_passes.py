"""round 6 (GPU box): how many passes the parallel-in-time DC block needs (msd_dc_filter_status) by content and batch size."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as G
from test_gpu_dc_parallel import content
pkg = G.load_package()
CH = 131072
for fmt, f in (("uc8", pkg.FMT_UC8), ("sc16", pkg.FMT_SC16)):
    bps = 2 if fmt == "uc8" else 4
    for kind in ("noise", "offset", "strong", "constant", "alternating", "random", "step"):
        line = []
        for n in (CH, 8 * CH, 128 * CH):
            iq = content(kind, fmt, 2 * n, seed=5)
            dem = pkg.Demodulator(fmt=f, max_batch_samples=128 * CH, dc_filter=True, flags=0)
            for k in range(2):   # the second call starts from the state the first one left
                dem.convert(iq[k * n * bps:(k + 1) * n * bps], n)
                ex, passes, guessed, blocks = dem.dc_filter_status()
                line.append("%s%d/%d" % ("" if ex else "FELL THROUGH ", passes, blocks))
        print("%-5s %-12s passes/blocks at %d, %d, %d samples (first call, second call): %s" % (fmt, kind, CH, 8 * CH, 128 * CH, "  ".join(line)), flush=True)
