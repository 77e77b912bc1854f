import os, sys, time
os.environ["MSD_RESOLVE_TRACE"] = "1"
sys.path.insert(0, "/root/repo")
import numpy as np
import __graft_entry__ as G
pkg = G.load_package()
CH = 131072; OV = 326; NB = 12
iq = pkg.siggen.generate(pkg.siggen.make_cfg(seed=10901), NB * CH)
conv = pkg.Demodulator(max_batch_samples=CH)
mags = []
for k in range(NB):
    m, ml, mp = conv.convert(iq[2 * k * CH:2 * (k + 1) * CH], CH)
    mags.append((m, ml, mp))
dem = pkg.Demodulator(max_batch_samples=NB * CH, flags=pkg.capi.CFG_TRACE)
bufs = []
prev = np.zeros(OV, np.uint16)
for k, (m, ml, mp) in enumerate(mags):
    data = np.concatenate([prev, m]); prev = m[-OV:]
    bufs.append((data, OV + CH, OV, k * CH * 5, k * 54, ml, mp))
for rep in range(4):
    t0 = time.perf_counter()
    out = pkg.capi.demodulate_magbufs(dem, bufs)
    print("call %d: %.1f us for %d buffers, %d messages" % (rep, (time.perf_counter() - t0) * 1e6, NB, len(out)), file=sys.stderr)
