import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import __graft_entry__ as g
pkg = g.load_package()
n = 48 * 131072
for fmt in (pkg.FMT_UC8, pkg.FMT_SC16):
    iq = pkg.siggen.generate(pkg.siggen.make_cfg(seed=5, fmt=fmt), n)
    d = torch.from_numpy(iq).to("cuda:0")
    dem = pkg.Demodulator(fmt=fmt, nfix_crc=1, dc_filter=True, max_batch_samples=16 * 131072, message_capacity=1 << 18)
    pkg.replay_device(dem, d.data_ptr(), n, 16 * 131072)
    dem.reset(); torch.cuda.synchronize()
    t0 = time.perf_counter(); got = pkg.replay_device(dem, d.data_ptr(), n, 16 * 131072); dt = time.perf_counter() - t0
    print("dcfilter fmt %d: %.1f MS/s (%d msgs)" % (fmt, n / dt / 1e6, len(got)))
