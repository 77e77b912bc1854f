"""PCIe-inclusive rates (for DESIGN.md 5.1; never reported as bench `value`)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g
pkg = g.load_package()
n, batch = 1 << 29, 1 << 26
iq = pkg.siggen.generate(pkg.siggen.make_cfg(seed=10901), n)
# (a) raw H2D bandwidth, pinned, 128 MiB pieces
pin = torch.from_numpy(iq[: batch * 2]).pin_memory()
dst = torch.empty(batch * 2, dtype=torch.uint8, device="cuda:0")
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(10): dst.copy_(pin, non_blocking=True)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
print("pinned H2D %.1f GB/s -> UC8 bound %.1f GSamples/s" % (batch * 2 / dt / 1e9, batch / dt / 1e9))
# (b) msd_submit_host: pageable host memory, synchronous per batch
dem = pkg.Demodulator(fmt=pkg.FMT_UC8, nfix_crc=0, max_batch_samples=batch, message_capacity=1 << 21)
for rep in range(2):
    dem.reset(); t = time.perf_counter(); off = 0; nm = 0
    while off < n:
        m = min(batch, n - off)
        nm += len(dem.submit_host(iq[off * 2:(off + m) * 2], m, last=off + m >= n)); off += m
    dt = time.perf_counter() - t
print("msd_submit_host (pageable, synchronous): %.2f GSamples/s, %d messages" % (n / dt / 1e9, nm))
