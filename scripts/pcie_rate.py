"""PCIe-inclusive rates (for DESIGN.md 5.1; never reported as bench `value`)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g
pkg = g.load_package()
n, batch = 1 << 29, 1 << 26
iq = pkg.siggen.generate(pkg.siggen.make_cfg(seed=10901), n)
# (b) msd_submit_host: pageable host memory, synchronous per batch
dem = pkg.Demodulator(fmt=pkg.FMT_UC8, nfix_crc=0, max_batch_samples=batch, message_capacity=1 << 21)
for rep in range(2):
    dem.reset(); t = time.perf_counter(); off = 0; nm = 0
    while off < n:
        m = min(batch, n - off)
        nm += len(dem.submit_host(iq[off * 2:(off + m) * 2], m, last=off + m >= n)); off += m
    dt = time.perf_counter() - t
print("msd_submit_host (pageable, synchronous): %.2f GSamples/s, %d messages" % (n / dt / 1e9, nm))

# (c) msd_launch_host / msd_collect from page-locked memory that already holds the samples (an SDR driver or
#     a reader thread fills such buffers directly): uploads overlap the kernels, three batches in flight
nb = (n + batch - 1) // batch
bufs = [dem.host_buffer(batch * 2) for _ in range(nb)]
for i, b in enumerate(bufs):
    m = min(batch, n - i * batch)
    b[: m * 2] = iq[i * batch * 2:(i * batch + m) * 2]
for rep in range(3):
    dem.reset(); t = time.perf_counter(); nm = 0; inflight = 0
    for i, b in enumerate(bufs):
        m = min(batch, n - i * batch)
        if inflight == pkg.capi.PIPELINE_DEPTH:
            nm += len(dem.collect(copy=False)); inflight -= 1
        dem.launch_host(b, m, last=i == nb - 1); inflight += 1
    while inflight:
        nm += len(dem.collect(copy=False)); inflight -= 1
    dt = time.perf_counter() - t
print("msd_launch_host (page-locked, 3 in flight): %.2f GSamples/s = %.1f GB/s over the link, %d messages"
      % (n / dt / 1e9, n * 2 / dt / 1e9, nm))
