"""Host-side timeline of one replay step (where do the milliseconds of a batch go?)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g
pkg = g.load_package()
n, batch = 1 << 29, 1 << 26
iq = pkg.siggen.generate(pkg.siggen.make_cfg(seed=10901), n)
d = torch.from_numpy(iq).to("cuda:0"); torch.cuda.synchronize()
dem = pkg.Demodulator(fmt=pkg.FMT_UC8, nfix_crc=0, max_batch_samples=batch, message_capacity=1 << 21,
                      stream=torch.cuda.current_stream().cuda_stream)
for rep in range(2):
    dem.reset(); ev = []; off = 0; infl = 0; t0 = time.perf_counter()
    while True:
        m = min(batch, n - off); last = off + m >= n
        if infl == 2:
            a = time.perf_counter(); dem.collect(); ev.append(("collect", time.perf_counter() - a)); infl -= 1
        a = time.perf_counter(); dem.launch_device(d.data_ptr() + off * 2, m, last); ev.append(("launch", time.perf_counter() - a)); infl += 1
        off += m
        if last: break
    while infl:
        a = time.perf_counter(); dem.collect(); ev.append(("collect", time.perf_counter() - a)); infl -= 1
    tot = time.perf_counter() - t0
print("total ms", tot * 1e3)
for k, v in ev: print(k, round(v * 1e3, 3))
