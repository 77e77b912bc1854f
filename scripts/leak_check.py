import sys, importlib, numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
pkg = importlib.import_module("readsb-protobuf_amd")
iq = pkg.siggen.generate(pkg.siggen.make_cfg(seed=9, msgs_per_sec=5000), 6 * 131072 + 5)
d = torch.from_numpy(iq).to("cuda:0")
free0 = torch.cuda.mem_get_info()[0]
tot = 0
import os
N = int(os.environ.get("LEAK_N", "40"))
for k in range(N):
    dem = pkg.Demodulator(nfix_crc=k % 3, mode_ac=k % 2, dc_filter=(k % 4 == 3), decode_fields=(k % 5 == 0),
                          max_batch_samples=4 * 131072, message_capacity=1 << 14)
    dem.launch_device(d.data_ptr(), 4 * 131072, last=False)
    dem.launch_device(d.data_ptr() + 8 * 131072, 2 * 131072 + 5, last=True)
    a = dem.collect_fields()[0] if k % 5 == 0 else dem.collect()
    b = dem.collect_fields()[0] if k % 5 == 0 else dem.collect()
    tot += len(a) + len(b)
    dem.close() if hasattr(dem, "close") else None
    del dem
torch.cuda.synchronize()
free1 = torch.cuda.mem_get_info()[0]
print("messages", tot, "device memory delta MiB", (free0 - free1) / 2**20)
