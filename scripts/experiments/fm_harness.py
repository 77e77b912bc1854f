"""Round-4 debugging aid (GPU box): the float-sum kernels alone on random SC16 samples, against numpy's sequential
float32 accumulation.  Usage: python scripts/experiments/fm_harness.py [nbuffers] [amplitude]"""
import ctypes, os, sys
import numpy as np
import torch

root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = ctypes.CDLL(os.environ.get("MSD_LIBMODES_HIP") or os.path.join(root, "readsb-protobuf_amd", "csrc", "libmodes_hip.so"))
lib.msd_fm_work_bytes.restype = ctypes.c_size_t
lib.msd_fm_work_bytes.argtypes = [ctypes.c_uint32]
lib.msd_launch_float_means.restype = ctypes.c_int
lib.msd_launch_float_means.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32,
                                       ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 512
amp = float(sys.argv[2]) if len(sys.argv) > 2 else 600.0
tail = int(sys.argv[3]) if len(sys.argv) > 3 else 0          # samples cut off the last buffer
L = 131072
rng = np.random.default_rng(5)
n = nb * L - tail
iq = np.clip(rng.normal(0, amp, size=(n, 2)), -32768, 32767).astype(np.int16)
d_iq = torch.from_numpy(iq).cuda()
d_out = torch.zeros(2 * nb, dtype=torch.float32, device="cuda")
work = torch.zeros(lib.msd_fm_work_bytes(nb), dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
rc = lib.msd_launch_float_means(1, d_iq.data_ptr(), n, L, nb, d_out.data_ptr(), None, work.data_ptr(), 0, None)
torch.cuda.synchronize()
print("rc", rc)
got = d_out.cpu().numpy().reshape(nb, 2)
bad = 0
for b in list(range(min(nb, 6))) + [nb - 1]:
    x = iq[b * L:(b + 1) * L].astype(np.float32) * np.float32(1 / 32768.0)
    sq = x[:, 0] * x[:, 0] + x[:, 1] * x[:, 1]
    sq = np.minimum(sq, np.float32(1.0)).astype(np.float32)
    m = np.sqrt(sq).astype(np.float32)
    want = (np.add.accumulate(m, dtype=np.float32)[-1], np.add.accumulate(sq, dtype=np.float32)[-1])
    ok = want[0] == got[b, 0] and want[1] == got[b, 1]
    bad += not ok
    print(b, want, tuple(got[b]), "ok" if ok else "DIFFERENT")
print("different:", bad)
