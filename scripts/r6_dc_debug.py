"""round 6 (GPU box): the exact prefix of the parallel-in-time DC block pass by pass (msd_launch_dcfilter_parallel called directly with
1, 2, ... passes queued), to compare with the numpy replica of the walk."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as G
from test_gpu_dc_parallel import content
pkg = G.load_package()
L = C.CDLL(pkg.capi.LIB_PATH)
L.msd_dcp_work_bytes.restype = C.c_size_t; L.msd_dcp_work_bytes.argtypes = [C.c_uint64, C.c_uint32]
L.msd_launch_dcfilter_parallel.restype = C.c_int
L.msd_launch_dcfilter_parallel.argtypes = [C.c_int, C.c_void_p, C.c_uint64, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_void_p]
n = 1 << 20
blk = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
kind = sys.argv[2] if len(sys.argv) > 2 else "noise"
hist = {}
for fmt, f, bps in (("uc8", pkg.FMT_UC8, 2), ("sc16", pkg.FMT_SC16, 4)):
    iq = content(kind, fmt, 2 * n, seed=5)[: n * bps]
    d_iq = torch.from_numpy(iq.copy()).cuda()
    wb = L.msd_dcp_work_bytes(n, blk)
    work = torch.zeros(wb, dtype=torch.uint8, device="cuda")
    mag = torch.zeros(n, dtype=torch.int16, device="cuda"); sq = torch.zeros(n, dtype=torch.float32, device="cuda")
    b = np.float32(np.exp(-2 * np.pi / 2.4e6)); a = np.float32(1.0 - float(b))
    for k in range(1, 21):
        state = torch.zeros(2, dtype=torch.float32, device="cuda")
        rc = L.msd_launch_dcfilter_parallel(f, d_iq.data_ptr(), n, float(a), float(b), state.data_ptr(), mag.data_ptr(), sq.data_ptr(), work.data_ptr(), blk, k, 0, None)
        torch.cuda.synchronize()
        ctl = work[:64].cpu().numpy().view(np.uint32)
        print(fmt, "passes queued", k, "rc", rc, "done", ctl[0], "done_ch", ctl[1:3], "frontier", ctl[3:5], "walks", ctl[9:11], "guessed (all passes)", ctl[11], flush=True)
        nb = (n + blk - 1) // blk
        Sg = work[256:256 + 8 * nb].cpu().numpy().view(np.uint32).reshape(2, nb).copy()
        hist.setdefault(fmt, []).append(Sg)
        e0 = 256 + ((nb * 24 + 255) & ~255)
        if fmt == "uc8" and k in (5, 6, 7):
            hist.setdefault(fmt + "_E", []).append(work[e0:e0 + 2 * nb * 64 * 16].cpu().numpy().view(np.float32).reshape(2, nb, 64, 4).copy())
        if ctl[0]:
            break
np.savez(os.path.join(ROOT, "gpurun_out", "dc", "S_gpu_%s_%d.npz" % (kind, blk)), **{k: np.stack(v) for k, v in hist.items()})
