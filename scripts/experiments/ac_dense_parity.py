"""Dense Mode A/C (30 000 to 400 000 replies per second, mostly overlapping) through the HIP path and the oracle, both
resolve stages, UC8 and SC16: python scripts/experiments/ac_dense_parity.py (GPU box)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import __graft_entry__ as g
from test_gpu_parity import assert_same
pkg=g.load_package(); O=g.load_oracle()
for gr in ("1","0"):
    os.environ["MSD_GPU_RESOLVE"]=gr
    for ac in (30000, 100000, 400000):
        for fmt,of in ((pkg.FMT_UC8,O.FMT_UC8),(pkg.FMT_SC16,O.FMT_SC16)):
            n=20*131072+999
            iq=pkg.siggen.generate(pkg.siggen.make_cfg(seed=ac, fmt=fmt, msgs_per_sec=500, ac_per_sec=ac, noise_fs=0.01), n)
            dem=pkg.Demodulator(fmt=fmt, nfix_crc=1, mode_ac=1, max_batch_samples=8*131072, message_capacity=1<<20)
            got=pkg.replay_device(dem, torch.from_numpy(iq).to("cuda:0").data_ptr(), n, 8*131072)
            want,ws=O.Oracle(of,58,1,1).replay(iq,cap=1<<20)
            assert_same(got, dem.stats(), want, ws)
            print("gpu_resolve",gr,"ac/s",ac,"fmt",fmt,"msgs",len(want),"modeac",ws["demod_modeac"],dem.timing()["reruns"],dem.timing()["resolve_fallback"])
