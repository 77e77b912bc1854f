"""The RCCL leg of bench.py on the hardware there is: one rank under torch.distributed.run, forced through the N > 1
branch (MSD_BENCH_FORCE_DIST=1) with --dist-backend nccl -- init_process_group("nccl", device_id=...), the barriers
around the timed region, reduce_job's MAX / SUM all-reduces on DEVICE tensors, all_gather_object over RCCL,
destroy_process_group.  RCCL accepts a communicator of one rank, so the first 8-GPU run is not the first run of that
code (tests/test_gpu_bench_two_ranks.py covers two ranks over gloo)."""
import json
import os
import subprocess
import sys

import pytest

from test_gpu_bench_two_ranks import free_port

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_one_rank_through_rccl(torch_cuda):
    n = 1 << 27
    env = dict(os.environ, MSD_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
           "--settle-seconds", "1", "--samples", str(n), "--dist-backend", "nccl", "--no-cpu-baseline", "--no-check", "--no-also"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["dist_backend"] == "nccl"      # the branch was taken, over RCCL
    ranks = d["ranks"]
    assert len(ranks) == 1 and ranks[0]["rank"] == 0 and ranks[0]["seed"] == 10901    # all_gather_object came back
    assert d["messages_per_step"] == ranks[0]["messages"] > 1000                       # SUM over one rank
    assert abs(d["ms_per_step"] - ranks[0]["ms_per_step"]) / d["ms_per_step"] < 0.05   # MAX over one rank
    assert abs(d["value"] - n / (d["ms_per_step"] * 1e-3) / 1e6) / d["value"] < 1e-3
