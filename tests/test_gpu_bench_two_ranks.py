"""bench.py's N>1 branch on the hardware there is: two ranks under torch.distributed.run, both on cuda:0
(MSD_BENCH_DEVICE_OVERRIDE=0), bookkeeping over gloo (RCCL refuses two ranks on one device).  Everything the 8-GPU
run will execute except the RCCL transport runs here: rendezvous, one capture and one context per rank, the barrier
around the timed region, MAX(time) / SUM(counts), the per-rank gather, the CPU slices."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.gpu
def test_bench_two_ranks_one_gpu(torch_cuda):
    n = 1 << 27
    env = dict(os.environ, MSD_BENCH_DEVICE_OVERRIDE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--settle-seconds", "1", "--samples", str(n), "--dist-backend", "gloo"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]   # rank 0 prints, nobody else
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak" and d["dist_backend"] == "gloo"
    ranks = sorted(d["ranks"], key=lambda r: r["rank"])
    assert [r["rank"] for r in ranks] == [0, 1]
    assert [r["seed"] for r in ranks] == [10901, 10902]           # one capture per rank
    assert ranks[0]["messages"] != ranks[1]["messages"] and min(r["messages"] for r in ranks) > 1000
    assert d["messages_per_step"] == ranks[0]["messages"] + ranks[1]["messages"]   # SUM over ranks
    pr = d["per_rank_ms_per_step"]
    assert abs(pr["max"] - max(r["ms_per_step"] for r in ranks)) < 1e-3 and pr["min"] <= pr["max"]
    assert abs(d["ms_per_step"] - pr["max"]) / pr["max"] < 0.05                    # MAX over ranks is what is reported
    assert abs(d["value"] - 2 * n / (d["ms_per_step"] * 1e-3) / 1e6) / d["value"] < 1e-3   # whole-job samples / MAX time
    cpus = [r["cpus"] for r in ranks]
    if all(cpus):                                                 # sysfs told: the ranks' CPU slices are disjoint
        spans = [tuple(int(x) for x in c.split("-")) for c in cpus]
        assert spans[0][1] < spans[1][0] or spans[1][1] < spans[0][0], cpus
    # round 6: an N > 1 line carries its own evidence -- every rank's list against the oracle on its own capture ...
    per = d["message_set_diff_vs_oracle_per_rank"]
    assert [x["rank"] for x in per] == [0, 1] and [x["seed"] for x in per] == [10901, 10902]
    assert all(x["diff"] == 0 and x["messages"] > 1000 and x["buffers"] == 256 for x in per), per
    assert per[0]["messages"] != per[1]["messages"]               # two captures, two lists
    assert d["message_set_diff_vs_oracle"] == 0 and d["messages_checked"] == per[0]["messages"] + per[1]["messages"]
    # ... and the CPU baseline of SURVEY.md 8(d) form (b): N two-thread streams on 2 N cores
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 4 and len(cb["streams"]) == 2 and cb["value"] > 0
    for st in cb["streams"]:
        assert st["cores"] == 2 and st["messages"] > 1000 and st["reader_thread_cpu_s"] > 0 and st["demod_thread_cpu_s"] > 0
    pinned_cpus = [c for st in cb["streams"] for c in st.get("cpus", [])]
    assert len(pinned_cpus) == len(set(pinned_cpus))              # a CPU of its own for every thread (when there are enough)
    assert cb["thread_cpu_s"]["demodulators"] >= cb["thread_cpu_s"]["readers"]     # the demodulator is the busy thread
    assert "also" not in d                                        # the other workloads: rank 0 at N=1 only
