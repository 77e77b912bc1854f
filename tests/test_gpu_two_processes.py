"""Several processes, one context each, ONE GPU, at the same time: the path the multi-GPU bench takes per rank
(one process per GPU) with the additional stress of sharing the device -- every process must still
deliver exactly its own capture's messages.  Nothing is shared between the contexts (no globals in libmodes_hip.so),
the device only time-slices their kernels."""
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_two_processes_two_contexts_one_gpu(torch_cuda, tmp_path):
    start = str(tmp_path / "go")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "two_proc_worker.py"), str(seed), "40", start],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, cwd=ROOT)
             for seed in (20901, 20902)]
    t0 = time.time()
    while not all(os.path.exists(f"{start}.{seed}.ready") for seed in (20901, 20902)) and time.time() - t0 < 240:
        if any(p.poll() is not None for p in procs):
            break
        time.sleep(0.05)
    open(start, "w").close()
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for p, out in zip(procs, outs):
        assert p.returncode == 0 and "identical to the oracle" in out, out[-3000:]
