"""The N>1 path on CPU: two ranks over gloo, one independent capture per rank, no data-path
collective -- only the benchmark's MAX(time) / SUM(counts) bookkeeping."""
import os
import socket

import numpy as np
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, nsamples, out):
    import torch.distributed as dist
    import __graft_entry__ as g
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg, O = g.load_package(), g.load_oracle()
    seed = pkg.sharding.capture_seed(rank)
    iq = pkg.siggen.generate(pkg.siggen.make_cfg(seed=seed), nsamples, nthreads=2)
    # stand-in for this rank's GPU: the oracle (tests may use it); every rank owns its filter/clock
    msgs, _ = O.Oracle(O.FMT_UC8, 58, 0, 0).replay(iq)
    dist.barrier()
    elapsed, total_msgs, total_samples = pkg.sharding.reduce_job(0.5 + rank, len(msgs), nsamples)
    out[rank] = (seed, len(msgs), elapsed, total_msgs, total_samples)
    dist.destroy_process_group()


def test_two_ranks_independent_captures():
    world, n = 2, 2 * 131072
    port = _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, port, n, out), nprocs=world, join=True)
        res = dict(out)
    assert [res[r][0] for r in range(world)] == [10901, 10902]           # distinct captures
    assert res[0][1] != res[1][1] or res[0][1] > 0
    for r in range(world):
        assert res[r][2] == 1.5                                          # MAX over ranks
        assert res[r][3] == res[0][1] + res[1][1]                        # SUM of messages
        assert res[r][4] == world * n                                    # weak scaling: samples add up


def test_reduce_is_identity_without_process_group(pkg):
    assert pkg.sharding.reduce_job(1.25, 7, 100) == (1.25, 7, 100)


def _one_rank_group(port, q):
    import torch.distributed as dist
    import __graft_entry__ as graft
    pkg = graft.load_package()
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
    q.put(pkg.sharding.reduce_job(1.25, 7, 100))
    dist.destroy_process_group()


def test_a_group_of_one_rank_goes_through_the_collectives(pkg):
    """bench.py's MSD_BENCH_FORCE_DIST leg: with a process group of one rank the reductions still run (and give the rank's own numbers)."""
    import multiprocessing as mp
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_one_rank_group, args=(port, q))
    p.start()
    got = q.get(timeout=120)
    p.join(60)
    assert got == (1.25, 7, 100) and p.exitcode == 0
