"""How the path shards across GPUs: one independent capture (its own ICAO filter and clock, like a
separate readsb process) per rank, no data-path collective.  The only communication is the
benchmark's own bookkeeping: a barrier, MAX over ranks of the elapsed time, SUM of the counts."""
import torch
import torch.distributed as dist

BASE_SEED = 10901  # captures 10901..10908 for the 8-GPU configuration


def capture_seed(rank: int) -> int:
    return BASE_SEED + rank


def reduce_job(elapsed_s: float, messages: int, samples: int, device=None):
    """(max elapsed, total messages, total samples) over all ranks; identity without a process group."""
    if not (dist.is_available() and dist.is_initialized()):
        return elapsed_s, messages, samples
    # (a group of one rank still goes through the collectives: that is how the RCCL leg is exercised on a one-GPU box)
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    c = torch.tensor([messages, samples], dtype=torch.int64, device=device)
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    return float(t.item()), int(c[0].item()), int(c[1].item())
