"""The little torch.distributed plumbing bench.py needs for N > 1: one process
per GPU, no data-path collective (the plan of one problem is one dependent
chain of greedy steps -- DESIGN.md "Multi-GPU": replicas only); ranks only
agree on the slowest rank's time."""
import torch
import torch.distributed as dist


def max_over_ranks(seconds):
    """Wall time of the slowest rank (MAX all-reduce; RCCL on GPUs, gloo on CPU)."""
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def replica_seed(rank):
    """Every rank plans an instance of the same shape; kept as a hook for
    per-rank variation of the synthetic input."""
    return 1000003 * (rank + 1)
