"""torch.distributed plumbing for N > 1 (one process per GPU):

  * shard_plan_rccl(planner, dist): the ranks join the library's own RCCL communicator, after which
    one plan's region chains are sharded over them (include/blance_hip.h "one plan on several GPUs");
  * gloo_allreduce(dist): the caller-provided collective for GPU-less tests (the SIMT emulator's
    "device" memory is host memory, summed through a gloo all-reduce);
  * max_over_ranks(seconds): the slowest rank's wall time for bench.py.
"""
import ctypes

import numpy as np
import torch
import torch.distributed as dist


def max_over_ranks(seconds):
    """Wall time of the slowest rank (MAX all-reduce; RCCL on GPUs, gloo on CPU)."""
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def shard_plan_rccl(planner, d=dist):
    """Plans made through `planner` from now on run sharded over the ranks of `d` (RCCL inside the
    library).  Every rank must upload the same problem and make the same calls."""
    return planner.comm_init_rccl(d)


def gloo_allreduce(d=dist):
    """allreduce(address, count) over host memory, for hip.Planner.comm_set_callback()."""
    def allreduce(ptr, count):
        arr = np.ctypeslib.as_array((ctypes.c_int32 * count).from_address(ptr))
        t = torch.from_numpy(arr)
        d.all_reduce(t, op=d.ReduceOp.SUM)         # in place: t shares arr's memory
    return allreduce
