"""Plumbing for N > 1 ranks of ONE plan (include/blance_hip.h "one plan on several GPUs"):

  * shard_plan_rccl(planner, dist): one process per GPU; the ranks join the library's own RCCL
    communicator, after which a plan's region chains are sharded over them;
  * gloo_collectives(planner, dist): the embedder's collectives for the GPU-less tests -- the SIMT
    emulator's "device" memory is host memory, summed / gathered through gloo.  Refused on the
    gfx950 build (its buffers are device memory);
  * LocalGroup(n): n contexts on ONE device driven by n threads of one process; the collectives stage
    the device buffers through the host (hipMemcpy).  This is how the sharded code path (region_base
    > 0, the load-vector exchange, the all-gather of output slices) runs real gfx950 kernels on a
    one-GPU box;
  * max_over_ranks(seconds): the slowest rank's wall time for bench.py.
"""
import ctypes
import threading

import numpy as np


def max_over_ranks(seconds):
    """Wall time of the slowest rank (MAX all-reduce; RCCL on GPUs, gloo on CPU)."""
    import torch
    import torch.distributed as dist
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def shard_plan_rccl(planner, d=None):
    """Plans made through `planner` from now on run sharded over the ranks of `d` (RCCL inside the
    library).  Every rank must upload the same problem and make the same calls."""
    if d is None:
        import torch.distributed as d
    return planner.comm_init_rccl(d)


def gloo_collectives(planner, d=None):
    """(allreduce, allgather) over HOST memory for hip.Planner.comm_set_callback(): emulator build only."""
    import torch
    if d is None:
        import torch.distributed as d
    if not planner.is_emulated():
        raise RuntimeError("gloo collectives read the planner's buffers as host memory: emulator build only "
                           "(on a GPU use shard_plan_rccl, or LocalGroup for contexts of one device)")

    def view(ptr, count):
        return torch.from_numpy(np.ctypeslib.as_array((ctypes.c_int32 * count).from_address(ptr)))

    def allreduce(ptr, count):
        d.all_reduce(view(ptr, count), op=d.ReduceOp.SUM)         # in place: the tensor shares the buffer

    def allgather(ptr, per_rank):
        world, rank = d.get_world_size(), d.get_rank()
        t = view(ptr, per_rank * world)
        parts = [torch.empty(per_rank, dtype=torch.int32) for _ in range(world)]
        d.all_gather(parts, t[rank * per_rank:(rank + 1) * per_rank].clone())
        for r in range(world):
            t[r * per_rank:(r + 1) * per_rank] = parts[r]
    return allreduce, allgather


class LocalGroup:
    """n ranks as n threads of this process, every rank a context of its own on the same device (or the
    emulator).  rank_collectives(r) gives rank r's (allreduce, allgather); the buffers are staged through
    the host with hipMemcpy (plain memmove under the emulator)."""

    def __init__(self, n_ranks, emulated):
        self.n = n_ranks
        self.bar = threading.Barrier(n_ranks)
        self.slots = [None] * n_ranks
        self.result = None
        self.calls = 0
        if emulated:
            self._d2h = lambda host, dev, nbytes: ctypes.memmove(host, dev, nbytes)
            self._h2d = lambda dev, host, nbytes: ctypes.memmove(dev, host, nbytes)
        else:
            rt = ctypes.CDLL("libamdhip64.so")
            rt.hipMemcpy.restype = ctypes.c_int
            rt.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]

            def d2h(host, dev, nbytes):
                if rt.hipMemcpy(host, dev, nbytes, 2):
                    raise RuntimeError("hipMemcpy D2H failed")

            def h2d(dev, host, nbytes):
                if rt.hipMemcpy(dev, host, nbytes, 1):
                    raise RuntimeError("hipMemcpy H2D failed")
            self._d2h, self._h2d = d2h, h2d

    def _fetch(self, ptr, count):
        a = np.empty(count, dtype=np.int32)
        self._d2h(a.ctypes.data, ptr, 4 * count)
        return a

    def rank_collectives(self, rank):
        def allreduce(ptr, count):
            self.slots[rank] = self._fetch(ptr, count)
            if self.bar.wait() == 0:
                self.result = np.sum(np.stack(self.slots), axis=0, dtype=np.int64).astype(np.int32)
                self.calls += 1
            self.bar.wait()
            self._h2d(ptr, self.result.ctypes.data, 4 * count)
            self.bar.wait()

        def allgather(ptr, per_rank):
            a = np.empty(per_rank, dtype=np.int32)
            self._d2h(a.ctypes.data, ptr + 4 * per_rank * rank, 4 * per_rank)
            self.slots[rank] = a
            if self.bar.wait() == 0:
                self.result = np.concatenate(self.slots)
                self.calls += 1
            self.bar.wait()
            self._h2d(ptr, self.result.ctypes.data, 4 * per_rank * self.n)
            self.bar.wait()
        return allreduce, allgather

    def run(self, planners, fn):
        """fn(rank, planner) on n threads; returns the list of results (an exception of any rank is re-raised)."""
        out, err = [None] * self.n, [None] * self.n

        def body(r):
            try:
                out[r] = fn(r, planners[r])
            except BaseException as e:          # noqa: BLE001 -- re-raised below
                err[r] = e
                self.bar.abort()
        th = [threading.Thread(target=body, args=(r,)) for r in range(self.n)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        for e in err:
            if e is not None and not isinstance(e, threading.BrokenBarrierError):
                raise e
        for e in err:
            if e is not None:
                raise e
        return out


def local_sharded_planners(n_ranks, make_planner):
    """n contexts (make_planner() each) wired into one LocalGroup; returns (group, planners)."""
    planners = [make_planner() for _ in range(n_ranks)]
    grp = LocalGroup(n_ranks, planners[0].is_emulated())
    for r, pl in enumerate(planners):
        ar, ag = grp.rank_collectives(r)
        pl.comm_set_callback(r, n_ranks, ar, ag)
    return grp, planners
