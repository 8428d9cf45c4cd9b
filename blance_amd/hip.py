"""ctypes binding of the C ABI (include/blance_hip.h) exported by
blance_amd/lib/libblance_hip.so -- the hand-written HIP planner for gfx950.

There is no CPU fallback: if the library is missing or no device is visible
the calls raise.  (tests/simt builds the same kernel source against a SIMT
emulator for GPU-less logic tests; that library is loaded only by tests, through
the `lib_path` argument.)
"""
import ctypes as C
import os

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libblance_hip.so")

EXPORTS = ["blance_abi_version", "blance_last_error", "blance_result_capacity", "blance_validate",
           "blance_ctx_create", "blance_ctx_destroy", "blance_plan", "blance_upload",
           "blance_plan_resident", "blance_download", "blance_calc_moves", "blance_plan_stats_get",
           "blance_comm_unique_id", "blance_comm_init_rccl", "blance_comm_set", "blance_comm_stats",
           "blance_is_emulated", "blance_host_alloc", "blance_host_free", "blance_comm_time_ms", "blance_host_trim"]

_libs = {}


class BlanceError(RuntimeError):
    def __init__(self, status, text):
        RuntimeError.__init__(self, "blance status %d: %s" % (status, text))
        self.status = status


def load_library(path=None):
    path = path or LIB_PATH
    lib = _libs.get(path)
    if lib is not None:
        return lib
    if not os.path.exists(path):
        raise ImportError("%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950); there is no CPU fallback" % path)
    lib = C.CDLL(path)
    lib.blance_abi_version.restype = C.c_int
    lib.blance_last_error.restype = C.c_char_p
    lib.blance_result_capacity.restype = C.c_int64
    lib.blance_result_capacity.argtypes = [C.POINTER(abi.Problem)]
    lib.blance_validate.restype = C.c_int
    lib.blance_validate.argtypes = [C.POINTER(abi.Problem)]
    lib.blance_ctx_create.restype = C.c_int
    lib.blance_ctx_create.argtypes = [C.POINTER(abi.Options), C.POINTER(C.c_void_p)]
    lib.blance_ctx_destroy.restype = None
    lib.blance_ctx_destroy.argtypes = [C.c_void_p]
    for name in ("blance_plan",):
        getattr(lib, name).restype = C.c_int
        getattr(lib, name).argtypes = [C.c_void_p, C.POINTER(abi.Problem), C.POINTER(abi.Result)]
    lib.blance_upload.restype = C.c_int
    lib.blance_upload.argtypes = [C.c_void_p, C.POINTER(abi.Problem)]
    lib.blance_plan_resident.restype = C.c_int
    lib.blance_plan_resident.argtypes = [C.c_void_p, C.POINTER(abi.Result)]
    lib.blance_download.restype = C.c_int
    lib.blance_download.argtypes = [C.c_void_p, C.POINTER(abi.Result)]
    lib.blance_plan_stats_get.restype = C.c_int
    lib.blance_plan_stats_get.argtypes = [C.c_void_p, C.POINTER(abi.PlanStats)]
    lib.blance_calc_moves.restype = C.c_int
    lib.blance_calc_moves.argtypes = [C.c_void_p, C.POINTER(abi.MovesProblem), C.POINTER(abi.MovesResult)]
    lib.blance_comm_unique_id.restype = C.c_int
    lib.blance_comm_unique_id.argtypes = [C.c_void_p]
    lib.blance_comm_init_rccl.restype = C.c_int
    lib.blance_comm_init_rccl.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    lib.blance_comm_set.restype = C.c_int
    lib.blance_comm_set.argtypes = [C.c_void_p, C.POINTER(abi.Comm)]
    lib.blance_comm_stats.restype = C.c_int
    lib.blance_comm_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.blance_is_emulated.restype = C.c_int
    lib.blance_comm_time_ms.restype = C.c_int
    lib.blance_comm_time_ms.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    lib.blance_host_alloc.restype = C.c_void_p
    lib.blance_host_alloc.argtypes = [C.c_size_t]
    lib.blance_host_free.restype = None
    lib.blance_host_free.argtypes = [C.c_void_p]
    lib.blance_host_trim.restype = None
    lib.blance_host_trim.argtypes = []
    if lib.blance_abi_version() != abi.ABI_VERSION:
        raise ImportError("ABI version mismatch")
    _libs[path] = lib
    return lib


class _PinnedBlock:
    """One block of blance_host_alloc; goes back to the library when the last numpy view of it is collected."""

    def __init__(self, lib, nbytes):
        self.lib = lib
        self.p = lib.blance_host_alloc(nbytes)
        if not self.p:
            raise MemoryError("blance_host_alloc(%d) failed" % nbytes)

    def __del__(self):
        try:
            if self.p:
                self.lib.blance_host_free(self.p)
                self.p = None
        except Exception:
            pass


class HostArena:
    """Page-locked host arrays from blance_host_alloc (include/blance_hip.h, ABI 5): numpy views the device copies from / to
    by DMA where they lie.  Every array keeps its block alive (the ctypes buffer it is a view of owns the block), so an array
    that outlives the arena -- a result kept after the arena is closed -- never aliases a block that was handed out again;
    a block goes back to the library when its last view is collected."""

    def __init__(self, lib_path=None):
        self.lib = load_library(lib_path)
        self.n_blocks = 0

    def empty(self, n, dtype):
        import numpy as np
        dt = np.dtype(dtype)
        nbytes = max(int(n), 1) * dt.itemsize
        block = _PinnedBlock(self.lib, nbytes)
        buf = (C.c_char * nbytes).from_address(block.p)
        buf._block = block                                  # the view's base owns the block
        self.n_blocks += 1
        return np.frombuffer(buf, dtype=dt, count=int(n))

    def copy_of(self, arr):
        a = self.empty(arr.size, arr.dtype)
        a[...] = arr.reshape(-1)
        return a

    def close(self):
        """Kept for callers of the earlier interface: the blocks are owned by the arrays, nothing to release here."""


class Planner:
    """One blance_ctx: a planner bound to one gfx950 device."""

    def __init__(self, device_id=0, engine=abi.ENGINE_AUTO, lib_path=None, force_threads=0,
                 chain_min_parts=0, seq_speculation=True, tree="auto", planes=True, stay_top="auto", periodic=True, queue=True,
                 shard_one_rank=False):
        self.lib = load_library(lib_path)
        opt = abi.Options()
        opt.engine = engine
        opt.device_id = device_id
        opt.reserved[0] = force_threads      # workgroup size of the sequential pass (0 = auto)
        opt.reserved[1] = chain_min_parts    # smallest pass run as region chains (0 = default)
        # test knobs: 1 = k_pass_seq without verified stays; k_pass_tree (flat passes): 2 = never,
        # 4 = every general step scores all nodes, 8 = also when a k_pass_seq workgroup size is forced,
        # 16 = every general step decodes its record (none served from the validating lane's registers)
        # 32 = the all-blank chain pass on k_pass_chain_blank (lane minima) instead of k_pass_chain_planes
        # 64 = never k_stay_by_top (a chain pass of stays verified per top priority node), 128 = try it in every
        # chain pass with NumPartitions > 0
        # 256 = the all-blank chain pass WITHOUT its periodic form (k_period.h: a pass whose step records repeat walks two
        # periods and copies the rest -- the default; also BLANCE_PERIODIC=0)
        # 512 = flat passes with k <= 2 never on k_pass_queue (k_pass_tree / k_pass_seq take them, as before round 4);
        # 1024 = k_pass_queue without its lean walk (every step that does not stay through its general code)
        # 2048 = ... and every general step of it scoring every node; 4096 = its lean walk as compiled C++ only (the device
        # build walks the plain k = 2 steps in hand-written assembly, k_queue_walk.h); 8192 = its window always rebuilt by the
        # exact selection (one helper wave), never by the helper waves' striped form
        # 16384 = a communicator of ONE rank takes the sharded branch of every chain pass (both collectives execute:
        # how ncclAllReduce / ncclAllGather run on a one-GPU box)
        # 32768 = its walking wave copies a batch's row bit maps itself (the helper waves do since round 6)
        qbits = {True: 0, "on": 0, False: 512, "off": 512, "general": 1024, "dense": 1024 | 2048, "lean-cpp": 4096,
                 "exact-rebuild": 8192, "bits-self": 32768}[queue]
        opt.reserved[2] = qbits | (16384 if shard_one_rank else 0) | (0 if periodic else 256) | {"auto": 0, "off": 64, "force": 128}[stay_top] | (0 if planes else 32) | (0 if seq_speculation else 1) | {"auto": 0, "off": 2, "dense": 4 | 8, "on": 8, "long": 8 | 16,
                                                          "dense-long": 4 | 8 | 16}[tree]
        h = C.c_void_p()
        self._check(self.lib.blance_ctx_create(C.byref(opt), C.byref(h)))
        self._h = h

    def _check(self, st):
        if st != abi.OK:
            raise BlanceError(st, (self.lib.blance_last_error() or b"").decode())

    def close(self):
        if getattr(self, "_h", None):
            self.lib.blance_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- one plan on several ranks (include/blance_hip.h "one plan on several GPUs") ----
    def comm_init_rccl(self, dist):
        """Join the library's RCCL communicator: one process per GPU, `dist` an initialised
        torch.distributed (any backend) that carries the 128-byte id from rank 0 to the others."""
        rank, world = dist.get_rank(), dist.get_world_size()
        buf = C.create_string_buffer(128)
        if rank == 0:
            self._check(self.lib.blance_comm_unique_id(buf))
        box = [buf.raw]
        dist.broadcast_object_list(box, src=0)
        ident = C.create_string_buffer(box[0], 128)
        self._check(self.lib.blance_comm_init_rccl(self._h, world, rank, ident))
        return world

    def comm_init_rccl_one_rank(self):
        """A RCCL communicator of this one rank (no torch.distributed needed).  With Planner(shard_one_rank=True) the
        chain passes then take their sharded branch and really issue ncclAllReduce / ncclAllGather."""
        buf = C.create_string_buffer(128)
        self._check(self.lib.blance_comm_unique_id(buf))
        self._check(self.lib.blance_comm_init_rccl(self._h, 1, 0, buf))
        return 1

    def comm_set_callback(self, rank, n_ranks, allreduce, allgather=None):
        """An embedder's collectives: allreduce(address, count) sums `count` int32 values in place over the
        ranks; allgather(address, count_per_rank) completes n_ranks blocks of which this rank's is filled in
        (None: the library sums zero-padded outputs with allreduce instead).  The addresses are device
        memory of this context (host memory under the SIMT emulator)."""
        def _wrap(fn):
            def _cb(_user, ptr, count):
                try:
                    fn(ptr, count)
                    return 0
                except Exception:                      # no exception may cross the C boundary
                    import traceback
                    traceback.print_exc()
                    return 1
            return abi.ALLREDUCE_FN(_cb)
        self._comm_cb = (_wrap(allreduce), _wrap(allgather) if allgather else abi.ALLREDUCE_FN())   # keep the trampolines alive
        comm = abi.Comm(rank, n_ranks, self._comm_cb[0], None, self._comm_cb[1])
        self._check(self.lib.blance_comm_set(self._h, C.byref(comm)))

    def comm_stats(self):
        """(collectives made, int32 words moved) by this context's sharded plans so far."""
        calls, words = C.c_int64(0), C.c_int64(0)
        self._check(self.lib.blance_comm_stats(self._h, C.byref(calls), C.byref(words)))
        return int(calls.value), int(words.value)

    def comm_time_ms(self):
        """Device ms spent inside RCCL collectives by this context's plans so far."""
        ms = C.c_double(0.0)
        self._check(self.lib.blance_comm_time_ms(self._h, C.byref(ms)))
        return float(ms.value)

    def is_emulated(self):
        return bool(self.lib.blance_is_emulated())

    def comm_clear(self):
        self._check(self.lib.blance_comm_set(self._h, None))

    def validate(self, fp):
        return self.lib.blance_validate(C.byref(fp.as_struct()))

    def plan_into(self, fp, res):
        """blance_plan with the caller's result buffers (a FlatResult made for this problem shape) written again."""
        self._fp = fp
        self._check(self.lib.blance_plan(self._h, C.byref(fp.as_struct()), C.byref(res.struct)))
        return res

    def plan(self, fp):
        """blance_plan(): host buffers in, host buffers out."""
        res = abi.FlatResult(fp)
        self._check(self.lib.blance_plan(self._h, C.byref(fp.as_struct()), C.byref(res.struct)))
        return res

    def plan_stats(self, n_states):
        """Per-state load statistics of the map the last plan produced (blance_plan_stats_get):
        dict of numpy arrays load_min / load_max / load_sum / load_sumsq / nodes_used / unmet_slots / rule_violations
        plus n_nodes_next."""
        import numpy as np
        a = {k: np.zeros(max(n_states, 1), dtype=np.int64) for k in ("load_min", "load_max", "load_sum", "load_sumsq", "unmet_slots", "rule_violations")}
        used = np.zeros(max(n_states, 1), dtype=np.int32)
        st = abi.PlanStats()
        st.n_states = n_states
        for k, v in a.items():
            setattr(st, k, v.ctypes.data_as(C.POINTER(C.c_int64)))
        st.nodes_used = used.ctypes.data_as(C.POINTER(C.c_int32))
        self._check(self.lib.blance_plan_stats_get(self._h, C.byref(st)))
        out = {k: v[:n_states] for k, v in a.items()}
        out["nodes_used"] = used[:n_states]
        out["n_nodes_next"] = int(st.n_nodes_next)
        return out

    def calc_moves(self, n_states, favor_min_nodes, beg_off, beg_nodes, end_off, end_nodes):
        """blance_calc_moves(): CalcPartitionMoves for every partition (CSR over
        p * (n_states + 1) + state).  Returns (op_off, op_node, op_state, op_kind, device_ms)."""
        import numpy as np
        arr = [np.ascontiguousarray(a, dtype=np.int32) for a in (beg_off, beg_nodes, end_off, end_nodes)]
        keep = [a if a.size else np.zeros(1, dtype=np.int32) for a in arr]
        P = (arr[0].size - 1) // (n_states + 1)
        pb = abi.MovesProblem()
        pb.n_parts, pb.n_states, pb.favor_min_nodes = P, n_states, int(bool(favor_min_nodes))
        pb.beg_off, pb.beg_nodes, pb.end_off, pb.end_nodes = [a.ctypes.data_as(C.POINTER(C.c_int32)) for a in keep]
        cap = int(arr[0][-1]) + int(arr[2][-1])
        out = [np.zeros(P + 1, dtype=np.int32)] + [np.zeros(max(cap, 1), dtype=np.int32) for _ in range(3)]
        res = abi.MovesResult()
        res.op_off, res.op_node, res.op_state, res.op_kind = [a.ctypes.data_as(C.POINTER(C.c_int32)) for a in out]
        res.capacity = cap
        self._check(self.lib.blance_calc_moves(self._h, C.byref(pb), C.byref(res)))
        return out[0], out[1], out[2], out[3], float(res.device_ms)

    def upload(self, fp):
        self._fp = fp
        self._check(self.lib.blance_upload(self._h, C.byref(fp.as_struct())))

    def plan_resident(self):
        """Run the whole planNextMapEx loop on the uploaded problem; returns the
        timing/statistics part of blance_result."""
        r = abi.Result()
        self._check(self.lib.blance_plan_resident(self._h, C.byref(r)))
        return r

    def download(self, arena=None, into=None):
        """arena: a HostArena -- the result's arrays in page-locked memory (the device writes them by DMA); into: a FlatResult
        of an earlier download of the same problem shape, written again (no allocation)."""
        res = into if into is not None else abi.FlatResult(self._fp, arena)
        self._check(self.lib.blance_download(self._h, C.byref(res.struct)))
        return res
