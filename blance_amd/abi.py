"""ctypes view of include/blance_hip.h (the C-ABI structs and status codes).

Kept field-for-field in sync with the header; tests/test_abi.py checks the
struct sizes against the sizes the C compiler reports.
"""
import ctypes as C

import numpy as np

ABI_VERSION = 6

OK = 0
ERR_BAD_ARG = -1
ERR_UNSUPPORTED = -2
ERR_CAPACITY = -3
ERR_DEVICE = -4
ERR_NO_DEVICE = -5
ERR_COMM = -6

LIST_ABSENT, LIST_NIL, LIST_SET = 0, 1, 2
BOOSTER_NONE, BOOSTER_CBGT = 0, 1
ENGINE_AUTO, ENGINE_SEQUENTIAL = 0, 1

_i32p = C.POINTER(C.c_int32)
_u8p = C.POINTER(C.c_uint8)


class Problem(C.Structure):
    _fields_ = [
        ("n_nodes", C.c_int32), ("n_nodes_ext", C.c_int32), ("n_states", C.c_int32),
        ("n_parts", C.c_int32), ("n_prev", C.c_int32), ("n_loads", C.c_int32),
        ("n_rules", C.c_int32), ("n_vertices", C.c_int32), ("max_iterations", C.c_int32),
        ("partition_weights_nil", C.c_int32), ("nodes_to_add_nil", C.c_int32),
        ("hierarchy_rules_nil", C.c_int32), ("booster_kind", C.c_int32), ("top_state", C.c_int32),
        ("state_priority", _i32p), ("state_constraints", _i32p), ("state_stickiness", _i32p),
        ("state_has_stickiness", _u8p),
        ("node_removed", _u8p), ("node_added", _u8p), ("node_weight", _i32p),
        ("node_has_weight", _u8p),
        ("part_order", _i32p), ("part_weight", _i32p), ("part_has_weight", _u8p),
        ("part_in_prev", _u8p), ("part_prev_never_equal", _u8p),
        ("assign_off", _i32p), ("assign_nodes", _i32p), ("assign_kind", _u8p),
        ("prev_off", _i32p), ("prev_nodes", _i32p), ("prev_kind", _u8p),
        ("load_state", _i32p), ("load_node", _i32p), ("load_weight", _i32p),
        ("load_first_sweep_only", _u8p),
        ("rule_off", _i32p), ("rule_inc", _i32p), ("rule_exc", _i32p),
        ("vertex_empty", C.c_int32),
        ("vertex_parent", _i32p), ("vertex_leaf_lo", _i32p), ("vertex_leaf_hi", _i32p),
        ("node_leaf_pos", _i32p),
    ]


class Result(C.Structure):
    _fields_ = [
        ("out_off", _i32p), ("out_nodes", _i32p), ("out_kind", _u8p), ("out_capacity", C.c_int64),
        ("warn_part", _i32p), ("warn_state", _i32p), ("warn_capacity", C.c_int64),
        ("n_warnings", C.c_int64),
        ("iterations", C.c_int32), ("converged", C.c_int32),
        ("device_ms", C.c_double), ("total_ms", C.c_double),
        ("steps_total", C.c_int64), ("steps_sequential", C.c_int64), ("steps_batched", C.c_int64),
        ("kernel_launches", C.c_int64),
        ("pass_kernel_ms", C.c_double), ("pass_kernel_launches", C.c_int64),
        ("flat_pass_ms", C.c_double), ("flat_passes", C.c_int64),
        ("blank_pass_ms", C.c_double), ("blank_pass_launches", C.c_int64),
        ("stay_pass_ms", C.c_double), ("stay_pass_launches", C.c_int64),
        ("host_syncs", C.c_int64),
    ]


OP_ADD, OP_DEL, OP_PROMOTE, OP_DEMOTE = 0, 1, 2, 3
OP_NAMES = ["add", "del", "promote", "demote"]          # NodeStateOp.Op, moves.go:17-21


class MovesProblem(C.Structure):
    _fields_ = [("n_parts", C.c_int32), ("n_states", C.c_int32), ("favor_min_nodes", C.c_int32),
                ("beg_off", _i32p), ("beg_nodes", _i32p), ("end_off", _i32p), ("end_nodes", _i32p)]


class MovesResult(C.Structure):
    _fields_ = [("op_off", _i32p), ("op_node", _i32p), ("op_state", _i32p), ("op_kind", _i32p),
                ("capacity", C.c_int64), ("device_ms", C.c_double)]


class Options(C.Structure):
    _fields_ = [("engine", C.c_int32), ("device_id", C.c_int32), ("reserved", C.c_int32 * 6)]


I32_FIELDS = [n for n, t in Problem._fields_ if t is _i32p]
U8_FIELDS = [n for n, t in Problem._fields_ if t is _u8p]
SCALAR_FIELDS = [n for n, t in Problem._fields_ if t is C.c_int32]


def _ptr(arr, ctype):
    return arr.ctypes.data_as(C.POINTER(ctype))


class FlatProblem:
    """A blance_problem held as numpy arrays (the SoA the ABI describes) plus the
    name tables needed to un-intern a result."""

    def __init__(self):
        self.scalars = {n: 0 for n in SCALAR_FIELDS}
        self.arrays = {}
        # name tables (optional; synthetic problems may leave them None)
        self.node_names = None
        self.state_names = None
        self.part_names = None
        self._struct = None

    def set(self, name, arr):
        if name in I32_FIELDS or name in U8_FIELDS:
            dt = np.int32 if name in I32_FIELDS else np.uint8
            src = np.asarray(arr)
            if src.size and src.dtype != dt:       # never wrap silently (a stickiness of 2**31, say)
                info = np.iinfo(dt)
                if src.min() < info.min or src.max() > info.max:
                    raise OverflowError("%s: value outside %s" % (name, np.dtype(dt).name))
            a = np.ascontiguousarray(src, dtype=dt)
        else:
            raise KeyError(name)
        if a.size == 0:          # keep a valid pointer for empty arrays
            a = np.zeros(1, dtype=a.dtype)[:0].copy()
        self.arrays[name] = a
        self._struct = None

    def __getattr__(self, name):
        d = self.__dict__
        if "scalars" in d and name in d["scalars"]:
            return d["scalars"][name]
        if "arrays" in d and name in d["arrays"]:
            return d["arrays"][name]
        raise AttributeError(name)

    def as_struct(self):
        if self._struct is None:
            s = Problem()
            for k, v in self.scalars.items():
                setattr(s, k, int(v))
            self._keep = []
            for n in I32_FIELDS + U8_FIELDS:
                a = self.arrays.get(n)
                if a is None:
                    a = np.zeros(0, dtype=np.int32 if n in I32_FIELDS else np.uint8)
                if a.size == 0:
                    a = np.zeros(1, dtype=a.dtype)   # non-NULL pointer, zero logical length
                self._keep.append(a)
                setattr(s, n, _ptr(a, C.c_int32 if n in I32_FIELDS else C.c_uint8))
            self._struct = s
        return self._struct

    def pin(self, arena):
        """A copy of this problem with every array in page-locked memory of `arena` (hip.HostArena): blance_upload then copies
        by DMA from where they lie instead of staging them.  This problem itself is left as it is."""
        import copy
        q = copy.copy(self)
        q.scalars = dict(self.scalars)
        q.arrays = {n: (arena.copy_of(a) if a.size else a) for n, a in self.arrays.items()}
        q._struct = None
        q._keep = []
        return q

    def save_npz(self, path):
        """The problem as one .npz file (arrays + scalars; no name tables): how bench.py hands a problem to a child process."""
        np.savez(path, __scalars__=np.array([[k, int(v)] for k, v in self.scalars.items()], dtype=object), **self.arrays)

    @staticmethod
    def load_npz(path):
        fp = FlatProblem()
        with np.load(path, allow_pickle=True) as z:
            for k, v in z["__scalars__"]:
                fp.scalars[str(k)] = int(v)
            for n in z.files:
                if n != "__scalars__":
                    fp.set(n, z[n])
        return fp

    def result_capacity(self):
        P, M = self.scalars["n_parts"], self.scalars["n_states"]
        if P * M == 0:
            return 0
        lens = np.diff(self.arrays["assign_off"].astype(np.int64)).reshape(P, M)
        k = np.maximum(self.arrays["state_constraints"].astype(np.int64), 0)[None, :]
        return int(np.maximum(lens, k).sum())


class FlatResult:
    """Caller-owned output buffers for one blance_plan call."""

    def __init__(self, prob, arena=None):
        P, M = prob.scalars["n_parts"], prob.scalars["n_states"]
        cap = prob.result_capacity()
        new = (lambda n, dt: np.zeros(n, dtype=dt)) if arena is None else arena.empty
        self._arena = arena
        self.out_off = new(P * M + 1, np.int32)
        self.out_nodes = new(max(cap, 1), np.int32)
        self.out_kind = new(max(P * M, 1), np.uint8)
        self.warn_part = new(max(P * M, 1), np.int32)
        self.warn_state = new(max(P * M, 1), np.int32)
        s = Result()
        s.out_off = _ptr(self.out_off, C.c_int32)
        s.out_nodes = _ptr(self.out_nodes, C.c_int32)
        s.out_kind = _ptr(self.out_kind, C.c_uint8)
        s.out_capacity = cap
        s.warn_part = _ptr(self.warn_part, C.c_int32)
        s.warn_state = _ptr(self.warn_state, C.c_int32)
        s.warn_capacity = P * M
        self.struct = s
        self.P, self.M = P, M

    @property
    def iterations(self):
        return int(self.struct.iterations)

    @property
    def converged(self):
        return bool(self.struct.converged)

    @property
    def n_warnings(self):
        return int(self.struct.n_warnings)

    def lists(self):
        """[(p, m)] -> (kind, np.array of node ids)."""
        off = self.out_off
        return [[(int(self.out_kind[p * self.M + m]),
                  self.out_nodes[off[p * self.M + m]:off[p * self.M + m + 1]].copy())
                 for m in range(self.M)] for p in range(self.P)]

    def warnings(self):
        n = self.n_warnings
        return list(zip(self.warn_part[:n].tolist(), self.warn_state[:n].tolist()))

    def digest(self):
        """SHA-256 over (out_kind, out_off, out_nodes[:total]) -- a compact
        bit-exactness check between two implementations of the same problem."""
        import hashlib
        h = hashlib.sha256()
        total = int(self.out_off[self.P * self.M]) if self.P * self.M else 0
        h.update(self.out_kind[:self.P * self.M].tobytes())
        h.update(self.out_off[:self.P * self.M + 1].tobytes())
        h.update(self.out_nodes[:total].tobytes())
        n = self.n_warnings
        h.update(np.asarray([self.iterations, int(self.converged), n], dtype=np.int64).tobytes())
        h.update(self.warn_part[:n].tobytes())
        h.update(self.warn_state[:n].tobytes())
        return h.hexdigest()


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64)


class Comm(C.Structure):
    """blance_comm: an embedder's collectives (int32 sum all-reduce, all-gather) for a plan sharded over ranks."""
    _fields_ = [("rank", C.c_int32), ("n_ranks", C.c_int32), ("allreduce_sum_i32", ALLREDUCE_FN), ("user", C.c_void_p),
                ("allgather_i32", ALLREDUCE_FN)]


class PlanStats(C.Structure):
    """blance_plan_stats of include/blance_hip.h."""
    _fields_ = [("n_states", C.c_int32), ("n_nodes_next", C.c_int32),
                ("load_min", C.POINTER(C.c_int64)), ("load_max", C.POINTER(C.c_int64)),
                ("load_sum", C.POINTER(C.c_int64)), ("load_sumsq", C.POINTER(C.c_int64)),
                ("nodes_used", C.POINTER(C.c_int32)), ("unmet_slots", C.POINTER(C.c_int64)),
                ("rule_violations", C.POINTER(C.c_int64))]
