"""Build + load oracle/blance_oracle.c (test infrastructure only)."""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libblance_oracle.so")
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "blance_oracle.c")
    hdr = os.path.join(_HERE, "..", "include", "blance_hip.h")
    stale = (not os.path.exists(_SO)) or any(
        os.path.getmtime(f) > os.path.getmtime(_SO) for f in (src, hdr))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


def load():
    global _lib
    if _lib is None:
        build()
        lib = C.CDLL(_SO)
        lib.blance_oracle_plan.restype = C.c_int
        lib.blance_oracle_plan.argtypes = [C.c_void_p, C.c_void_p]
        lib.blance_oracle_plan_ex.restype = C.c_int
        lib.blance_oracle_plan_ex.argtypes = [C.c_void_p, C.c_void_p, C.c_int64,
                                              C.POINTER(C.c_int64), C.POINTER(C.c_double)]
        lib.blance_oracle_result_capacity.restype = C.c_int64
        lib.blance_oracle_result_capacity.argtypes = [C.c_void_p]
        _lib = lib
    return _lib


def plan(flat_problem):
    """Run the C oracle on a blance_amd.abi.FlatProblem; returns a FlatResult."""
    from blance_amd import abi
    lib = load()
    res = abi.FlatResult(flat_problem)
    ps = flat_problem.as_struct()
    st = lib.blance_oracle_plan(C.byref(ps), C.byref(res.struct))
    if st != 0:
        raise RuntimeError("oracle status %d" % st)
    return res


def timed_sample(flat_problem, step_limit):
    """Time the first `step_limit` findBestNodes calls (bounded CPU baseline).
    Returns (steps_done, seconds, finished)."""
    from blance_amd import abi
    lib = load()
    res = abi.FlatResult(flat_problem)
    ps = flat_problem.as_struct()
    steps = C.c_int64(0)
    secs = C.c_double(0.0)
    st = lib.blance_oracle_plan_ex(C.byref(ps), C.byref(res.struct), int(step_limit),
                                   C.byref(steps), C.byref(secs))
    if st not in (0, 1):
        raise RuntimeError("oracle status %d" % st)
    return steps.value, secs.value, st == 0
