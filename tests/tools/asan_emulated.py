#!/usr/bin/env python3
"""The kernels under the SIMT emulator built with AddressSanitizer: "device" memory is the host's heap there, so a kernel
that reads or writes outside a buffer of the library (hipMalloc = malloc in tests/simt/hip_emu.h) is reported with file
and line -- the check a GPU does not make until a page is missing.  (It found k_pass_chain's stage prefetch reading up to
928 bytes past the record array for a last stage shorter than 11 steps; fixed in round 3.)

    python tests/tools/asan_emulated.py [n_cases] [seed0]        # re-executes itself under LD_PRELOAD=libasan.so

Cases: the generator of stress_gpu.py at sizes that put short and ragged stages on the chain kernels (chain_min_parts=1)
plus its usual emulator sizes; fresh plan and rebalance, digests compared with the oracle as well."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
HERE = os.path.join(ROOT, "tests")
SO = os.path.join(HERE, "simt", "_build", "libblance_emu_asan.so")
SRC = os.path.join(HERE, "simt", "emu_lib.cpp")


def build():
    deps = [SRC, os.path.join(HERE, "simt", "hip_emu.h"), os.path.join(ROOT, "include", "blance_hip.h")]
    csrc = os.path.join(ROOT, "blance_amd", "csrc")
    deps += [os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".h", ".hip"))]
    if os.path.exists(SO) and all(os.path.getmtime(d) <= os.path.getmtime(SO) for d in deps):
        return SO
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    subprocess.check_call(["g++", "-x", "c++", "-std=c++17", "-O1", "-g1", "-fsanitize=address", "-fno-omit-frame-pointer",
                           "-fPIC", "-shared", "-ffp-contract=off", "-Wno-unknown-pragmas", "-o", SO, SRC])
    return SO


def libasan():
    return subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()


def worker(n, s0):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.join(HERE, "tools"))
    sys.argv = [sys.argv[0], "--emulated"]
    import random
    import stress_gpu
    from blance_amd import hip, problem
    from oracle import loader
    bad = 0
    for seed in range(s0, s0 + n):
        rng = random.Random(seed)
        small = rng.random() < 0.7
        os.environ["STRESS_P"] = rng.choice(["3,11,23,70", "5,17,130,263", "9,40,301,517"]) if small else "257,520,900"
        kw = rng.choice([{}, {}, {"planes": False}, {"stay_top": "force"}, {"stay_top": "off"}])
        pl = hip.Planner(lib_path=SO, chain_min_parts=1 if small else 8, **kw)
        nodes, old, rm, model, opts, fresh = stress_gpu.case(seed)
        fp1 = problem.build_problem({}, fresh, old, [], old, model, **opts)
        r1 = pl.plan(fp1)
        bad += r1.digest() != loader.plan(fp1).digest()
        plan1, _ = problem.decode_result(fp1, r1)
        fp2 = problem.build_problem(plan1, plan1, nodes, [x for x in rm if x in old], [x for x in nodes if x not in old],
                                    model, **opts)
        bad += pl.plan(fp2).digest() != loader.plan(fp2).digest()
        pl.close()
    print("asan emulated: %d cases (2 plans each), %d digest mismatches, no memory error reported" % (n, bad), flush=True)
    return 1 if bad else 0


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    if os.environ.get("BLANCE_ASAN_WORKER") == "1":
        return worker(n, s0)
    build()
    env = dict(os.environ, BLANCE_ASAN_WORKER="1", LD_PRELOAD=libasan(),
               ASAN_OPTIONS="detect_leaks=0:detect_stack_use_after_return=0:abort_on_error=1")
    return subprocess.call([sys.executable, os.path.abspath(__file__), str(n), str(s0)], env=env)


if __name__ == "__main__":
    sys.exit(main())
