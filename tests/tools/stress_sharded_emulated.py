#!/usr/bin/env python3
"""One-off: the sharded plan (G ranks as G contexts on threads, kernels under the SIMT emulator, the two collectives of
a chain pass through dist_util.LocalGroup) on the random hierarchy cases of stress_gpu.py, fresh plan and rebalance,
against the CPU oracle.      python tests/tools/stress_sharded_emulated.py [n_cases] [seed0]"""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
sys.argv.append("--emulated")
import stress_gpu                                    # noqa: E402  (its generator, at the emulator's scale)
from blance_amd import dist_util, hip, problem       # noqa: E402
from oracle import loader                            # noqa: E402
from test_simt_emulated import build_emu             # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 20
    s0 = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 0
    emu = build_emu()
    bad = done = 0
    t0 = time.time()
    seed = s0
    while done < n:
        nodes, old, rm, model, opts, fresh = stress_gpu.case(seed)
        seed += 1
        if not opts.get("hierarchy_rules"):
            continue                                     # flat passes do not shard
        done += 1
        G = random.Random(seed).choice([2, 3, 4, 8])
        fp1 = problem.build_problem({}, fresh, old, [], old, model, **opts)
        want1 = loader.plan(fp1)
        grp, planners = dist_util.local_sharded_planners(G, lambda: hip.Planner(lib_path=emu, chain_min_parts=8))

        def work(rank, pl):
            r1 = pl.plan(fp1)
            plan1, _ = problem.decode_result(fp1, r1)
            rm2 = [x for x in rm if x in old]
            add2 = [x for x in nodes if x not in old]
            fp2 = problem.build_problem(plan1, plan1, nodes, rm2, add2, model, **opts)
            r2 = pl.plan(fp2)
            return r1.digest(), r2.digest(), fp2, pl.comm_stats()[0]
        res = grp.run(planners, work)
        for pl in planners:
            pl.close()
        want2 = loader.plan(res[0][2]).digest()
        ok = all(d1 == want1.digest() and d2 == want2 for d1, d2, _, _ in res)
        print("seed %d G=%d P=%d N=%d k=%s rules=%s: %s (collectives %d)" % (
            seed - 1, G, fp1.n_parts, len(nodes), model["replica"]["constraints"], opts["hierarchy_rules"]["replica"],
            "ok" if ok else "MISMATCH", res[0][3]), flush=True)
        bad += not ok
    print("mismatches: %d of %d sharded cases, %.1f s" % (bad, n, time.time() - t0))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
