#!/usr/bin/env python3
"""One-off GPU stress: mid-size random problems (thousands of partitions, up to a
thousand nodes; regular and ragged hierarchies, weights, removals, rebalances)
through the HIP planner vs the CPU oracle, bit for bit.  Run on the GPU box:
    python tests/tools/stress_gpu.py [n_cases] [seed0] [--flat-heavy]
With --emulated the same generator, at a tenth of the partitions, runs the kernels under the SIMT emulator on the CPU
(tests/simt): hours of idle CPU find what the fixed seeds of the test-suite do not."""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from blance_amd import hip, problem, synth          # noqa: E402
from oracle import loader                            # noqa: E402


FLAT_HEAVY = "--flat-heavy" in sys.argv
if FLAT_HEAVY:
    sys.argv.remove("--flat-heavy")
EMULATED = "--emulated" in sys.argv
if EMULATED:
    sys.argv.remove("--emulated")


def case(seed):
    rng = random.Random(seed)
    rack = rng.choice([2, 4, 8, 16])
    rpz = rng.choice([2, 3, 4, 8])
    zones = rng.choice([2, 3, 4, 8, 16])
    N = rack * rpz * zones - rng.choice([0, 0, 1, 3])
    P = rng.choice([257, 520, 900, 2100]) if EMULATED else rng.choice([2500, 4000, 9000, 20000])
    if os.environ.get("STRESS_P"):                  # e.g. STRESS_P=5,23,70,130: chains with short last stages
        P = rng.choice([int(x) for x in os.environ["STRESS_P"].split(",")])
    nodes = ["n%04d" % i for i in range(N)]
    hier = synth.hierarchy_names(N, rack=rack, racks_per_zone=rpz, zones_per_dc=4)
    k = rng.choice([1, 2, 2, 3])
    model = {"primary": {"priority": 0, "constraints": 1}, "replica": {"priority": 1, "constraints": k}}
    rule = rng.choice([(2, 1), (2, 1), (1, 0), (2, 0), (3, 1), None])
    if FLAT_HEAVY:                                  # the workgroup pass: flat, weighted, up to 8 nodes per thread
        rule = rng.choice([None, None, (2, 1)])
        if rng.random() < 0.3:
            N = rng.choice([300, 700]) if EMULATED else rng.choice([1500, 3000, 5000])
            nodes = ["n%04d" % i for i in range(N)]
            hier = synth.hierarchy_names(N, rack=rack, racks_per_zone=rpz, zones_per_dc=4)
            P = rng.choice([200, 400]) if EMULATED else rng.choice([2500, 4000])
    rules = None if rule is None else {"replica": [{"includeLevel": rule[0], "excludeLevel": rule[1]}]}
    opts = dict(node_hierarchy=hier if rules else None, hierarchy_rules=rules)
    if rng.random() < (0.7 if FLAT_HEAVY else 0.4):
        opts["partition_weights"] = {str(i): rng.choice([1, 2, 3, 5]) for i in range(P) if rng.random() < 0.6}
    if rng.random() < (0.6 if FLAT_HEAVY else 0.3):
        opts["node_weights"] = {n: rng.choice([1, 1, 2, 3]) for n in nodes}
    if rng.random() < 0.3:
        opts["state_stickiness"] = {"primary": rng.choice([1, 5, 100]), "replica": rng.choice([1, 10])}
    fresh = {str(i): {"name": str(i), "nodesByState": {}} for i in range(P)}
    rm = rng.sample(nodes, rng.choice([0, 1, N // 10]))
    old = [n for n in nodes if n not in rng.sample(nodes, rng.choice([0, N // 10]))]
    return nodes, old, rm, model, opts, fresh


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    import json
    kw = json.loads(os.environ.get("STRESS_PLANNER_KW", "{}"))     # e.g. '{"planes": false}', '{"stay_top": "force"}'
    if EMULATED:
        from test_simt_emulated import build_emu
        pl = hip.Planner(lib_path=build_emu(), **kw)
    else:
        pl = hip.Planner(device_id=0, **kw)
    bad = 0
    t0 = time.time()
    for seed in range(s0, s0 + n):
        nodes, old, rm, model, opts, fresh = case(seed)
        fp1 = problem.build_problem({}, fresh, old, [], old, model, **opts)
        r1 = pl.plan(fp1)
        o1 = loader.plan(fp1)
        ok1 = r1.digest() == o1.digest()
        plan1, _ = problem.decode_result(fp1, r1)
        rm2 = [x for x in rm if x in old]
        add2 = [x for x in nodes if x not in old]
        fp2 = problem.build_problem(plan1, plan1, nodes, rm2, add2, model, **opts)
        r2 = pl.plan(fp2)
        o2 = loader.plan(fp2)
        ok2 = r2.digest() == o2.digest()
        print("seed %d P=%d N=%d k=%s rules=%s: fresh %s (it %d, bulk %d/%d) rebalance %s (it %d, bulk %d/%d)"
              % (seed, fp1.n_parts, len(nodes), model["replica"]["constraints"], opts.get("hierarchy_rules"),
                 "ok" if ok1 else "MISMATCH", r1.iterations, r1.struct.steps_batched, r1.struct.steps_total,
                 "ok" if ok2 else "MISMATCH", r2.iterations, r2.struct.steps_batched, r2.struct.steps_total), flush=True)
        bad += (not ok1) + (not ok2)
    print("mismatches: %d of %d plans, %.1f s" % (bad, 2 * n, time.time() - t0))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
