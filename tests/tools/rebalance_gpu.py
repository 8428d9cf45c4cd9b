#!/usr/bin/env python3
"""Rebalance timing on the GPU box (BASELINE.json config 5's shape at one-GPU size):
plan a cluster, remove / add a tenth of the nodes, re-plan from the first plan.
    python tests/tools/rebalance_gpu.py P N [--weighted] [--flat] [--check]
--weighted keeps config 5's partition weights, node weights and stickiness,
--flat drops the hierarchy, --check compares the rebalance with the CPU oracle."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from blance_amd import hip, problem, synth          # noqa: E402


def main():
    P, N = int(sys.argv[1]), int(sys.argv[2])
    flags = set(sys.argv[3:])
    c = synth.rebalance_case(P=P, N=N, hierarchy="--flat" not in flags)
    if "--weighted" not in flags:
        c["partitionWeights"] = c["nodeWeights"] = c["stateStickiness"] = None
    fresh = {p: {"name": p, "nodesByState": {}} for p in c["partitions"]}
    opts = dict(partition_weights=c["partitionWeights"], state_stickiness=c["stateStickiness"],
                node_weights=c["nodeWeights"], node_hierarchy=c["nodeHierarchy"], hierarchy_rules=c["hierarchyRules"])
    pl = hip.Planner(lib_path=os.environ.get("BLANCE_DEV_LIB"), force_threads=int(os.environ.get("BLANCE_FORCE_T", "0")))   # developer builds (phase profile)
    fp1 = problem.build_problem({}, fresh, c["oldNodes"], [], c["oldNodes"], c["model"], **opts)
    r1 = pl.plan(fp1)
    r1 = pl.plan(fp1)
    print("initial  : sweeps %d  %.1f ms  bulk %d / %d" % (r1.iterations, r1.struct.device_ms,
                                                           r1.struct.steps_batched, r1.struct.steps_total))
    plan1, _ = problem.decode_result(fp1, r1)
    fp2 = problem.build_problem(plan1, plan1, c["nodesAll"], c["nodesToRemove"], c["nodesToAdd"], c["model"], **opts)
    r2 = pl.plan(fp2)
    print("rebalance: sweeps %d  %.1f ms  bulk %d / %d  warnings %d" % (
        r2.iterations, r2.struct.device_ms, r2.struct.steps_batched, r2.struct.steps_total, r2.n_warnings))
    if "--check" in flags:
        from oracle import loader
        print("initial matches oracle  :", r1.digest() == loader.plan(fp1).digest())
        print("rebalance matches oracle:", r2.digest() == loader.plan(fp2).digest())


if __name__ == "__main__":
    main()
