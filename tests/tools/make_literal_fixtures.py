#!/usr/bin/env python3
"""tests/golden/literal_oracle_cases.json: results of the LITERAL string-keyed oracle (oracle/blance_ref.py -- the
line-by-line restatement of plan.go that the reference's 69 golden tables pin) on the cases of tests/literal_cases.py at
4,096 partitions x 256 nodes.  Pure Python, one core, a few minutes.  The fixture holds digests only (canonical JSON of the
result map and of the warnings, sweeps, converged); tests rebuild the inputs from the same deterministic generators.
Usage: python tests/tools/make_literal_fixtures.py"""
import copy
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import literal_cases as L             # noqa: E402
from oracle import blance_ref as R    # noqa: E402


def run(case):
    info = {}
    t0 = time.time()
    out, w = R.run_case(copy.deepcopy(case), info)
    return out, w or {}, info, time.time() - t0


def entry(name, case, out, w, info, dt):
    e = {"case": name, "partitions": len(case["partitionsToAssign"]), "nodes": len(case["nodesAll"]),
         "map_sha256": L.canonical_sha(out), "warnings_sha256": L.warnings_sha(w), "n_warning_keys": len(w),
         "iterations": info["iterations"], "converged": bool(info["converged"]), "literal_oracle_seconds": round(dt, 1)}
    print(json.dumps(e), flush=True)
    return e


def main():
    entries = []
    for name, case in (("named_weighted", L.case_named_weighted()), ("node_weights", L.case_node_weights())):
        out, w, info, dt = run(case)
        entries.append(entry(name, case, out, w, info, dt))
    from blance_amd import synth
    base = synth.config_case(3, P=L.P_LITERAL, N=L.N_LITERAL)
    plan, w, info, dt = run(base)
    entries.append(entry("config3", base, plan, w, info, dt))
    reb = L.case_rebalance(plan)
    out, w, info, dt = run(reb)
    entries.append(entry("rebalance_of_config3", reb, out, w, info, dt))
    with open(os.path.join(ROOT, "tests", "golden", "literal_oracle_cases.json"), "w") as f:
        json.dump({"made_by": "tests/tools/make_literal_fixtures.py: oracle/blance_ref.py (literal, string keyed) on the generators of "
                              "tests/literal_cases.py", "cases": entries}, f, indent=1)


if __name__ == "__main__":
    main()
