#!/usr/bin/env python3
"""Config 3 at its full size on the CPU oracle: the numbers tests/test_properties.py asserts for the device result
(which is bit-identical to this one: tests/golden/config_digests.json).  ~4 minutes on one core."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from blance_amd import synth          # noqa: E402
from oracle import loader             # noqa: E402
import properties                     # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
fp = synth.config_flat(3, P=P, N=N)
t0 = time.time()
r = loader.plan(fp)
props = properties.config3_properties(fp, r)
t1 = time.time()
r2 = loader.plan(synth.replan_problem(fp, r))
print(json.dumps({"partitions": P, "nodes": N, "digest": r.digest(), "iterations": r.iterations, "properties": props,
                  "replan_digest": r2.digest(), "replan_iterations": r2.iterations, "replan_converged": bool(r2.converged),
                  "replan_same_lists": bool(properties.same_lists(r, r2, P, 2)),
                  "plan_s": round(t1 - t0, 1), "replan_s": round(time.time() - t1, 1)}))
