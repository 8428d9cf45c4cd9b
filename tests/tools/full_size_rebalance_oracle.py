#!/usr/bin/env python3
"""Config 3 at its full size, then the rebalance after a tenth of the nodes left (the scenario of
tests/test_properties.py::_moves_scenario), on the CPU oracle: digest and sweep count of the rebalanced map for
tests/golden/config3_full_size_properties.json ("rebalance").  Several minutes on one core."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from blance_amd import synth          # noqa: E402
from oracle import loader             # noqa: E402
import properties                     # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
fp = synth.config_flat(3, P=P, N=N)
r1 = loader.plan(fp)
fp2 = synth.replan_problem(fp, r1)
rm = np.zeros(N, dtype=np.uint8)
rm[np.arange(N) % 10 == 3] = 1
fp2.set("node_removed", rm)
t0 = time.time()
r2 = loader.plan(fp2)
print(json.dumps({"partitions": P, "nodes": N, "plan_digest": r1.digest(), "removed": "node ids with id % 10 == 3",
                  "digest": r2.digest(), "iterations": r2.iterations, "converged": bool(r2.converged),
                  "warnings": int(r2.n_warnings), "properties": properties.plan_properties(fp2, r2),
                  "rebalance_s": round(time.time() - t0, 1)}))
