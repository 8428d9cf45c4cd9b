#!/usr/bin/env python3
"""Config 3 with scrambled non-numeric partition names and Zipf partition weights at its full size
(blance_amd/synth.py: config3_named_weighted_flat; workload (b) of bench.py's "general_regime" block) on the CPU
oracle: digest, sweeps and warnings for tests/golden/config3_general_regime.json.  Minutes on one core.
Usage: python tests/tools/general_regime_oracle.py [P N]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from blance_amd import synth          # noqa: E402
from oracle import loader             # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 2 else 1 << 20
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
fp = synth.config3_named_weighted_flat(P, N)
t0 = time.time()
r = loader.plan(fp)
entry = {"partitions": P, "nodes": N, "workload": "config 3 with scrambled non-numeric partition names and Zipf partition weights",
         "digest": r.digest(), "iterations": r.iterations, "converged": bool(r.converged), "warnings": int(r.n_warnings),
         "oracle_seconds": round(time.time() - t0, 1), "made_by": "tests/tools/general_regime_oracle.py (CPU oracle, one core)"}
print(json.dumps(entry), flush=True)
if P == 1 << 20 and N == 4096:
    with open(os.path.join(ROOT, "tests", "golden", "config3_general_regime.json"), "w") as f:
        json.dump({"named_weighted": entry}, f, indent=1)
