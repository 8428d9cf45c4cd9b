#!/usr/bin/env python3
"""Generates tests/golden/config_digests.json: the CPU oracle's result digest
(abi.FlatResult.digest) for BASELINE.json's synthetic configs at FULL size, so
GPU tests and bench.py can check bit-parity at 1M partitions without re-running
the 2.5-minute single-core oracle.  Usage: python tests/tools/make_config_digests.py"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from blance_amd import synth          # noqa: E402
from oracle import loader             # noqa: E402

out = {}
for cfg in (1, 2, 3):
    fp = synth.config_flat(cfg)
    t = time.time()
    r = loader.plan(fp)
    out["config%d" % cfg] = {"partitions": fp.n_parts, "nodes": fp.n_nodes, "iterations": r.iterations,
                            "warnings": r.n_warnings, "digest": r.digest(),
                            "oracle_seconds": round(time.time() - t, 2)}
    print(cfg, out["config%d" % cfg], flush=True)
with open(os.path.join(ROOT, "tests", "golden", "config_digests.json"), "w") as f:
    json.dump(out, f, indent=1)
