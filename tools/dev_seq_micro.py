#!/usr/bin/env python3
"""Developer aid: every step through k_pass_seq (BLANCE_ENGINE_SEQUENTIAL), for instruction-count
profiling with rocprofv3 --pmc.  python tools/dev_seq_micro.py P N [weighted]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from blance_amd import abi, hip, problem, synth          # noqa: E402

P, N = int(sys.argv[1]), int(sys.argv[2])
c = synth.config_case(2, P=P, N=N)
if len(sys.argv) > 3:
    c["nodeWeights"] = {n: 1 + (i % 3) for i, n in enumerate(c["nodesAll"])}
fp = synth.case_to_flat(c)
pl = hip.Planner(engine=abi.ENGINE_SEQUENTIAL, lib_path=os.environ.get("BLANCE_DEV_LIB"),
                 force_threads=int(os.environ.get("BLANCE_FORCE_T", "0")))
r = pl.plan(fp)
print("sweeps %d  steps %d  device %.2f ms  -> %.2f us/step" % (
    r.iterations, r.struct.steps_total, r.struct.device_ms, 1e3 * r.struct.device_ms / r.struct.steps_total))
