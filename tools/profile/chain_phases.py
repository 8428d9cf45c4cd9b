#!/usr/bin/env python3
"""Developer aid: one PlanNextMap of config 3 (or P N given) through devbuild/libblance_prof.so built with
tools/profile/build_prof.sh tu_chain (-DBLANCE_PHASE_PROF: shader-clock totals per phase of chain 0, printed by the kernel).
Phases of k_pass_chain: 0 stage records into LDS; 12..17 a stay round (12 entry, 13 command + barrier A, 14 marks + barrier M,
15 the test, 16 barrier B + verdicts, 17 the prefix's bumps); 18 stage tail, 19 outputs to HBM; 1..11 the general step.
    python tools/profile/chain_phases.py [P N]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from blance_amd import hip, synth          # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 2 else None
N = int(sys.argv[2]) if len(sys.argv) > 2 else None
pl = hip.Planner(lib_path=os.path.join(ROOT, "devbuild", "libblance_prof.so"))
r = pl.plan(synth.config_flat(3, P=P, N=N))
print("sweeps %d, %.2f ms" % (r.iterations, r.struct.device_ms))
