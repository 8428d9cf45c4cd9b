#!/usr/bin/env python3
"""Developer aid: run one PlanNextMap of config 3 (or P N given) through a library built with
-DBLANCE_PHASE_PROF (shader-clock totals per phase of chain 0 / the workgroup pass, printed by
the kernels).  Build: hipcc ... -DBLANCE_PHASE_PROF -o devbuild/libblance_prof.so blance_amd/csrc/blance_hip.hip
    python tools/dev_phase_profile.py [P N]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from blance_amd import hip, synth          # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 2 else None
N = int(sys.argv[2]) if len(sys.argv) > 2 else None
pl = hip.Planner(lib_path=os.path.join(ROOT, "devbuild", "libblance_prof.so"))
r = pl.plan(synth.config_flat(3, P=P, N=N))
print("sweeps %d, %.2f ms" % (r.iterations, r.struct.device_ms))
