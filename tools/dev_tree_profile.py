#!/usr/bin/env python3
"""Developer aid: config 5's generator (weighted flat rebalance) at P x N through a library whose
k_pass_tree was built with -DBLANCE_PHASE_PROF (devbuild/libblance_prof.so): per launch the kernel
prints its step / general-step / walk counters and shader-clock totals per phase.
    python tools/dev_tree_profile.py [P N]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from blance_amd import hip, synth          # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 2 else 1 << 17
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
lib = os.path.join(ROOT, "devbuild", "libblance_prof.so")
pl = hip.Planner(lib_path=lib if os.path.exists(lib) else None)
fp1 = synth.config5_initial(P, N)
r1 = pl.plan(fp1)
print("initial  : sweeps %d  device %.1f ms  bulk %d of %d" % (r1.iterations, r1.struct.device_ms, r1.struct.steps_batched, r1.struct.steps_total), flush=True)
fp2 = synth.config5_rebalance(fp1, r1, P, N)
r2 = pl.plan(fp2)
print("rebalance: sweeps %d  device %.1f ms  bulk %d of %d" % (r2.iterations, r2.struct.device_ms, r2.struct.steps_batched, r2.struct.steps_total), flush=True)
