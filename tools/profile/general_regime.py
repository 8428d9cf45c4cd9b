#!/usr/bin/env python3
"""Developer aid: config 3 with scrambled names + Zipf partition weights (bench.py's general_regime workload b) at P x N
through devbuild/libblance_prof.so (or the library BLANCE_DEV_LIB names) when it exists (per-launch statistics and phase clocks of k_pass_queue), BLANCE_TRACE style;
BLANCE_DEV_PRODUCT=1: through the product library.
    python tools/profile/general_regime.py [P N]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from blance_amd import hip, synth          # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 2 else 1 << 20
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
lib = os.environ.get("BLANCE_DEV_LIB") or os.path.join(ROOT, "devbuild", "libblance_prof.so")
pl = hip.Planner(lib_path=lib if (os.path.exists(lib) and not os.environ.get("BLANCE_DEV_PRODUCT")) else None)
fp = synth.config3_named_weighted_flat(P, N)
t = time.time()
r = pl.plan(fp)
print("named+weighted: sweeps %d  device %.1f ms  bulk %d of %d  (%.1f s)" % (r.iterations, r.struct.device_ms, r.struct.steps_batched, r.struct.steps_total, time.time() - t), flush=True)
