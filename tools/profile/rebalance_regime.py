#!/usr/bin/env python3
"""Developer aid: config 3's plan rebalanced after every tenth node left (bench.py's general_regime workload a) at P x N, with
the driver's trace (BLANCE_TRACE=1) / the queue kernel's statistics (BLANCE_QUEUE_STATS=1) when those are set; through
devbuild/libblance_prof.so when it exists and BLANCE_DEV_PROF=1.
    python tools/profile/rebalance_regime.py [P N [calls]]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from blance_amd import hip, synth          # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 2 else 1 << 20
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 1
lib = os.path.join(ROOT, "devbuild", "libblance_prof.so")
lib = lib if (os.environ.get("BLANCE_DEV_PROF") and os.path.exists(lib)) else None
trace = os.environ.pop("BLANCE_TRACE", None)          # the headline plan itself: quiet (the trace flag is read when a context is made)
stats = os.environ.pop("BLANCE_QUEUE_STATS", None)
quiet = hip.Planner()
fp = synth.config_flat(3, P, N)
res = quiet.plan(fp)
print("headline plan: sweeps %d  device %.2f ms" % (res.iterations, res.struct.device_ms), flush=True)
fp2 = synth.config3_rebalance_flat(fp, res)
quiet.close()
if trace:
    os.environ["BLANCE_TRACE"] = trace
if stats:
    os.environ["BLANCE_QUEUE_STATS"] = stats
pl = hip.Planner(lib_path=lib)
for i in range(calls):
    t = time.time()
    r = pl.plan(fp2)
    print("rebalance: sweeps %d  device %.1f ms  pass kernels %.1f ms  flat passes %.1f ms  bulk %d of %d  (%.2f s)" % (
        r.iterations, r.struct.device_ms, r.struct.pass_kernel_ms, r.struct.flat_pass_ms, r.struct.steps_batched, r.struct.steps_total,
        time.time() - t), flush=True)
