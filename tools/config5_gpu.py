#!/usr/bin/env python3
"""BASELINE.json config 5 at its named size on the GPU box: the initial plan over the old nodes,
then the rebalance from it; both digests against tests/golden/config_digests.json (made by
tests/tools/make_config5_digest.py with the CPU oracle, ~8 minutes each).
    python tools/config5_gpu.py [P N]      (other sizes: timings only)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from blance_amd import hip, synth          # noqa: E402


def main():
    P = int(sys.argv[1]) if len(sys.argv) > 2 else 1 << 20
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    with open(os.path.join(ROOT, "tests", "golden", "config_digests.json")) as f:
        want = json.load(f).get("config5") if (P, N) == (1 << 20, 4096) else None
    pl = hip.Planner()
    t = time.time()
    fp1 = synth.config5_initial(P, N)
    print("built initial problem in %.1f s" % (time.time() - t), flush=True)
    r1 = pl.plan(fp1)
    print("initial  : sweeps %d  device %.1f ms  verified/bulk %d of %d steps  %.2f M assignments/s" % (
        r1.iterations, r1.struct.device_ms, r1.struct.steps_batched, r1.struct.steps_total,
        synth.assignments(fp1) / r1.struct.device_ms / 1e3), flush=True)
    t = time.time()
    fp2 = synth.config5_rebalance(fp1, r1, P, N)
    print("built rebalance problem in %.1f s" % (time.time() - t), flush=True)
    r2 = pl.plan(fp2)
    print("rebalance: sweeps %d  device %.1f ms  verified/bulk %d of %d steps  %.2f M assignments/s" % (
        r2.iterations, r2.struct.device_ms, r2.struct.steps_batched, r2.struct.steps_total,
        synth.assignments(fp2) / r2.struct.device_ms / 1e3), flush=True)
    if want:
        ok1 = r1.digest() == want["initial"]["digest"] and r1.iterations == want["initial"]["iterations"]
        ok2 = r2.digest() == want["rebalance"]["digest"] and r2.iterations == want["rebalance"]["iterations"]
        print("initial matches the oracle digest  :", ok1)
        print("rebalance matches the oracle digest:", ok2)
        print("CPU oracle: %.0f s and %.0f s on one core" % (want["initial"]["oracle_seconds"], want["rebalance"]["oracle_seconds"]))
        return 0 if ok1 and ok2 else 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
