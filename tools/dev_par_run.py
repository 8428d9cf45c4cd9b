#!/usr/bin/env python3
"""developer aid: config 5 (or P N) on the GPU box with the product library -- timings, digests against
tests/golden/config_digests.json -- and, with --stats, once more with devbuild/libblance_parstats.so
(tools/dev_build_par_stats.sh), whose k_pass_par prints its counters per launch.
    python tools/dev_par_run.py [P N] [--stats] [--tree nopar]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from blance_amd import hip, synth          # noqa: E402


def run(pl, P, N, want, label):
    fp1 = synth.config5_initial(P, N)
    r1 = pl.plan(fp1)
    print("%s initial  : sweeps %d  device %.1f ms  verified/bulk %d of %d steps" % (
        label, r1.iterations, r1.struct.device_ms, r1.struct.steps_batched, r1.struct.steps_total), flush=True)
    fp2 = synth.config5_rebalance(fp1, r1, P, N)
    r2 = pl.plan(fp2)
    print("%s rebalance: sweeps %d  device %.1f ms  verified/bulk %d of %d steps" % (
        label, r2.iterations, r2.struct.device_ms, r2.struct.steps_batched, r2.struct.steps_total), flush=True)
    print("%s digests %s %s" % (label, r1.digest()[:16], r2.digest()[:16]))
    if want:
        print("%s initial matches the oracle digest  :" % label, r1.digest() == want["initial"]["digest"])
        print("%s rebalance matches the oracle digest:" % label, r2.digest() == want["rebalance"]["digest"], flush=True)
    return r1.digest(), r2.digest()


def main():
    a = [x for x in sys.argv[1:] if not x.startswith("--")]
    P, N = (int(a[0]), int(a[1])) if len(a) >= 2 else (1 << 20, 4096)
    tree = sys.argv[sys.argv.index("--tree") + 1] if "--tree" in sys.argv else "auto"
    a = [x for x in a if x != tree]
    with open(os.path.join(ROOT, "tests", "golden", "config_digests.json")) as f:
        want = json.load(f).get("config5") if (P, N) == (1 << 20, 4096) else None
    t = time.time()
    pl = hip.Planner(tree=tree)
    d = run(pl, P, N, want, "[product %s]" % tree)
    pl.close()
    if "--compare" in sys.argv:
        pl = hip.Planner(tree="nopar")
        d2 = run(pl, P, N, want, "[product nopar]")
        print("same result with and without k_pass_par:", d == d2)
        pl.close()
    if "--stats" in sys.argv:
        pl = hip.Planner(lib_path=os.path.join(ROOT, "devbuild", "libblance_parstats.so"), tree=tree)
        run(pl, P, N, want, "[stats]")
        pl.close()
    print("total %.0f s" % (time.time() - t))


if __name__ == "__main__":
    main()
