#!/usr/bin/env python3
"""BASELINE.json config 5 at its named size (1,048,576 partitions x 4,096 nodes, flat, Zipf
partition weights, node weights, stickiness; 410 nodes removed + 410 added): the CPU oracle's
digests of the initial plan over the old nodes and of the rebalance from that plan, stored in
tests/golden/config_digests.json so the GPU test can check bit-parity at full size without the
oracle's minutes.  Usage: python tests/tools/make_config5_digest.py [P N]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from blance_amd import synth          # noqa: E402
from oracle import loader             # noqa: E402


def main():
    P = int(sys.argv[1]) if len(sys.argv) > 2 else 1 << 20
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    t = time.time()
    fp1 = synth.config5_initial(P, N)
    t1 = time.time()
    r1 = loader.plan(fp1)
    t2 = time.time()
    fp2 = synth.config5_rebalance(fp1, r1, P, N)
    t3 = time.time()
    r2 = loader.plan(fp2)
    t4 = time.time()
    entry = {"partitions": P, "nodes": N,
             "initial": {"iterations": r1.iterations, "warnings": r1.n_warnings, "digest": r1.digest(),
                         "oracle_seconds": round(t2 - t1, 1)},
             "rebalance": {"iterations": r2.iterations, "warnings": r2.n_warnings, "digest": r2.digest(),
                           "oracle_seconds": round(t4 - t3, 1)},
             "build_seconds": [round(t1 - t, 1), round(t3 - t2, 1)]}
    print(entry, flush=True)
    if P == 1 << 20 and N == 4096:
        path = os.path.join(ROOT, "tests", "golden", "config_digests.json")
        with open(path) as f:
            d = json.load(f)
        d["config5"] = entry
        with open(path, "w") as f:
            json.dump(d, f, indent=1)


if __name__ == "__main__":
    main()
