#!/usr/bin/env python3
"""Throughput of blance_calc_moves (CalcPartitionMoves for every partition, moves.go:41-136) on the GPU box:
begMap = config 3's plan, endMap = the same plan with every partition's replicas rotated and a tenth of the
primaries swapped with a replica (adds, deletes, promotions and demotions).  python tools/moves_bench.py [P]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from blance_amd import hip, synth          # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
pl = hip.Planner()
res = pl.plan(synth.config_flat(3, P=P, N=4096))
beg = np.array(res.out_nodes[:3 * P], dtype=np.int32).reshape(P, 3)          # primary, replica, replica
end = beg.copy()
end[:, 1], end[:, 2] = beg[:, 2], (beg[:, 1] + 1) % 4096                       # one replica kept (reordered), one moved
swap = np.arange(P) % 10 == 0
end[swap, 0], end[swap, 1] = beg[swap, 1], beg[swap, 0]                        # promotion + demotion
M = 2


def csr(a):
    off = np.zeros(P * (M + 1) + 1, dtype=np.int32)
    lens = np.zeros((P, M + 1), dtype=np.int32)
    lens[:, 0], lens[:, 1] = 1, 2
    off[1:] = np.cumsum(lens.reshape(-1))
    return off, a.reshape(-1).astype(np.int32)


bo, bn = csr(beg)
eo, en = csr(end)
for favor in (False, True):
    pl.calc_moves(M, favor, bo, bn, eo, en)                                     # warm-up
    t0 = time.perf_counter()
    op_off, _, _, _, dev_ms = pl.calc_moves(M, favor, bo, bn, eo, en)
    dt = time.perf_counter() - t0
    print("favorMinNodes=%s: %d partitions, %d moves, kernel %.3f ms, call (H2D + kernel + D2H + compaction) %.1f ms "
          "-> %.1f M partitions/s per call" % (favor, P, int(op_off[-1]), dev_ms, dt * 1e3, P / dt / 1e6))
