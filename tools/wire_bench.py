#!/usr/bin/env python3
"""Throughput of the PartitionMap JSON codec (include/blance_wire.h) on a planner-sized map:
P partitions, primary + 2 replicas over N nodes (config 3's result shape), host only.
    python tools/wire_bench.py [P] [N]
Prints MB/s of decode (bytes -> interned arrays) and encode (arrays -> bytes), with Python's
json module (C accelerated, builds objects, no interning) beside it for scale."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np                        # noqa: E402
from blance_amd import wire               # noqa: E402


def main():
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    keys = [b"%d" % i for i in range(P)]
    nodes = [b"n%04d" % i for i in range(N)]
    i = np.arange(P, dtype=np.int64)
    entry_nodes = np.empty(3 * P, dtype=np.int32)
    entry_nodes[0::3] = i % N
    entry_nodes[1::3] = (i * 7 + 16) % N
    entry_nodes[2::3] = (i * 13 + 32) % N
    entry_state = np.tile(np.array([0, 1], dtype=np.int32), P)
    entry_off = np.empty(2 * P + 1, dtype=np.int64)
    entry_off[0::2] = 3 * np.arange(P + 1)
    entry_off[1::2] = 3 * np.arange(P) + 1
    t0 = time.perf_counter()
    doc = wire.encode_arrays(keys, keys, np.full(P, wire.LIST, np.uint8), 2 * np.arange(P + 1), [b"primary", b"replica"],
                             nodes, entry_state, np.full(2 * P, wire.LIST, np.uint8), entry_off, entry_nodes)
    mb = len(doc) / 1e6
    print("document: %d partitions, %.1f MB" % (P, mb))
    best_d = best_e = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        m = wire.decode(doc)
        best_d = min(best_d, time.perf_counter() - t0)
        t0 = time.perf_counter()
        out = m.encode()
        best_e = min(best_e, time.perf_counter() - t0)
        assert out == doc
        m.close()
    print("blance_wire_decode: %.3f s  %.0f MB/s   (bytes -> interned CSR, %d node names, %d states)"
          % (best_d, mb / best_d, N, 2))
    print("blance_wire_encode: %.3f s  %.0f MB/s   (sorted-key check, escape scan, one output buffer)" % (best_e, mb / best_e))
    t0 = time.perf_counter()
    obj = json.loads(doc)
    t_l = time.perf_counter() - t0
    t0 = time.perf_counter()
    json.dumps(obj, separators=(",", ":"))
    t_s = time.perf_counter() - t0
    print("python json.loads : %.3f s  %.0f MB/s   json.dumps: %.3f s  %.0f MB/s" % (t_l, mb / t_l, t_s, mb / t_s))


if __name__ == "__main__":
    main()
