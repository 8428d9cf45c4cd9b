#!/usr/bin/env python3
"""Developer aid: blance_upload / blance_download of BASELINE config 3 timed on the device -- pageable arrays (staged) and
page-locked ones (blance_host_alloc); the first call of each (allocations) apart from the later ones.
    python tools/dev_transfers.py [P N]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from blance_amd import hip, synth          # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 2 else 1 << 20
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
fp = synth.config_flat(3, P=P, N=N)
pl = hip.Planner()
nbytes = sum(a.nbytes for a in fp.arrays.values())


def timed(label, fp, arena):
    ups, downs = [], []
    res = None
    for i in range(4):
        t = time.perf_counter()
        pl.upload(fp)
        ups.append((time.perf_counter() - t) * 1e3)
        r = pl.plan_resident()
        t = time.perf_counter()
        res = pl.download(arena, into=res)           # (the first call allocates the result arrays, the others reuse them)
        downs.append((time.perf_counter() - t) * 1e3)
    out_bytes = res.out_off.nbytes + res.out_nodes.nbytes + res.out_kind.nbytes
    print("%s: upload %.1f MB: first %.2f ms, then %s ms | download %.1f MB: first (allocates the result arrays) %.2f ms, then %s ms | device %.2f ms" % (
        label, nbytes / 1e6, ups[0], " ".join("%.2f" % x for x in ups[1:]), out_bytes / 1e6, downs[0],
        " ".join("%.2f" % x for x in downs[1:]), r.device_ms), flush=True)
    return res.digest()


d1 = timed("pageable   ", fp, None)
arena = hip.HostArena()
t = time.perf_counter()
fpp = synth.config_flat(3, P=P, N=N).pin(arena)
print("pinning the problem's arrays: %.1f ms" % ((time.perf_counter() - t) * 1e3))
d2 = timed("page-locked", fpp, arena)
print("digests equal:", d1 == d2)
# the whole call, host buffers in, host buffers out
from blance_amd import abi                  # noqa: E402
for label, f, ar in (("pageable", fp, None), ("page-locked", fpp, arena)):
    res = abi.FlatResult(f, ar)
    for i in range(4):
        t = time.perf_counter()
        r = pl.plan_into(f, res)
        print("blance_plan %s, buffers reused: %.2f ms (device %.2f)" % (label, (time.perf_counter() - t) * 1e3, r.struct.device_ms), flush=True)
