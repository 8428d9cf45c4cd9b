#!/usr/bin/env python3
"""Summaries of rocprofv3 runs (rocpd SQLite databases) for profiles/:
    profile_summary.py trace <results.db> <calls> <out.txt> <title>
    profile_summary.py pmc   <out.txt> <out.json|-> <title> <db> [<db> ...]
`pmc` sums every counter found in the given databases per kernel (one row per kernel, one column per
counter) -- counters collected in separate passes simply come from separate databases."""
import hashlib
import json
import os
import re
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def kernel_sources():
    """Every source of libblance_hip.so (all of blance_amd/csrc/*.h and *.hip: a list kept by hand missed k_queue_walk.h once)."""
    import glob
    d = os.path.join(ROOT, "blance_amd", "csrc")
    return sorted(os.path.relpath(f, ROOT) for f in glob.glob(os.path.join(d, "*.h")) + glob.glob(os.path.join(d, "*.hip")))


def source_hash():
    h = hashlib.sha256()
    for f in kernel_sources():
        with open(os.path.join(ROOT, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def tables(cur):
    t = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    return lambda k: [x for x in t if k in x][0]


def short(name):
    name = re.sub(r"\(.*", "", name)
    name = re.sub(r"^_ZN6blance\d+", "", name)
    return name[:60]


def trace(db, calls, out, title):
    cur = sqlite3.connect(db).cursor()
    g = tables(cur)
    rows = cur.execute(
        "select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start) "
        "from %s d join %s s on d.kernel_id=s.id group by s.kernel_name order by 3 desc" % (g("kernel_dispatch"), g("kernel_symbol"))).fetchall()
    tot = sum(r[2] for r in rows)
    lines = [title, "kernel sources sha256[:16] %s" % source_hash(),
             "total kernel time %.3f ms (%.3f ms per PlanNextMap call over %d calls)" % (tot / 1e6, tot / 1e6 / calls, calls),
             "%-62s %6s %12s %12s %12s %12s %7s" % ("kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "share")]
    for r in rows:
        lines.append("%-62s %6d %12.3f %12.1f %12.1f %12.1f %6.1f%%" % (short(r[0]), r[1], r[2] / 1e6, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / tot))
    # where the device idles between kernels (host round trips, launch latency): gaps between consecutive dispatches
    try:
        disp = cur.execute("select d.start, d.end from %s d order by d.start" % g("kernel_dispatch")).fetchall()
        gaps, end = [], None
        for s0, e0 in disp:
            if end is not None and s0 > end:
                gaps.append(s0 - end)
            end = e0 if end is None or e0 > end else end
        lines.append("")
        lines.append("device idle between consecutive kernels (a gap = the next kernel starts after everything before it has ended), %d calls:" % calls)
        for lo, hi, what in ((0, 5e3, "under 5 us: launch back to back"), (5e3, 2e4, "5 - 20 us"), (2e4, 1e5, "20 - 100 us: a host round trip inside a call"),
                             (1e5, 2e6, "0.1 - 2 ms: between calls (the host's work between two blance_plan_resident), uploads"),
                             (2e6, 1e18, "over 2 ms")):
            sel = [x for x in gaps if lo <= x < hi]
            lines.append("   %-92s %5d gaps, %9.3f ms in total, %7.3f ms per call" % (what, len(sel), sum(sel) / 1e6, sum(sel) / 1e6 / calls))
    except Exception as e:                                   # (older databases: no timestamps table layout we know)
        lines.append("(no gap statistics: %s)" % str(e)[:80])
    # kernels that ran BESIDE a longer one (the second stream: k_stay_by_top's work list made beside the chain kernel)
    try:
        disp = cur.execute("select s.kernel_name, d.start, d.end from %s d join %s s on d.kernel_id=s.id order by d.start"
                           % (g("kernel_dispatch"), g("kernel_symbol"))).fetchall()
        inside = {}
        for i, (n0, s0, e0) in enumerate(disp):
            if e0 - s0 < 100e3:
                continue                                     # hosts of at least 0.1 ms
            for n1, s1, e1 in disp[i + 1:]:
                if s1 >= e0:
                    break
                if e1 <= e0:
                    k = (short(n0), short(n1))
                    v = inside.setdefault(k, [0, 0.0])
                    v[0] += 1
                    v[1] += e1 - s1
        lines.append("")
        lines.append("kernels that started and ended while a longer kernel (>= 0.1 ms) was running -- another stream of the same context:")
        for (n0, n1), (cnt, ns) in sorted(inside.items(), key=lambda kv: -kv[1][1]):
            lines.append("   inside %-50s %-50s %4d launches, %8.3f ms" % (n0[:50], n1[:50], cnt, ns / 1e6))
        if not inside:
            lines.append("   (none)")
    except Exception as e:
        lines.append("(no overlap statistics: %s)" % str(e)[:80])
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:12]))


def pmc(out, out_json, title, dbs):
    per = {}            # kernel -> {counter: sum, "calls": n}
    counters = []
    for db in dbs:
        cur = sqlite3.connect(db).cursor()
        g = tables(cur)
        rows = cur.execute(
            "select s.kernel_name, i.name, count(distinct d.id), sum(e.value) from %s e join %s i on e.pmc_id=i.id join %s d on "
            "e.event_id=d.event_id join %s s on d.kernel_id=s.id group by s.kernel_name, i.name"
            % (g("pmc_event"), g("info_pmc"), g("kernel_dispatch"), g("kernel_symbol"))).fetchall()
        for k, c, n, v in rows:
            per.setdefault(k, {"calls": n})[c] = v
            per[k]["calls"] = max(per[k]["calls"], n)
            if c not in counters:
                counters.append(c)
    order = sorted(per, key=lambda k: -max(per[k].get(c, 0) for c in counters))
    lines = [title, "kernel sources sha256[:16] %s" % source_hash(),
             "%-62s %6s " % ("kernel", "calls") + " ".join("%18s" % c for c in counters)]
    for k in order:
        lines.append("%-62s %6d " % (short(k), per[k]["calls"]) + " ".join("%18.0f" % per[k].get(c, 0) for c in counters))
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:10]))
    if out_json != "-":
        m = re.search(r"\((\d+) PlanNextMap call", title)
        json.dump({"source_hash": source_hash(), "title": title, "plan_calls": int(m.group(1)) if m else 1,
                   "git_head": subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip(),
                   "kernels": {short(k): per[k] for k in order}}, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    if sys.argv[1] == "trace":
        trace(sys.argv[2], int(sys.argv[3]), sys.argv[4], sys.argv[5])
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5:])
