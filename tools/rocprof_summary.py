#!/usr/bin/env python3
"""Per-kernel summary (calls, total, average, share) of a rocprofv3
--kernel-trace run, read from the rocpd SQLite database it writes
(<dir>/<name>_results.db).  Usage: tools/rocprof_summary.py <results.db> [calls_per_step]"""
import re
import sqlite3
import sys


def summarize(path, per=1):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tables = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [x for x in tables if "kernel_dispatch" in x][0]
    ks = [x for x in tables if "kernel_symbol" in x][0]
    rows = cur.execute(
        "select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), "
        "max(d.end-d.start) from %s d join %s s on d.kernel_id=s.id group by s.kernel_name order by 3 desc"
        % (kd, ks)).fetchall()
    tot = sum(r[2] for r in rows)
    out = ["total kernel time %.3f ms (%.3f ms per PlanNextMap call over %d calls)" % (tot / 1e6, tot / 1e6 / per, per),
           "%-72s %6s %12s %12s %12s %12s %7s" % ("kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "share")]
    for r in rows:
        name = re.sub(r"\(.*", "", r[0])[:72]
        out.append("%-72s %6d %12.3f %12.1f %12.1f %12.1f %6.1f%%"
                   % (name, r[1], r[2] / 1e6, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / tot))
    return "\n".join(out)


if __name__ == "__main__":
    print(summarize(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 1))
