#!/usr/bin/env python3
"""Developer aid: in a rocprofv3 kernel-trace database, print the dispatches around the slowest dispatch of a kernel
whose name contains <pattern>:  dev_trace_neighbors.py <results.db> <pattern> [n]"""
import sqlite3, sys
db, pat = sys.argv[1], sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4
cur = sqlite3.connect(db).cursor()
t = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
g = lambda k: [x for x in t if k in x][0]
rows = cur.execute("select d.start, d.end, s.kernel_name, d.grid_size_x, d.workgroup_size_x from %s d join %s s on d.kernel_id=s.id order by d.start" % (g("kernel_dispatch"), g("kernel_symbol"))).fetchall()
idx = max((i for i, r in enumerate(rows) if pat in r[2]), key=lambda i: rows[i][1] - rows[i][0])
for i in range(max(0, idx - n), min(len(rows), idx + n + 1)):
    r = rows[i]
    print("%s %9.1f us  grid %8d x %4d  %s" % ("->" if i == idx else "  ", (r[1] - r[0]) / 1e3, r[3], r[4], r[2][:70]))
