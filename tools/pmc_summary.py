#!/usr/bin/env python3
"""HBM traffic per kernel from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; each run
with --kernel-trace only).  Writes profiles/r1_pmc_hbm_traffic.txt and r1_pmc_traffic.json.
Usage: tools/pmc_summary.py <fetch results.db> <write results.db>"""
import json
import os
import re
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(path, counter):
    cur = sqlite3.connect(path).cursor()
    t = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    g = lambda k: [x for x in t if k in x][0]
    kd, ks, pe, ip = g("kernel_dispatch"), g("kernel_symbol"), g("pmc_event"), g("info_pmc")
    return cur.execute(
        "select s.kernel_name, count(*), sum(e.value) from %s e join %s i on e.pmc_id=i.id join %s d on "
        "e.event_id=d.event_id join %s s on d.kernel_id=s.id where i.name='%s' group by s.kernel_name order by 3 desc"
        % (pe, ip, kd, ks, counter)).fetchall()


def main():
    f, w = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    wd = {r[0]: r for r in w}
    lines = ["rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, each with --kernel-trace only) -- "
             "python bench.py --steps 1 --warmup 0 --no-cpu-baseline",
             "config 3, 1,048,576 partitions x 4,096 nodes, one PlanNextMap call (3 sweeps).  Counter unit: KB as reported.",
             "gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reads half of a wide coalesced stream's bytes -> "
             "doubled below; WRITE_SIZE uncalibrated, taken as is.", "",
             "%-66s %6s %14s %14s" % ("kernel", "calls", "FETCH_SIZE_KB", "WRITE_SIZE_KB")]
    cf = cw = cc = tf = tw = 0
    for r in f:
        ww = wd.get(r[0])
        wv = ww[2] if ww else 0.0
        tf += r[2]
        tw += wv
        lines.append("%-66s %6d %14.1f %14.1f" % (re.sub(r"\(.*", "", r[0])[:66], r[1], r[2], wv))
        if "k_pass_chain" in r[0]:
            cf += r[2]
            cw += wv
            cc += r[1]
    lines.append("%-66s %6s %14.1f %14.1f" % ("ALL KERNELS", "", tf, tw))
    per = (2 * cf + cw) * 1024 / max(cc, 1)
    alg = 1048576 * (4096 * (16 + 4 * 2) + 40)
    lines += ["", "k_pass_chain (all variants): %d launches, FETCH %.1f KB (x2 = %.1f MB), WRITE %.1f KB -> %.1f MB of HBM "
              "traffic per launch" % (cc, cf, 2 * cf / 1024, cw, per / 1e6),
              "algorithmic bytes per launch (SURVEY.md 8d, replica pass): %.0f MB -> measured traffic / algorithmic = %.5f"
              % (alg / 1e6, per / alg),
              "whole call: FETCH x2 + WRITE = %.1f MB" % ((2 * tf + tw) * 1024 / 1e6)]
    open(os.path.join(ROOT, "profiles", "r1_pmc_hbm_traffic.txt"), "w").write("\n".join(lines) + "\n")
    json.dump({"kernel": "k_pass_chain", "launches": cc, "fetch_size_kb": cf, "write_size_kb": cw,
               "hbm_bytes_per_launch": per, "correction": "FETCH_SIZE x2 (gfx950), WRITE_SIZE as reported",
               "source": "profiles/r1_pmc_hbm_traffic.txt", "workload": "config 3, 1048576 x 4096"},
              open(os.path.join(ROOT, "profiles", "r1_pmc_traffic.json"), "w"), indent=1)
    print("\n".join(lines[-3:]))


if __name__ == "__main__":
    main()
