#!/usr/bin/env python
"""Turns a rocprofv3 results database (rocprofv3 --kernel-trace [--stats] [--pmc ...] -d DIR -o NAME)
into the plain-text per-kernel summary that is committed under profiles/.

  python tools/prof_summary.py gpurun_out/prof1/r01_results.db > profiles/r01_kernel_stats.txt
"""
import sqlite3
import sys


def main(path):
    con = sqlite3.connect(path)
    cur = con.cursor()
    rows = cur.execute(
        "select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start), "
        "max(vgpr_count), max(sgpr_count), max(lds_size), max(grid_x), max(grid_y), max(workgroup_x) "
        "from kernels group by name order by 6 desc").fetchall()
    total = sum(r[5] for r in rows) or 1
    print("# rocprofv3 kernel-trace summary of %s" % path)
    print("%-72s %7s %12s %12s %12s %10s %6s %5s %5s %7s %12s" % (
        "kernel", "calls", "avg_ns", "min_ns", "max_ns", "total_ms", "pct", "vgpr", "sgpr", "lds", "grid"))
    for r in rows:
        name = r[0].replace("rmclhip::(anonymous namespace)::", "").replace("void ", "")
        print("%-72s %7d %12.1f %12.1f %12.1f %10.3f %6.2f %5s %5s %7s %12s" % (
            name[:72], r[1], r[2], r[3], r[4], r[5] / 1e6, 100.0 * r[5] / total, r[6], r[7], r[8],
            "%sx%s/%s" % (r[9], r[10], r[11])))
    try:
        pm = cur.execute(
            "select k.name, p.counter_name, avg(p.counter_value), count(*) from pmc_events p join kernels k "
            "on p.dispatch_id = k.dispatch_id group by k.name, p.counter_name order by k.name").fetchall()
    except sqlite3.Error:
        pm = []
    if pm:
        print("\n# PMC counters (average per dispatch)")
        for name, cname, val, n in pm:
            name = name.replace("rmclhip::(anonymous namespace)::", "").replace("void ", "")
            print("%-72s %-28s %18.1f  (n=%d)" % (name[:72], cname, val, n))


if __name__ == "__main__":
    main(sys.argv[1])
