#!/usr/bin/env python3
"""Turns a rocprofv3 result database (rocpd sqlite, ROCm 7.2 default output) into the text summary
that is committed under profiles/: per-kernel stats (calls, total/avg/min/max ns, %) and, if PMC
counters were collected, per-kernel per-dispatch-average counter values.
usage: rocprof_summary.py <results.db> [kernel-name-filter]"""
import sqlite3, sys, collections

def main():
    con = sqlite3.connect(sys.argv[1])
    filt = sys.argv[2] if len(sys.argv) > 2 else ""
    rows = con.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall() \
        if "duration" not in [r[1] for r in con.execute("pragma table_info(kernels)")] else \
        con.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    print("## kernel-trace stats (ns)")
    print("%-70s %6s %14s %14s %12s %12s %7s" % ("kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct"))
    for name, n, s, a, lo, hi in rows:
        if filt in name:
            print("%-70s %6d %14.0f %14.1f %12.0f %12.0f %6.2f%%" % (name[:70], n, s, a, lo, hi, 100.0 * s / tot))
    try:
        pm = con.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name order by 1,2").fetchall()
    except sqlite3.Error:
        pm = []
    if pm:
        print("\n## PMC counters (average per dispatch)")
        for k, c, v, n in pm:
            if filt in k:
                print("%-60s %-22s %18.1f  (%d dispatches)" % (k[:60], c, v, n))

if __name__ == "__main__":
    main()
