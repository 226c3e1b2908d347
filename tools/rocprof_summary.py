#!/usr/bin/env python
"""Turn a rocprofv3 (ROCm 7.2, rocpd sqlite) result DB into a small text summary for profiles/.

    python tools/rocprof_summary.py gpurun_out/prof_ls/ls_results.db [title] > profiles/r01_ls_kernel_stats.txt
"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    title = sys.argv[2] if len(sys.argv) > 2 else db
    con = sqlite3.connect(db)
    print("# %s" % title)
    print("# source: rocprofv3 result db %s" % db)
    rows = con.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    if rows:
        print("\n## kernel-trace --stats (durations in microseconds)")
        print("%-60s %8s %14s %12s %8s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
        for name, calls, tot, avg, pct in rows:
            print("%-60s %8d %14.1f %12.1f %8.3f" % (name.split("(")[0][-60:], calls, tot, avg, pct))
    try:
        rows = con.execute(
            "select kernel_name, counter_name, sum(value), count(*), avg(duration) from counters_collection "
            "group by kernel_name, counter_name order by kernel_name, counter_name").fetchall()
    except sqlite3.Error:
        rows = []
    if rows:
        print("\n## PMC counters (value summed over dispatches and over XCDs/SEs as rocprofv3 reports them)")
        print("%-44s %-24s %20s %10s %14s" % ("kernel", "counter", "sum", "dispatches", "avg_kernel_ns"))
        for k, c, v, n, d in rows:
            print("%-44s %-24s %20.0f %10d %14.0f" % (k.split("(")[0][-44:], c, v, n, d or 0))


if __name__ == "__main__":
    main()
