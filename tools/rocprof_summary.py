#!/usr/bin/env python
"""Turn a rocprofv3 (ROCm 7.2, rocpd sqlite) result DB into a small text summary for profiles/.

    python tools/rocprof_summary.py gpurun_out/prof_ls/ls_results.db [title] [--skip-frac F] > profiles/r01_ls_kernel_stats.txt

--timeline N: the last N kernel dispatches in launch order with start offset, duration and the idle gap before each (all in
microseconds) — shows host-side bubbles between launches that the per-kernel sums hide.
--skip-frac F (e.g. 0.25): a second table over the dispatches that START after the first fraction F of the traced time span
— the steady state without the cold warm-up launches, which is what the bench line's per-step time corresponds to.
"""
import sqlite3
import sys


def main():
    argv = list(sys.argv[1:])
    skip = 0.0
    if "--skip-frac" in argv:
        i = argv.index("--skip-frac")
        skip = float(argv[i + 1])
        del argv[i:i + 2]
    timeline = 0
    if "--timeline" in argv:
        i = argv.index("--timeline")
        timeline = int(argv[i + 1])
        del argv[i:i + 2]
    db = argv[0]
    title = argv[1] if len(argv) > 1 else db
    con = sqlite3.connect(db)
    print("# %s" % title)
    print("# source: rocprofv3 result db %s" % db)
    rows = con.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    if rows:
        print("\n## kernel-trace --stats (durations in microseconds)")
        print("%-60s %8s %14s %12s %8s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
        for name, calls, tot, avg, pct in rows:
            print("%-60s %8d %14.1f %12.1f %8.3f" % (name.split("(")[0][-60:], calls, tot, avg, pct))
    if skip > 0.0:
        t0, t1 = con.execute("select min(start), max(end) from kernels").fetchone()
        cut = t0 + skip * (t1 - t0)
        rows = con.execute("select name, count(*), sum(end - start) / 1000.0, avg(end - start) / 1000.0 from kernels "
                           "where start >= ? group by name order by 3 desc", (cut,)).fetchall()
        tot_all = sum(r[2] for r in rows) or 1.0
        print("\n## steady state: dispatches starting after the first %.0f %% of the traced span (durations in microseconds)" % (100 * skip))
        print("%-60s %8s %14s %12s %8s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
        for name, calls, tot, avg in rows:
            print("%-60s %8d %14.1f %12.1f %8.3f" % (name.split("(")[0][-60:], calls, tot, avg, 100.0 * tot / tot_all))
    if timeline > 0:
        rows = con.execute("select name, start, end from kernels order by start").fetchall()[-timeline:]
        print("\n## timeline of the last %d dispatches (microseconds; gap = idle time since the previous dispatch ended)" % len(rows))
        print("%-60s %12s %12s %10s" % ("kernel", "start_us", "dur_us", "gap_us"))
        base, prev_end = rows[0][1], rows[0][1]
        busy = gap_tot = 0.0
        for name, st, en in rows:
            gap = max(0.0, (st - prev_end) / 1000.0)
            print("%-60s %12.1f %12.1f %10.1f" % (name.split("(")[0][-60:], (st - base) / 1000.0, (en - st) / 1000.0, gap))
            busy += (en - st) / 1000.0
            gap_tot += gap
            prev_end = max(prev_end, en)
        print("# span %.1f us, kernels %.1f us, idle gaps %.1f us" % ((prev_end - base) / 1000.0, busy, gap_tot))
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
