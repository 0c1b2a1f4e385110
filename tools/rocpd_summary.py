#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd SQLite outputs into the text files kept under profiles/.

    tools/rocpd_summary.py <dir with trace/, pmc_fetch/, pmc_write/ subdirs> [> profiles/rNN_x.txt]
"""
import glob
import os
import sqlite3
import sys


def db_of(d):
    f = glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)
    return f[0] if f else None


def kernel_stats(db):
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                     "from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    out = ["%-10s %-12s %-10s %-10s %-10s %-6s  %s" % ("calls", "total_us", "avg_us", "min_us", "max_us", "%", "kernel")]
    for name, n, s, a, mn, mx in rows:
        out.append("%-10d %-12.1f %-10.3f %-10.3f %-10.3f %-6.2f  %s" % (n, s / 1e3, a / 1e3, mn / 1e3, mx / 1e3,
                                                                      100.0 * s / tot, name[:140]))
    return "\n".join(out)


def pmc_stats(db):
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, counter_name, count(*), avg(value), min(value), max(value) "
                     "from counters_collection group by kernel_name, counter_name order by kernel_name").fetchall()
    out = ["%-12s %-8s %-14s %-14s %-14s  %s" % ("counter", "calls", "avg", "min", "max", "kernel")]
    for k, cn, n, a, mn, mx in rows:
        out.append("%-12s %-8d %-14.3f %-14.3f %-14.3f  %s" % (cn, n, a, mn, mx, k[:120]))
    return "\n".join(out)


def main():
    root = sys.argv[1]
    for sub, fn, title in (("trace", kernel_stats, "rocprofv3 --kernel-trace --stats (durations)"),
                           ("pmc_fetch", pmc_stats, "rocprofv3 --pmc FETCH_SIZE (KB per dispatch, as reported)"),
                           ("pmc_write", pmc_stats, "rocprofv3 --pmc WRITE_SIZE (KB per dispatch, as reported)")):
        d = os.path.join(root, sub)
        db = db_of(d) if os.path.isdir(d) else None
        if not db:
            continue
        print("== %s ==" % title)
        print(fn(db))
        print()
    for log in sorted(glob.glob(os.path.join(root, "bench_*.log"))):
        lines = [l for l in open(log) if l.startswith("{")]
        if lines:
            print("== %s ==" % os.path.basename(log))
            print(lines[-1].strip())
            print()


if __name__ == "__main__":
    main()
