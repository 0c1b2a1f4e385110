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


def traffic_json(root, out_path, workload):
    """profiles/traffic_*.json: HBM bytes per launch of the plain and the chained step kernel from the two
    PMC passes (FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM, calibrated on the known read volume)."""
    import json
    f, w = db_of(os.path.join(root, "pmc_fetch")), db_of(os.path.join(root, "pmc_write"))
    out = {"workload": workload, "fetch_correction": "x2 (gfx950 FETCH_SIZE reports half of a coalesced stream; "
           "calibrated: the step kernels read exactly 8192 KB of state + action per launch)",
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes of `python bench.py --prewarm 0 "
                     "--warmup 64 --steps 640 --no-cpu-baseline` (tools/gpu_profile_round.sh); `fused` = one 64-step "
                     "steps_quad_kernel / steps_kernel launch, `chain` / `plain` = one single-step launch"}
    # `plain`: what env.step() launches — step_quad_kernel for the RockSample family at this size, else step_kernel<., ., false>
    for key, pat, pat2 in (("plain", "%step_kernel<%>, _, false>(%", "%step_quad_kernel<%"),
                           ("chain", "%step_kernel<%>, _, true>(%", ""), ("fused", "%steps%kernel<%", "")):
        vals = []
        for db, cn in ((f, "FETCH_SIZE"), (w, "WRITE_SIZE")):
            r = sqlite3.connect(db).execute("select avg(value), count(*) from counters_collection where counter_name=? "
                                            "and (kernel_name like ? or kernel_name like ?)", (cn, pat, pat2)).fetchone()
            vals.append(r)
        if vals[0][0] is None:
            continue
        out[key] = {"fetch_size_kb_reported": vals[0][0], "write_size_kb_reported": vals[1][0],
                    "dispatches_sampled": [vals[0][1], vals[1][1]],
                    "hbm_bytes_per_launch": int((2 * vals[0][0] + vals[1][0]) * 1024)}
    json.dump(out, open(out_path, "w"), indent=1)


def main():
    if len(sys.argv) > 2 and sys.argv[2] == "--traffic":
        return traffic_json(sys.argv[1], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else "")
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
