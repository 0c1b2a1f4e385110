#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc passes of tools/gpu_pmc_valu.sh into profiles/<tag>_pmc_valu.{json,txt}.

    tools/pmc_valu_summary.py <work dir with one sub-directory per workload> <out dir>

Per workload and kernel: dispatches sampled, average duration, the raw counter averages per launch and what follows
from them —
  valu_per_launch          SQ_INSTS_VALU: vector instructions issued by all waves of one launch (wave-instructions)
  wave_insts_per_s         valu_per_launch / duration: the integer-op rate SURVEY.md §8d asks for (x 64 = lane-ops/s)
  issue_peak               256 CUs x 4 SIMDs x 2.4 GHz / 4 cycles per wave64 instruction = 6.144e11 wave-instructions/s
  valu_issue_frac          wave_insts_per_s / issue_peak
  clock_ghz                GRBM_GUI_ACTIVE / 8 XCDs / duration: the clock the launch really ran at — for launches of at least
                           MIN_WINDOW_US only: the counter runs over the profiler's sampling window, which is longer than a
                           short launch (round 5's summary derived 3.7-4.7 "GHz" and a VALU busy of 1.02-1.17 for 35 us launches)
  valu_busy_frac           SQ_ACTIVE_INST_VALU (quad-cycles, summed over waves) x 4 / (1024 SIMDs x duration x clock), same
                           launches only; a value above 1 is reported as not derivable, never printed as a fraction
bench.py reads the JSON (`valu_per_launch`, keyed by workload and kernel) and divides by the kernel time it measures
live, as it does with the recorded HBM byte counters.
"""
import glob
import json
import os
import sqlite3
import sys

SIMDS = 256 * 4
NOMINAL_HZ = 2.4e9
ISSUE_PEAK = SIMDS * NOMINAL_HZ / 4.0
MIN_WINDOW_US = 150.0        # launches shorter than this do not fill the GRBM counter's window: no clock, no busy fraction
KERNELS = ("%rollout_kernel%", "%heuristic_steps_kernel%", "%steps%kernel%")


def db_of(d):
    f = glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)
    return f[0] if f else None


def counters(db, last=None):
    """-> {kernel: {counter: (dispatches, avg value, avg duration ns)}} for the kernels of interest; `last`: only the last
    that many dispatches of each kernel (the timed region of a run whose earlier launches are in another phase)."""
    c = sqlite3.connect(db)
    out = {}
    for pat in KERNELS:
        if last is None:
            rows = c.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) from "
                             "counters_collection where kernel_name like ? group by kernel_name, counter_name", (pat,)).fetchall()
        else:
            rows = []
            for k, cn in c.execute("select distinct kernel_name, counter_name from counters_collection where kernel_name like ?", (pat,)):
                v = c.execute("select value, duration from counters_collection where kernel_name=? and counter_name=? "
                              "order by start desc limit ?", (k, cn, last)).fetchall()
                rows.append((k, cn, len(v), sum(x[0] for x in v) / len(v), sum(x[1] for x in v) / len(v)))
        for k, cn, n, a, d in rows:
            out.setdefault(k, {})[cn] = (n, a, d)
    return out


def short(k):
    k = k.replace("void pomdp::", "").replace("pomdp::", "")
    return k.split("(")[0].strip()


def main():
    root, out_dir = sys.argv[1], sys.argv[2]
    merge_from = sys.argv[3] if len(sys.argv) > 3 else None      # an earlier pmc_valu.json: its other workloads are carried over
    cmds = {}
    cf = os.path.join(root, "commands.txt")
    if os.path.exists(cf):
        for line in open(cf):
            name, _, cmd = line.partition(": ")
            cmds[name] = "python bench.py " + cmd.strip()
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import csrc_sha
    res = {"source": "tools/gpu_pmc_valu.sh: rocprofv3 --kernel-trace --pmc <one counter group per pass> -- python bench.py ...; "
                     "averages over the dispatches of each kernel (MI355X)",
           "csrc_sha256": csrc_sha(),      # the kernel sources these counters were recorded from (bench.py: counters_stale)
           "issue_peak_wave_insts_per_s": ISSUE_PEAK,
           "issue_peak_formula": "256 CUs x 4 SIMDs x 2.4e9 Hz / 4 cycles per wave64 VALU instruction",
           "workloads": {}}
    lines = []
    for wd in sorted(glob.glob(os.path.join(root, "*"))):
        if not os.path.isdir(wd):
            continue
        name = os.path.basename(wd)
        merged = {}
        last = None
        if name.startswith("heuristic_"):          # the timed region = the last ceil(K / steps per launch) launches of the run (bench.py: heuristic_mode)
            import re
            from gym_pomdp_amd._native import FUSE_MAX_DEFAULT
            m = re.search(r"--steps (\d+)", cmds.get(name, ""))
            last = -(-int(m.group(1)) // FUSE_MAX_DEFAULT) if m else None
        for sub in ("p1", "p2", "p3", "p4"):          # p3 / p4: the store-path stall counters (headline set only)
            db = db_of(os.path.join(wd, sub))
            if not db:
                continue
            for k, v in counters(db, last).items():
                merged.setdefault(k, {}).update({cn: (n, a, d, sub) for cn, (n, a, d) in v.items()})
        entry = {"command": cmds.get(name), "kernels": {}}
        for k, v in merged.items():
            if "SQ_INSTS_VALU" not in v:
                continue
            n1, valu, d1 = v["SQ_INSTS_VALU"][:3]
            dur_s = d1 * 1e-9
            ke = {"dispatches": n1, "duration_us": d1 / 1e3,
                  "counters_per_launch": {cn: x[1] for cn, x in sorted(v.items())},
                  "valu_per_launch": valu, "wave_insts_per_s": valu / dur_s,
                  "valu_issue_frac": valu / dur_s / ISSUE_PEAK}
            if "SQ_WAVES" in v:
                ke["waves"] = v["SQ_WAVES"][1]
                ke["valu_per_wave"] = valu / v["SQ_WAVES"][1]
            if "GRBM_GUI_ACTIVE" in v:
                d2 = v["GRBM_GUI_ACTIVE"][2] * 1e-9
                clk = v["GRBM_GUI_ACTIVE"][1] / 8.0 / d2
                ke["duration_us_pass2"] = d2 * 1e6
                if d2 * 1e6 >= MIN_WINDOW_US and clk <= 1.05 * NOMINAL_HZ:
                    ke["clock_ghz"] = clk / 1e9
                    if "SQ_ACTIVE_INST_VALU" in v:
                        busy = v["SQ_ACTIVE_INST_VALU"][1] * 4.0 / (SIMDS * dur_s * clk)
                        if busy <= 1.0:
                            ke["valu_busy_frac"] = busy
                else:
                    ke["clock_note"] = "launch shorter than the GRBM counter's window (%.0f us < %.0f us) or clock above nominal: " \
                                       "clock and VALU busy not derived" % (d2 * 1e6, MIN_WINDOW_US)
            entry["kernels"][short(k)] = ke
            lines.append("%-18s %-58s n=%-5d %9.1f us  VALU/launch %.4e  %.3e wave-insts/s = %.3f of issue peak%s%s"
                         % (name, short(k)[:58], n1, d1 / 1e3, valu, valu / dur_s, valu / dur_s / ISSUE_PEAK,
                            "  clk %.2f GHz" % ke["clock_ghz"] if "clock_ghz" in ke else "",
                            "  VALU busy %.3f" % ke["valu_busy_frac"] if "valu_busy_frac" in ke else ""))
        # HBM bytes (FETCH_SIZE x2 per the guide's gfx950 correction + WRITE_SIZE), when the passes exist
        f, w = db_of(os.path.join(wd, "pmc_fetch")), db_of(os.path.join(wd, "pmc_write"))
        if f and w:
            tr = {}
            for pat in KERNELS:
                for db, cn in ((f, "FETCH_SIZE"), (w, "WRITE_SIZE")):
                    for k, n, a in sqlite3.connect(db).execute(
                            "select kernel_name, count(*), avg(value) from counters_collection where counter_name=? and "
                            "kernel_name like ? group by kernel_name", (cn, pat)):
                        tr.setdefault(short(k), {})[cn] = (n, a)
            for k, v in tr.items():
                if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
                    entry.setdefault("traffic", {})[k] = {
                        "fetch_size_kb_reported": v["FETCH_SIZE"][1], "write_size_kb_reported": v["WRITE_SIZE"][1],
                        "dispatches_sampled": [v["FETCH_SIZE"][0], v["WRITE_SIZE"][0]],
                        "hbm_bytes_per_launch": int((2 * v["FETCH_SIZE"][1] + v["WRITE_SIZE"][1]) * 1024)}
                    lines.append("%-18s %-58s HBM bytes per launch %d (2 x FETCH_SIZE + WRITE_SIZE)"
                                 % (name, k[:58], entry["traffic"][k]["hbm_bytes_per_launch"]))
        # the bench line each pass printed (value, mean steps per simulation ...)
        for sub in ("p1",):
            log = os.path.join(wd, sub + ".log")
            if os.path.exists(log):
                js = [l for l in open(log) if l.startswith("{")]
                if js:
                    try:
                        j = json.loads(js[-1])
                        entry["bench_line_under_pmc"] = {"value": j.get("value"), "ms_per_step": j.get("ms_per_step"),
                                                         "config": {k: j["config"].get(k) for k in
                                                                    ("workload", "mean_steps_per_simulation", "lanes_per_gpu")
                                                                    if k in j.get("config", {})}}
                    except Exception:  # noqa: BLE001
                        pass
        entry["csrc_sha256"] = res["csrc_sha256"]
        res["workloads"][name] = entry
    if merge_from and os.path.exists(merge_from):
        old = json.load(open(merge_from))
        for name, entry in old.get("workloads", {}).items():
            if name not in res["workloads"]:
                entry.setdefault("csrc_sha256", old.get("csrc_sha256"))
                res["workloads"][name] = entry
    os.makedirs(out_dir, exist_ok=True)
    json.dump(res, open(os.path.join(out_dir, "pmc_valu.json"), "w"), indent=1)
    with open(os.path.join(out_dir, "pmc_valu.txt"), "w") as fo:
        fo.write("# tools/gpu_pmc_valu.sh — rocprofv3 --pmc passes (MI355X); issue peak = 256 x 4 x 2.4e9 / 4 = %.4e wave-instructions/s\n" % ISSUE_PEAK)
        for name, cmd in cmds.items():
            fo.write("# %s: %s\n" % (name, cmd))
        fo.write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
