#!/usr/bin/env python3
"""Static check of the fused step loops (runs without a GPU): a storing loop must not wait on vmcnt.

gfx9 counts loads and stores on one counter, so an `s_waitcnt vmcnt(..)` inside a step loop waits for the previous step's
stores — every step.  The compiler puts such a wait there whenever it cannot prove that the loads issued
before the loop have landed (a load inside the loop, or a loop with two entries); the kernels therefore settle their loads
explicitly (wait_loads, kernels_common.hip.h).  This tool compiles the fused translation units to assembly and lists, per
kernel, the vmcnt waits that sit in a basic block belonging to a loop.

    python tools/check_loop_waits.py [unit.hip ...]      (default: every fused_*.hip)
exit status 1 if a fused step kernel (name contains "steps_") waits inside a loop."""
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "gym_pomdp_amd", "csrc")
UNITS = ["fused_rock.hip", "fused_stochrock.hip", "fused_tag.hip", "fused_battleship.hip", "fused_misc.hip"]


def assembly(unit, out_dir="/tmp/pomdp_loop_waits"):
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, unit + ".s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                           "--cuda-device-only", "-S", "-o", out, os.path.join(CSRC, unit)], stderr=subprocess.DEVNULL)
    return open(out).read()


def loop_waits(text):
    """-> {demangled kernel name: ([line numbers of vmcnt waits inside loop blocks], number of s_setprio)}"""
    lines = text.split("\n")
    names = [m.group(1) for l in lines for m in [re.match(r"^(_Z\w+):", l)] if m]
    dem = dict(zip(names, subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()))
    res, cur, in_loop = {}, None, False
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            cur, in_loop = dem[m.group(1)], False
            res[cur] = ([], 0)
            continue
        if cur is None:
            continue
        if re.match(r"^\.LBB\d+_\d+:", l) or l.startswith("; %bb."):
            in_loop = ("in Loop" in l) or ("Loop Header" in l)
        s = l.strip()
        if s.startswith("s_endpgm"):
            cur = None
            continue
        if in_loop and s.startswith("s_waitcnt") and "vmcnt" in s:
            res[cur][0].append(i + 1)
        if s.startswith("s_setprio"):
            res[cur] = (res[cur][0], res[cur][1] + 1)
    return res


def main(units):
    bad = 0
    with ThreadPoolExecutor(max_workers=4) as ex:
        texts = list(ex.map(assembly, units))
    for unit, text in zip(units, texts):
        for name, (waits, prio) in loop_waits(text).items():
            if "steps_" not in name:
                continue
            flag = "WAITS IN LOOP at lines %s" % waits if waits else "ok"
            print("%-22s %-100s s_setprio x%d  %s" % (unit, name[:100], prio, flag))
            bad += bool(waits)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:] or UNITS))
