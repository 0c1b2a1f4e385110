#!/usr/bin/env python3
"""Registers / occupancy / LDS of every kernel in the product library, from hipcc's -Rpass-analysis=kernel-resource-usage
(dev aid; cross-compiles without a GPU).  usage: python tools/kernel_resources.py [filter substring] [-D...]"""
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from gym_pomdp_amd import _native  # noqa: E402
from concurrent.futures import ThreadPoolExecutor  # noqa: E402


def remarks(unit, defs=()):
    cmd = ["/opt/rocm/bin/hipcc"] + _native.HIPCC_FLAGS + ["-c", "-o", "/dev/null", os.path.join(REPO, "gym_pomdp_amd/csrc", unit),
                                                          "-Rpass-analysis=kernel-resource-usage"] + list(defs)
    return subprocess.run(cmd, capture_output=True, text=True).stderr


def collect(defs=(), units=None):
    """-> [{"kernel": demangled name without arguments, "vgpr", "sgpr", "occupancy", "lds", "scratch"}] for every kernel of the
    translation units (default: all of the product library's)"""
    units = list(units or _native.UNITS)
    with ThreadPoolExecutor(max_workers=len(units)) as ex:
        out = "\n".join(ex.map(lambda u: remarks(u, defs), units))
    cur, rows = None, []
    for line in out.splitlines():
        m = re.search(r"remark: +(.*?) \[-Rpass", line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            cur = {"name": t.split(":", 1)[1].strip()}
            rows.append(cur)
        elif cur is not None and ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip()
    names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.splitlines()
    res = []
    for r, n in zip(rows, names):
        n = re.sub(r"\(.*", "", n).replace("pomdp::", "").replace("void ", "")
        res.append({"kernel": n, "vgpr": r.get("VGPRs", "?"), "sgpr": r.get("TotalSGPRs", r.get("SGPRs", "?")),
                    "occupancy": r.get("Occupancy [waves/SIMD]", "?"), "lds": r.get("LDS Size [bytes/block]", "?"),
                    "scratch": r.get("ScratchSize [bytes/lane]", "?")})
    return res


if __name__ == "__main__":
    flt = [a for a in sys.argv[1:] if not a.startswith("-D")]
    defs = [a for a in sys.argv[1:] if a.startswith("-D")]
    print("%-5s %-5s %-4s %-7s %-7s %s" % ("VGPR", "SGPR", "occ", "LDS", "scratch", "kernel"))
    for r in collect(defs):
        if flt and not all(f in r["kernel"] for f in flt):
            continue
        print("%-5s %-5s %-4s %-7s %-7s %s" % (r["vgpr"], r["sgpr"], r["occupancy"], r["lds"], r["scratch"], r["kernel"][:150]))
