#!/usr/bin/env python3
"""Registers / occupancy / LDS of every kernel in the product library, from hipcc's -Rpass-analysis=kernel-resource-usage
(dev aid; cross-compiles without a GPU).  usage: python tools/kernel_resources.py [filter substring] [-D...]"""
import os
import re
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
flt = [a for a in sys.argv[1:] if not a.startswith("-D")]
defs = [a for a in sys.argv[1:] if a.startswith("-D")]
sys.path.insert(0, REPO)
from gym_pomdp_amd import _native  # noqa: E402
from concurrent.futures import ThreadPoolExecutor  # noqa: E402


def remarks(unit):
    cmd = ["/opt/rocm/bin/hipcc"] + _native.HIPCC_FLAGS + ["-c", "-o", "/dev/null", os.path.join(REPO, "gym_pomdp_amd/csrc", unit),
                                                          "-Rpass-analysis=kernel-resource-usage"] + defs
    return subprocess.run(cmd, capture_output=True, text=True).stderr


with ThreadPoolExecutor(max_workers=len(_native.UNITS)) as ex:
    out = "\n".join(ex.map(remarks, _native.UNITS))
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
print("%-5s %-5s %-4s %-7s %-7s %s" % ("VGPR", "SGPR", "occ", "LDS", "scratch", "kernel"))
for r, n in zip(rows, names):
    n = re.sub(r"\(.*", "", n).replace("pomdp::", "").replace("void ", "")
    if flt and not all(f in n for f in flt):
        continue
    print("%-5s %-5s %-4s %-7s %-7s %s" % (r.get("VGPRs", "?"), r.get("TotalSGPRs", r.get("SGPRs", "?")), r.get("Occupancy [waves/SIMD]", "?"),
                                         r.get("LDS Size [bytes/block]", "?"), r.get("ScratchSize [bytes/lane]", "?"), n[:150]))
