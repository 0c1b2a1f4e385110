#!/usr/bin/env python3
"""Where the time of ONE fused RockSample launch goes (dev aid): a library built with -DPOMDP_DEV_TIMELINE stamps the 100 MHz
wall clock per workgroup of steps_quad_kernel at entry (0), after the staging barrier (1), after the (position, action) table
is built (2), after the step loop (3), after the state stores are issued (4) and acknowledged (5).
argv: library path, steps per launch [, log2 lanes]."""
import ctypes as C
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from gym_pomdp_amd import _native  # noqa: E402
_native.LIB_PATH = os.path.abspath(sys.argv[1])
import gym_pomdp_amd as gpa  # noqa: E402

k = int(sys.argv[2])
n = 1 << int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 20
kw = dict(board_size=15, num_rocks=15) if len(sys.argv) > 4 and sys.argv[4] == "rock15" else {}
e = gpa.make("Rock-v0", batch_size=n, seed=0, reuse_buffers=True, **kw)
e.reset()
lib = _native.lib()
buf = torch.zeros(8 * 8192, dtype=torch.int64, device=e.device)
tr = e.collect_synthetic(k)
for i in range(20):
    e.collect_synthetic(k, out=tr)
torch.cuda.synchronize()
assert lib.pomdp_dev_timeline(C.c_void_p(buf.data_ptr())) == 0
for i in range(3):
    e.collect_synthetic(k, out=tr)
torch.cuda.synchronize()
t = buf.cpu().numpy().reshape(-1, 8)
t = t[t[:, 0] != 0][:, :6].astype(np.float64)
t0 = t[:, 0].min()
t = (t - t0) * 0.01       # microseconds
print("rock%s, %%d lanes, %%d steps per launch, %%d workgroups, kernel %%s, library %%s" % ("15" if kw else "") % (
#, %d workgroups, kernel %s, library %s" % (
    n, k, len(t), lib.pomdp_last_fused_kernel().decode(), os.path.basename(sys.argv[1])))
names = ["entry", "staged+barrier", "table built", "loop done", "state stores issued", "stores acked"]
for j in range(6):
    c = t[:, j]
    print("  %-22s min %6.2f  p10 %6.2f  median %6.2f  p90 %6.2f  max %6.2f" % (names[j], c.min(), np.percentile(c, 10), np.median(c), np.percentile(c, 90), c.max()))
d = np.diff(t, axis=1)
print("  per-workgroup phase lengths (median): " + "  ".join("%s %.2f" % (a, np.median(d[:, j])) for j, a in enumerate(["entry->staged", "->table", "->loop", "->state stores", "->acked"])))
print("  loop per step (median workgroup): %.3f us;  last ack - first entry: %.2f us" % (np.median(d[:, 2]) / k, t[:, 5].max()))
# which CU ran which workgroup (HW_ID: bits 8-11 cu_id, 12 sh_id, 13-15 se_id on gfx9; XCC_ID bits 0-3)
raw = buf.cpu().numpy().reshape(-1, 8)
raw = raw[raw[:, 0] != 0]
hw, xcc = raw[:, 6], raw[:, 7] & 15
cu = (hw >> 8) & 15
sh = (hw >> 12) & 1
se = (hw >> 13) & 7
cuid = xcc * 1000 + se * 100 + sh * 20 + cu
dur = (raw[:, 3] - raw[:, 2]) * 0.01 / k
uniq, cnt = np.unique(cuid, return_counts=True)
print("  distinct CUs used: %d; workgroups per CU: %s" % (len(uniq), dict(zip(*np.unique(cnt, return_counts=True)))))
for c in np.unique(cnt):
    sel = np.isin(cuid, uniq[cnt == c])
    print("    CUs holding %d workgroups: loop %.3f us per step (median), loop-done at %.1f us (median)" % (
        c, np.median(dur[sel]), np.median((raw[sel, 3] - raw[:, 0].min()) * 0.01)))
for x in range(8):
    sel = xcc == x
    if sel.any():
        print("    XCD %d: %4d workgroups on %3d CUs, loop %.3f us per step (median), blockIdx %% 8 = %s" % (
            x, sel.sum(), len(np.unique(cuid[sel])), np.median(dur[sel]), np.unique(np.arange(len(raw))[sel] % 8)))
