#!/usr/bin/env python3
"""Where the time of ONE env.step() launch goes (dev aid): a library built with -DPOMDP_DEV_TIMELINE
(tools/ab_build.sh lib tl -DPOMDP_DEV_TIMELINE) stamps the 100 MHz wall clock per workgroup at kernel entry (0), after the
table-staging barrier (1), after the lane steps / first use of the loads (2), after the reset pass (3), after the stores
are issued (4) and after they are acknowledged (5).  argv: library path, env name, log2 lanes."""
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

ENVS = {"rock": ("Rock-v0", {}), "tag": ("Tag-v0", {}), "tiger": ("Tiger-v0", {}), "network": ("Network-v0", {})}
env_id, kw = ENVS[sys.argv[2]]
n = 1 << int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 20
e = gpa.make(env_id, batch_size=n, seed=0, reuse_buffers=True, **kw)
e.reset()
lib = _native.lib()
buf = torch.zeros(8 * 8192, dtype=torch.int64, device=e.device)
acts = []
for j in range(8):
    e.call_counter += 1
    acts.append(e.synthetic_actions().clone())
for i in range(20):
    e.step(acts[i & 7])
torch.cuda.synchronize()
assert lib.pomdp_dev_timeline(C.c_void_p(buf.data_ptr())) == 0
for i in range(3):
    e.step(acts[i & 7])
torch.cuda.synchronize()
t = buf.cpu().numpy().reshape(-1, 8)
t = t[t[:, 0] != 0][:, :6].astype(np.float64)
t0 = t[:, 0].min()
t = (t - t0) * 0.01       # microseconds
print("%s, %d lanes, %d workgroups, library %s" % (sys.argv[2], n, len(t), os.path.basename(sys.argv[1])))
names = ["entry", "staged+barrier", "loads used / lane steps", "reset pass", "stores issued", "stores acked"]
for k in range(6):
    c = t[:, k]
    print("  %-26s min %5.2f  p10 %5.2f  median %5.2f  p90 %5.2f  max %5.2f" % (names[k], c.min(), np.percentile(c, 10), np.median(c), np.percentile(c, 90), c.max()))
d = np.diff(t, axis=1)
print("  per-workgroup phase lengths (median): " + "  ".join("%s %.2f" % (a, np.median(d[:, k])) for k, a in enumerate(["entry->staged", "->steps", "->resets", "->stores", "->acked"])))
