#!/usr/bin/env python3
"""Round 4: the fused launch's time by TRAJECTORY LAYOUT and by where the buffer lies (dev aid; the successor of
round 3's placement probe, profiles/r03b_placement.txt).  One process, one env per workload (2^20 lanes); for each layout (columns = four write
streams, blocked = the same 13 B per lane-step in one stream, packed = 4 B records) PP_ALLOCS fresh trajectory buffers are
allocated with the earlier ones kept alive (fresh physical pages each time) and a K-step launch is timed into each (HIP
events, best of 5 x 10 launches).  Prints us per launch per buffer, min / median / max, the spread, and the fill rate of
every buffer.  usage: gpu_layout_probe.py [K=64] [envs=rock]   (envs: comma list of bench.py workload keys)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import gym_pomdp_amd as gpa  # noqa: E402
from bench import WORKLOADS  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ENVS = (sys.argv[2] if len(sys.argv) > 2 else "rock").split(",")
ALLOCS = int(os.environ.get("PP_ALLOCS", "8"))
LAYOUTS = os.environ.get("PP_LAYOUTS", "columns,blocked,packed").split(",")
n = int(os.environ.get("PP_LANES", str(1 << 20)))
x = torch.zeros(1 << 26, device="cuda")
for _ in range(300):
    x.add_(1.0)
torch.cuda.synchronize()


def time_it(e, tr):
    for _ in range(10):
        e.collect_synthetic(K, out=tr)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            e.collect_synthetic(K, out=tr)
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / 10 * 1e3)
    return best


def fill_rate(t):
    flat = t.view(-1)
    for _ in range(3):
        flat.fill_(1)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        flat.fill_(1)
    b.record()
    torch.cuda.synchronize()
    return flat.numel() * flat.element_size() * 10 / (a.elapsed_time(b) * 1e-3) / 1e9


from gym_pomdp_amd import _native  # noqa: E402
for key in ENVS:
    env_id, kwargs, label, _, _ = WORKLOADS[key]
    e = gpa.make(env_id, batch_size=n, seed=0, reuse_buffers=True, **kwargs)
    e.reset()
    print("%s, %d lanes, %d steps per launch, %d fresh allocations per layout (earlier ones kept alive)" % (label, n, K, ALLOCS))
    keep = []
    for rnd in range(int(os.environ.get("PP_ROUNDS", "1"))):
        for layout in LAYOUTS:
            ts, fr = [], []
            for i in range(ALLOCS):
                tr = e.trajectory_buffers(K, layout)
                keep.append(tr)
                ts.append(time_it(e, tr))
                fr.append(fill_rate(tr["ob"] if layout == "columns" else tr["traj"]))
            kern = _native.lib().pomdp_last_fused_kernel().decode()
            s = sorted(ts)
            by = {"columns": 13.0, "blocked": 13.0, "packed": 4.0}[layout]
            print("  %-8s %s | min %.1f median %.1f max %.1f us per launch (%.3f us per step, %.2f TB/s at %g B) spread %.1f %% | fill %.2f-%.2f TB/s | %s"
                  % (layout, " ".join("%.1f" % t for t in ts), s[0], s[len(s) // 2], s[-1], s[len(s) // 2] / K,
                     by * n * K / (s[len(s) // 2] * 1e-6) / 1e12, by, 100.0 * (s[-1] - s[0]) / s[0], min(fr) / 1e3, max(fr) / 1e3, kern), flush=True)
    del keep, e
    torch.cuda.empty_cache()
