#!/usr/bin/env python3
"""us per fused launch of k steps (HIP events, best of 7 x 20 launches) for the library in $POMDP_LIB (dev aid for
tools/ab_build.sh variants).  env: LT_ENV (rock), LT_LG (20), LT_KS (20,64), LT_LAYOUT (packed)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from gym_pomdp_amd import _native  # noqa: E402
if os.environ.get("POMDP_LIB"):
    _native.LIB_PATH = os.path.abspath(os.environ["POMDP_LIB"])
import gym_pomdp_amd as gpa  # noqa: E402

ENVS = {"rock": ("Rock-v0", {}), "rock15": ("Rock-v0", dict(board_size=15, num_rocks=15)), "tag": ("Tag-v0", {}),
        "tiger": ("Tiger-v0", {}), "network": ("Network-v0", {}),
        "battleship": ("Battleship-v0", dict(board_size=(10, 10), max_len=5)), "battleship5": ("Battleship-v0", {}),
        "stochrock": ("StochasticRock-v0", {})}
x = torch.zeros(1 << 26, device="cuda")
t_end = __import__("time").time() + 1.5            # clocks up before the first timing
while __import__("time").time() < t_end:
    x.add_(1.0)
torch.cuda.synchronize()
for name in os.environ.get("LT_ENV", "rock").split(","):
    env_id, kw = ENVS[name]
    e = gpa.make(env_id, batch_size=1 << int(os.environ.get("LT_LG", "20")), seed=0, reuse_buffers=True, **kw)
    e.reset()
    for k in [int(x) for x in os.environ.get("LT_KS", "20,64").split(",")]:
        tr = e.collect_synthetic(k, layout=os.environ.get("LT_LAYOUT", "packed"))
        for _ in range(30):
            e.collect_synthetic(k, out=tr)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(7):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20):
                e.collect_synthetic(k, out=tr)
            b.record()
            torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b) / 20 * 1e3)
        print("%-22s %-10s k=%d: %7.2f us per launch, %.3f us per step  %s" % (os.path.basename(_native.LIB_PATH), name, k, best, best / k,
                                                                          _native.lib().pomdp_last_fused_kernel().decode()), flush=True)
    del e
