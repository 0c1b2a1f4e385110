#!/usr/bin/env python3
"""Checksums of what a library variant's fused launches leave (dev aid for tools/ab_build.sh variants): every variant of the
same contract must print the same lines.  python tools/gpu_variant_check.py lib.so [env ...]"""
import hashlib
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402
from gym_pomdp_amd import _native  # noqa: E402
if sys.argv[1] != "-":
    _native.LIB_PATH = os.path.abspath(sys.argv[1])
import gym_pomdp_amd as gpa  # noqa: E402

ENVS = {"rock": ("Rock-v0", {}), "rock15": ("Rock-v0", dict(board_size=15, num_rocks=15)), "tag": ("Tag-v0", {}),
        "tiger": ("Tiger-v0", {}), "network": ("Network-v0", {}),
        "battleship": ("Battleship-v0", dict(board_size=(10, 10), max_len=5)), "battleship5": ("Battleship-v0", {}),
        "stochrock": ("StochasticRock-v0", {})}
for name in (sys.argv[2:] or ["rock", "tag", "tiger", "network", "battleship"]):
    env_id, kw = ENVS[name]
    for lg in (17, 20):
        try:
            e = gpa.make(env_id, batch_size=1 << lg, seed=3, **kw)
        except Exception as ex:  # noqa: BLE001
            print(name, "make failed:", ex)
            break
        e.reset()
        h = hashlib.sha256()
        for k in (70, 20):
            tr = e.collect_synthetic(k)
            torch.cuda.synchronize()
            for key in ("action", "ob", "reward", "done_u8"):
                h.update(tr[key][:k].contiguous().cpu().numpy().tobytes())
        h.update(e.state.cpu().numpy().tobytes() if hasattr(e, "state") else b"")
        print("%-11s 2^%d %s %s" % (name, lg, h.hexdigest()[:16], _native.lib().pomdp_last_fused_kernel().decode()), flush=True)
