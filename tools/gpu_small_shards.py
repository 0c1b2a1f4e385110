#!/usr/bin/env python3
"""Fused-launch time per step at the shard sizes a 2^20-lane batch leaves per GPU (2^17 .. 2^20, and below), for every
library variant given: python tools/gpu_small_shards.py [libA.so libB.so ...]   (default: the product library).
Variants come from tools/ab_build.sh (e.g. -DPOMDP_QUAD_MIN_LANES=4096: the quad-per-thread loops from 4096 lanes up).
Prints us per step of collect_synthetic($SHARD_K or 256, layout=$SHARD_LAYOUT or "packed"; "returns": collect_returns) by HIP
events, and the kernel the launcher picked.  SHARD_TAPE=1: the same launches on a tape of the caller's actions (collect_tape)."""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

ENVS = [("rock", "Rock-v0", {}), ("rock15", "Rock-v0", dict(board_size=15, num_rocks=15)), ("stochrock", "StochasticRock-v0", {}), ("tag", "Tag-v0", {}),
        ("tiger", "Tiger-v0", {}), ("network", "Network-v0", {}),
        ("battleship", "Battleship-v0", dict(board_size=(10, 10), max_len=5))]


def one(lib_path):
    import torch
    from gym_pomdp_amd import _native
    if lib_path:
        _native.LIB_PATH = lib_path
    if os.environ.get("AB_ABI"):        # an older revision's library with the same struct layouts (tools/ab_build.sh rev)
        _native.ABI_VERSION = int(os.environ["AB_ABI"])
    import gym_pomdp_amd as gpa
    L = _native.lib()
    print("library: %s" % _native.LIB_PATH)
    only = [x for x in os.environ.get("SHARD_ENVS", "").split(",") if x]
    for name, env_id, kw in ENVS:
        if only and name not in only:
            continue
        sizes = [int(x) for x in os.environ["SHARD_SIZES"].split(",")] if os.environ.get("SHARD_SIZES") else [1 << lg for lg in (14, 16, 17, 18, 19, 20)]
        for n in sizes:
            lg = n.bit_length() - 1
            e = gpa.make(env_id, batch_size=n, seed=0, reuse_buffers=True, **kw)
            e.reset()
            K = int(os.environ.get("SHARD_K", "256"))
            layout = os.environ.get("SHARD_LAYOUT", "packed")
            tape = None
            if os.environ.get("SHARD_TAPE"):           # the caller's actions (pomdp_collect_tape*): K rows of uniform random bytes
                tape = torch.randint(0, e.action_space.n, (K, n), dtype=torch.uint8, device="cuda")
            if layout == "returns":                    # no trajectory: the episode-return reduction (pomdp_collect_returns)
                tr = e.collect_returns(K)
                run = (lambda: e.collect_tape(tape, stats=tr)) if tape is not None else (lambda: e.collect_returns(K, stats=tr))
            else:
                tr = e.collect_synthetic(K, layout=layout)
                run = (lambda: e.collect_tape(tape, out=tr, layout=layout)) if tape is not None else (lambda: e.collect_synthetic(K, out=tr))
            for _ in range(8):
                run()
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(4):
                    run()
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / (4 * K) * 1e3)
            print("%-10s %s lanes: %7.3f us/step  %8.3e lane-steps/s  %s" % (name, ("2^%d" % lg) if n == 1 << lg else str(n), best, n / best * 1e6,
                                                                               L.pomdp_last_fused_kernel().decode()), flush=True)
            del e, tr


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--one":
        one(sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] != "-" else None)
    else:
        for lib in (sys.argv[1:] or ["-"]):
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "--one", lib])
