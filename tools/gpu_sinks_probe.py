#!/usr/bin/env python3
"""What a fused launch costs per step by SINK (csrc/traj_out.hip.h) and by steps per launch (pomdp_fuse_max):
    python tools/gpu_sinks_probe.py            (envs: $SINK_ENVS, sizes: $SINK_SIZES, sinks: $SINK_SINKS, k: $SINK_K)
us per step by HIP events of `k`-step launches — columns / blocked / packed / narrow trajectories, the returns-only sink
(collect_returns) and packed + pomdp_decode_packed (records -> int32 columns, the decode pass alone as well) — and the kernel
the launcher picked."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

ENVS = [("rock", "Rock-v0", {}), ("rock15", "Rock-v0", dict(board_size=15, num_rocks=15)), ("tag", "Tag-v0", {}),
        ("tiger", "Tiger-v0", {}), ("network", "Network-v0", {}),
        ("battleship", "Battleship-v0", dict(board_size=(10, 10), max_len=5)), ("battleship5", "Battleship-v0", {})]


def timed(fn, steps_per_call, calls=6, rounds=5):
    import torch
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(calls):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / (calls * steps_per_call) * 1e3)
    return best


def main():
    import torch
    import gym_pomdp_amd as gpa
    from gym_pomdp_amd import _native
    L = _native.lib()
    only = [x for x in os.environ.get("SINK_ENVS", "").split(",") if x]
    sizes = [int(x) for x in os.environ.get("SINK_SIZES", str(1 << 20)).split(",")]
    sinks = [x for x in os.environ.get("SINK_SINKS", "columns,packed,narrow,returns,decode").split(",") if x]
    ks = [int(x) for x in os.environ.get("SINK_K", "64,128,256").split(",")]
    for name, env_id, kw in ENVS:
        if only and name not in only:
            continue
        for n in sizes:
            for k in ks:
                L.pomdp_fuse_max(k)
                e = gpa.make(env_id, batch_size=n, seed=0, reuse_buffers=True, **kw)
                e.reset()
                row = []
                for sink in sinks:
                    if sink == "returns":
                        st = gpa.EpisodeStats(e)
                        us = timed(lambda: e.collect_returns(k, st), k)
                    elif sink == "decode":
                        tr = e.collect_synthetic(k, layout="packed")
                        cols = e.trajectory_buffers(k)
                        us_d = timed(lambda: e.decode_trajectory(tr, into=cols), k)
                        us = timed(lambda: (e.collect_synthetic(k, out=tr), e.decode_trajectory(tr, into=cols)), k)
                        row.append("decode-only %.3f" % us_d)
                        del tr, cols
                    else:
                        tr = e.collect_synthetic(k, layout=sink)
                        us = timed(lambda: e.collect_synthetic(k, out=tr), k)
                        del tr
                    row.append("%s %.3f" % ("packed+decode" if sink == "decode" else sink, us))
                    if sink in ("packed", "returns"):
                        row.append("[%s]" % L.pomdp_last_fused_kernel().decode())
                lg = n.bit_length() - 1
                print("%-11s %-8s k=%-3d  %s" % (name, ("2^%d" % lg) if n == 1 << lg else n, k, "  ".join(row)), flush=True)
                del e
    L.pomdp_fuse_max(_native.FUSE_MAX_DEFAULT)


if __name__ == "__main__":
    main()
