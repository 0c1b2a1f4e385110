#!/usr/bin/env python3
"""Kernel time of ONE step per launch through the different launch structures (dev aid; run under rocprofv3 --kernel-trace
--stats to read kernel durations, or stand-alone for HIP-event times of back-to-back calls):
  env.step()                    step_kernel<Env, LPT>
  rollout_synthetic(1, fuse=0)  step_kernel<Env, LPT, chain>
  collect_synthetic(1)          the fused kernels with k = 1 (quad-per-thread where the env has one)"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402
from gym_pomdp_amd import _native  # noqa: E402
if len(sys.argv) > 1 and sys.argv[1] != "-":                 # a variant library built by tools/ab_build.sh
    _native.LIB_PATH = os.path.abspath(sys.argv[1])
import gym_pomdp_amd as gpa  # noqa: E402

n = 1 << (int(sys.argv[2]) if len(sys.argv) > 2 else 20)
print("library: %s, %d lanes" % (_native.LIB_PATH, n))
for name, env_id, kw in (("rock", "Rock-v0", {}), ("rock15", "Rock-v0", dict(board_size=15, num_rocks=15)),
                         ("stochrock", "StochasticRock-v0", {}), ("tag", "Tag-v0", {}), ("tiger", "Tiger-v0", {}), ("network", "Network-v0", {})):
    if len(sys.argv) > 3 and name not in sys.argv[3].split(","):
        continue
    e = gpa.make(env_id, batch_size=n, seed=0, reuse_buffers=True, **kw)
    e.reset()
    ring = []
    for j in range(16):
        e.call_counter += 1
        ring.append(e.synthetic_actions().clone())
    tr = e.collect_synthetic(1)

    def timed(fn, reps=400):
        for _ in range(50):
            fn(0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for i in range(reps):
            fn(i)
        e1.record()
        host = (time.perf_counter() - t0) / reps * 1e6
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3, host

    a = timed(lambda i: e.step(ring[i & 15]))
    b = timed(lambda i: e.rollout_synthetic(1, fuse=False))
    c = timed(lambda i: e.collect_synthetic(1, out=tr))
    print("%-8s env.step %.2f us (host issue %.2f)   chained single %.2f (host %.2f)   fused k=1 %.2f (host %.2f)" % ((name,) + a + b + c), flush=True)
