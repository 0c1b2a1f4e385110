#!/usr/bin/env python3
"""Does the placement of the trajectory rows matter to the store stream? (dev aid)  The fused launch writes row s of three
int32 columns and one byte column per step; the rows of a column are `pitch` elements apart and the columns are separate
allocations.  Tries pitches n, n + 1024, n + 16384 + 1024 and column bases staggered by 0 / 4 KB / 68 KB."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402
import gym_pomdp_amd as gpa  # noqa: E402
from gym_pomdp_amd import _native  # noqa: E402

n, K = 1 << 20, 64
L = _native.lib()
for env_id in ("Rock-v0", "Tiger-v0"):
    e = gpa.make(env_id, batch_size=n, seed=0, reuse_buffers=True)
    e.reset()
    for pitch in (n, n + 1024, n + 16384 + 1024):
        for stagger in (0, 1024, 17408):          # elements (int32): 0, 4 KB, 68 KB
            big = (K + 2) * pitch + 4 * 17408 + 64
            act = torch.zeros(big, dtype=torch.int32, device="cuda")
            ob = torch.zeros(big, dtype=torch.int32, device="cuda")[stagger:]
            rew = torch.zeros(big, dtype=torch.int32, device="cuda")[2 * stagger:]
            done = torch.zeros(big, dtype=torch.uint8, device="cuda")[3 * stagger:]
            def call():
                rc = L.pomdp_collect_synthetic(_native.ENV_KIND[e.env_name], e._params_ref, e._state.data_ptr(), act.data_ptr(),
                                               ob.data_ptr(), rew.data_ptr(), done.data_ptr(), e._err.data_ptr(), n, e._seed, 0, e._t, K,
                                               pitch, 1, torch._C._cuda_getCurrentRawStream(0))
                assert rc == 0
                e._t += K
            for _ in range(10):
                call()
            torch.cuda.synchronize()
            best = 1e9
            for rep in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    call()
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / (10 * K) * 1e3)
            print("%-9s pitch n+%-6d column stagger %6d B: %.3f us/step" % (env_id, pitch - n, 4 * stagger, best), flush=True)
            del act, ob, rew, done
