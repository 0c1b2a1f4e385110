import sys, time
sys.path.insert(0, "/root/repo")
import torch
import gym_pomdp_amd as gpa
for env_id in ("Rock-v0", "Tag-v0", "Tiger-v0"):
    e = gpa.make(env_id, seed=0)
    e.reset()
    k, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < 1.0:
        _, _, d, _ = e.step(e.action_space.sample())
        if d:
            e.reset()
        k += 1
    print(env_id, "scalar loop: %.0f steps/s" % (k / (time.perf_counter() - t0)))
