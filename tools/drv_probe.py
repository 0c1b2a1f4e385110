import sys, time, ctypes as C
sys.path.insert(0, "/root/repo")
import torch
import gym_pomdp_amd as gpa
from gym_pomdp_amd import _native
n = 1 << 20
env = gpa.make("Rock-v0", batch_size=n, device="cuda:0", seed=0, reuse_buffers=True)
env.reset()
act = torch.empty(n, dtype=torch.int32, device="cuda")
L = _native.lib()
def run(k, stream):
    t0 = env.call_counter
    env.call_counter = t0 + k
    rc = L.pomdp_rollout_synthetic(0, env._params_ref, env._state.data_ptr(), act.data_ptr(), env._ob.data_ptr(), env._reward.data_ptr(), env._done.data_ptr(), env._err.data_ptr(), n, 0, 0, 0, t0, k, 1, stream)
    assert rc == 0
for name, strm in (("null", None), ("torch side stream", torch.cuda.Stream())):
    h = None if strm is None else strm.cuda_stream
    run(200, h); torch.cuda.synchronize()
    for k in (100, 1000, 5000):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        w0 = time.perf_counter()
        if strm is None:
            e0.record(); run(k, h); e1.record()
        else:
            e0.record(strm); run(k, h); e1.record(strm)
        w1 = time.perf_counter()
        torch.cuda.synchronize()
        w2 = time.perf_counter()
        print("%s: k=%d  %.3f us/step by events, wall %.3f, cpu issue %.3f us/launch" % (name, k, e0.elapsed_time(e1) * 1e3 / k, (w2 - w0) * 1e6 / k, (w1 - w0) * 1e6 / k))
