import sys, time
sys.path.insert(0, "/root/repo")
import torch
import gym_pomdp_amd as gpa
from gym_pomdp_amd import _native
e = gpa.make("Rock-v0", seed=0)
e.reset()
stream = torch.cuda.current_stream(e.device)
ev = torch.cuda.Event()
def launch(a, t):
    ptrs, hp = e._ptrs, e._host_ptrs
    rc = e._step_fn(e._params_ref, ptrs[0], e._action_table.data_ptr() + 4 * a, hp[0], hp[1], hp[2], ptrs[4], 1, e._seed, 0, t, 1, stream.cuda_stream)
    assert rc == 0
for mode in ("stream.synchronize", "event.synchronize", "event.query spin", "host flag spin"):
    k, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < 1.0:
        e._host_out[3] = 0
        launch(k % 13, k + 5)
        if mode == "stream.synchronize":
            stream.synchronize()
        elif mode == "event.synchronize":
            ev.record(); ev.synchronize()
        elif mode == "event.query spin":
            ev.record()
            while not ev.query():
                pass
        else:
            ev.record()
            while not ev.query():
                pass
        k += 1
    print("%-22s %.0f launches+syncs/s" % (mode, k / (time.perf_counter() - t0)))
# breakdown: launch only
k, t0 = 0, time.perf_counter()
while k < 20000:
    launch(k % 13, k + 5); k += 1
torch.cuda.synchronize()
print("launch only: %.2f us per launch (async)" % ((time.perf_counter() - t0) * 1e6 / k))
