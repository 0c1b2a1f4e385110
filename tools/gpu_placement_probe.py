#!/usr/bin/env python3
"""Does a 64-step collection's time depend on WHERE its trajectory buffer lies?  (dev aid)  One process, one env (RockSample(7,8),
2^20 lanes): (a) the buffer re-allocated several times with the old ones kept alive (fresh physical pages each time), (b) carved
out of one 6 GB pool at different offsets.  Prints us per 64-step launch (HIP events, best of 5 x 10 launches) per placement."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import gym_pomdp_amd as gpa  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = 1 << 20
e = gpa.make("Rock-v0", batch_size=n, seed=0, reuse_buffers=True)
e.reset()
x = torch.zeros(1 << 26, device="cuda")
for _ in range(200):
    x.add_(1.0)
torch.cuda.synchronize()


def time_it(tr):
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
    """GB/s of a plain torch fill over the tensor's storage (a pure 16-byte store stream over the same pages)"""
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


from gym_pomdp_amd.envs import base as _base  # noqa: E402
print("(a) fresh allocations of trajectory_buffers(%d), earlier ones kept alive, by column stagger" % K)
for round_ in range(int(os.environ.get('PP_ROUNDS', '2'))):
    for stg in [int(x) for x in os.environ.get("PP_ALLOC_STAGGERS", "4096,2048").split(",")]:
        _base.STAGGER_BYTES = stg
        keep, ts = [], []
        for i in range(int(os.environ.get('PP_ALLOCS', '6'))):
            tr = e.trajectory_buffers(K)
            keep.append(tr)
            ts.append(time_it(tr))
        print("  stagger %5d B: %s  median %.2f us per launch (%.3f per step)" % (
            stg, " ".join("%.1f" % t for t in ts), sorted(ts)[len(ts) // 2], sorted(ts)[len(ts) // 2] / K), flush=True)
        del keep, tr
        torch.cuda.empty_cache()
if os.environ.get("PP_ALLOC_ONLY"):
    sys.exit(0)
print("(b) one pool, column stagger and row pitch")
pool = torch.zeros(6 << 30, dtype=torch.uint8, device="cuda")
base = pool.data_ptr()


def carve(off, gap, pitch):
    o = off
    cols = []
    for dt, rows, esz in ((torch.int32, K + 1, 4), (torch.int32, K, 4), (torch.int32, K, 4), (torch.uint8, K, 1)):
        nbytes = rows * pitch * esz
        cols.append(pool[o:o + nbytes].view(dt).view(rows, pitch)[:, :n])
        o += -(-nbytes // 4096) * 4096 + gap
    a, ob, r, d = cols
    return {"action": a, "ob": ob, "reward": r, "done_u8": d, "done": d.view(torch.bool)}


def time_pitch(tr, pitch):
    e.collect_synthetic(K, out=tr)                 # binds the buffers
    for key, bound in e._collect_cache.items():
        if key[0] == tr["action"].data_ptr():
            bound[0].pitch = pitch
    return time_it(tr)


staggers = [int(x) for x in os.environ.get("PP_STAGGERS", "0,512,1024,1536,2048,2560,3072,4096,5120,6144,8192,12288,16384,20480,24576,32768,49152").split(",")]
pitches = [n + int(x) for x in os.environ.get("PP_PITCH_PADS", "0,1024").split(",")]
for rep in range(2):
    for pitch in pitches:
        for gap in staggers:
            tr = carve(0, gap, pitch)
            print("  rep %d pitch n+%-5d column stagger %8d B: %.2f us per launch" % (rep, pitch - n, gap, time_pitch(tr, pitch)), flush=True)
