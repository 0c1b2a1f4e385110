#!/usr/bin/env python3
"""Random search over the RELATIVE starts of the four trajectory columns inside one pool (dev aid): which offsets make the four
rows a step writes side by side get along?  Prints 'd1 d2 d3 pad us' per sample: extra bytes in front of the ob / reward / done
columns (multiples of 256), row pitch pad in elements, us per K-step launch.  argv: K samples seed"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import gym_pomdp_amd as gpa  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 64
S = int(sys.argv[2]) if len(sys.argv) > 2 else 400
rs = np.random.RandomState(int(sys.argv[3]) if len(sys.argv) > 3 else 0)
n = 1 << 20
e = gpa.make("Rock-v0", batch_size=n, seed=0, reuse_buffers=True)
e.reset()
x = torch.zeros(1 << 26, device="cuda")
for _ in range(300):
    x.add_(1.0)
torch.cuda.synchronize()
pool = torch.zeros(5 << 30, dtype=torch.uint8, device="cuda")


def carve(d, pitch):
    o, cols = 0, []
    for q, (dt, rows, esz) in enumerate(((torch.int32, K + 1, 4), (torch.int32, K, 4), (torch.int32, K, 4), (torch.uint8, K, 1))):
        o += d[q]
        nbytes = rows * pitch * esz
        cols.append(pool[o:o + nbytes].view(dt).view(rows, pitch)[:, :n])
        o += -(-nbytes // 4096) * 4096
    a, ob, r, dn = cols
    return {"action": a, "ob": ob, "reward": r, "done_u8": dn, "done": dn.view(torch.bool)}


def time_it(tr, pitch):
    e.collect_synthetic(K, out=tr)
    for key, bound in e._collect_cache.items():
        if key[0] == tr["action"].data_ptr():
            bound[0].pitch = pitch
    for _ in range(4):
        e.collect_synthetic(K, out=tr)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(6):
            e.collect_synthetic(K, out=tr)
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / 6 * 1e3)
    e._collect_cache.clear()
    return best


pads = [int(x) for x in os.environ.get("OS_PADS", "0").split(",")]
gran = int(os.environ.get("OS_GRAN", "256"))
span = int(os.environ.get("OS_SPAN", str(8 << 20)))
print("# pool base 0x%x K %d" % (pool.data_ptr(), K))
for i in range(S):
    d = [0] + [int(rs.randint(0, span // gran)) * gran for _ in range(3)]
    pad = pads[rs.randint(len(pads))]
    t = time_it(carve(d, n + pad), n + pad)
    print("%d %d %d %d %.2f" % (d[1], d[2], d[3], pad, t), flush=True)
