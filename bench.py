#!/usr/bin/env python3
"""bench.py — env-steps/s of the batched step() hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--env rock|tag|battleship|tiger|network]

Workload (BASELINE.json metric): RockSample(7,8), 2^20 lanes per GPU, i.i.d. uniform random
actions from the synthetic policy (its kernel is inside the timed region), auto-reset on done.
A "step" is one pass of the hot path over the whole batch.  The steps are issued by the library's C-side
drivers, by default as trajectory collection (pomdp_collect_layout): up to pomdp_fuse_max() (256) consecutive steps
inside one launch, every step's action / ob / reward / done written to its own [step][lane] row in the layout --layout
names (packed 4-byte records by default), a lane's state in registers between its steps.  `--collect 0` runs the same fused launches with every step overwriting the same
N-element outputs (pomdp_rollout_synthetic; what per-step launches leave), `--fuse 0` launches every step
separately, `--host-loop python` times the same steps through env.step() instead.
N > 1: one process per GPU, lanes sharded by global lane id, no data-path collective — only the timing
barrier / max-over-ranks (gloo, host side).  `python bench.py --gpus N` starts its N ranks itself (it re-executes
under `python -m torch.distributed.run`); started under torch.distributed.run already, it uses the ranks it was
given.  Scaling is weak (every GPU owns --lanes-per-gpu lanes; `value` is the whole-job rate); for N > 1 the line
also carries `strong_scaling`: the same K steps on a batch of --lanes-per-gpu lanes IN TOTAL split over the N GPUs.

Timing protocol: for each master seed in --seeds (default 0,1,2) the env is re-seeded and reset, W
warm-up steps run, and the region "barrier, sync, K steps, sync" is timed R times (R = --repeats, default 31 for
K < 1000 else 7; even regions by wall clock, odd ones by HIP events for the kernel time); every wall-clock region is
max-reduced over the ranks; a seed's figure is the median of its regions and the
line's `value` / `ms_per_step` the median over the seeds (config.seed_values has all of them and the min / max).
Every buffer the timed steps write is allocated, and touched by the same chunking of K, before the first timed
region.

Prints ONE JSON line on rank 0 with the driver's contract keys plus `roofline` (the tighter of HBM — algorithmic bytes /
HIP-event time of the timed kernel — and VALU issue — recorded instructions per launch / the same time), `layouts`
(the other sinks of the same workload, and what a consumer of int32 columns gets from the records: packed_plus_decode),
`configs` (BASELINE.json's other configs at their per-GPU sizes, and the returns-only reduction of the headline workload)
and `cpu_baseline` (the C oracle, OpenMP, on this box's host cores, N == 1 only).
"""
import argparse
import json
import os
import sys
import time

# kernel arguments in device memory (the step kernels take their parameter tables by value, ~1 KB per launch): read
# by the HIP runtime when it initialises, so it has to be in the environment before torch touches the GPU.  Measured:
# 7.2 us per chained step with it, 7.5-7.7 without, 9.0 with the arguments in host memory (round 2's probe)
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

import torch  # noqa: E402

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
SIMD_HZ = 256 * 4 * 2.4e9  # VALU issue: 256 CUs x 4 SIMDs x 2.4 GHz cycles per second


def _latest(pattern):
    """The newest profiles/<pattern>: the one recorded from THIS tree's kernel sources if there is one (its csrc_sha256),
    else the last by name (round tags sort: r04q < r05a) — never by mtime, which a fresh checkout scrambles."""
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "profiles", pattern)))
    if not files:
        return None
    sha = csrc_sha()
    for f in reversed(files):
        try:
            if json.load(open(f)).get("csrc_sha256") == sha:
                return f
        except Exception:  # noqa: BLE001
            pass
    return files[-1]


def csrc_sha():
    """sha256 over the kernel sources (gym_pomdp_amd/csrc/**, include/pomdp_hip.h), name-sorted: what a recorded PMC pass is
    tied to.  tools/pmc_valu_summary.py writes it into profiles/*_pmc_valu.json when the counters are recorded; a tree that
    differs makes every figure derived from those counters `counters_stale` (and tests/test_host_logic.py fail)."""
    import hashlib
    h = hashlib.sha256()
    root = os.path.join(REPO, "gym_pomdp_amd", "csrc")
    files = [os.path.join(d, f) for d, _, fs in os.walk(root) for f in fs if f.endswith((".hip", ".h"))]
    for f in sorted(files) + [os.path.join(REPO, "include", "pomdp_hip.h")]:
        h.update(os.path.relpath(f, REPO).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def counters_stale(pmc_path, workload=None):
    """True when the recorded counters (of one workload of the file: each entry carries the hash it was recorded under, so a
    partial re-recording can be merged into an older file) were taken from other kernel sources than this tree's, or carry
    no hash at all."""
    try:
        d = json.load(open(pmc_path))
        sha = d.get("csrc_sha256")
        if workload is not None:
            sha = d["workloads"][workload].get("csrc_sha256", sha)
        return sha != csrc_sha()
    except Exception:  # noqa: BLE001
        return True


def valu_roofline(workload, kernel_prefix, lanes, launch_ms):
    """The VALU-issue roofline of one launch ("report int-op rate" for the compute-bound kernels):
    achieved = vector instructions the launch issues (SQ_INSTS_VALU per launch, RECORDED by the committed rocprofv3 --pmc
    pass of the same command, profiles/*_pmc_valu.json) / the launch time measured in this run; peak = SIMD cycles per
    second / the average issue cycles per instruction of the kernel's hot loop (profiles/*_isa_mix.json: its static
    instruction mix priced with tools/valu_microbench's measured costs — 2 cycles for v_add / v_and / shifts / v_mov,
    4 for the rest).  None when no recorded pass matches this workload."""
    pmc, mix = _latest("*_pmc_valu.json"), _latest("*_isa_mix.json")
    if not pmc or not mix:
        return None
    w = json.load(open(pmc))["workloads"].get(workload)
    if not w or (w.get("bench_line_under_pmc") or {}).get("config", {}).get("lanes_per_gpu") not in (None, lanes):
        return None
    # the workload's dominant kernel of that family (a run's warm-up launches may be another instantiation of it)
    cands = [k for k in w["kernels"] if k.startswith(kernel_prefix)]
    name = max(cands, key=lambda k: w["kernels"][k].get("dispatches", 0)) if cands else None
    m = json.load(open(mix))["kernels"].get(name) if name else None
    if not m:
        return None
    insts, cpi = w["kernels"][name]["valu_per_launch"], m["avg_cycles_per_instruction"]
    achieved = insts / (launch_ms * 1e-3)
    return {"bound": "valu", "achieved": achieved, "peak": SIMD_HZ / cpi, "unit": "wave-instructions/s",
            "frac": achieved * cpi / SIMD_HZ, "insts_per_launch": insts, "launch_ms": launch_ms,
            # the same count against the programming guide's flat figure — every wave64 VALU instruction 2 cycles — beside the
            # priced one: `frac` says how full the issue slots are for THIS mix as measured, `frac_flat_2cyc` how far the
            # instruction count is from a machine that issued everything at the guide's rate
            "frac_flat_2cyc": achieved * 2.0 / SIMD_HZ,
            # the recorded pass's own measure of how busy the vector ALUs were (SQ_ACTIVE_INST_VALU x 4 / SIMD cycles at the measured
            # clock; launches of at least 150 us only — tools/pmc_valu_summary.py), for kernels whose static mix prices badly
            "valu_busy_recorded": w["kernels"][name].get("valu_busy_frac"),
            "cycles_per_instruction": cpi, "lane_ops_per_s": achieved * 64, "kernel": name,
            "counters_stale": counters_stale(pmc, workload) or json.load(open(mix)).get("csrc_sha256") != csrc_sha(),
            "source": "instructions per launch recorded (not measured in this run): profiles/%s [%s]; issue cost of the "
                      "kernel's instruction mix: profiles/%s" % (os.path.basename(pmc), workload, os.path.basename(mix))}

# env -> (make id, ctor kwargs, workload label, algorithmic bytes per env-step [SURVEY.md §8d], dtype)
WORKLOADS = {
    "rock": ("Rock-v0", {}, "RockSample(7,8)", 21, "int32"),
    "rock15": ("Rock-v0", dict(board_size=15, num_rocks=15), "RockSample(15,15)", 29, "int32"),
    "stochrock": ("StochasticRock-v0", {}, "StochasticRock(7,8)", 21, "int32"),
    "tag": ("Tag-v0", {}, "Tag-v0 (5x10 grid, 1 opponent)", 21, "int32"),
    "battleship": ("Battleship-v0", dict(board_size=(10, 10), max_len=5), "BattleShip 10x10 max_len=5", 61, "int32"),
    "battleship5": ("Battleship-v0", {}, "BattleShip 5x5 max_len=3 (the reference's default)", 29, "int32"),
    "tiger": ("Tiger-v0", {}, "Tiger-v0", 21, "int32"),
    "network": ("Network-v0", {}, "Network-v0 (10 machines)", 21, "int32"),
}
ORACLE_NAME = {"rock": "rock", "rock15": "rock", "stochrock": "stochrock", "tag": "tag", "battleship": "battleship", "battleship5": "battleship", "tiger": "tiger",
               "network": "network"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000, help="timed steps per region (SURVEY.md §8d: T = 1000 after 10 warm-up steps)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--env", default="rock", choices=sorted(WORKLOADS))
    ap.add_argument("--lanes-per-gpu", type=int, default=1 << 20)
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default, what the driver's contract asks for): --lanes-per-gpu lanes on every GPU.  "
                         "strong: --lanes-per-gpu is the TOTAL batch (SURVEY.md §8d reads the metric that way), split "
                         "over the GPUs")
    ap.add_argument("--seed", type=int, default=None, help="a single master seed (shorthand for --seeds S)")
    ap.add_argument("--seeds", default="0,1,2", help="master seeds of the timing protocol (SURVEY.md §8d: 0, then 1 and 2)")
    ap.add_argument("--repeats", type=int, default=0,
                    help="timed regions of K steps per seed (median reported); 0 = 31 if K < 1000 else 7")
    ap.add_argument("--host-loop", default="c", choices=["c", "python"],
                    help="who issues the two launches of a step: the C rollout driver or a python loop over env.step()")
    ap.add_argument("--mode", default="step", choices=["step", "rollout", "heuristic"],
                    help="step: the headline metric.  rollout: BASELINE.json configs[4] shape — every launch runs "
                         "sims-per-root random rollouts of --depth steps from each root (fused kernel, state in registers).  "
                         "heuristic: every lane follows the env's own _generate_preferred(history) policy "
                         "(use_heuristic=True; rock / rock15 / tag), up to pomdp_fuse_max() steps per fused launch")
    ap.add_argument("--depth", type=int, default=64)
    ap.add_argument("--sims-per-root", type=int, default=1024)
    ap.add_argument("--action-seed", type=int, default=None, help="policy key (default: the env seed)")
    ap.add_argument("--prewarm", type=float, default=2.0,
                    help="seconds of untimed launches before the W warm-up steps: a GPU coming out of idle runs its first "
                         "~0.1-1 s below full clock (a compute-only kernel like the fused rollout is up to 1.4x slower there)")
    ap.add_argument("--phase-steps", type=int, default=4096,
                    help="heuristic mode: untimed steps between reset() and the warm-up, so that the timed region is a fixed "
                         "stretch of the process with the lanes spread over their episodes")
    ap.add_argument("--fuse", type=int, default=1, choices=[0, 1],
                    help="1: the C driver may run up to 64 consecutive steps inside one launch (steps_kernel: every step's "
                         "outputs still computed and written, state in registers between steps); 0: one launch per step")
    ap.add_argument("--collect", type=int, default=1, choices=[0, 1],
                    help="step mode, fused launches: 1 = keep every step's action / ob / reward / done in [128][N] "
                         "trajectory buffers (env.collect_synthetic); 0 = every step overwrites the same N-element "
                         "outputs, as per-step launches do (env.rollout_synthetic).  Same bytes written either way")
    ap.add_argument("--layout", default="packed", choices=["columns", "blocked", "packed", "narrow", "returns"],
                    help="trajectory layout of the collected steps (include/pomdp_hip.h: POMDP_LAYOUT_*).  packed (default): one "
                         "32-bit record per lane-step (action | ob << 8 | reward code << 16 | done << 24), 4 B instead of 13 — the fused "
                         "loop is then bound by instruction issue; columns: the default ABI's four int32 / float / uint8 columns "
                         "(four write streams); blocked: the columns' 13 bytes as one stream (256-lane blocks); narrow: the record's "
                         "four bytes as four typed planes (uint8 / int8 tensors, nothing to decode); returns: NO trajectory — per lane "
                         "the reference callers' reduction r += discount * rw, one return per episode (pomdp_collect_returns).  The "
                         "line carries all of them under `layouts` / `configs.returns_only`")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the `layouts` and `configs` blocks (the other layouts of this workload; Tag / BattleShip / the C5 "
                         "rollout) that the default single-GPU run appends after its timed regions")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--share-gpus", action="store_true",
                    help="allow fewer visible GPUs than ranks (ranks then share devices: a development aid, e.g. two ranks on a "
                         "one-GPU box); without it such a run is an error, so a line that says n_gpus: N ran on N GPUs")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline sample")
    return ap.parse_args()


def prewarm(args, run, dev, k=64):
    """Untimed clock spin-up: keep the GPU busy with the workload's own launches for --prewarm seconds."""
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < args.prewarm:
        run(k)
        torch.cuda.synchronize(dev)


def median(xs):
    xs = sorted(xs)
    m = len(xs) // 2
    return xs[m] if len(xs) % 2 else 0.5 * (xs[m - 1] + xs[m])


def timed_regions(run, k, repeats, dev, cp):
    """`repeats` times: barrier + device sync, K steps, device sync.  Even regions are timed by wall clock (max over ranks),
    odd ones by HIP events on the launch stream (their records would otherwise sit inside the wall-clock window: ~3 us
    of host time each, which is 4 % of a 20-step region); a single region carries both.
    Returns ([wall seconds], [event milliseconds])."""
    walls, evs = [], []
    for r in range(repeats):
        with_events = (r & 1) == 1 or repeats == 1
        torch.cuda.synchronize(dev)
        cp.barrier()
        if with_events:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record()
            run(k)
            e1.record()
            torch.cuda.synchronize(dev)
            el = time.perf_counter() - t0
            cp.barrier()
            evs.append(e0.elapsed_time(e1))
            if repeats == 1:
                walls.append(cp.max(el))
        else:
            t0 = time.perf_counter()
            run(k)
            torch.cuda.synchronize(dev)
            el = time.perf_counter() - t0
            cp.barrier()
            walls.append(cp.max(el))
    return walls, evs


def self_launch(args):
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks (one per GPU) and hand over."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    raise SystemExit(subprocess.call(cmd, env=env))


CPU_BASELINE_CHILD = r"""
import json, os, sys, time
sys.path.insert(0, sys.argv[1])
from oracle import oracle_lib as ol
env_key, kwargs, seed, budget_s = sys.argv[2], json.loads(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5])
usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
o = ol.OracleEnv(env_key, **kwargs)
n = 1 << 20
t1, s1, t0 = 0.0, 0, time.perf_counter()          # one core: at least half a second of it (a cold first pass reads low)
o.bench_loop(1 << 16, 8, seed, 1)
while time.perf_counter() - t0 < 0.6:
    t1 += o.bench_loop(1 << 16, 8, seed + s1, 1)
    s1 += 8
cands = sorted({c for c in (usable, usable // 2, 128, 64, 32, 16, 8, 4) if 1 <= c <= usable}, reverse=True)
share = budget_s / len(cands)
table = []
for threads in cands:                      # strictly wall-clock bounded: chunks of steps until this count's share is spent
    steps, el, t0 = 0, 0.0, time.perf_counter()
    while time.perf_counter() - t0 < share:
        el += o.bench_loop(n, 8, seed + steps, threads)
        steps += 8
    table.append({"threads": threads, "value": n * steps / el, "steps": steps, "seconds": el})
print(json.dumps({"usable": usable, "one_core": (1 << 16) * s1 / t1, "table": table}))
"""


def cpu_baseline(env_key, kwargs, seed, budget_s):
    """The C oracle (a port of the reference's step()/reset(), pinned to reference traces) on the host
    cores this process may use, same workload shape (2^20 lanes, synthetic policy, auto-reset), the loop
    entirely in C (oracle/pomdp_oracle.c: or_bench_loop — every thread owns a chunk of lanes for the whole run, no
    barrier between steps), bounded sample, every thread count tried is in the line.  Runs in a child process so that
    its OpenMP runtime starts with its own settings (threads pinned to cores, passive waits) and torch's is left alone."""
    import subprocess
    env = dict(os.environ, OMP_PROC_BIND="close", OMP_PLACES="cores", OMP_WAIT_POLICY="passive")
    env.pop("OMP_NUM_THREADS", None)
    out = subprocess.run([sys.executable, "-c", CPU_BASELINE_CHILD, REPO, ORACLE_NAME[env_key], json.dumps(kwargs), str(seed),
                          str(budget_s)], capture_output=True, text=True, env=env, timeout=60 + 4 * budget_s)
    if out.returncode:
        raise RuntimeError("cpu_baseline child failed: %s" % out.stderr[-2000:])
    r = json.loads(out.stdout.strip().splitlines()[-1])
    table, usable = r["table"], r["usable"]
    best = max(table, key=lambda t: t["value"])
    quota = None
    try:                                                   # a cgroup CPU quota explains a curve that flattens early
        with open("/sys/fs/cgroup/cpu.max") as f:
            q = f.read().split()
            quota = None if q[0] == "max" else float(q[0]) / float(q[1])
    except Exception:  # noqa: BLE001
        pass
    usable_cpus = usable if quota is None else min(usable, int(quota))
    return {"value": best["value"], "unit": "env-steps/s", "cores": best["threads"], "usable_cpus": usable_cpus, "kind": "port",
            "cores_note": "`cores` = the OpenMP threads of the best row of by_threads; `usable_cpus` = the CPUs this process can "
                          "actually run on at once (min of its affinity mask and the cgroup quota) — threads beyond that time-share",
            "sample": "%d lanes x %d steps of the same workload on the C oracle (OpenMP, %d threads pinned with "
                      "OMP_PROC_BIND=close OMP_PLACES=cores, %.1f s; %d CPUs usable by this process%s; best of the thread counts "
                      "in by_threads)" % (1 << 20, best["steps"], best["threads"], best["seconds"], usable,
                                          "" if quota is None else ", cgroup quota %.0f CPUs" % quota),
            "by_threads": table, "cgroup_cpu_quota": quota,
            "single_core": {"value": r["one_core"], "unit": "env-steps/s", "cores": 1},
            "reference_python_recorded": {"value": 6.0e4, "unit": "env-steps/s", "cores": 1,
                                          "note": "reference's own Python loop, RockSample(7,8), measured in the "
                                                  "build container (BASELINE.md); it cannot travel to this box"}}


def rollout_roofline(args, lanes, launch_ms, hbm_achieved, alg_per_sim):
    """The fused rollout is compute-bound by construction (the state lives in registers for the whole simulation): its
    roofline is VALU issue; the HBM figure is kept beside it to show how far from memory-bound the launch is."""
    hbm = {"bound": "hbm", "achieved": hbm_achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_achieved / HBM_PEAK_GBS,
           "traffic": None, "algorithmic_bytes_per_simulation": alg_per_sim}
    v = valu_roofline("rollout_%s" % args.env, "rollout_kernel<", lanes, launch_ms)
    if v is None:
        return dict(hbm, kernel="rollout_kernel<%s>" % args.env, kernel_ms=launch_ms,
                    note="no recorded VALU counters for this workload (profiles/*_pmc_valu.json): HBM figure only")
    return dict(v, traffic=None, kernel_ms=launch_ms, hbm=hbm,
                note="kernel_ms = HIP events over the timed region / launches (the step count's reduction included, "
                     "~5 % of it); lane-steps per second is `value`")


def rollout_mode(args, env, cp, dev, rank, world, label):
    """configs[4]-shaped run: roots x sims-per-root lanes per GPU, one fused rollout launch per "step"."""
    n = args.lanes_per_gpu
    sims = args.sims_per_root
    roots_n = n // sims
    import gym_pomdp_amd as gpa
    roots_env = env
    roots_env.reset()
    for _ in range(4):                                   # move the roots off the start state
        roots_env.step(roots_env.synthetic_actions())
    roots = roots_env.state[:, :roots_n].contiguous()
    total = torch.zeros((), dtype=torch.int64, device=dev)

    bufs = [None]

    def run(k, count):
        for _ in range(k):
            bufs[0] = r = env.rollout(args.depth, sims_per_root=sims, roots=roots, lane_offset=rank * n, out=bufs[0])
            if count:
                total.add_(r["n_steps"].sum())

    args.prewarm = max(args.prewarm, 3.0)     # a compute-only kernel: the first seconds after idle run at ~half speed
    prewarm(args, lambda k: run(k, True), dev)        # with the step count's reduction in: its first launches are slow too
    run(args.warmup, True)
    torch.cuda.synchronize(dev)
    total.zero_()
    cp.barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    run(args.steps, True)
    ev1.record()
    torch.cuda.synchronize(dev)
    elapsed = cp.max(time.perf_counter() - t0)
    cp.barrier()
    launch_ms = ev0.elapsed_time(ev1) / args.steps
    steps_done = cp.sum(int(total.item()))
    # a fused rollout moves next to nothing — the root's state words in, 21 B of results out per
    # simulation of up to `depth` steps — so the HBM figure only shows how far from memory-bound it is; what bounds
    # the kernel is VALU issue (~200 instructions per lane-step)
    alg_per_sim = 21.0 + 4.0 * env.state_words / sims
    achieved = alg_per_sim * roots_n * sims / (launch_ms * 1e-3) / 1e9
    if rank == 0:
        print(json.dumps({
            "metric": "env steps/sec (whole node), fused random rollouts", "value": steps_done / elapsed,
            "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": "%s: %d roots x %d simulations per GPU, depth <= %d, uniform policy over "
                                   "_generate_legal(), one fused rollout launch per step" % (label, roots_n, sims, args.depth),
                       "lanes_per_gpu": roots_n * sims, "mean_steps_per_simulation": steps_done / (args.steps * roots_n * sims * world),
                       "parallelism": "lane-shard x%d, no collectives" % world},
            "roofline": rollout_roofline(args, roots_n * sims, launch_ms, achieved, alg_per_sim)}), flush=True)
    cp.close()


def heuristic_roofline(args, n, kern_ms, hbm_achieved, alg):
    hbm = {"bound": "hbm", "achieved": hbm_achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_achieved / HBM_PEAK_GBS,
           "traffic": None, "algorithmic_bytes_per_step": alg}
    spl = fuse_max()
    v = valu_roofline("heuristic_%s" % args.env, "heuristic_steps_kernel<", n, kern_ms * spl)
    if v is None:
        return dict(hbm, kernel="heuristic_steps_kernel (up to %d steps per launch)" % spl, kernel_ms=kern_ms)
    return dict(v, traffic=None, kernel_ms=kern_ms, steps_per_launch=spl, hbm=hbm,
                note="the tighter of the two bounds is reported first; launch_ms = kernel_ms x %d steps per launch" % spl)


def heuristic_mode(args, gpa, env_id, kwargs, cp, dev, rank, world, label, n, lane_offset):
    """Every lane runs the reference's heuristic rollout loop (rock.py:557-573): choice(_generate_preferred(history)),
    step, side statistics, history.append — one pomdp_heuristic_steps launch per step."""
    is_rock = env_id == "Rock-v0"

    def make():
        e = gpa.make(env_id, batch_size=n, device=dev, seed=args.seed, lane_offset=lane_offset, reuse_buffers=True,
                     **dict(kwargs, **(dict(use_heuristic=True) if is_rock else {})))
        e.reset()
        return e, gpa.History(e)

    chunk = fuse_max()

    def stepper(e, h):
        def run(k):
            while k > 0:
                c = min(k, chunk)            # whole launches of pomdp_fuse_max() steps
                e.heuristic_steps(h, c)
                k -= c
        return run

    # The cost of a step depends on where the lanes are in their episodes (the first hundred steps after a reset are
    # CHECK-heavy and cost several times more), so the timed region is a FIXED stretch of the process: reset, --phase-steps
    # untimed steps (default 4096: lanes are spread over their episodes by then), W warm-up steps, K timed steps — the same
    # launches whether or not a profiler slows the run down (tools/gpu_pmc_valu.sh records the instruction counters of
    # exactly these launches).  The clock spin-up (--prewarm seconds) therefore runs on a scratch env.
    scratch = make()
    prewarm(args, stepper(*scratch), dev)
    del scratch
    env, hist = make()
    run = stepper(env, hist)
    run(args.phase_steps)
    run(args.warmup)
    torch.cuda.synchronize(dev)
    cp.barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    run(args.steps)
    ev1.record()
    torch.cuda.synchronize(dev)
    elapsed = cp.max(time.perf_counter() - t0)
    cp.barrier()
    kern_ms = ev0.elapsed_time(ev1) / args.steps
    # algorithmic bytes per lane-step: with many steps per launch the lane's state and history words
    # stay in registers, so a step must write action, ob, reward, done = 13 B; state in / out once per launch (a CHECK's
    # statistics update — one rock's five fields and two sums — is not counted, nor the 40 B of history words per lane
    # read and written once per launch)
    alg = 13.0 + 8.0 * env.state_words / float(chunk)
    achieved = alg * n / (kern_ms * 1e-3) / 1e9
    if rank == 0:
        print(json.dumps({
            "metric": "env steps/sec (whole node), heuristic policy", "value": n * world * args.steps / elapsed,
            "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "int32 (+ f64 side statistics)", "data": "synthetic",
            "config": {"workload": "%s batch=%d lanes per GPU, every lane follows _generate_preferred(history) "
                                   "(use_heuristic=True), auto-reset, up to %d steps per fused launch" % (label, n, chunk),
                       "lanes_per_gpu": n, "phase_steps": args.phase_steps, "mean_history_size": float(hist._size.float().mean().item()),
                       "parallelism": "lane-shard x%d, no collectives" % world},
            "roofline": heuristic_roofline(args, n, kern_ms, achieved, alg)}), flush=True)
    cp.close()


def recorded_traffic(env_key, layout, spl, kernel_prefix):
    """HBM bytes of one fused launch of `spl` steps in `layout`, from the FETCH_SIZE / WRITE_SIZE passes of
    tools/gpu_pmc_valu.sh (2 x FETCH_SIZE + WRITE_SIZE, the guide's gfx950 correction) — recorded, not measured in this run."""
    pmc = _latest("*_pmc_valu.json")
    if not pmc:
        return None, None
    key = valu_workload_key(env_key, spl, layout)            # (recorded at 2^20 lanes only)
    w = json.load(open(pmc))["workloads"].get(key, {})
    for k, v in (w.get("traffic") or {}).items():
        if k.startswith(kernel_prefix):
            return v["hbm_bytes_per_launch"], "recorded, not measured in this run%s: profiles/%s [%s: %s]" % (
                " (STALE: the kernel sources changed since)" if counters_stale(pmc, key) else "", os.path.basename(pmc), key, k)
    return None, None


def fused_alg_bytes(bytes_per_step, spl, layout):
    """Algorithmic bytes per lane-step of a fused launch of `spl` steps: the sink's output bytes per step (action, ob,
    reward: 4 B each + done: 1 B = 13; packed / narrow: 4; returns: none) + what the launch moves once — the state in and
    out (SURVEY.md §8d's per-step figure minus its 13 B of per-step columns ... minus the action it READS, which a fused
    launch generates), columns only: the row of first actions it writes, returns only: the lane's statistics in and out
    (three float64 and two int32 rows: 64 B)."""
    out_b = {"packed": 4.0, "narrow": 4.0, "returns": 0.0}.get(layout, 13.0)
    once = bytes_per_step - 9 - (0 if layout == "columns" else 4) + (64 if layout == "returns" else 0)
    return out_b + once / float(spl)


def fuse_max(env_key=None, layout=None):
    """steps per fused launch of the library's C-side drivers (include/pomdp_hip.h: pomdp_fuse_max; for a trajectory
    collection of `env_key` in `layout`: pomdp_fuse_steps — the 13-byte layouts of the store-bound envs stay at 64)"""
    from gym_pomdp_amd import _native
    if env_key is None or layout not in _native.LAYOUTS:
        return int(_native.lib().pomdp_fuse_max(0))
    return int(_native.lib().pomdp_fuse_steps(_native.ENV_KIND[ORACLE_NAME[env_key].replace("stochrock", "rock")], _native.LAYOUTS[layout]))


def valu_workload_key(env_key, spl, layout, n=1 << 20):
    """name of the recorded PMC workload (tools/gpu_pmc_valu.sh) of a fused step launch: steps per launch, env, layout and —
    when it is not the metric's 2^20 — the shard size"""
    size = "" if n == 1 << 20 else ("_2e%d" % (n.bit_length() - 1) if n & (n - 1) == 0 else "_%d" % n)
    return "step%d_%s%s%s" % (spl, env_key, "" if layout == "columns" else "_" + layout, size)


def step_rooflines(env_key, bytes_per_step, layout, n, kern_ms, spl, fused, fused_kernel):
    """Both rooflines of the timed step kernel: HBM (algorithmic bytes / live kernel time) and, when the kernel's VALU count
    is on record, VALU issue.  `primary` = the tighter one (HBM when there is no usable VALU record)."""
    alg = fused_alg_bytes(bytes_per_step, spl, layout) if fused else float(bytes_per_step)
    achieved = alg * n / (kern_ms * 1e-3) / 1e9
    hbm = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS}
    valu = None
    if fused and fused_kernel:
        valu = valu_roofline(valu_workload_key(env_key, spl, layout, n), fused_kernel.split("<")[0] + "<", n, kern_ms * spl)
    primary = hbm
    if valu is not None and not valu["counters_stale"] and valu["frac"] > hbm["frac"]:
        primary = {k: valu[k] for k in ("bound", "achieved", "peak", "unit", "frac")}
    return {"alg_bytes": alg, "hbm": hbm, "valu": valu, "primary": primary}


def quick_step_config(args, gpa, _native, cp, dev, env_key, n, lane_offset, seed, layout, k=None):
    """One workload measured briefly with the headline's protocol (reseed + reset, W warm-up steps, an untimed pass, then 9
    regions of K steps bracketed by device syncs: even ones by wall clock, odd ones by HIP events) -> a `configs` / `layouts`
    entry.  K = --steps, at most one launch's worth (`layouts`: like for like with the headline); `configs` pass K = 512 — two
    256-step launches per region, whatever --steps is — so that their figures are those of the stand-alone runs of the same
    workloads (profiles/*_bench_envs.jsonl) and not a function of how the driver sliced the headline.  layout "returns": the
    returns-only sink (collect_returns); "packed_plus_decode": packed records, then pomdp_decode_packed into int32 columns."""
    env_id, kwargs, label, bytes_per_step, _ = WORKLOADS[env_key]
    k = min(args.steps, StepWorkload.CHUNK) if k is None else k
    wl = StepWorkload(args, gpa, env_id, kwargs, dev, n, lane_offset, seed, layout=layout, max_steps=k)
    wl.reseed(seed)
    wl.run(args.warmup)
    wl.run(k)
    walls, evs = timed_regions(wl.run, k, 9, dev, cp)
    kernel = _native.lib().pomdp_last_fused_kernel().decode()
    spl = min(fuse_max(env_key, layout), StepWorkload.CHUNK, k)
    kern_ms = median(evs) / k
    if layout == "packed_plus_decode":
        # two kernels per chunk: the packed producer and the decode pass (4 B read + 13 B written per lane-step); the pass is
        # timed by itself as well — it is a pure stream, so its roofline is HBM at 17 B
        c = min(k, StepWorkload.CHUNK)
        view = wl._view(c)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        wl.env.decode_trajectory(view, into=wl.cols)
        torch.cuda.synchronize(dev)
        e0.record()
        for _ in range(5):
            wl.env.decode_trajectory(view, into=wl.cols)
        e1.record()
        torch.cuda.synchronize(dev)
        dec_ms = e0.elapsed_time(e1) / (5 * c)
        del wl
        torch.cuda.empty_cache()
        alg = fused_alg_bytes(bytes_per_step, spl, "packed") + 17.0
        hbm = alg * n / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        return {"workload": "%s batch=%d, packed records then pomdp_decode_packed into the default int32 columns, %d steps per region"
                            % (label, n, k),
                "value": n * k / median(walls), "unit": "env-steps/s", "ms_per_step": median(walls) / k * 1e3,
                "kernel": "%s + decode_packed_kernel<true>" % kernel, "kernel_ms": kern_ms, "steps_per_launch": spl, "bytes_per_lane_step": alg,
                "decode_kernel_ms": dec_ms, "decode_hbm_frac": 17.0 * n / (dec_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "roofline": {"bound": "hbm", "frac": hbm, "hbm_frac": hbm, "valu_frac": None, "counters_stale": None},
                "note": "what a consumer that needs int32 / float tensors gets FROM RECORDS; writing the columns directly "
                        "(layouts.columns) is faster, reading the narrow layout's typed planes in place (layouts.narrow) faster still"}
    rf = step_rooflines(env_key, bytes_per_step, layout, n, kern_ms, spl, True, kernel)
    del wl
    torch.cuda.empty_cache()
    return {"workload": "%s batch=%d, %s, %d steps per region" % (label, n, "returns-only sink (per-lane episode statistics, no "
                                                                   "trajectory)" if layout == "returns" else layout + " layout", k),
            "value": n * k / median(walls), "unit": "env-steps/s", "ms_per_step": median(walls) / k * 1e3,
            "kernel": kernel, "kernel_ms": kern_ms, "steps_per_launch": spl, "bytes_per_lane_step": rf["alg_bytes"],
            "roofline": {"bound": rf["primary"]["bound"], "frac": rf["primary"]["frac"], "hbm_frac": rf["hbm"]["frac"],
                         "valu_frac": None if rf["valu"] is None else rf["valu"]["frac"],
                         "valu_busy_recorded": None if rf["valu"] is None else rf["valu"].get("valu_busy_recorded"),
                         "counters_stale": None if rf["valu"] is None else rf["valu"]["counters_stale"]}}


def quick_tape_config(args, gpa, _native, cp, dev, env_key, n, seed, layout, k=1024):
    """The fused launches on the CALLER's actions (pomdp_collect_tape*: rock.py:562-566, `env.step(action)` with whatever the
    caller chose): the headline's protocol on a tape of uniform random bytes in [0, n_actions) that torch generated — 256
    rows, replayed for every 256-step launch of the K-step regions — in the packed sink or the returns-only sink.  What the
    tape costs against the synthetic policy: one 4-byte load per quad-step instead of one Philox block — `vs_synthetic` is
    the ratio of the two kernels' times under THIS protocol (same env, same sink, same regions of four 256-step launches: a
    region starts on an idle GPU, so the first call's host time is inside it — 3 % of a 1024-step region, 7 % of a 512-step one)."""
    env_id, kwargs, label, bytes_per_step, _ = WORKLOADS[env_key]
    e = gpa.make(env_id, batch_size=n, device=dev, seed=seed, reuse_buffers=True, **kwargs)
    e.reset()
    rows = StepWorkload.CHUNK
    tape = torch.randint(0, e.action_space.n, (rows, n), dtype=torch.uint8, device=dev)
    sink = gpa.EpisodeStats(e) if layout == "returns" else e.trajectory_buffers(rows, layout)

    def run(steps):
        left = steps
        while left > 0:
            c = min(left, rows)
            if layout == "returns":
                e.collect_tape(tape[:c], stats=sink)
            else:
                e.collect_tape(tape[:c], out=sink, layout=layout)
            left -= c

    def run_synthetic(steps):
        left = steps
        while left > 0:
            c = min(left, rows)
            if layout == "returns":
                e.collect_returns(c, stats=sink)
            else:
                e.collect_synthetic(c, out=sink)
            left -= c

    run(args.warmup)
    run(k)
    walls, evs = timed_regions(run, k, 9, dev, cp)
    kernel = _native.lib().pomdp_last_fused_kernel().decode()
    run_synthetic(k)
    _, evs_syn = timed_regions(run_synthetic, k, 9, dev, cp)
    spl = min(fuse_max(env_key, layout), rows, k)
    kern_ms = median(evs) / k
    alg = fused_alg_bytes(bytes_per_step, spl, layout) + 1.0          # + the tape's byte
    hbm = alg * n / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
    bad = e.invalid_action_count()
    del e, tape, sink
    torch.cuda.empty_cache()
    return {"workload": "%s batch=%d, actions from a caller's tape (uint8 [steps][lanes] in HBM), %s, %d steps per region"
                        % (label, n, "returns-only sink" if layout == "returns" else layout + " layout", k),
            "value": n * k / median(walls), "unit": "env-steps/s", "ms_per_step": median(walls) / k * 1e3, "kernel": kernel,
            "kernel_ms": kern_ms, "steps_per_launch": spl, "bytes_per_lane_step": alg, "invalid_actions": bad,
            "synthetic_kernel_ms": median(evs_syn) / k, "vs_synthetic": median(evs) / median(evs_syn),
            "roofline": {"bound": "hbm", "frac": hbm, "hbm_frac": hbm, "valu_frac": None, "counters_stale": None,
                         "note": "no PMC record for the tape kernels: compare kernel_ms with the same sink under the synthetic policy"}}


def quick_rollout_config(args, gpa, dev, env_key, roots_n, sims, depth, seed):
    """BASELINE.json configs[4] per GPU: roots_n x sims random rollouts of <= depth steps in one fused launch, 40 timed
    launches (HIP events) after 5 untimed ones."""
    env_id, kwargs, label, _, _ = WORKLOADS[env_key]
    e = gpa.make(env_id, batch_size=roots_n, device=dev, seed=seed, reuse_buffers=True, **kwargs)
    e.reset()
    for _ in range(4):
        e.step(e.synthetic_actions())
    roots = e.state.clone()
    out, total = None, torch.zeros((), dtype=torch.int64, device=dev)
    for _ in range(5):                     # the step count's reduction too: torch loads that kernel at its first use (~0.1 s)
        out = e.rollout(depth, sims_per_root=sims, roots=roots, out=out)
        total.add_(out["n_steps"].sum())
    total.zero_()
    torch.cuda.synchronize(dev)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(40):
        out = e.rollout(depth, sims_per_root=sims, roots=roots, out=out)
        total.add_(out["n_steps"].sum())
    ev1.record()
    torch.cuda.synchronize(dev)
    wall = time.perf_counter() - t0
    launch_ms = ev0.elapsed_time(ev1) / 40
    steps_done = int(total.item())
    v = valu_roofline("rollout_%s" % env_key, "rollout_kernel<", roots_n * sims, launch_ms)
    del e, out
    torch.cuda.empty_cache()
    return {"workload": "%s: %d roots x %d simulations, depth <= %d, uniform policy over _generate_legal(), one fused rollout "
                        "launch per step" % (label, roots_n, sims, depth),
            "value": steps_done / wall, "unit": "env-steps/s (lane-steps of the simulations)", "kernel": "rollout_kernel<%s>" % env_key,
            "kernel_ms": launch_ms, "mean_steps_per_simulation": steps_done / (40.0 * roots_n * sims),
            "roofline": {"bound": "valu" if v else None, "frac": v["frac"] if v else None,
                         "counters_stale": v["counters_stale"] if v else None}}


def quick_plan_config(args, gpa, dev, env_key, roots_n, sims, depth, seed):
    """BASELINE.json configs[4] as the caller runs it — "a POMCP-style 1024-simulation rollout PER REAL STEP": every timed step
    is env.plan_step(): the fused rollout launch from the roots' live state, the on-device reduction of each root's
    simulations to action values (pomdp_plan_reduce: visits / mean return by first action, argmax) and the real step of the
    roots with the chosen actions (pomdp_<env>_step).  20 planned steps timed by wall clock and by HIP events after 3 untimed
    ones; the reduction and the real step are then timed on their own (HIP events) for their share."""
    env_id, kwargs, label, _, _ = WORKLOADS[env_key]
    e = gpa.make(env_id, batch_size=roots_n, device=dev, seed=seed, reuse_buffers=True, auto_reset=True, **kwargs)
    e.reset()
    for _ in range(4):
        e.step(e.synthetic_actions())
    out = None
    for _ in range(3):
        out = e.plan_step(depth, sims_per_root=sims, out=out)[4]
    torch.cuda.synchronize(dev)
    k = 20
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(k):
        e.plan_step(depth, sims_per_root=sims, out=out)
    ev1.record()
    torch.cuda.synchronize(dev)
    wall = time.perf_counter() - t0
    step_ms = ev0.elapsed_time(ev1) / k
    from gym_pomdp_amd import _native
    import ctypes
    po = ctypes.byref(out["_plan_out"])
    stream = torch.cuda.current_stream(dev).cuda_stream
    ev0.record()
    for _ in range(k):
        _native.lib().pomdp_plan_reduce(out["sim_ret"].data_ptr(), out["sim_first_action"].data_ptr(), roots_n, sims, e.action_space.n, po, stream)
    ev1.record()
    torch.cuda.synchronize(dev)
    reduce_ms = ev0.elapsed_time(ev1) / k
    best = out["best"].clone()
    ev0.record()
    for _ in range(k):
        e.step(best)
    ev1.record()
    torch.cuda.synchronize(dev)
    real_ms = ev0.elapsed_time(ev1) / k
    visited = float((out["visits"] > 0).sum().item()) / roots_n
    del e, out
    torch.cuda.empty_cache()
    return {"workload": "%s: %d roots, each real step planned by %d random rollouts of <= %d steps (uniform over _generate_legal()), "
                        "action values reduced on the device, roots stepped with the argmax" % (label, roots_n, sims, depth),
            "value": roots_n * k / wall, "unit": "planned real env-steps/s", "ms_per_planned_step": wall / k * 1e3,
            "simulations_per_s": roots_n * sims * k / wall, "kernel_ms": step_ms,
            "kernels": "rollout_kernel<%s> + plan_reduce_kernel + the env's step kernel" % env_key,
            "share": {"rollout": max(0.0, step_ms - reduce_ms - real_ms) / step_ms, "reduce": reduce_ms / step_ms, "real_step": real_ms / step_ms},
            "reduce_kernel_ms": reduce_ms, "reduce_hbm_frac": 12.0 * roots_n * sims / (reduce_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "real_step_ms": real_ms, "visited_actions_per_root": visited}


class StepWorkload(object):
    """K consecutive steps of one env shard under the synthetic policy, every buffer allocated up front."""

    CHUNK = 256          # steps per C-driver call = one full launch when fused (pomdp_fuse_max)

    def __init__(self, args, gpa, env_id, kwargs, dev, n, lane_offset, seed, layout=None, max_steps=0):
        self.args, self.dev, self.n = args, dev, n
        self.layout = layout or args.layout
        self.env = gpa.make(env_id, batch_size=n, device=dev, seed=seed, lane_offset=lane_offset, reuse_buffers=True, **kwargs)
        self.actions = torch.empty(n, dtype=torch.int32, device=dev)
        self.shared_key = args.action_seed is None
        self.collect = bool(args.collect) and bool(args.fuse) and args.host_loop == "c" and self.shared_key
        self.chained = args.host_loop == "c" and self.shared_key
        self.fused = self.chained and bool(args.fuse)
        self.views = {}
        self.stats = self.cols = None
        if self.collect:
            # one [CHUNK + 1][n] trajectory buffer per column; a call of c steps writes the first c (+ 1) rows
            c = min(self.CHUNK, max(args.steps, args.warmup, max_steps, 1))
            if self.layout == "returns":                               # nothing per step: the lanes' episode statistics only
                self.stats, self.traj = gpa.EpisodeStats(self.env), None
            elif self.layout == "packed_plus_decode":                  # records, then one device pass into int32 columns
                self.traj, self.cols = self.env.trajectory_buffers(c, "packed"), self.env.trajectory_buffers(c)
            else:
                self.traj = self.env.trajectory_buffers(c, self.layout)    # one allocation (columns: starts staggered, envs/base.py)
        else:
            self.layout = "columns"

    def _view(self, c):
        v = self.views.get(c)
        if v is None:
            t = self.traj
            if self.layout == "columns":
                v = {"action": t["action"][:c + 1], "ob": t["ob"][:c], "reward": t["reward"][:c], "done_u8": t["done_u8"][:c]}
                v["done"] = v["done_u8"].view(torch.bool)
            else:
                v = dict(t, traj=t["traj"][:c])
            self.views[c] = v
        return v

    def traj_tensor(self):
        """a flat view of (one column of) the buffer the timed launches write: what store_ceiling fills"""
        if self.stats is not None:
            return self.stats.acc.view(-1)
        return (self.traj["ob"] if self.layout == "columns" else self.traj["traj"]).view(-1)

    def reseed(self, seed):
        self.env.seed(seed)
        self.env.reset()

    def action_seed(self):
        return self.env._seed if self.shared_key else self.args.action_seed

    def run(self, k):
        env, left = self.env, k
        if self.collect:
            while left > 0:
                c = min(left, self.CHUNK)
                if self.stats is not None:
                    env.collect_returns(c, self.stats)
                else:
                    env.collect_synthetic(c, out=self._view(c))
                    if self.cols is not None:
                        env.decode_trajectory(self._view(c), into=self.cols)
                left -= c
        elif self.args.host_loop == "python":
            for _ in range(k):
                env.synthetic_actions(out=self.actions, seed=self.action_seed())
                env.step(self.actions)
        else:
            while left > 0:
                c = min(left, self.CHUNK)
                env.rollout_synthetic(c, action_seed=self.action_seed(), actions=self.actions, fuse=self.fused)
                left -= c

    def measure(self, seeds, cp):
        """The §8d protocol.  -> (per-seed [(seed, median wall s, min, max, median event ms)])."""
        args, k = self.args, self.args.steps
        repeats = args.repeats or (31 if k < 1000 else 7)
        rows = []
        for i, seed in enumerate(seeds):
            self.reseed(seed)
            if i == 0:
                prewarm(args, self.run, self.dev, k)      # same chunking as the timed region: every row it writes is touched
            self.run(args.warmup)
            self.run(k)                                    # untimed pass over exactly the timed call sequence
            walls, evs = timed_regions(self.run, k, repeats, self.dev, cp)
            rows.append((seed, median(walls), min(walls), max(walls), median(evs)))
        return rows, repeats


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    seeds = [args.seed] if args.seed is not None else [int(x) for x in args.seeds.split(",") if x != ""]
    args.seed = seeds[0]
    n_dev = torch.cuda.device_count()
    if n_dev < world and not args.share_gpus:
        raise SystemExit("bench.py: %d ranks but %d visible GPU(s) — one GPU per rank, or --share-gpus to let ranks share "
                         "devices (development only: the line would still say n_gpus: %d)" % (world, n_dev, world))
    dev_index = local_rank % max(n_dev, 1)      # --share-gpus: fewer GPUs than ranks, ranks share devices
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    import gym_pomdp_amd as gpa
    from gym_pomdp_amd import _native, sharding
    cp = sharding.ControlPlane()    # gloo, host side: barrier + max of the timings; no data-path collective
    if not os.path.exists(_native.LIB_PATH):      # normally prebuilt by __graft_entry__.build(); never a CPU fallback
        if rank == 0:
            _native.build(verbose=True)
        cp.barrier()

    env_id, kwargs, label, bytes_per_step, dtype = WORKLOADS[args.env]
    if args.scaling == "strong":
        lane_offset, n = sharding.shard_range(args.lanes_per_gpu, rank, world)   # the total batch split over the GPUs
        assert n == args.lanes_per_gpu // world, "strong scaling needs a batch divisible by 4 * gpus"
    else:
        n = args.lanes_per_gpu
        lane_offset, count = sharding.shard_range(n * world, rank, world)        # n lanes on every GPU
        assert count == n
    if args.mode == "heuristic":
        return heuristic_mode(args, gpa, env_id, kwargs, cp, dev, rank, world, label, n, lane_offset)
    if args.mode == "rollout":
        env = gpa.make(env_id, batch_size=n, device=dev, seed=args.seed, lane_offset=lane_offset, reuse_buffers=True, **kwargs)
        return rollout_mode(args, env, cp, dev, rank, world, label)

    shards = cp.gather((lane_offset, n, dev_index))        # who owns which global lanes, on which device
    wl = StepWorkload(args, gpa, env_id, kwargs, dev, n, lane_offset, seeds[0])
    env = wl.env
    rows, repeats = wl.measure(seeds, cp)
    elapsed = median([r[1] for r in rows])
    timed_kernel_ms = median([r[4] for r in rows]) / args.steps
    fused_kernel = _native.lib().pomdp_last_fused_kernel().decode() if wl.fused else None
    chained, fused, collect = wl.chained, wl.fused, wl.collect
    action_seed = wl.action_seed()

    # ---- the same K steps on a batch of --lanes-per-gpu lanes IN TOTAL, split over the GPUs (N > 1, weak runs) -------
    strong = None
    if world > 1 and args.scaling == "weak" and args.lanes_per_gpu % (4 * world) == 0:
        off_s, n_s = sharding.shard_range(args.lanes_per_gpu, rank, world)
        wl_s = StepWorkload(args, gpa, env_id, kwargs, dev, n_s, off_s, seeds[0])
        rows_s, _ = wl_s.measure(seeds[:1], cp)
        strong = {"total_lanes": args.lanes_per_gpu, "lanes_per_gpu": n_s, "value": args.lanes_per_gpu * args.steps / rows_s[0][1],
                  "unit": "env-steps/s", "ms_per_step": rows_s[0][1] / args.steps * 1e3, "seed": seeds[0],
                  "kernel": _native.lib().pomdp_last_fused_kernel().decode() if wl_s.fused else None}
        # How far the shard's launch is from its floor: a small shard has two to four waves per SIMD and is
        # bound by instruction issue at best — floor = the launch's recorded vector instructions x the cycles its mix costs
        # / the SIMDs' cycles, i.e. frac_of_floor is the VALU-issue fraction of the shard's kernel (None: no recorded pass
        # for this shard size and launch length, tools/gpu_pmc_valu.sh shards).
        strong["frac_of_floor"] = strong["floor_source"] = None
        if wl_s.fused and strong["kernel"]:
            spl_s = min(fuse_max(args.env, wl_s.layout), StepWorkload.CHUNK, args.steps)
            v = valu_roofline(valu_workload_key(args.env, spl_s, wl_s.layout, n_s), strong["kernel"].split("<")[0] + "<", n_s,
                              rows_s[0][4] / args.steps * spl_s)
            if v is not None:
                strong.update({"frac_of_floor": v["frac"], "kernel_ms": rows_s[0][4] / args.steps, "floor_ms_per_step": v["frac"] * rows_s[0][4] / args.steps,
                               "counters_stale": v["counters_stale"], "floor_source": v["source"]})
        del wl_s

    # ---- the single-step kernels, for reference: HIP events on their stream ------------------------------------------
    # A ring of pre-generated action batches keeps the action distribution of the timed region
    # (i.i.d. uniform every step); replaying ONE batch would make lanes repeat their action forever.
    k1 = min(max(args.steps, 256), 1024)
    ring = []
    for j in range(16):
        a = torch.empty(n, dtype=torch.int32, device=dev)
        env.call_counter = env.call_counter + 1
        env.synthetic_actions(out=a, seed=action_seed)
        ring.append(a)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for j in range(64):
        env.step(ring[j & 15])
    torch.cuda.synchronize(dev)
    ev0.record()
    for j in range(k1):
        env.step(ring[j & 15])
    ev1.record()
    torch.cuda.synchronize(dev)
    plain_ms = ev0.elapsed_time(ev1) / k1
    plain_achieved = bytes_per_step * n / (plain_ms * 1e-3) / 1e9
    kern_ms = timed_kernel_ms if chained else plain_ms
    spl = min(fuse_max(args.env, wl.layout if collect else None), StepWorkload.CHUNK, args.steps) if fused else 1
    layout = wl.layout
    rf = step_rooflines(args.env, bytes_per_step, layout, n, kern_ms, spl, fused, fused_kernel)
    chain1_ms = None
    if fused:      # the same chained steps launched one by one (step_kernel<., chain>), for reference
        env.rollout_synthetic(64, action_seed=action_seed, actions=wl.actions, fuse=False)
        torch.cuda.synchronize(dev)
        ev0.record()
        env.rollout_synthetic(k1, action_seed=action_seed, actions=wl.actions, fuse=False)
        ev1.record()
        torch.cuda.synchronize(dev)
        chain1_ms = ev0.elapsed_time(ev1) / k1
    invalid = env.invalid_action_count()
    # What a write stream reaches on THIS device into THIS buffer (measured here, after the timed regions): a plain fill of the
    # trajectory buffer (columns: its ob column) — one 16-byte store stream over the very pages the timed launches wrote.
    store_ceiling = None
    if collect:
        col = wl.traj_tensor()
        for _ in range(3):
            col.fill_(0)
        torch.cuda.synchronize(dev)
        ev0.record()
        for _ in range(10):
            col.fill_(0)
        ev1.record()
        torch.cuda.synchronize(dev)
        store_ceiling = {"value": col.numel() * col.element_size() * 10 / (ev0.elapsed_time(ev1) * 1e-3) / 1e9, "unit": "GB/s",
                         "how": "torch fill over the trajectory buffer%s (%d MB, one 16-byte store stream), HIP events; "
                                "untimed, after the regions" % ("'s ob column" if layout == "columns" else "", col.numel() * col.element_size() >> 20)}

    # ---- the same workload in the other trajectory layouts, and BASELINE.json's configs[2..4] at their per-GPU sizes --------
    # (single-GPU default runs only; short HIP-event measurements of the same protocol, after the headline's timed regions)
    layouts_block = configs_block = None
    extras = world == 1 and not args.no_extras and collect
    if extras:
        layouts_block = {}
        # (the 4-byte sinks first: the 13-byte layouts are store-heavy launches under which the clock drops to 2.1-2.3 GHz, and an
        #  instruction-bound launch measured right after them reads 10-15 % slower than it does on its own)
        for lay in ("packed", "narrow", "columns", "blocked", "packed_plus_decode"):
            if lay == layout:
                layouts_block[lay] = {"value": n * args.steps / elapsed, "ms_per_step": elapsed / args.steps * 1e3, "kernel": fused_kernel,
                                      "kernel_ms": kern_ms, "bytes_per_lane_step": rf["alg_bytes"],
                                      "roofline": {"bound": rf["primary"]["bound"], "frac": rf["primary"]["frac"], "hbm_frac": rf["hbm"]["frac"],
                                                   "valu_frac": None if rf["valu"] is None else rf["valu"]["frac"]}, "headline": True}
            else:
                layouts_block[lay] = quick_step_config(args, gpa, _native, cp, dev, args.env, n, lane_offset, seeds[0], lay)
    if extras and args.env == "rock":
        configs_block = {
            "tag": quick_step_config(args, gpa, _native, cp, dev, "tag", 1 << 20, 0, seeds[0], layout, k=512),            # configs[2]
            "battleship": quick_step_config(args, gpa, _native, cp, dev, "battleship", 1 << 19, 0, seeds[0], layout, k=512),   # configs[3]: 2^22 / 8 GPUs
            "rollout_rock15": quick_rollout_config(args, gpa, dev, "rock15", 2048, 1024, 64, seeds[0]),                 # configs[4]: 2^24 / 8 GPUs
            "plan_rock15": quick_plan_config(args, gpa, dev, "rock15", 2048, 1024, 64, seeds[0]),                      # ... as planned real steps
            # the headline workload on the CALLER's actions instead of the synthetic policy's (pomdp_collect_tape*)
            "tape_packed": quick_tape_config(args, gpa, _native, cp, dev, "rock", 1 << 20, seeds[0], "packed"),
            "tape_returns": quick_tape_config(args, gpa, _native, cp, dev, "rock", 1 << 20, seeds[0], "returns"),
            # the headline workload reduced on the fly to what the reference's callers keep of it (network.py:175-191): no trajectory
            "returns_only": quick_step_config(args, gpa, _native, cp, dev, "rock", 1 << 20, 0, seeds[0], "returns", k=512)}

    # the recorded PMC figure belongs to a launch of the recorded shape only (2^20 lanes, this many steps per fused launch)
    traffic, traffic_src = (None, None)
    kprefix = (fused_kernel or "").split("<")[0] + "<" if fused else "step_kernel<"
    if fused and n == 1 << 20:
        traffic, traffic_src = recorded_traffic(args.env, layout, spl, kprefix)
    rank_kernel_ms = cp.gather(timed_kernel_ms)            # a slow GPU shows in the one line the driver keeps
    if rank == 0:
        total_lanes = n * world
        metric = "env steps/sec (whole node)"
        if args.env == "rock":          # the headline: BASELINE.json's metric string, verbatim
            try:
                with open(os.path.join(REPO, "BASELINE.json")) as f:
                    metric = json.load(f)["metric"]
            except Exception:  # noqa: BLE001
                metric = "env steps/sec (whole node), RockSample(7,8) batch=2^20, 1/2/4/8 MI355X"
        seed_values = [total_lanes * args.steps / r[1] for r in rows]
        if fused:
            kernel_name = "%s — %d chained steps per launch: step + next-step policy, every step's outputs written%s, " \
                          "state in registers between steps; the only kernel of the timed region" % (
                              fused_kernel, spl, (" to its own trajectory row (%s layout)" % layout) if collect else " over the previous step's")
        elif chained:
            kernel_name = "step_kernel<%s, chain> (step + next-step policy, the launch of the timed region)" % args.env
        else:
            kernel_name = "step_kernel<%s>" % args.env
        primary, hbm, valu = rf["primary"], rf["hbm"], rf["valu"]
        roof = dict(primary)
        roof.update({
            "traffic": traffic, "traffic_unit": "bytes per %d-step launch" % spl if fused else "bytes per launch",
            "traffic_source": traffic_src,
            "hbm": hbm, "valu": valu,
            "tighter_bound": None if valu is None else ("valu" if valu["frac"] > hbm["frac"] else "hbm"),
            "counters_stale": None if valu is None else valu["counters_stale"],
            "store_ceiling": store_ceiling,
            "kernel_ms_by_rank": rank_kernel_ms,
            "kernel": kernel_name,
            "kernel_ms": kern_ms, "algorithmic_bytes_per_step": rf["alg_bytes"],
            "algorithmic_bytes_per_step_unfused": bytes_per_step,
            "steps_per_launch": spl,
            "launch_ms": kern_ms * spl,
            "chained_step_kernel": None if chain1_ms is None else {
                "kernel": "step_kernel<%s, chain> (the same steps, one launch each)" % args.env,
                "kernel_ms": chain1_ms, "achieved": bytes_per_step * n / (chain1_ms * 1e-3) / 1e9,
                "frac": bytes_per_step * n / (chain1_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
            "plain_step_kernel": {"kernel": "%s<%s> (what env.step() launches)" % (
                "step_quad_kernel" if args.env in ("rock", "rock15", "stochrock") and n >= (1 << 19) and n % 1024 == 0
                else "step_kernel", args.env),
                                  "kernel_ms": plain_ms, "achieved": plain_achieved,
                                  "frac": plain_achieved / HBM_PEAK_GBS},
            "note": "kernel_ms: HIP events on the launch stream around each timed region of %d back-to-back "
                    "steps (median over regions and seeds) / steps; launch_ms = kernel_ms x steps_per_launch; "
                    "algorithmic bytes: the layout's bytes of outputs per lane-step (13, packed: 4) + state in/out (columns: "
                    "and the first actions) once per fused launch, the full per-step figure for the single-step kernels; the "
                    "object's own bound/achieved/peak/frac are the TIGHTER of the two rooflines when the VALU counters of this "
                    "kernel are on record (profiles/*_pmc_valu.json, csrc hash checked), else HBM; both are kept under "
                    "`hbm` / `valu`; plain_step_kernel: env.step() on a ring of 16 pre-generated action batches"
                    % args.steps})
        out = {
            "metric": metric,
            "value": total_lanes * args.steps / elapsed,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": dtype,
            "data": "synthetic",
            "config": {"workload": "%s batch=%d lanes per GPU (%d total, %s scaling), uniform random actions from the "
                                   "synthetic policy (generated inside the timed launches), auto-reset, every step's "
                                   "action / ob / reward / done kept in the %s trajectory layout"
                                   % (label, n, total_lanes, args.scaling, layout),
                       "lanes_per_gpu": n, "total_lanes": total_lanes, "host_loop": args.host_loop, "visible_gpus": n_dev,
                       "steps_per_launch": spl, "trajectories_kept": collect, "trajectory_layout": layout if collect else None,
                       "trajectory_bytes_per_lane_step": {"packed": 4, "narrow": 4, "returns": 0}.get(layout, 13) if collect else None,
                       "seeds": seeds, "repeats": repeats,
                       "protocol": "per seed: reseed + reset, W warm-up steps, one untimed pass of K steps, then `repeats` "
                                   "timed regions of exactly K steps (barrier + device sync on both sides, max over ranks); "
                                   "value = median over seeds of the per-seed median region",
                       "seed_values": {"per_seed": seed_values, "min": min(seed_values), "median": median(seed_values),
                                       "max": max(seed_values),
                                       "region_ms": [{"seed": r[0], "median": r[1] * 1e3, "min": r[2] * 1e3, "max": r[3] * 1e3}
                                                     for r in rows]},
                       "untimed_prewarm_s": args.prewarm, "parallelism": "lane-shard x%d, no collectives" % world,
                       "shards": [{"rank": r, "lane_offset": s[0], "lanes": s[1], "device": s[2]} for r, s in enumerate(shards)],
                       "host_cpus": os.cpu_count(),
                       "host_cpus_usable": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None},
            "roofline": roof,
            "invalid_actions": invalid,
        }
        if layouts_block is not None:
            out["layouts"] = layouts_block
        if configs_block is not None:
            out["configs"] = configs_block
        if strong is not None:
            out["strong_scaling"] = strong
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.env, kwargs, args.seed, args.cpu_seconds)
            # the like-for-like "one env, python loop" usage of the reference, through this build's public API
            # (batch_size=1: python scalars in and out, one launch + one sync per step)
            e1 = gpa.make(env_id, device=dev, seed=args.seed, **kwargs)
            e1.reset()
            k, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < 1.0:
                _, _, d, _ = e1.step(e1.action_space.sample())
                if d:
                    e1.reset()
                k += 1
            out["cpu_baseline"]["scalar_api_loop"] = {
                "value": k / (time.perf_counter() - t0), "unit": "env-steps/s",
                "note": "gym_pomdp_amd batch_size=1 python loop on the GPU (launch + sync per step); the reference's "
                        "own python loop is in reference_python_recorded"}
        print(json.dumps(out), flush=True)
    cp.close()


if __name__ == "__main__":
    main()
