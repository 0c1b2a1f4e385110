"""Host-side logic that needs no GPU: the C-ABI library loads and exports every symbol the header
declares, params structs mirror the header, config validation mirrors the reference's constructor
behaviour, spaces follow gym's rules, and the product path fails loudly without a GPU."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, REPO


def test_library_exports_every_declared_symbol():
    from gym_pomdp_amd import _native
    _native.build()
    hdr = open(os.path.join(REPO, "include", "pomdp_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(pomdp_[a-z_0-9]+)\s*\(", hdr)))
    assert declared, "no declarations parsed"
    lib = C.CDLL(_native.LIB_PATH)
    for sym in declared:
        assert hasattr(lib, sym), sym
    assert sorted(_native.SYMBOLS) == declared
    L = _native.lib()
    assert L.pomdp_abi_version() == _native.ABI_VERSION
    assert L.pomdp_error_string(0) == b"ok"
    assert b"bad argument" in L.pomdp_error_string(-1)


def test_bad_arguments_are_rejected_without_a_gpu():
    from gym_pomdp_amd import _native
    from gym_pomdp_amd.envs import rock
    L = _native.lib()
    p = rock.make_params()[0]
    assert L.pomdp_rock_step(C.byref(p), None, None, None, None, None, None, 16, 0, 0, 0, 1, None) == -1
    assert L.pomdp_rock_reset(C.byref(p), None, None, 16, 0, 0, 0, None) == -1
    p.size = 99
    assert L.pomdp_rock_reset(C.byref(p), None, None, 16, 0, 0, 0, None) == -2
    assert L.pomdp_synthetic_actions(None, 16, 0, 0, 0, 13, None) == -1
    q = rock.make_params()[0]
    q.rock_x[0] = 9                       # off a 7x7 board
    assert L.pomdp_rock_reset(C.byref(q), None, None, 16, 0, 0, 0, None) == -2
    from gym_pomdp_amd import _native as nn
    b = nn.BattleShipParams()
    b.x_size, b.y_size, b.max_len = 3, 3, 3
    assert L.pomdp_battleship_reset(C.byref(b), None, None, 16, 0, 0, 0, None) == -2


def test_struct_sizes_match_header_layout():
    from gym_pomdp_amd import _native as n
    assert C.sizeof(n.RockParams) == 16 + 16 + 16 + 256 + 32 * 8 + 32 * 8 + 16
    assert C.sizeof(n.TagParams) == 24
    assert C.sizeof(n.BattleShipParams) == 16 + 16 + 12 * 16
    assert C.sizeof(n.TigerParams) == 8
    assert C.sizeof(n.NetworkParams) == 8 + 32 * 4 + 3 * 8
    assert n.RockParams.grid.offset == 48 and n.RockParams.thr.offset == 304


def test_ctypes_structs_match_the_header_as_gcc_lays_it_out(tmp_path):
    """Every struct of include/pomdp_hip.h against its ctypes mirror: sizeof and the offset of every field, from a
    program gcc compiles against the header itself."""
    import subprocess
    from gym_pomdp_amd import _native as n
    pairs = {"pomdp_rock_params": n.RockParams, "pomdp_tag_params": n.TagParams, "pomdp_battleship_params": n.BattleShipParams,
             "pomdp_tiger_params": n.TigerParams, "pomdp_network_params": n.NetworkParams, "pomdp_step_args": n.StepArgs,
             "pomdp_collect_args": n.CollectArgs, "pomdp_rock_belief": n.RockBelief, "pomdp_history": n.HistoryPtrs,
             "pomdp_returns": n.Returns, "pomdp_traj_args": n.TrajArgs, "pomdp_return_stats": n.ReturnStats}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "pomdp_hip.h"', 'int main(void) {']
    for cname, cls in pairs.items():
        lines.append('printf("%s %%zu\\n", sizeof(%s));' % (cname, cname))
        for f in cls._fields_:
            lines.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (cname, f[0], cname, f[0]))
    lines.append('return 0; }')
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(REPO, "include"), "-o", str(exe), str(src)])
    got = dict(l.split() for l in subprocess.check_output([str(exe)], text=True).splitlines())
    for cname, cls in pairs.items():
        assert int(got[cname]) == C.sizeof(cls), cname
        for f in cls._fields_:
            assert int(got["%s.%s" % (cname, f[0])]) == getattr(cls, f[0]).offset, (cname, f[0])


def test_tables_match_captured_thresholds():
    from gym_pomdp_amd import tables
    with open(os.path.join(GOLDEN, "thresholds.json")) as f:
        thr = json.load(f)
    assert list(tables.ROCK_THR) == thr["rock_thr"]
    assert tables.TAG_MOVE_THR == thr["tag_move"]["thr"]
    assert tables.NET_FAIL_THR == thr["net_fail"]["thr"]
    assert tables.NET_FAIL_NEIGHBOUR_THR == thr["net_fail_neighbour"]["thr"]
    assert tables.NET_OBS_THR == thr["net_obs"]["thr"]
    assert tables.TIGER_LISTEN_THR == thr["tiger_listen"]["thr"]
    assert tables.bernoulli_threshold(.8) == (tables.TAG_MOVE_THR, "le")
    assert tables.bernoulli_threshold(.1) == (tables.NET_FAIL_THR, "gt")


def test_rock_params_follow_reference_config():
    from gym_pomdp_amd.envs import rock
    with open(os.path.join(GOLDEN, "edge_cases.json")) as f:
        edge = json.load(f)
    for bs, k in ((7, 8), (7, 7), (7, 6), (3, 3), (11, 11), (15, 15), (2, 1), (4, 3)):
        if edge["rock.ctor_%d_%d" % (bs, k)] == "ok":
            p, words, nA, nO = rock.make_params(bs, k)
            assert (nA, nO) == (5 + k, 3) and words == (1 if k <= 12 else 2)
        else:
            with pytest.raises(AssertionError):
                rock.make_params(bs, k)
    with pytest.raises(IndexError):      # passes the reference's assert, IndexError in its reset()
        rock.make_params(4, 4)
    p = rock.make_params(15, 15)[0]
    assert p.grid[1 * 16 + 2] == 3       # (1,2) is listed twice: the later index wins (rock.py:110-111)
    assert p.grid[12 * 16 + 2] == 15     # stamped although num_rocks == 15
    p = rock.make_params(7, 8)[0]
    assert (p.start_x, p.start_y) == (0, 3) and p.grid[6 * 16 + 3] == 3 and p.grid[0] == -1


def test_network_params_follow_reference_topology():
    from gym_pomdp_amd.envs import network
    with open(os.path.join(GOLDEN, "edge_cases.json")) as f:
        edge = json.load(f)
    p = network.make_params(10, 3)[0]
    for i, nb in enumerate(edge["network.make_3legs_10"]):
        assert p.nb_mask[i] == sum(1 << j for j in nb)
    assert p.deg_gt2_mask == 1      # only machine 0 has more than 2 neighbours
    ring = network.make_params(5, 1)[0]
    assert ring.nb_mask[0] == (1 << 1) | (1 << 4) and ring.deg_gt2_mask == 0


def test_other_params_validation():
    from gym_pomdp_amd.envs import battleship, tag
    assert battleship.make_params((10, 10), 5)[1:] == (12, 100, 2)      # occupied + visited + next-episode mask words
    assert battleship.make_params((5, 5), 3)[1:] == (3, 25, 2)
    with pytest.raises(ValueError):
        battleship.make_params((12, 12), 5)
    with pytest.raises(ValueError):          # the reference's reset() would never terminate on this board
        battleship.make_params((3, 3), 3)
    assert tag.make_params()[1:] == (1, 5, 30)
    with pytest.raises(ValueError):
        tag.make_params(num_opponents=5)
    with pytest.raises(ValueError):
        tag.make_params(board_size=(8, 8))


def test_discrete_space_follows_gym_rules():
    from gym_pomdp_amd.spaces import Discrete
    d = Discrete(5)
    assert d.n == 5 and d.contains(0) and d.contains(4) and not d.contains(5) and not d.contains(-1)
    assert d.contains(np.int64(3)) and d.contains(np.array(2)) and not d.contains(np.array([2]))
    assert not d.contains(1.0) and not d.contains(np.float32(1)) and not d.contains("1")
    assert all(0 <= d.sample() < 5 for _ in range(50))
    assert Discrete(3) == Discrete(3) and Discrete(3) != Discrete(4)


def test_registry_and_loud_failure_without_gpu():
    import torch
    import gym_pomdp_amd as gpa
    assert sorted(gpa.registry) == ["Battleship-v0", "Network-v0", "Rock-v0", "StochasticRock-v0", "Tag-v0", "Tiger-v0"]
    with pytest.raises(KeyError):
        gpa.make("Pocman-v0")
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no GPU"):
            gpa.make("Rock-v0", batch_size=4)


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under gym_pomdp_amd/ may reference it."""
    pkg = os.path.join(REPO, "gym_pomdp_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(root, f)).read()
                assert "oracle" not in src.replace("no CPU fallback", ""), os.path.join(root, f)


def test_stream_and_device_handles_fall_back_to_the_public_api():
    """The hot paths take torch's raw stream handle and current device through two private C functions; they are resolved
    once at import and anything missing falls back to torch.cuda.current_stream(dev).cuda_stream / current_device()."""
    import torch
    from gym_pomdp_amd.envs import base

    class Nothing(object):
        pass

    raw, cur = base._resolve_fast_handles(Nothing())
    assert raw is base._public_raw_stream and cur is torch.cuda.current_device

    class Half(object):
        _cuda_getDevice = staticmethod(lambda: 3)
        _cuda_getCurrentRawStream = "not callable"

    raw, cur = base._resolve_fast_handles(Half())
    assert raw is base._public_raw_stream and cur() == 3
    assert callable(base._raw_stream) and callable(base._current_device)
    src = open(os.path.join(REPO, "gym_pomdp_amd", "envs", "base.py")).read()
    assert src.count("torch._C") == 1, "private torch internals are touched in _resolve_fast_handles only"


def test_fused_step_loops_never_wait_for_their_own_stores():
    """Static property of the compiled kernels (tools/check_loop_waits.py, no GPU needed): inside the step loop of the
    headline kernel and of its small-shard sibling there is no `s_waitcnt vmcnt` — gfx9 counts loads and stores on one
    counter, so such a wait would stall every step on the previous step's stores — and the priority ladder (LoopPrio:
    four s_setprio, one per segment) is compiled in, outside the step loop."""
    import sys
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import check_loop_waits as clw
    res = clw.loop_waits(clw.assembly("fused_rock.hip"))
    seen = taped = taped_small = 0
    for name, (waits, prio) in res.items():
        if "steps_quad_kernel<pomdp::RockEnv<1, false>, " in name or "steps_kernel<pomdp::RockEnv<1, false>, 1, true, true, " in name:
            assert prio == 4, (name, prio)
            if "TapeQuad" in name:
                # the tape-driven loops (pomdp_collect_tape*) read one row per step: requested at the top of the step, first touched
                # at its end — ONE wait per iteration, a whole step after the previous step's stores were issued (traj_out.hip.h)
                taped += 1
                # (the quad loop reads its tape two steps ahead in a loop unrolled by two — TapeQuadAhead: one wait per step, two
                # per iteration)
                two_ahead = "TapeQuad, 4>(" in name
                assert len(waits) == (2 if two_ahead else 1), (name, waits)
                continue
            if name.split("(")[0].rstrip().endswith(", true>"):
                # the one-lane-per-thread loop on a tape (last template argument): unrolled by four, the tape read four rows ahead
                # — it waits for its rows, four to nine times per iteration, never for nothing
                taped_small += 1
                assert 4 <= len(waits) <= 9, (name, waits)
                continue
            seen += 1
            assert waits == [], (name, waits)
    # the quad kernel and the pooled one with each of the five sinks (traj_out.hip.h: three layouts of round 4, Narrow, Returns),
    # the half-quad-per-thread form of the quad kernel (the shards between 3 * 2^17 and 3 * 2^18 lanes) with Packed / Narrow
    assert seen == 12, sorted(res)
    assert taped == 7, sorted(res)          # the quad loop on a tape, each sink; its half-quad form, two sinks
    assert taped_small == 5, sorted(res)    # the small shards' loop on a tape, each sink


def test_step_loops_keep_nothing_in_scratch_memory():
    """Static property of the compiled kernels (tools/kernel_resources.py: hipcc's resource remarks, no GPU needed): no step kernel
    spills to, or indexes an array in, scratch memory.  Round 6 shipped the general tape loop for a day with its time-shared words
    in a 32-byte scratch array read at a run-time index — and a `s_waitcnt vmcnt(0)` behind every read: results stay right, the GPU
    suite stays green, only the time shows it.  Known and documented exceptions: the heuristic policy's loop and
    BattleShip's returns sink with a quad of three- or four-word boards per thread (128 registers)."""
    import sys
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import kernel_resources as kr
    rows = kr.collect()
    assert len(rows) > 300
    known = lambda k: k.startswith("heuristic_steps_kernel<") or (k.startswith("battleship_steps_quad_kernel<") and "Returns<" in k and ", 4, false>" in k)
    bad = [(r["kernel"], r["scratch"]) for r in rows if r["scratch"] not in ("0", "?") and not known(r["kernel"])]
    assert not bad, bad
    assert sum(1 for r in rows if r["kernel"].startswith(("steps_kernel<", "steps_quad_kernel<"))) > 150


def test_library_override_by_environment_variable():
    """GYM_POMDP_AMD_LIB points the package at another build of the library (tools/ab_build.sh variants for same-box A/B
    runs); unset, the in-tree product library is what loads."""
    import subprocess
    import sys
    from gym_pomdp_amd import _native
    code = "from gym_pomdp_amd import _native; print(_native.LIB_PATH)"
    env = dict(os.environ, GYM_POMDP_AMD_LIB="/nonexistent/libpomdp_hip_x.so")
    out = subprocess.run([sys.executable, "-c", code], cwd=REPO, env=env, capture_output=True, text=True).stdout.strip()
    assert out == "/nonexistent/libpomdp_hip_x.so"
    env.pop("GYM_POMDP_AMD_LIB")
    out = subprocess.run([sys.executable, "-c", code], cwd=REPO, env=env, capture_output=True, text=True).stdout.strip()
    assert out == os.path.join(REPO, "gym_pomdp_amd", "_lib", "libpomdp_hip.so") == _native.LIB_PATH or os.environ.get("GYM_POMDP_AMD_LIB")


def test_recorded_counters_belong_to_this_tree():
    """bench.py's VALU roofline and `traffic` divide / quote counters RECORDED by tools/gpu_pmc_valu.sh (profiles/*_pmc_valu.json)
    and an instruction mix compiled by tools/isa_mix.py (profiles/*_isa_mix.json).  Both files carry the sha256 of the kernel
    sources they were taken from; a tree whose sources differ makes `roofline.counters_stale` true in the bench line — and
    fails here, for the workloads the driver's line is built from (the headline kernel in both timed launch shapes)."""
    import json
    import sys
    sys.path.insert(0, REPO)
    import bench
    sha = bench.csrc_sha()
    assert len(sha) == 64 and sha == bench.csrc_sha()
    pmc, mix = bench._latest("*_pmc_valu.json"), bench._latest("*_isa_mix.json")
    assert pmc and mix
    assert json.load(open(mix)).get("csrc_sha256") == sha, "re-run tools/isa_mix.py --json profiles/<tag>_isa_mix.json"
    for wl in ("step20_rock_packed", "step256_rock_packed"):
        assert not bench.counters_stale(pmc, wl), "kernel sources changed since %s [%s] was recorded: re-run tools/gpu_pmc_valu.sh <tag> headline" % (pmc, wl)
        v = bench.valu_roofline(wl, "steps_quad_kernel<", 1 << 20, 0.04 if wl.startswith("step20") else 0.38)
        assert v is not None and v["counters_stale"] is False and v["kernel"] == "steps_quad_kernel<RockEnv<1, false>, Packed, SyntheticQuad, 4>" and 0.3 < v["frac"] < 1.0, v
    # the shards of a 2^20-lane batch over 8 / 4 / 2 GPUs, in the default and the driver's launch shape: what
    # `strong_scaling.frac_of_floor` of a multi-GPU line is computed from
    # (2^19 lanes: the half-quad-per-thread form of the quad loop since round 6; below, the one-lane-per-thread loop)
    for lg, launch_us, family in ((17, (12.0, 100.0), "steps_kernel<"), (18, (16.0, 140.0), "steps_kernel<"), (19, (26.0, 240.0), "steps_quad_kernel<")):
        for spl, us in zip((20, 256), launch_us):
            key = bench.valu_workload_key("rock", spl, "packed", 1 << lg)
            assert key == "step%d_rock_packed_2e%d" % (spl, lg)
            v = bench.valu_roofline(key, family, 1 << lg, us * 1e-3)
            assert v is not None and v["counters_stale"] is False and 0.3 < v["frac"] < 1.0, (key, v)
    # a changed source file flips the flag (the hash covers every file under csrc/ and the C header)
    assert bench.counters_stale(pmc, "no_such_workload")
    t, _ = bench.recorded_traffic("rock", "packed", 20, "steps_quad_kernel<")
    assert t is not None and 0.9 < t / (4.4 * (1 << 20) * 20) < 1.25, t
