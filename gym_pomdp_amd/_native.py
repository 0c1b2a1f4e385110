"""ctypes binding of libpomdp_hip.so (C ABI: include/pomdp_hip.h).

The HIP library is the product: there is no CPU fallback.  `lib()` raises if the
shared object is missing or does not export the ABI this package was written for.
"""
import ctypes as C
import os
import shutil
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
_REPO = os.path.dirname(_PKG)
INTREE_LIB_PATH = os.path.join(_PKG, "_lib", "libpomdp_hip.so")
# GYM_POMDP_AMD_LIB: another build of the same library (tools/ab_build.sh variants for same-box A/B runs).  It redirects
# what lib() LOADS; build() writes the in-tree path unless it is given `out` (a rebuild never overwrites a variant).
LIB_PATH = os.environ.get("GYM_POMDP_AMD_LIB") or INTREE_LIB_PATH
# one object per translation unit (built in parallel), linked into one shared library
UNITS = ["api.hip", "step_rock.hip", "step_other.hip", "fused_rock.hip", "fused_stochrock.hip", "fused_tag.hip",
         "fused_battleship.hip", "fused_misc.hip", "planner.hip"]
HEADERS = ["kernels_common.hip.h", "traj_out.hip.h", "step_impl.hip.h", "fused_impl.hip.h", "envs.hip.h", "envs_common.hip.h", "philox.hip.h",
           "envs/rock.hip.h", "envs/tag.hip.h", "envs/battleship.hip.h", "envs/tiger.hip.h", "envs/network.hip.h"]
SOURCES = [os.path.join(_PKG, "csrc", f) for f in UNITS + HEADERS]
HEADER = os.path.join(_REPO, "include", "pomdp_hip.h")
ABI_VERSION = 14

POMDP_AUTO_RESET = 1
POMDP_FUSE_STEPS = 2
POMDP_ROLLOUT_ALL_ACTIONS = 1
LAYOUTS = {"columns": 0, "blocked": 1, "packed": 2, "narrow": 3}     # POMDP_LAYOUT_*
FUSE_MAX_DEFAULT = 256
ENV_KIND = {"rock": 0, "tag": 1, "battleship": 2, "tiger": 3, "network": 4}

# every symbol include/pomdp_hip.h declares
SYMBOLS = [
    "pomdp_abi_version", "pomdp_error_string", "pomdp_last_fused_kernel",
    "pomdp_rock_reset", "pomdp_rock_step", "pomdp_tag_reset", "pomdp_tag_step",
    "pomdp_battleship_reset", "pomdp_battleship_step", "pomdp_tiger_reset", "pomdp_tiger_step",
    "pomdp_network_reset", "pomdp_network_step", "pomdp_step", "pomdp_step_sync", "pomdp_reset_sync", "pomdp_stream_sync", "pomdp_synthetic_actions", "pomdp_philox_blocks",
    "pomdp_rollout_synthetic", "pomdp_collect_synthetic", "pomdp_collect", "pomdp_collect_layout", "pomdp_collect_traj", "pomdp_packed_reward", "pomdp_decode_packed", "pomdp_collect_returns", "pomdp_collect_tape", "pomdp_collect_tape_layout", "pomdp_collect_tape_returns", "pomdp_fuse_max", "pomdp_fuse_steps", "pomdp_legal_actions", "pomdp_rollout", "pomdp_plan", "pomdp_plan_reduce", "pomdp_compute_prob",
    "pomdp_rock_belief_reset", "pomdp_rock_belief_refresh", "pomdp_rock_belief_update", "pomdp_rock_select_target", "pomdp_history_clear",
    "pomdp_history_append", "pomdp_preferred_actions", "pomdp_pick_actions", "pomdp_heuristic_steps",
]


class RockParams(C.Structure):
    _fields_ = [("size", C.c_int32), ("num_rocks", C.c_int32), ("start_x", C.c_int32), ("start_y", C.c_int32),
                ("rock_x", C.c_int8 * 16), ("rock_y", C.c_int8 * 16), ("grid", C.c_int8 * 256),
                ("thr", C.c_uint64 * 32), ("eff", C.c_double * 32),
                ("stochastic", C.c_int32), ("act_gt", C.c_int32), ("act_thr", C.c_uint64)]


class TagParams(C.Structure):
    _fields_ = [("num_opponents", C.c_int32), ("obs_cells", C.c_int32), ("move_thr", C.c_uint64), ("move_gt", C.c_int32),
                ("reserved", C.c_int32)]


class BattleShipParams(C.Structure):
    _fields_ = [("x_size", C.c_int32), ("y_size", C.c_int32), ("max_len", C.c_int32), ("reserved", C.c_int32),
                ("col0", C.c_uint32 * 4), ("vpat", (C.c_uint32 * 4) * 12)]


class TigerParams(C.Structure):
    _fields_ = [("listen_thr", C.c_uint64)]


class NetworkParams(C.Structure):
    _fields_ = [("n_machines", C.c_int32), ("deg_gt2_mask", C.c_uint32), ("nb_mask", C.c_uint32 * 32),
                ("fail_thr", C.c_uint64), ("fail_nb_thr", C.c_uint64), ("obs_thr", C.c_uint64)]


class StepArgs(C.Structure):        # pomdp_step_args
    _fields_ = [("env", C.c_int32), ("flags", C.c_int32), ("params", C.c_void_p), ("state", C.c_void_p), ("ob", C.c_void_p),
                ("reward", C.c_void_p), ("done", C.c_void_p), ("err", C.c_void_p), ("n", C.c_int64), ("seed", C.c_uint64),
                ("lane0", C.c_uint32), ("reserved", C.c_uint32)]


class CollectArgs(C.Structure):     # pomdp_collect_args
    _fields_ = [("env", C.c_int32), ("flags", C.c_int32), ("params", C.c_void_p), ("state", C.c_void_p), ("action", C.c_void_p),
                ("ob", C.c_void_p), ("reward", C.c_void_p), ("done", C.c_void_p), ("err", C.c_void_p), ("n", C.c_int64),
                ("pitch", C.c_int64), ("seed", C.c_uint64), ("lane0", C.c_uint32), ("reserved", C.c_uint32)]


class TrajArgs(C.Structure):        # pomdp_traj_args
    _fields_ = [("env", C.c_int32), ("flags", C.c_int32), ("layout", C.c_int32), ("reserved", C.c_int32), ("params", C.c_void_p),
                ("state", C.c_void_p), ("traj", C.c_void_p), ("err", C.c_void_p), ("n", C.c_int64), ("pitch", C.c_int64),
                ("seed", C.c_uint64), ("lane0", C.c_uint32), ("reserved2", C.c_uint32)]


class RockBelief(C.Structure):      # pomdp_rock_belief: device pointers, [num_rocks][n]
    _fields_ = [(k, C.c_void_p) for k in ("count", "measured", "lkv", "lkw", "prob_valuable", "check_ok")]


class HistoryPtrs(C.Structure):     # pomdp_history: device pointers + the window size of a bounded history
    _fields_ = [(k, C.c_void_p) for k in ("size", "last_action", "last_ob", "total_sample", "total_move", "move_ok", "ring", "head")] + \
               [("max_size", C.c_int32), ("reserved", C.c_int32)]


class ReturnStats(C.Structure):     # pomdp_return_stats
    _fields_ = [("discount", C.c_double), ("acc", C.c_void_p), ("cnt", C.c_void_p), ("pitch", C.c_int64)]


class Tape(C.Structure):            # pomdp_tape
    _fields_ = [("actions", C.c_void_p), ("stride", C.c_int64)]


class PlanOut(C.Structure):         # pomdp_plan_out
    _fields_ = [("q", C.c_void_p), ("visits", C.c_void_p), ("best", C.c_void_p), ("value", C.c_void_p), ("stride", C.c_int32),
                ("reserved", C.c_int32)]


class Returns(C.Structure):         # pomdp_returns
    _fields_ = [("discount", C.c_double), ("ret", C.c_void_p), ("disc", C.c_void_p), ("ret_done", C.c_void_p)]


def hipcc_path():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC"]


def build(force=False, verbose=False, defines=(), out=None, jobs=None):
    """Compile the HIP library for gfx950 in-tree (hipcc cross-compiles without a GPU): every translation unit of
    csrc/ to its own object, in parallel, then one link.  `defines` (-DNAME=VALUE strings) and `out` build same-box A/B
    variants (tools/ab_build.sh)."""
    from concurrent.futures import ThreadPoolExecutor
    lib_path = out or INTREE_LIB_PATH
    deps = SOURCES + [HEADER]
    if (not force and not defines and os.path.exists(lib_path)
            and all(os.path.getmtime(lib_path) >= os.path.getmtime(d) for d in deps)):
        return lib_path
    os.makedirs(os.path.dirname(lib_path), exist_ok=True)
    obj_dir = os.path.join(os.path.dirname(lib_path), "obj_" + os.path.splitext(os.path.basename(lib_path))[0])
    os.makedirs(obj_dir, exist_ok=True)
    hipcc = hipcc_path()

    def compile_one(unit):
        obj = os.path.join(obj_dir, os.path.splitext(unit)[0] + ".o")
        cmd = [hipcc] + HIPCC_FLAGS + list(defines) + ["-c", "-o", obj, os.path.join(_PKG, "csrc", unit)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=jobs or min(len(UNITS), os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, UNITS))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return lib_path


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "gym_pomdp_amd: HIP library %s is missing — build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH)
    if LIB_PATH != INTREE_LIB_PATH:
        import sys
        print("gym_pomdp_amd: loading the library variant GYM_POMDP_AMD_LIB=%s" % LIB_PATH, file=sys.stderr)
    L = C.CDLL(LIB_PATH)
    missing = [s for s in SYMBOLS if not hasattr(L, s)]
    if missing:
        raise RuntimeError("gym_pomdp_amd: %s does not export %s" % (LIB_PATH, missing))
    L.pomdp_abi_version.restype = C.c_int
    if L.pomdp_abi_version() != ABI_VERSION:
        raise RuntimeError("gym_pomdp_amd: ABI version mismatch (%d != %d); rebuild the HIP library"
                           % (L.pomdp_abi_version(), ABI_VERSION))
    L.pomdp_error_string.restype = C.c_char_p
    L.pomdp_error_string.argtypes = [C.c_int]
    L.pomdp_last_fused_kernel.restype = C.c_char_p
    L.pomdp_last_fused_kernel.argtypes = []
    vp, i64, u64, u32, ci = C.c_void_p, C.c_int64, C.c_uint64, C.c_uint32, C.c_int
    for env in ("rock", "tag", "battleship", "tiger", "network"):
        r = getattr(L, "pomdp_%s_reset" % env)
        r.restype = ci
        r.argtypes = [vp, vp, vp, i64, u64, u32, u64, vp]
        s = getattr(L, "pomdp_%s_step" % env)
        s.restype = ci
        s.argtypes = [vp, vp, vp, vp, vp, vp, vp, i64, u64, u32, u64, ci, vp]
    L.pomdp_step.restype = ci
    L.pomdp_step.argtypes = [vp, vp, u64, vp]
    L.pomdp_stream_sync.restype = ci
    L.pomdp_stream_sync.argtypes = [vp]
    L.pomdp_step_sync.restype = ci
    L.pomdp_step_sync.argtypes = [vp, vp, u64, vp]
    L.pomdp_reset_sync.restype = ci
    L.pomdp_reset_sync.argtypes = [ci, vp, vp, vp, i64, u64, u32, u64, vp]
    L.pomdp_synthetic_actions.restype = ci
    L.pomdp_synthetic_actions.argtypes = [vp, i64, u64, u32, u64, u32, vp]
    L.pomdp_rollout_synthetic.restype = ci
    L.pomdp_rollout_synthetic.argtypes = [ci, vp, vp, vp, vp, vp, vp, vp, i64, u64, u64, u32, u64, i64, ci, vp]
    L.pomdp_collect_synthetic.restype = ci
    L.pomdp_collect_synthetic.argtypes = [ci, vp, vp, vp, vp, vp, vp, vp, i64, u64, u32, u64, i64, i64, ci, vp]
    L.pomdp_collect.restype = ci
    L.pomdp_collect.argtypes = [vp, u64, i64, vp]
    L.pomdp_collect_layout.restype = ci
    L.pomdp_collect_layout.argtypes = [ci, vp, vp, vp, vp, i64, u64, u32, u64, i64, i64, ci, ci, vp]
    L.pomdp_collect_traj.restype = ci
    L.pomdp_collect_traj.argtypes = [vp, u64, i64, vp]
    L.pomdp_packed_reward.restype = C.c_double
    L.pomdp_packed_reward.argtypes = [ci, u32]
    L.pomdp_decode_packed.restype = ci
    L.pomdp_decode_packed.argtypes = [ci, vp, i64, i64, i64, vp, vp, vp, vp, i64, vp]
    L.pomdp_collect_returns.restype = ci
    L.pomdp_collect_returns.argtypes = [ci, vp, vp, vp, vp, i64, u64, u32, u64, i64, ci, vp]
    L.pomdp_collect_tape.restype = ci
    L.pomdp_collect_tape.argtypes = [ci, vp, vp, vp, vp, vp, vp, vp, i64, u64, u32, u64, i64, i64, ci, vp]
    L.pomdp_collect_tape_layout.restype = ci
    L.pomdp_collect_tape_layout.argtypes = [ci, vp, vp, vp, vp, vp, i64, u64, u32, u64, i64, i64, ci, ci, vp]
    L.pomdp_collect_tape_returns.restype = ci
    L.pomdp_collect_tape_returns.argtypes = [ci, vp, vp, vp, vp, vp, i64, u64, u32, u64, i64, ci, vp]
    L.pomdp_fuse_max.restype = ci
    L.pomdp_fuse_max.argtypes = [ci]
    L.pomdp_fuse_steps.restype = ci
    L.pomdp_fuse_steps.argtypes = [ci, ci]
    L.pomdp_legal_actions.restype = ci
    L.pomdp_legal_actions.argtypes = [ci, vp, vp, vp, vp, i64, ci, vp]
    L.pomdp_compute_prob.restype = ci
    L.pomdp_compute_prob.argtypes = [ci, vp, vp, vp, vp, vp, i64, vp]
    L.pomdp_rollout.restype = ci
    L.pomdp_rollout.argtypes = [ci, vp, vp, i64, i64, ci, C.c_double, ci, u64, u32, u64, vp, vp, vp, vp, vp, vp]
    L.pomdp_plan.restype = ci
    L.pomdp_plan.argtypes = [ci, vp, vp, i64, i64, ci, C.c_double, ci, u64, u32, u64, vp, vp, vp, vp]
    L.pomdp_plan_reduce.restype = ci
    L.pomdp_plan_reduce.argtypes = [vp, vp, i64, i64, ci, vp, vp]
    L.pomdp_rock_belief_reset.restype = ci
    L.pomdp_rock_belief_reset.argtypes = [vp, vp, vp, i64, vp]
    L.pomdp_rock_belief_refresh.restype = ci
    L.pomdp_rock_belief_refresh.argtypes = [vp, vp, i64, vp]
    L.pomdp_rock_belief_update.restype = ci
    L.pomdp_rock_belief_update.argtypes = [vp, vp, vp, vp, vp, vp, i64, ci, vp]
    L.pomdp_rock_select_target.restype = ci
    L.pomdp_rock_select_target.argtypes = [vp, vp, vp, vp, i64, vp]
    L.pomdp_history_clear.restype = ci
    L.pomdp_history_clear.argtypes = [ci, vp, vp, vp, i64, vp]
    L.pomdp_history_append.restype = ci
    L.pomdp_history_append.argtypes = [ci, vp, vp, vp, vp, vp, vp, i64, ci, vp]
    L.pomdp_preferred_actions.restype = ci
    L.pomdp_preferred_actions.argtypes = [ci, vp, vp, vp, vp, vp, vp, i64, ci, vp]
    L.pomdp_pick_actions.restype = ci
    L.pomdp_pick_actions.argtypes = [vp, vp, ci, vp, i64, u64, u32, u64, vp]
    L.pomdp_heuristic_steps.restype = ci
    L.pomdp_heuristic_steps.argtypes = [ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, u64, u32, u64, i64, ci, vp]
    L.pomdp_philox_blocks.restype = ci
    L.pomdp_philox_blocks.argtypes = [vp, vp, i64, vp]
    _lib = L
    return L


def check(code, what):
    if code != 0:
        raise RuntimeError("gym_pomdp_amd: %s failed: %s (code %d)"
                           % (what, lib().pomdp_error_string(code).decode(), code))
