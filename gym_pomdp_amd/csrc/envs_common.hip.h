// envs_common.hip.h — what the five env headers share (envs.hip.h includes them all).
// The envs are per-lane transition / observation / reward functions of the five envs,
// written against the packed int32 lane state documented in include/pomdp_hip.h.
// Each Env type plugs into the generic kernels in step_impl.hip.h, fused_impl.hip.h and planner.hip:
//
//   Params   plain-C params struct (kernarg, wave-uniform)
//   Shared   lookup tables staged into LDS once per workgroup
//   State    the lane's state words, in registers
//   Reward   int32_t or float
//
// Reference semantics are cited per function (paths relative to gym_pomdp/envs/);
// the quirks catalogued in SURVEY.md §9 are reproduced on purpose.
#pragma once
#include "../../include/pomdp_hip.h"
#include "philox.hip.h"
#include <type_traits>

namespace pomdp {

constexpr uint64_t TWO52 = 4503599627370496ull;

// The synthetic policy's action of global lane `lane` at the call counter in `akey` (stream ACTION,
// one Philox block per 4 consecutive lanes): what pomdp_synthetic_actions writes for that lane.
__device__ __forceinline__ int synthetic_action(const RngKey &akey, uint32_t lane, uint32_t n_actions)
{
    const uint4 w = philox4x32_10(lane >> 2, akey.t_lo, akey.t_hi, (uint32_t)POMDP_STREAM_ACTION << 24, akey.k0, akey.k1);
    const uint32_t sel = lane & 3u;
    return (int)__umulhi(sel == 0 ? w.x : sel == 1 ? w.y : sel == 2 ? w.z : w.w, n_actions);
}

// default for envs without a cooperative reset: reset, then every lane derives its own next action
template <class Env>
__device__ __forceinline__ void reset_where_chain_default(const typename Env::Shared &sh, const typename Env::Params &p,
                                                          typename Env::State &st, bool fresh, const RngKey &key,
                                                          uint32_t lane, const RngKey &akey, uint32_t n_actions,
                                                          int &next_action)
{
    Env::reset_where(sh, p, st, fresh, key, lane);
    next_action = synthetic_action(akey, lane, n_actions);
}

// index of the n-th (0-based) set bit of m, branch-free: a binary search on popcounts (n < popc(m)).  MINW = 2: m has
// bits at even positions only (RockSample's spread rock sets), so the search stops at windows of two.
template <int MINW = 1>
__device__ __forceinline__ int nth_set_bit(uint32_t m, int n)
{
    int pos = 0;
#pragma unroll
    for (int w = 16; w >= MINW; w >>= 1) {
        const int c = __popc(m & ((1u << w) - 1u));
        const bool up = n >= c;
        n -= up ? c : 0;
        pos += up ? w : 0;
        m = up ? (m >> w) : m;
    }
    return pos;
}

} // namespace pomdp
