// envs/network.hip.h — Network (gym_pomdp/envs/network.py): the lane functions the generic kernels of step_impl.hip.h / fused_impl.hip.h / planner.hip call.
// Included by envs.hip.h (which holds the Env interface description and the shared helpers).
#pragma once
#include "../envs_common.hip.h"

namespace pomdp {

struct NetworkEnv {
    using Params = pomdp_network_params;
    using Reward = float;
    static constexpr int WORDS = 1;
    static constexpr const char *NAME = "NetworkEnv";
    static constexpr bool POOLED_LPT2 = false;
    static constexpr bool QUAD_STEP = false;   // step_impl.hip.h: step_quad_kernel
    static constexpr bool HAS_ROCKS = false;   // a bounded History keeps a window of transitions for this env (history_push)
    static constexpr bool POOLED_ANY_LPT = false;
    static constexpr bool QUAD_SENSOR = false;
    static constexpr bool NEVER_DONE = true;   // network.py:113 — the Returns sink banks nothing (traj_out.hip.h)
    // nbf[k][v]: the machines that see a failed neighbour when the down machines among 4 k .. 4 k + 3 are the set v
    // (network.py:82-85) — the OR over the nibbles of ~state replaces a loop over the machines
    struct Shared { uint32_t nbf[8][16]; };
    struct State { uint32_t w; };

    static __device__ __forceinline__ void stage(Shared &sh, const Params &p, int tid)
    {
        if (tid >= 128) return;
        const int k = tid >> 4, v = tid & 15;
        uint32_t m = 0;
        for (int i = 0; i < p.n_machines; ++i) m |= (((p.nb_mask[i] >> (4 * k)) & (uint32_t)v) != 0u ? 1u : 0u) << i;
        sh.nbf[k][v] = m;
    }
    static __device__ __forceinline__ int n_actions(const Params &p) { return 2 * p.n_machines + 1; }
    // The reward byte of a Packed trajectory record (traj_out.hip.h).  Network's reward is float32(base - cost) with base =
    // 2 per up machine with more than two neighbours + 1 per other up machine (<= 64) and cost 0 (no action), .1 (ping) or 2.5
    // (reboot): network.py:87-92, 103, 110.  code = kind * 68 + base, kind 0 / 1 / 2 in that order; pomdp_packed_reward() and
    // the host's table turn it back into the same float.  From the float itself: the fractional part names the kind.
    static constexpr int REWARD_BASES = 68;
    static __device__ __forceinline__ uint32_t reward_code(int kind, int base) { return (uint32_t)(kind * REWARD_BASES + base); }
    static __device__ __forceinline__ uint32_t reward_code(Reward r)
    {
        const float f = floorf(r), frac = r - f;                                // 0 | .9 (base - .1) | .5 (base - 2.5)
        const int kind = frac == 0.f ? 0 : (frac == .5f ? 2 : 1);
        return reward_code(kind, (int)f + (kind == 0 ? 0 : kind == 1 ? 1 : 3));
    }
    // ... and back, as the float64 the reference computes — `reward -= .1` / `reward -= 2.5` on the integer base
    // (network.py:87-92, 103, 110) — which is what its callers add up (traj_out.hip.h: the Returns sink); the reward
    // COLUMN holds the float32 of this value
    static __device__ __forceinline__ double code_reward(uint32_t code)
    {
        const int kind = (int)code / REWARD_BASES;
        double r = (double)((int)code % REWARD_BASES);
        if (kind == 1) r -= .1;
        if (kind == 2) r -= 2.5;
        return r;
    }
    static __device__ __forceinline__ void load(State &st, const uint32_t *state, int64_t, uint32_t i) { st.w = ld_stream(state + i); }
    static __device__ __forceinline__ void store(const State &st, uint32_t *state, int64_t, uint32_t i, bool) { st_stream(state + i, st.w); }

    // network.py:61-69: all machines up, ob = OFF (0)
    static __device__ __forceinline__ int reset(const Shared &, const Params &p, State &st, const RngKey &, uint32_t)
    {
        st.w = p.n_machines >= 32 ? 0xFFFFFFFFu : ((1u << p.n_machines) - 1u);
        return 0;
    }
    static __device__ __forceinline__ int reset_ob(const Params &, const State &) { return 0; }

    // Called convergently by every lane of the wave; `fresh` marks the lanes that start a new episode.
    static __device__ __forceinline__ void reset_where(const Shared &sh, const Params &p, State &st, bool fresh,
                                                       const RngKey &key, uint32_t lane)
    {
        if (fresh) reset(sh, p, st, key, lane);
    }
    static __device__ __forceinline__ void reset_where_chain(const Shared &sh, const Params &p, State &st, bool fresh,
                                                             const RngKey &key, uint32_t lane, const RngKey &akey,
                                                             uint32_t n_actions, int &next_action)
    {
        reset_where_chain_default<NetworkEnv>(sh, p, st, fresh, key, lane, akey, n_actions, next_action);
    }
    // network.py:130-131: every action is legal
    static __device__ __forceinline__ int legal_count(const Shared &, const Params &p, const State &) { return n_actions(p); }
    static __device__ __forceinline__ int legal_nth(const Shared &, const Params &, const State &, int idx) { return idx; }

    // no heuristic: _generate_preferred is _generate_legal (0 = fall back to the legal list)
    static __device__ __forceinline__ uint32_t preferred_mask(const Shared &, const Params &, const State &,
                                                              const pomdp_rock_belief &, const pomdp_history &, int64_t,
                                                              uint32_t) { return 0u; }
    static __device__ __forceinline__ uint32_t preferred_mask(const Shared &, const Params &, const State &,
                                                              const pomdp_history &, int64_t, uint32_t, uint32_t, uint32_t,
                                                              int) { return 0u; }
    static __device__ __forceinline__ uint32_t preferred_mask(const Shared &, const Params &, const State &,
                                                              const pomdp_history &, int64_t, uint32_t, uint32_t, uint32_t,
                                                              int, int, int) { return 0u; }
    // network.py:43-55 _compute_prob
    static __device__ __forceinline__ double compute_prob(const Shared &, const Params &p, const State &st, int a, int ob)
    {
        if (a < 2 * p.n_machines) return ((int)((st.w >> (a >> 1)) & 1u) == ob) ? .95 : 1 - .95;
        return ob == 2 ? 1.0 : 0.0;
    }

    // ---- pieces of the step for the quad-per-thread fused loop (fused_impl.hip.h: network_steps_quad_kernel) ----------
    // Thresholds against the draw's HIGH word itself: k53 <= thr is decided by (H >> 5) < (thr >> 26), i.e. H < T with
    // T = (thr >> 26) << 5, unless H lies in [T, T + 32) — the tie (probability 2^-27) that asks for the low word.
    struct Thr { uint32_t fail, nb, obs; };
    static __device__ __forceinline__ Thr thresholds(const Params &p)
    {
        return Thr{(uint32_t)(p.fail_thr >> 26) << 5, (uint32_t)(p.fail_nb_thr >> 26) << 5, (uint32_t)(p.obs_thr >> 26) << 5};
    }
    // machines that see a failed neighbour (network.py:82-85), from the nibble tables
    static __device__ __forceinline__ uint32_t nb_failed_of(const Shared &sh, const Params &p, uint32_t s0)
    {
        const int M = p.n_machines;
        const uint32_t down = ~s0 & (M >= 32 ? 0xFFFFFFFFu : ((1u << M) - 1u));
        uint32_t nbf = 0;
        for (int k = 0; 4 * k < M; ++k) nbf |= sh.nbf[k][(down >> (4 * k)) & 15u];          // wave-uniform trip count
        return nbf;
    }
    // N consecutive draws of a lane's STEP stream (high words of one Philox block) applied to the next N up machines of
    // `todo` (network.py:94-99, index order): returns the machines that fail, removes the N from `todo`, and folds "some
    // draw was a tie" into `near` (the minimum of H - T over the draws; a tie has H - T < 32 — the caller then takes the
    // exact per-lane form, so a false alarm from a slot without a machine costs time, 2^-27 of the time, and nothing
    // else).  A slot without a machine (todo ran out) kills nothing.
    template <int N>
    static __device__ __forceinline__ uint32_t draws(const uint32_t (&H)[N], uint32_t &todo, uint32_t nbf, const Thr &T, uint32_t &near)
    {
        uint32_t kill = 0;
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const uint32_t lb = todo & (0u - todo);                              // this draw's machine; 0 = none left
            todo ^= lb;
            const uint32_t t = (nbf & lb) ? T.nb : T.fail;
            kill |= H[k] < t ? 0u : lb;                                          // fails iff k53 > thr
            near = min(near, H[k] - t);
        }
        return kill;
    }
    static __device__ __forceinline__ uint32_t draw4(const uint4 &h, uint32_t &todo, uint32_t nbf, const Thr &T, uint32_t &near)
    {
        const uint32_t H[4] = {h.x, h.y, h.z, h.w};
        return draws<4>(H, todo, nbf, T, near);
    }
    // the action's draw (network.py:106-109): word `w` of the same stream; truthful iff k53 <= obs_thr
    static __device__ __forceinline__ bool truthful_of(uint32_t w, const Thr &T, uint32_t &near)
    {
        near = min(near, w - T.obs);
        return w < T.obs;
    }
    // reward, observation and the reboot (network.py:87-92, 101-112) once the machine draws are in: s = state after the
    // failures, base = reward before the action's cost
    template <class RT>
    static __device__ __forceinline__ void finish(const Params &p, uint32_t &s, int a, int base, bool truthful, int &ob, RT &rew)
    {
        double r = (double)base;
        ob = 2;
        if (a < 2 * p.n_machines) {
            const int machine = a >> 1;
            if (a & 1) { r -= 2.5; s |= 1u << machine; ob = truthful; }
            else { r -= .1; const int up = (int)((s >> machine) & 1u); ob = truthful ? up : 1 - up; }
        }
        rew = (RT)r;
    }

    // network.py:71-114.  The reference draws one double per *up* machine in index order, then one for
    // the action.  Lanes iterate over the draws (j = 0, 1, ...), not over the machines: j is
    // wave-uniform, so the Philox block that feeds doubles 2q and 2q+1 is generated under a uniform
    // condition and the only divergence left is the per-lane number of up machines.
    // The step as the single-step kernels, the rollouts and the generic fused loop run it: the FIRST block of the lane's
    // stream is computed unconditionally and its four words applied straight-line (draw4) — under a random policy 98 % of the
    // lanes need no more — then the wave loops, block by block, only while some lane still has machines to draw for or its
    // action's draw ahead.  A draw decided by its low word (2^-27) sends the lane through step_exact.
    template <class RT>
    static __device__ __forceinline__ void step(const Shared &sh, const Params &p, State &st, int a,
                                                const RngKey &key, uint32_t lane, int &ob, RT &rew, int &done)
    {
        const uint32_t s0 = st.w;
        const int n_up = __popc(s0), base = n_up + __popc(s0 & p.deg_gt2_mask);       // network.py:87-92
        const uint32_t nbf = nb_failed_of(sh, p, s0);
        const Thr T = thresholds(p);
        const bool has_action = a < 2 * p.n_machines;
        uint32_t todo = s0, near = 0xFFFFFFFFu, kill = 0;
        bool pend = has_action, truthful = false;
        int left = n_up;
        uint4 h = stream_block(key, lane, POMDP_STREAM_STEP, 0u);
        for (uint32_t blk = 1;; ++blk) {
            kill |= draw4(h, todo, nbf, T, near);
            if (pend && left < 4) {                                              // the action's draw: the word after the last machine's
                uint32_t w = left == 1 ? h.y : h.x;
                w = left == 2 ? h.z : w;
                w = left == 3 ? h.w : w;
                truthful = truthful_of(w, T, near);
                pend = false;
            }
            if (!__any(todo != 0u || pend)) break;                               // wave-uniform
            left = __popc(todo);
            h = stream_block(key, lane, POMDP_STREAM_STEP, 2u * blk);
        }
        if (near < 32u) { step_exact(sh, p, st, a, key, lane, ob, rew, done); return; }
        uint32_t s = s0 & ~kill;
        finish(p, s, a, base, truthful, ob, rew);
        done = 0;
        st.w = s;
    }

    // the same, one draw at a time with the low words at hand: the exact form (ties; and what the fast forms are checked against)
    template <class RT>
    static __device__ __forceinline__ void step_exact(const Shared &sh, const Params &p, State &st, int a,
                                                      const RngKey &key, uint32_t lane, int &ob, RT &rew, int &done)
    {
        const uint32_t s0 = st.w;
        const int M = p.n_machines;
        // reward: 2 per up machine with > 2 neighbours, 1 per other up machine   network.py:87-92
        const int n_up = __popc(s0);
        double r = (double)(n_up + __popc(s0 & p.deg_gt2_mask));
        // machines whose neighbourhood has a failure, from the pre-update state    network.py:82-85
        const uint32_t down = ~s0 & (M >= 32 ? 0xFFFFFFFFu : ((1u << M) - 1u));
        uint32_t nb_failed = 0;
        for (int k = 0; 4 * k < M; ++k) nb_failed |= sh.nbf[k][(down >> (4 * k)) & 15u];      // wave-uniform trip count
        const bool has_action = a < 2 * M;
        const int n_draws = n_up + (has_action ? 1 : 0);
        // Split word layout: double j compares by its high word — element j & 3 of block 2 (j >> 2) —
        // and needs its low word (same element of the next block) only on a tie, probability 2^-27 per draw.  One
        // Philox block therefore serves four draws instead of two.  Thresholds as (high 27 bits, low 26 bits).
        // Draw j belongs to the j-th up machine (lowest set bit of `todo`), draw n_up to the action.
        constexpr uint32_t LO = (1u << 26) - 1u;
        const uint32_t th_fail = (uint32_t)(p.fail_thr >> 26), tl_fail = (uint32_t)p.fail_thr & LO;
        const uint32_t th_nb = (uint32_t)(p.fail_nb_thr >> 26), tl_nb = (uint32_t)p.fail_nb_thr & LO;
        const uint32_t th_obs = (uint32_t)(p.obs_thr >> 26), tl_obs = (uint32_t)p.obs_thr & LO;
        uint32_t todo = s0, s = s0;
        uint4 blk = make_uint4(0, 0, 0, 0);
        bool truthful = false;
        for (int j = 0; __any(j < n_draws); ++j) {                               // j is wave-uniform
            if ((j & 3) == 0) blk = stream_block(key, lane, POMDP_STREAM_STEP, 2u * (uint32_t)(j >> 2));
            const uint32_t H = (j & 3) == 0 ? blk.x : (j & 3) == 1 ? blk.y : (j & 3) == 2 ? blk.z : blk.w;
            const uint32_t lb = todo & (0u - todo);                              // this draw's machine; 0 = none left
            const bool machine_draw = lb != 0u;                                  // network.py:94-99, else the action's draw
            const bool nbf = (nb_failed & lb) != 0u;
            const uint32_t th = machine_draw ? (nbf ? th_nb : th_fail) : th_obs;
            const uint32_t kh = H >> 5;
            bool le = kh < th;                                                   // k53 <= thr, decided by the high word
            if (kh == th && j < n_draws) {                                       // tie: fetch the low word
                const uint32_t tl = machine_draw ? (nbf ? tl_nb : tl_fail) : tl_obs;
                const uint4 lo = stream_block(key, lane, POMDP_STREAM_STEP, 2u * (uint32_t)(j >> 2) + 1u);
                const uint32_t L = (j & 3) == 0 ? lo.x : (j & 3) == 1 ? lo.y : (j & 3) == 2 ? lo.z : lo.w;
                le = (L >> 6) <= tl;
            }
            s &= le ? 0xFFFFFFFFu : ~lb;                                         // fails iff k > thr (lb == 0: nothing)
            truthful = (j == n_up) ? le : truthful;                              // only read when the action draws
            todo ^= lb;
        }
        ob = 2;
        if (has_action) {                                                        // network.py:101-112
            const int machine = a >> 1;
            if (a & 1) { r -= 2.5; s |= 1u << machine; ob = truthful; }
            else { r -= .1; const int up = (int)((s >> machine) & 1u); ob = truthful ? up : 1 - up; }
        }
        rew = (RT)r;    // float32(float64 value) for the step kernel, the float64 itself for rollouts
        done = 0;
        st.w = s;
    }
};

} // namespace pomdp
