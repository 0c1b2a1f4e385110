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

    // ---- the step's random words (ABI 12) ------------------------------------------------------------------------
    // step() draws one double per UP machine, in index order, then one for the action (network.py:94-109), each compared with
    // a threshold.  16 random bits decide such a comparison unless they EQUAL the threshold's top 16 bits (2^-16 per draw), so
    // the top 16 bits of double j are a half — upper for even j, lower for odd j — of the lane's element of a block shared
    // by the four lanes of a quad (stream STEP, counter word 0 = lane >> 2, block j >> 1): one Philox block serves two draws
    // of each of four lanes.  The 37 bits below them come from the lane's own stream STEP_LO (block j >> 1, elements 2 (j & 1)
    // and 2 (j & 1) + 1) and are generated on a tie only.
    static constexpr uint32_t STREAM_STEP_LO = POMDP_STREAM_STEP_LO;
    static constexpr uint32_t TIE = 1u << 16;               // H - T < TIE: the 16 bits equal the threshold's
    static __device__ __forceinline__ uint4 quad_block(const RngKey &key, uint32_t lane, uint32_t b)
    {
        return stream_block(key, lane >> 2, POMDP_STREAM_STEP, b);
    }
    static __device__ __forceinline__ uint32_t elem(const uint4 &w, uint32_t e) { return e == 0 ? w.x : e == 1 ? w.y : e == 2 ? w.z : w.w; }
    // Thresholds against a draw's 16 bits held in the TOP half of a 32-bit word H (whatever lies below): k53 <= thr is
    // decided by H < T with T = (thr >> 37) << 16, unless H - T < TIE.
    struct Thr { uint32_t fail, nb, obs; };
    static __device__ __forceinline__ Thr thresholds(const Params &p)
    {
        return Thr{(uint32_t)(p.fail_thr >> 37) << 16, (uint32_t)(p.fail_nb_thr >> 37) << 16, (uint32_t)(p.obs_thr >> 37) << 16};
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
    // The two draws of one word W of the lane (upper half: the earlier draw) applied to the next two up machines of `todo`
    // (network.py:94-99, index order): returns the machines that fail, removes the two from `todo`, and folds "some draw
    // was a tie" into `near` (the minimum of H - T over the draws; a tie has H - T < TIE — the caller then takes the exact
    // per-lane form, so a false alarm from a slot without a machine costs time, 2^-16 of the time, and nothing else).
    static __device__ __forceinline__ uint32_t draw2(uint32_t W, uint32_t &todo, uint32_t nbf, const Thr &T, uint32_t &near)
    {
        uint32_t kill = 0;
        const uint32_t H[2] = {W, W << 16};
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const uint32_t lb = todo & (0u - todo);                              // this draw's machine; 0 = none left
            todo ^= lb;
            const uint32_t t = (nbf & lb) ? T.nb : T.fail;
            kill |= H[k] < t ? 0u : lb;                                          // fails iff k53 > thr
            near = min(near, H[k] - t);
        }
        return kill;
    }
    // the action's draw (network.py:106-109) when it is draw `slot` (0 or 1) of word W; truthful iff k53 <= obs_thr
    static __device__ __forceinline__ bool truthful_of(uint32_t W, int slot, const Thr &T, uint32_t &near)
    {
        const uint32_t H = slot ? W << 16 : W;
        near = min(near, H - T.obs);
        return H < T.obs;
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

    // network.py:71-114 for one lane, as the single-step kernels, the rollouts, the heuristic loop and the generic fused
    // loop run it: the lane takes its words from its quad's blocks, two draws per block, and the wave loops block by block
    // while some lane still has machines to draw for or its action's draw ahead (j is wave-uniform; under a random policy a
    // lane has 1.4 machines up, so two blocks serve 98 % of the lanes).  A tie (2^-16 per draw) sends the lane through
    // step_exact.
    static __device__ __forceinline__ uint32_t machines_mask(const Params &p)
    {
        return p.n_machines >= 32 ? 0xFFFFFFFFu : ((1u << p.n_machines) - 1u);
    }
    template <class RT>
    static __device__ __forceinline__ void step(const Shared &sh, const Params &p, State &st, int a,
                                                const RngKey &key, uint32_t lane, int &ob, RT &rew, int &done)
    {
        step_from<0>(sh, p, st, a, key, lane, 0u, 0u, 0u, ob, rew, done);
    }
    // The same with the lane's words of the quad's first QUAD_WORDS blocks handed in: the one-lane-per-thread fused loop lets
    // lane e of a quad compute the blocks of step s + e once per four steps and passes the words round by DPP transposes
    // (steps_kernel) — three blocks per lane per four steps instead of per step.
    static constexpr int QUAD_WORDS = 3;
    template <class RT>
    static __device__ __forceinline__ void step_words(const Shared &sh, const Params &p, State &st, int a, const RngKey &key,
                                                      uint32_t lane, uint32_t W0, uint32_t W1, uint32_t W2, int &ob, RT &rew, int &done)
    {
        step_from<QUAD_WORDS>(sh, p, st, a, key, lane, W0, W1, W2, ob, rew, done);
    }
    template <int GIVEN, class RT>
    static __device__ __forceinline__ void step_from(const Shared &sh, const Params &p, State &st, int a, const RngKey &key,
                                                     uint32_t lane, uint32_t W0, uint32_t W1, uint32_t W2, int &ob, RT &rew, int &done)
    {
        const uint32_t s0 = st.w & machines_mask(p);                             // bits at or above n_machines are not machines
        const int n_up = __popc(s0), base = n_up + __popc(s0 & p.deg_gt2_mask);       // network.py:87-92
        const uint32_t nbf = nb_failed_of(sh, p, s0);
        const Thr T = thresholds(p);
        const bool has_action = a < 2 * p.n_machines;
        uint32_t todo = s0, near = 0xFFFFFFFFu, kill = 0;
        bool pend = has_action, truthful = false;
        auto word = [&](uint32_t W) {
            const int left = __popc(todo);
            kill |= draw2(W, todo, nbf, T, near);
            if (pend && left < 2) { truthful = truthful_of(W, left, T, near); pend = false; }   // the draw after the last machine's
        };
        if constexpr (GIVEN == 3) { word(W0); word(W1); word(W2); }
        for (uint32_t b = (uint32_t)GIVEN; __any(todo != 0u || pend); ++b)      // wave-uniform
            word(elem(quad_block(key, lane, b), lane & 3u));
        if (near < TIE) { step_exact(sh, p, st, a, key, lane, ob, rew, done); return; }
        uint32_t s = s0 & ~kill;
        finish(p, s, a, base, truthful, ob, rew);
        done = 0;
        st.w = s;
    }

    // the same, one draw at a time with all 53 bits at hand: the exact form (ties; and what the fast forms are checked against)
    template <class RT>
    static __device__ __forceinline__ void step_exact(const Shared &sh, const Params &p, State &st, int a,
                                                      const RngKey &key, uint32_t lane, int &ob, RT &rew, int &done)
    {
        const uint32_t s0 = st.w & machines_mask(p);                             // every launch shape reads the same machines (a caller's
                                                                                 // stray upper bits never become draws)
        const int M = p.n_machines;
        // reward: 2 per up machine with > 2 neighbours, 1 per other up machine   network.py:87-92
        const int n_up = __popc(s0);
        double r = (double)(n_up + __popc(s0 & p.deg_gt2_mask));
        // machines whose neighbourhood has a failure, from the pre-update state    network.py:82-85
        const uint32_t nb_failed = nb_failed_of(sh, p, s0);
        const bool has_action = a < 2 * M;
        const int n_draws = n_up + (has_action ? 1 : 0);
        // Draw j belongs to the j-th up machine (lowest set bit of `todo`), draw n_up to the action.  Its double, as numpy
        // builds it from two words: a = Q << 16 | X >> 16, b = Y, k53 = (a >> 5) << 26 | b >> 6 (network_step_words).
        uint32_t todo = s0, s = s0;
        uint4 qb = make_uint4(0, 0, 0, 0);
        bool truthful = false;
        for (int j = 0; __any(j < n_draws); ++j) {                               // j is wave-uniform
            if ((j & 1) == 0) qb = quad_block(key, lane, (uint32_t)(j >> 1));
            const uint32_t W = elem(qb, lane & 3u), Q = (j & 1) ? (W & 0xFFFFu) : (W >> 16);
            const uint32_t lb = todo & (0u - todo);                              // this draw's machine; 0 = none left
            const bool machine_draw = lb != 0u;                                  // network.py:94-99, else the action's draw
            const uint64_t thr = machine_draw ? ((nb_failed & lb) ? p.fail_nb_thr : p.fail_thr) : p.obs_thr;
            const uint32_t t16 = (uint32_t)(thr >> 37);
            bool le = Q < t16;                                                   // k53 <= thr, decided by the top 16 bits
            if (Q == t16 && j < n_draws) {                                       // tie: the lane's own low words
                const uint4 lo = stream_block(key, lane, STREAM_STEP_LO, (uint32_t)(j >> 1));
                const uint32_t X = (j & 1) ? lo.z : lo.x, Y = (j & 1) ? lo.w : lo.y;
                le = k53((Q << 16) | (X >> 16), Y) <= thr;
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
