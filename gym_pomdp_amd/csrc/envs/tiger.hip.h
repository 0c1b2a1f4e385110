// envs/tiger.hip.h — Tiger (gym_pomdp/envs/tiger.py): the lane functions the generic kernels of step_impl.hip.h / fused_impl.hip.h / planner.hip call.
// Included by envs.hip.h (which holds the Env interface description and the shared helpers).
#pragma once
#include "../envs_common.hip.h"

namespace pomdp {

struct TigerEnv {
    using Params = pomdp_tiger_params;
    using Reward = int32_t;
    static constexpr int WORDS = 1;
    static constexpr const char *NAME = "TigerEnv";
    static constexpr bool POOLED_LPT2 = false;
    static constexpr bool QUAD_STEP = false;   // step_impl.hip.h: step_quad_kernel
    static constexpr bool HAS_ROCKS = false;   // a bounded History keeps a window of transitions for this env (history_push)
    static constexpr bool POOLED_ANY_LPT = false;
    static constexpr bool QUAD_SENSOR = false;
    static constexpr bool QUAD_FUSED = true;      // fused_impl.hip.h: steps_quad_generic_kernel
    struct Shared { int unused; };
    // w: the tiger's door (what is stored).  rs: registers only — the fresh episode's door when this step ended the
    // episode (see step()), NO_RS otherwise.
    static constexpr uint32_t NO_RS = 0xFFFFFFFFu;
    struct State { uint32_t w; uint32_t rs = NO_RS; };

    static __device__ __forceinline__ void stage(Shared &, const Params &, int) {}
    static __device__ __forceinline__ int n_actions(const Params &) { return 3; }
    // the reward byte of a Packed trajectory record (traj_out.hip.h): the reward itself, an int8
    static __device__ __forceinline__ uint32_t reward_code(Reward r) { return (uint32_t)(int)r; }
    // ... and back, as the float64 the reference's callers add up (traj_out.hip.h: the Returns sink)
    static __device__ __forceinline__ double code_reward(uint32_t code) { return (double)(int8_t)code; }
    static __device__ __forceinline__ void load(State &st, const uint32_t *state, int64_t, uint32_t i) { st.w = ld_stream(state + i); }
    static __device__ __forceinline__ void store(const State &st, uint32_t *state, int64_t, uint32_t i, bool) { st_stream(state + i, st.w); }

    // Word contract (ABI 13, include/pomdp_hip.h): whatever a call with counter t draws — LISTEN's uniform(), the door a wrong
    // guess resamples, the door of the episode that reset() or the auto-reset after a right guess starts — it reads from the
    // QUAD's STEP stream (counter word 0 = lane >> 2, lane L element L & 3): the double's high word / the door's word from
    // block 0, the double's low word (a tie of the top 27 bits: 2^-27) from block 1.  One block serves four lanes.
    static constexpr int QUAD_WORD = 1;                                        // fused_impl.hip.h: steps_quad_generic_kernel shares the block
    static __device__ __forceinline__ uint4 quad_block(const RngKey &key, uint32_t lane, uint32_t block)
    {
        return stream_block(key, lane >> 2, POMDP_STREAM_STEP, block);
    }
    static __device__ __forceinline__ uint32_t elem(const uint4 &b, uint32_t e) { return e == 0 ? b.x : e == 1 ? b.y : e == 2 ? b.z : b.w; }
    static __device__ __forceinline__ uint32_t word(const RngKey &key, uint32_t lane) { return elem(quad_block(key, lane, 0u), lane & 3u); }
    // tiger.py:60-66: state = state_space.sample(); ob = NULL
    static __device__ __forceinline__ int reset(const Shared &, const Params &, State &st, const RngKey &key,
                                                uint32_t lane)
    {
        st.w = word(key, lane) & 1u;                                            // randint(2): mask 1, never rejects
        return 2;
    }
    static __device__ __forceinline__ int reset_ob(const Params &, const State &) { return 2; }

    // Called convergently by every lane of the wave; `fresh` marks the lanes that start a new episode.
    static __device__ __forceinline__ void reset_where(const Shared &sh, const Params &p, State &st, bool fresh,
                                                       const RngKey &key, uint32_t lane)
    {
        // a lane is fresh because the step just before (same key, same lane) opened the tiger's door; step() then left the new
        // episode's door — bit 0 of the lane's word of the quad's STEP block, which a terminal step does not otherwise read
        // (tiger.py:81-83 returns before any draw) — in st.rs: no second Philox block
        if (fresh) {
            if (st.rs != NO_RS) st.w = st.rs;
            else reset(sh, p, st, key, lane);
        }
        st.rs = NO_RS;
    }
    static __device__ __forceinline__ void reset_where_chain(const Shared &sh, const Params &p, State &st, bool fresh,
                                                             const RngKey &key, uint32_t lane, const RngKey &akey,
                                                             uint32_t n_actions, int &next_action)
    {
        reset_where_chain_default<TigerEnv>(sh, p, st, fresh, key, lane, akey, n_actions, next_action);
    }
    // tiger.py:111-112: every action is legal
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
    // tiger.py:125-138 _compute_prob
    static __device__ __forceinline__ double compute_prob(const Shared &, const Params &, const State &st, int a, int ob)
    {
        if (a == 2 && ob != 2) return ((int)(st.w & 1u) == ob) ? .85 : 1 - .85;
        if (a != 2 && ob == 2) return 1.0;
        return 0.0;
    }

    // tiger.py:72-88 step, 117-119 _sample_state, 140-149 _sample_ob, 155-172.
    // A step draws from exactly one place that matters: LISTEN one double; the wrong door one word (the state is resampled;
    // the uniform() drawn there affects nothing); the tiger's door nothing — but the auto-reset that follows draws the fresh
    // episode's door at the same call counter, parked in st.rs.  All of them read W, the lane's word of the quad's block.
    // lo(): the double's low word (block 1), asked for on a tie of the top 27 bits only.
    template <class RT, class LowWord>
    static __device__ __forceinline__ void step_word(const Params &p, State &st, int a, uint32_t W, LowWord lo, int &ob, RT &rew,
                                                     int &done)
    {
        const int tiger = (int)(st.w & 1u);
        const bool listen = a == 2, right = !listen && a == tiger;               // right: terminal, ob is the state
        const uint32_t door = W & 1u;                                            // randint(2): mask 1, never rejects
        const uint32_t kh = W >> 5, th = (uint32_t)(p.listen_thr >> 26);         // k53 > thr: the top 27 bits decide, but for a tie
        bool flip = kh > th;                                                     // p > .85
        if (listen && kh == th) flip = (lo() >> 6) > (uint32_t)(p.listen_thr & 0x3FFFFFFu);
        ob = right ? tiger : (listen ? (tiger ^ (int)flip) : 2);
        rew = right ? -20 : (listen ? -1 : 10);
        done = right;
        st.rs = right ? door : NO_RS;
        st.w = (listen | right) ? st.w : door;     // wrong door: state resampled
    }
    template <class RT>
    static __device__ __forceinline__ void step(const Shared &, const Params &p, State &st, int a,
                                                const RngKey &key, uint32_t lane, int &ob, RT &rew, int &done)
    {
        step_word(p, st, a, word(key, lane), [&]() { return elem(quad_block(key, lane, 1u), lane & 3u); }, ob, rew, done);
    }
    // the interface of the loops that time-share the quad's block (one lane per thread: lane e of a quad computes the block of
    // step s + e once per four steps and a 4 x 4 DPP transpose hands every lane its word of every step): the step and the
    // auto-reset given the lane's word W
    template <class RT>
    static __device__ __forceinline__ void step_w(const Shared &, const Params &p, State &st, int a, const RngKey &key, uint32_t lane,
                                                  uint32_t W, int &ob, RT &rew, int &done)
    {
        step_word(p, st, a, W, [&]() { return elem(quad_block(key, lane, 1u), lane & 3u); }, ob, rew, done);
    }
    static __device__ __forceinline__ void fresh_w(const Shared &, const Params &, State &st, bool fresh, const RngKey &, uint32_t, uint32_t W)
    {
        if (fresh) st.w = st.rs != NO_RS ? st.rs : (W & 1u);
        st.rs = NO_RS;
    }
};

} // namespace pomdp
