// envs/rock.hip.h — RockSample / StochasticRock (gym_pomdp/envs/rock.py): the lane functions the generic kernels of step_impl.hip.h / fused_impl.hip.h / planner.hip call.
// Included by envs.hip.h (which holds the Env interface description and the shared helpers).
#pragma once
#include "../envs_common.hip.h"

namespace pomdp {

// Word contract of the RockSample envs ("split layout").  Every draw RockSample makes is a numpy double
// = (high word H, low word L) -> k53 = (H >> 5) * 2^26 + (L >> 6).  H and L live in DIFFERENT Philox blocks, and every
// stream is shared by the four lanes of a quad: counter word 0 = lane >> 2, lane L uses element L & 3 of each block.
//   reset  (stream RESET): rock j of the lane: H = the lane's element of block 0 rotated right by 2 j + 2 bits, L = its
//                          element of block 1 under the same rotation — reset() only uses sign(U - .5), the top bit of H,
//                          i.e. bit 2 j + 1 of the element: ONE 32-bit word carries the statuses of all K <= 16 rocks,
//                          already at the upper bit of each rock's 2-bit code (codes = word & 0xAAAAAAAA);
//   step   (stream STEP):  double j (RockEnv: j = 0 the sensor; StochasticRockEnv: j = 0 the action gate, j = 1 the
//                          sensor): H = block 2 j, L = block 2 j + 1;
//   auto-reset (the reset that follows a done step inside that step's call counter): the same rotated pair, taken from
//                          the step's own SENSOR blocks — stream STEP, blocks b and b + 1, b = SENSOR_BLOCK — instead of
//                          stream RESET.  A step never makes both draws (a CHECK does not end the episode, rock.py:171-175,
//                          193), so the word is consumed exactly once either way.
// A comparison k53 <= thr is decided by H alone unless (H >> 5) == (thr >> 26), which happens with probability 2^-27
// per draw; only then is the L block generated.  So a quad's step — sensor draws AND the fresh episodes of its done lanes —
// costs ONE block besides its policy words: a thread that owns a quad has both thread-local (steps_quad_kernel), a lane of
// a one-lane-per-thread loop computes each once per four steps (quad_transpose4).
//
// STOCH selects StochasticRockEnv (rock.py:428-504).
template <int W, bool STOCH = false> // W = state words per lane: 1 (K <= 12) or 2
struct RockEnv {
    using Params = pomdp_rock_params;
    using Reward = int32_t;
    using S = typename std::conditional<W == 1, uint32_t, uint64_t>::type; // 32-bit ALU when one word is enough
    static constexpr int WORDS = W;
    static constexpr const char *NAME = STOCH ? (W == 1 ? "StochasticRockEnv<1>" : "StochasticRockEnv<2>") : (W == 1 ? "RockEnv<1>" : "RockEnv<2>");
    static constexpr bool POOLED_LPT2 = !STOCH;   // kernels_common.hip.h: Finisher<RockEnv, 2, .>
    static constexpr bool QUAD_STEP = true;   // step_impl.hip.h: step_quad_kernel
    static constexpr bool HAS_ROCKS = true;   // a bounded History keeps a window of transitions for this env (history_push)
    static constexpr bool POOLED_ANY_LPT = !STOCH; // ... and for any other number of lanes per thread >= 2
    static constexpr bool STOCHASTIC = STOCH;
    static constexpr bool QUAD_TAB = true;        // fused_impl.hip.h: steps_quad_kernel (RockEnv and StochasticRockEnv)
    struct Shared {
        uint2 thr[32];         // sensor threshold by L1 distance: .x = thr >> 26 (compared with H >> 5),
                               // .y = thr & (2^26 - 1) (compared with L >> 6 on a tie) — one 8-byte LDS read
        int8_t grid[256];      // rock id stamped at [x * 16 + y], -1 = none
        uint8_t rxy[16];       // rock j position, x | y << 4
        uint16_t rpos[16];     // the same as x | y << 8: one v_sad_u8 against the agent's bytes is the L1 distance
        // heuristic-policy launches only (stage_policy): rock sets by position, bit j = rock j
        uint32_t dir_y[16];    // [y]: rocks with ry > y | rocks with ry < y << 16
        uint32_t dir_x[16];    // [x]: rocks with rx < x | rocks with rx > x << 16
        uint32_t row[16];      // [y]: rocks with ry == y
    };
    struct State { S s; };
    // what the lane step leaves for the deferred sensor draw (pooled launches fetch H from the wave's task pass)
    struct Aux { uint32_t th; uint8_t r; bool good, want; };   // th: sensor threshold (high 27 bits) of a CHECK of rock r

    static constexpr uint32_t SENSOR_BLOCK = STOCH ? 2u : 0u;   // high words of the sensor draw (StochasticRock: block 0 gates the action)
    static constexpr uint32_t LO_MASK = (1u << 26) - 1u;
    static constexpr uint32_t HALF_HI = 1u << 26;            // 2^52 >> 26: the reset's "U > .5" threshold

    // One global-load latency: every thread fetches a slice of the kernarg-resident tables with
    // unconditional (index-wrapped) loads, all issued before the first LDS write, so the compiler
    // emits one s_waitcnt instead of one per predicated region; duplicate writers store equal values.
    // Split in two so that a kernel can put independent work between the table loads and their first use.
    struct Staged { int8_t g, rx, ry; uint64_t t; };
    static __device__ __forceinline__ Staged stage_load(const Params &p, int tid)
    {
        Staged r;
        r.g = p.grid[tid & 255];
        r.t = p.thr[tid & 31];
        r.rx = p.rock_x[tid & 15];
        r.ry = p.rock_y[tid & 15];
        return r;
    }
    static __device__ __forceinline__ void stage_store(Shared &sh, const Staged &r, int tid)
    {
        sh.grid[tid & 255] = r.g;
        sh.thr[tid & 31] = make_uint2((uint32_t)(r.t >> 26), (uint32_t)r.t & LO_MASK);
        sh.rxy[tid & 15] = (uint8_t)((r.rx & 15) | (r.ry << 4));
        sh.rpos[tid & 15] = (uint16_t)((r.rx & 15) | ((r.ry & 15) << 8));
    }
    static __device__ __forceinline__ void stage(Shared &sh, const Params &p, int tid) { stage_store(sh, stage_load(p, tid), tid); }
    static __device__ __forceinline__ int n_actions(const Params &p) { return 5 + p.num_rocks; }
    // the reward byte of a Packed trajectory record (traj_out.hip.h): the reward itself, an int8
    static __device__ __forceinline__ uint32_t reward_code(Reward r) { return (uint32_t)(int)r; }
    // ... and back, as the float64 the reference's callers add up (traj_out.hip.h: the Returns sink)
    static __device__ __forceinline__ double code_reward(uint32_t code) { return (double)(int8_t)code; }
    // the position tables preferred_mask reads; kernels that call it run this next to stage()
    static __device__ __forceinline__ void stage_policy(Shared &sh, const Params &p, int tid)
    {
        if (tid >= 16) return;
        uint32_t above = 0, below = 0, left = 0, right = 0, same = 0;
        for (int j = 0; j < p.num_rocks; ++j) {
            const int rx = p.rock_x[j], ry = p.rock_y[j];
            above |= (uint32_t)(ry > tid) << j;
            below |= (uint32_t)(ry < tid) << j;
            same |= (uint32_t)(ry == tid) << j;
            left |= (uint32_t)(rx < tid) << j;
            right |= (uint32_t)(rx > tid) << j;
        }
        sh.dir_y[tid] = above | (below << 16);
        sh.dir_x[tid] = left | (right << 16);
        sh.row[tid] = same;
    }

    static __device__ __forceinline__ void load(State &st, const uint32_t *state, int64_t n, uint32_t i)
    {
        st.s = ld_stream(state + i);
        if (W == 2) st.s |= (S)((uint64_t)ld_stream(state + n + i) << 32);
    }
    static __device__ __forceinline__ void store(const State &st, uint32_t *state, int64_t n, uint32_t i, bool)
    {
        st_stream(state + i, (uint32_t)st.s);
        if (W == 2) st_stream(state + n + i, (uint32_t)((uint64_t)st.s >> 32));
    }

    static __device__ __forceinline__ uint32_t elem(const uint4 &w, uint32_t e) { return e == 0 ? w.x : e == 1 ? w.y : e == 2 ? w.z : w.w; }

    // ---- split-layout draws ---------------------------------------------------------------------------------
    // k53 <= (th << 26 | tl)?  decided by the high word; `lo()` (the L word) is only evaluated on a tie
    template <class LowWord>
    static __device__ __forceinline__ bool k53_le(uint32_t H, uint32_t th, uint32_t tl, LowWord lo)
    {
        const uint32_t kh = H >> 5;
        bool r = kh < th;
        if (kh == th) r = (lo() >> 6) <= tl;                                  // probability 2^-27
        return r;
    }
    // status + 1 of a fresh rock, sign(k53 - 2^52) + 1, from its high word; 3 = undecided (needs the low word)
    static __device__ __forceinline__ uint32_t rock_code_hi(uint32_t H)
    {
        const uint32_t kh = H >> 5;
        return kh > HALF_HI ? 2u : (kh < HALF_HI ? 0u : 3u);
    }
    static __device__ __forceinline__ uint32_t rock_code_lo(uint32_t L) { return (L >> 6) ? 2u : 1u; }   // kh == 2^26 exactly
    // block `j2` (0 = high words, 1 = low words) of the rotated pair a reset of lane's quad reads: stream RESET for a
    // reset() call of its own; AUTO (the reset inside a done step's call counter): the step's sensor blocks
    template <bool AUTO = false>
    static __device__ __forceinline__ uint4 reset_block(const RngKey &key, uint32_t lane, uint32_t j2)
    {
        const uint32_t c3 = AUTO ? (((uint32_t)POMDP_STREAM_STEP << 24) | (SENSOR_BLOCK + j2)) : (((uint32_t)POMDP_STREAM_RESET << 24) | j2);
        return philox4x32_10(lane >> 2, key.t_lo, key.t_hi, c3, key.k0, key.k1);
    }
    // The 2-bit codes of all K <= 16 rocks (bit pair j = rock j) of lane `lane`'s fresh episode from its RESET word `w`
    // (element lane & 3 of reset_block(key, lane, 0)).  Rock j reads the word rotated right by 2 j + 2: its top bit is bit
    // 2 j + 1 of the word, and the code (status + 1) is twice that bit — the word masked to its odd bits IS the code
    // word — unless the rotated word lies in the 32 values just above 2^31: the tie the low word decides.
    template <bool AUTO>
    static __device__ __forceinline__ uint32_t reset_codes(const RngKey &key, uint32_t lane, int K)
    {
        return reset_codes<AUTO>(elem(reset_block<AUTO>(key, lane, 0u), lane & 3u), key, lane, K);
    }
    template <bool AUTO>
    static __device__ __forceinline__ uint32_t reset_codes(uint32_t w, const RngKey &key, uint32_t lane, int K)
    {
        uint32_t codes = w & 0xAAAAAAAAu;
        // A tied rotation is 1, twenty-six zeros, five free bits: at most six bits set in the word (2.7e-4 per word);
        // only then look closer
        if (__popc(w) <= 6) {
            bool have_lo = false;
            uint32_t l = 0;
            for (int j = 0; j < K; ++j) {
                const uint32_t rot = (uint32_t)(2 * j + 2) & 31u;
                if (rock_code_hi(__builtin_rotateright32(w, rot)) != 3u) continue;
                if (!have_lo) {
                    // The low-word block of an auto-reset IS the block a sensor tie of the same step reads: left as one
                    // expression, the compiler merges the two 2^-27 paths' Philox calls and computes the block
                    // unconditionally, every step (one lane per thread, 2^18 lanes: 0.79 -> 0.97 us per step).  The lane id
                    // passes through an opaque register move, so this call stays where it is needed.
                    uint32_t lane_ = lane;
                    asm volatile("" : "+v"(lane_));
                    l = elem(reset_block<AUTO>(rare_key(key), lane_, 1u), lane & 3u);
                    have_lo = true;
                }
                const uint32_t c = rock_code_lo(__builtin_rotateright32(l, rot));
                codes = (codes & ~(3u << (2 * j))) | (c << (2 * j));
            }
        }
        return codes & (K >= 16 ? 0xFFFFFFFFu : ((1u << (2 * K)) - 1u));      // rocks that exist (wave-uniform)
    }
    // block `j2` (0 = sensor / gate high words, 1 = their low words, 2 / 3 = StochasticRock's sensor) of lane's quad
    static __device__ __forceinline__ uint4 quad_block(const RngKey &key, uint32_t lane, uint32_t j2)
    {
        return philox4x32_10(lane >> 2, key.t_lo, key.t_hi, ((uint32_t)POMDP_STREAM_STEP << 24) | j2, key.k0, key.k1);
    }

    // the same for the four lanes of a quad inside a step (auto-reset) from its sensor block (R[e] = lane first + e's word):
    // ONE branch for the ties
    static __device__ __forceinline__ void reset_codes4(const uint32_t (&R)[4], const RngKey &key, uint32_t first, int K,
                                                        uint32_t (&codes)[4])
    {
        const uint32_t exist = K >= 16 ? 0xFFFFFFFFu : ((1u << (2 * K)) - 1u);
#pragma unroll
        for (int e = 0; e < 4; ++e) codes[e] = R[e] & 0xAAAAAAAAu & exist;
        if (min(min(__popc(R[0]), __popc(R[1])), min(__popc(R[2]), __popc(R[3]))) <= 6) {
#pragma unroll
            for (int e = 0; e < 4; ++e) codes[e] = reset_codes<true>(R[e], key, first + (uint32_t)e, K);
        }
    }
    // rock.py:236-241 reset -> 266-271 _get_init_state -> 78-86 Rock.__init__:
    // status_j = sign(U_j - .5), rocks in index order, one double each.
    static __device__ __forceinline__ int reset(const Shared &, const Params &p, State &st, const RngKey &key,
                                                uint32_t lane)
    {
        const uint64_t s = (uint32_t)p.start_x | ((uint32_t)p.start_y << 4);
        st.s = (S)(s | ((uint64_t)reset_codes<false>(key, lane, p.num_rocks) << 8));
        return 0; // Obs.NULL
    }
    static __device__ __forceinline__ int reset_ob(const Params &, const State &) { return 0; }   // what reset() returned

    // The auto-reset of a step: called convergently by every lane of the wave; `fresh` marks the lanes that start a new
    // episode.  ~1/8 of a wave's lanes reset in a given step under a random policy, so nearly every wave-step computes the
    // block; the quads' four lanes each derive it themselves — it is the block the step's sensor draw came from, and the
    // kernels that own or time-share a quad hand that one over instead (steps_quad_kernel, steps_kernel, fresh_state).
    static __device__ __forceinline__ void reset_where(const Shared &, const Params &p, State &st, bool fresh,
                                                       const RngKey &key, uint32_t lane)
    {
        if (!__any(fresh)) return;                                     // wave-uniform
        const uint32_t codes = reset_codes<true>(key, lane, p.num_rocks);
        if (fresh) st.s = (S)((uint64_t)((uint32_t)p.start_x | ((uint32_t)p.start_y << 4)) | ((uint64_t)codes << 8));
    }
    // ... plus the synthetic policy's action for the NEXT call counter (C-side rollout driver)
    static __device__ __forceinline__ void reset_where_chain(const Shared &sh, const Params &p, State &st, bool fresh,
                                                             const RngKey &key, uint32_t lane, const RngKey &akey,
                                                             uint32_t n_actions, int &next_action)
    {
        reset_where_chain_default<RockEnv>(sh, p, st, fresh, key, lane, akey, n_actions, next_action);
    }
    // the fresh episode's state inside a step from the lane's word of the step's sensor block (auto-reset)
    static __device__ __forceinline__ S fresh_state(const Params &p, uint32_t w, const RngKey &key, uint32_t lane)
    {
        return (S)((uint64_t)((uint32_t)p.start_x | ((uint32_t)p.start_y << 4)) | ((uint64_t)reset_codes<true>(w, key, lane, p.num_rocks) << 8));
    }

    // rock.py:273-291 _generate_legal, in the reference's list order: EAST, then NORTH / SOUTH / WEST when
    // in-grid, SAMPLE on an uncollected rock, then CHECK(grid[rock.pos]) per uncollected rock (rock order;
    // RockSample(15,15)'s duplicated coordinate makes CHECK 3 appear twice — kept, it weights the draw).
    static __device__ __forceinline__ int legal_count(const Shared &sh, const Params &p, const State &st, uint32_t &pre,
                                                      int &n_pre, uint32_t &alive)
    {
        const S s = st.s;
        const int x = (int)(s & 15u), y = (int)((s >> 4) & 15u), K = p.num_rocks;
        pre = 1u; n_pre = 1;                                                     // 3 bits per entry
        if (y + 1 < p.size) { pre |= 0u << (3 * n_pre); ++n_pre; }
        if (y - 1 >= 0) { pre |= 2u << (3 * n_pre); ++n_pre; }
        if (x - 1 >= 0) { pre |= 3u << (3 * n_pre); ++n_pre; }
        const int id = sh.grid[x * 16 + y];
        if (id >= 0 && id < K && ((uint32_t)(s >> (8 + 2 * (id & 15))) & 3u) != 1u) { pre |= 4u << (3 * n_pre); ++n_pre; }
        // uncollected rocks (code != 1), all K at once: with the 2-bit codes spread over even/odd bits,
        // "collected" is (low bit set, high bit clear); alive stays in spread form, rock j at bit 2 j
        const uint64_t r = (uint64_t)s >> 8;
        const uint64_t even = 0x5555555555555555ull;
        const uint64_t spread = ~(r & ~(r >> 1)) & even & ((1ull << (2 * K)) - 1ull);
        alive = (uint32_t)spread;                                      // K <= 16 rocks: 32 bits
        return n_pre + __popc(alive);
    }
    // the legal list in the form both its length and its idx-th entry come from: computed once per step by the loops that
    // need both (the rollout kernel)
    struct Legal { uint32_t pre, alive; int n_pre, count; };
    static __device__ __forceinline__ Legal legal_set(const Shared &sh, const Params &p, const State &st)
    {
        Legal L;
        L.count = legal_count(sh, p, st, L.pre, L.n_pre, L.alive);
        return L;
    }
    static __device__ __forceinline__ int legal_pick(const Shared &sh, const Legal &L, int idx)
    {
        if (idx < L.n_pre) return (int)((L.pre >> (3 * idx)) & 7u);
        // rock j sits at bit 2 j of `alive`: the (idx - n_pre)-th set bit, without a data-dependent loop
        const int j = nth_set_bit<2>(L.alive, idx - L.n_pre) >> 1;
        const uint32_t rxy = sh.rxy[j & 15];
        return 5 + sh.grid[(rxy & 15u) * 16 + (rxy >> 4)];
    }
    static __device__ __forceinline__ int legal_nth(const Shared &sh, const Params &p, const State &st, int idx)
    {
        return legal_pick(sh, legal_set(sh, p, st), idx);
    }
    static __device__ __forceinline__ int legal_count(const Shared &sh, const Params &p, const State &st)
    {
        uint32_t pre, alive; int n_pre;
        return legal_count(sh, p, st, pre, n_pre, alive);
    }

    // ---- heuristic-policy support ------------------------------------------------------
    // the "worth another CHECK" test of rock.py:371 on one rock's statistics
    static __device__ __forceinline__ bool check_ok(int measured, int count, double pv)
    {
        return measured < 5 && abs(count) < 2 && 0 < pv && pv < 1;
    }
    // rock.py:177-191: side statistics of the rock a CHECK just measured (CHECK does not move the agent, so the
    // stored position is the one the reading was taken from); keeps the rock's bit of the caller's copy of b.check_ok current
    static __device__ __forceinline__ void belief_update(const Shared &sh, const Params &p, const State &st, int a, int ob,
                                                         const pomdp_rock_belief &b, int64_t n, uint32_t i, uint32_t &ck)
    {
        const S s = st.s;
        const int x = (int)(s & 15u), y = (int)((s >> 4) & 15u), r = (a - 5) & 15;
        const uint32_t rxy = sh.rxy[r];
        const double eff = p.eff[abs(x - (int)(rxy & 15u)) + abs(y - (int)(rxy >> 4))];
        const int64_t k = (int64_t)r * n + i;
        double lkv = b.lkv[k], lkw = b.lkw[k];
        const int measured = b.measured[k] + 1;
        int count = b.count[k];
        if (ob == 2) { count += 1; lkv *= eff; lkw *= (1 - eff); }
        else         { count -= 1; lkw *= eff; lkv *= (1 - eff); }
        const double denom = (.5 * lkv) + (.5 * lkw);
        const double pv = (.5 * lkv) / denom;
        b.measured[k] = measured;
        b.count[k] = count;
        b.lkv[k] = lkv;
        b.lkw[k] = lkw;
        b.prob_valuable[k] = pv;
        const uint32_t bit = 1u << r;                                            // the caller keeps b.check_ok[i] in `ck`
        ck = check_ok(measured, count, pv) ? (ck | bit) : (ck & ~bit);
    }

    // rock.py:293-374 _generate_preferred with use_heuristic=True, as a bitmask over actions: every list the
    // heuristic builds is in ascending action order ([SAMPLE], [EAST], or N/E/S/W then the CHECKs by rock index);
    // 0 = the heuristic produced nothing and the caller falls back to _generate_legal() (rock.py:374-375).
    // Per-rock tests come from the two derived words b.check_ok / h.move_ok (bit j: total_move[j] >= 0, bit 16 + j:
    // total_sample[j] > 0): 16 bytes per lane, whatever K is.
    static __device__ __forceinline__ uint32_t preferred_mask(const Shared &sh, const Params &p, const State &st,
                                                              const pomdp_rock_belief &b, const pomdp_history &h,
                                                              int64_t n, uint32_t i)
    {
        return preferred_mask(sh, p, st, h, n, i, ld_stream(b.check_ok + i), ld_stream(h.move_ok + i), ld_stream(h.size + i));
    }
    static __device__ __forceinline__ uint32_t preferred_mask(const Shared &sh, const Params &p, const State &st,
                                                              const pomdp_history &h, int64_t n, uint32_t i, uint32_t ck,
                                                              uint32_t mv, int hsize, int, int)
    {
        return preferred_mask(sh, p, st, h, n, i, ck, mv, hsize);
    }
    // the same with the lane's three per-lane words already loaded (the fused kernel issues those loads up front)
    static __device__ __forceinline__ uint32_t preferred_mask(const Shared &sh, const Params &p, const State &st,
                                                              const pomdp_history &h, int64_t n, uint32_t i, uint32_t ck,
                                                              uint32_t mv, int hsize)
    {
        const S s = st.s;
        const int x = (int)(s & 15u), y = (int)((s >> 4) & 15u), K = p.num_rocks;
        const int id = sh.grid[x * 16 + y];
        // rock.py:300-313: SAMPLE when the CHECKs of the rock underfoot sum to > 0 — bit 16 + id of the derived word, no
        // load of total_sample inside the step loop (a load there waits for every store the loop has in flight)
        if (id >= 0 && id < K && ((uint32_t)(s >> (8 + 2 * (id & 15))) & 3u) != 1u && hsize != 0 && ((mv >> (16 + (id & 15))) & 1u))
            return 1u << 4;
        // uncollected rocks (code != 1), bit j = rock j: spread form as in legal_count, then the even bits squeezed together
        const uint64_t r = (uint64_t)s >> 8;
        uint32_t alive = (uint32_t)(~(r & ~(r >> 1)) & 0x5555555555555555ull & ((1ull << (2 * K)) - 1ull));
        alive = (alive | (alive >> 1)) & 0x33333333u;
        alive = (alive | (alive >> 2)) & 0x0F0F0F0Fu;
        alive = (alive | (alive >> 4)) & 0x00FF00FFu;
        alive = (alive | (alive >> 8)) & 0x0000FFFFu;
        const uint32_t am = alive & mv & 0xFFFFu;                                                 // rock.py:335: total >= 0
        if (!am) return 1u << 1;                                                                  // all_bad: rock.py:347-349
        // rock.py:338-345, per rock: north if above, else south if below, else (same row) west if left, else east if right
        const uint32_t dy = sh.dir_y[y], dx = sh.dir_x[x], same = am & sh.row[y];
        const bool north = (am & dy & 0xFFFFu) != 0, south = (am & (dy >> 16)) != 0;
        const bool west = (same & dx & 0xFFFFu) != 0, east = (same & (dx >> 16)) != 0;
        uint32_t m = (alive & ck) << 5;                                                           // rock.py:370-372
        if (y + 1 < p.size && north) m |= 1u << 0;                                                // rock.py:358-368
        if (east) m |= 1u << 1;
        if (y - 1 >= 0 && south) m |= 1u << 2;
        if (x - 1 >= 0 && west) m |= 1u << 3;
        return m;
    }

    // rock.py:389-399 _select_target; distances compared as dx^2 + dy^2 (the reference takes the square root of the
    // same integers, coord.py:83-85, and every candidate is below its initial bound of 2 * size)
    static __device__ __forceinline__ int select_target(const Shared &sh, const Params &p, const State &st,
                                                        const pomdp_rock_belief &b, int64_t n, uint32_t i)
    {
        const S s = st.s;
        const int x = (int)(s & 15u), y = (int)((s >> 4) & 15u), K = p.num_rocks;
        int best = 4 * p.size * p.size, best_rock = -1;
        for (int j = 0; j < K; ++j) {
            if (((uint32_t)(s >> (8 + 2 * j)) & 3u) == 1u || b.count[(int64_t)j * n + i] < 0) continue;
            const uint32_t rxy = sh.rxy[j];
            const int dx = x - (int)(rxy & 15u), dy = y - (int)(rxy >> 4), d2 = dx * dx + dy * dy;
            if (d2 < best) { best = d2; best_rock = j; }
        }
        return best_rock;
    }

    // rock.py:250-264 _compute_prob
    static __device__ __forceinline__ double compute_prob(const Shared &sh, const Params &p, const State &st, int a, int ob)
    {
        if (a <= 4) return ob == 0 ? 1.0 : 0.0;
        const S s = st.s;
        const int x = (int)(s & 15u), y = (int)((s >> 4) & 15u), r = (a - 5) & 15;
        const uint32_t rxy = sh.rxy[r];
        const double eff = p.eff[abs(x - (int)(rxy & 15u)) + abs(y - (int)(rxy >> 4))];
        const uint32_t code = (uint32_t)(s >> (8 + 2 * r)) & 3u;               // status + 1
        if ((ob == 2 && code == 2u) || (ob == 1 && code == 0u)) return eff;
        return 1 - eff;
    }

    // Everything of rock.py:123-194 except the sensor's Bernoulli draw: transition, reward, done, and (in `aux`) what
    // the draw will be compared with.  Branch-free: the three action classes (move / SAMPLE / CHECK) are all
    // evaluated and selected, so a wave with mixed actions — every wave, under a random policy — runs one straight line.
    template <class RT>
    static __device__ __forceinline__ void step_pre(const Shared &sh, const Params &p, State &st, int a, RT &rew,
                                                    int &done, Aux &aux)
    {
        const S s = st.s;
        const uint32_t x = (uint32_t)s & 15u, y = ((uint32_t)s >> 4) & 15u;
        const uint32_t size = (uint32_t)p.size, K = (uint32_t)p.num_rocks;
        // CHECK rock a-5 (rock.py:171-175, 401-407, 383-387; coord.py:79-81: L1 distance = sum of absolute byte differences)
        const int r = (a - 5) & 15;
        const uint32_t rp = sh.rpos[r];
        const uint32_t d = __builtin_amdgcn_sad_u8(x | (y << 8), rp, 0u);
        aux.th = sh.thr[d].x;
        aux.r = (uint8_t)r;
        aux.good = ((uint32_t)(s >> (8 + 2 * r)) & 3u) == 2u;
        aux.want = a > 4;
        const int penalty = STOCH ? 0 : -100;                                  // rock.py:117 / rock.py:432
        // SAMPLE (rock.py:160-169); ids >= K raise IndexError in the reference, "no rock" here (-1 wraps above K)
        const int id = sh.grid[x * 16 + y];
        const int sh_ = 8 + 2 * (id & 15);
        const uint32_t code = (uint32_t)(s >> sh_) & 3u;
        const bool sample_ok = ((uint32_t)id < K) & (code != 1u);
        // move: 0 N (0,+1)  1 E (+1,0)  2 S (0,-1)  3 W (-1,0)   (coord.py:101-110, rock.py:134-158): the steps are
        // 2-bit signed fields of two constants, zero for every other action (a < 16; is_move masks the rest)
        const uint32_t a2 = (uint32_t)a << 1;
        const uint32_t nx = x + (uint32_t)__builtin_amdgcn_sbfe(0xC4, a2, 2u), ny = y + (uint32_t)__builtin_amdgcn_sbfe(0x31, a2, 2u);
        const bool inside = max(nx, ny) < size;                                // -1 wraps
        const bool is_move = a < 4, is_sample = a == 4, east = a == 1;
        // The three action classes meet in lane-mask logic (scalar unit) and a handful of selects:
        // state ^= the bits that change — the position byte of a move that stays inside, or a sampled rock's code -> 1
        const bool moved = is_move & inside, left = is_move & !inside, sampled = is_sample & sample_ok,
                   missed = is_sample & !sample_ok;
        const uint32_t d_move = ((uint32_t)s & 0xFFu) ^ (nx | (ny << 4));
        const S d_sample = (S)(code ^ 1u) << sh_;                              // code 0 or 2 -> 1 (collected)
        st.s = s ^ (moved ? (S)d_move : (sampled ? d_sample : (S)0));
        // reward: +10 east exit / good rock, -10 bad rock, penalty for leaving elsewhere or sampling nothing, else 0
        const bool good_rock = code == 2u;
        int rw = ((left & east) | (sampled & good_rock)) ? 10 : 0;
        rw = (sampled & !good_rock) ? -10 : rw;
        rw = ((left & !east) | missed) ? penalty : rw;
        rew = rw;
        // done (rock.py:139-141, 193): a move that left the grid, or the -100 penalty
        if (STOCH) done = left & east;                                         // penalties never terminate (rock.py:503)
        else done = left | missed;
    }
    // the low 26 bits of the sensor threshold of a CHECK of rock r from where the lane stands (a CHECK does not move it):
    // only a tie asks for them, so they are looked up there instead of travelling with every step
    static __device__ __forceinline__ uint32_t thr_lo_of(const Shared &sh, const State &st, int r)
    {
        const uint32_t x = (uint32_t)st.s & 15u, y = ((uint32_t)st.s >> 4) & 15u;
        return sh.thr[__builtin_amdgcn_sad_u8(x | (y << 8), sh.rpos[r & 15], 0u)].y;
    }
    static constexpr int TAB_ACTIONS = W == 1 ? 17 : 21;                    // 5 + K: K <= 12 in one state word, <= 16 in two

    // ---- the lane step from a (position, action) -> outcome table, as one packed record -------------------------------------
    // Everything a step derives from the agent's cell and the action alone — where a move leads and whether it leaves the
    // board, which rock lies under a SAMPLE, the sensor threshold of a CHECK from here — is one 32-bit entry, built in LDS by
    // the workgroup when a multi-step launch starts (one thread per cell, one pass over the actions) and read with a single
    // lookup per lane-step.  With 4-byte trajectory records the fused loops are bound by instruction issue; a lane step that
    // fills (reward, done, aux) for a separate sensor / select stage cost ~49 vector instructions (round 3's step_tab, gone
    // since every table-driven launch takes this form).  step_rec produces the lane's Packed record
    // (traj_out.hip.h: action | ob << 8 | reward code << 16 | done << 24) and its new state in ~31, from a table whose
    // entries already hold, per (action, position), everything that does not depend on the rocks' codes:
    //   every entry: bits 28-30 = the OUTCOME CODE the step has when no uncollected rock is sampled ("fallback"), bit 31 =
    //                a rock with an id < K lies under a SAMPLE;
    //   a <  4: bits 0-7 = position byte XOR new position byte (0 if the move leaves); fallback 6 inside, 3 leaving east,
    //           1 leaving elsewhere (StochasticRock: 6, rock.py:432, 503);
    //   a == 4: bits 0-5 = bit offset of the cell's rock code in the state (bit 5: it lies in the upper word); fallback 1
    //           (StochasticRock: 6);
    //   a >= 5: bits 0-27 = the sensor threshold's high bits (thr >> 26 <= 2^27: a sensor that is always right at distance 0
    //           has thr = 2^53) at this distance; fallback 6 — the kernel compares against (H >> 5) | 6 << 28 (one
    //           v_alignbit_b32), so the threshold test and the tie test read the entry as it is.
    // Outcome codes: 0 = bad rock sampled (-10), 1 = penalty (-100, done), 2 = good rock sampled (+10), 3 = east exit (+10,
    // done), 6 = nothing (0) — a sampled rock's own code IS its outcome code, done is bit 0, and the reward byte is one
    // v_perm_b32 lookup in an 8-byte constant.
    static constexpr uint32_t REC_LUT_LO = 0x0A0A9CF6u, REC_LUT_HI = 0x00000000u;   // reward byte by outcome code 0..7
    struct RecTab { uint32_t e[TAB_ACTIONS][256]; };
    static __device__ __forceinline__ void build_rec_tab(RecTab &tab, const Shared &sh, const Params &p, int pos)
    {
        const uint32_t x = (uint32_t)pos & 15u, y = (uint32_t)pos >> 4, size = (uint32_t)p.size, K = (uint32_t)p.num_rocks;
        const int id = sh.grid[x * 16 + y];
        const uint32_t NOTHING = 6u << 28, PENALTY = (STOCH ? 6u : 1u) << 28, EXIT_EAST = 3u << 28;
        for (int a = 0; a < 5 + (int)K && a < TAB_ACTIONS; ++a) {
            uint32_t e;
            if (a < 4) {
                const uint32_t nx = x + (uint32_t)((a == 1) - (a == 3)), ny = y + (uint32_t)((a == 0) - (a == 2));
                const bool inside = max(nx, ny) < size;
                e = inside ? (((uint32_t)pos ^ (nx | (ny << 4))) | NOTHING) : (a == 1 ? EXIT_EAST : PENALTY);
            } else if (a == 4) {
                const bool rock = (uint32_t)id < K;
                e = (rock ? ((8u + 2u * (uint32_t)id) | 0x80000000u) : 0u) | PENALTY;
            } else {
                e = sh.thr[__builtin_amdgcn_sad_u8(x | (y << 8), sh.rpos[a - 5], 0u) & 31u].x | NOTHING;
            }
            tab.e[a][pos] = e;
        }
    }
    // rock.py:123-194 for one lane: s -> s' (the fresh episode `fresh` if the step ends this one), rec = the step's record.
    // H: the lane's sensor high word; `lo` yields its low word (a tie, 2^-27 per CHECK).  Two state words (K > 12): the rock
    // codes run on into the upper word (rock j at bits 8 + 2 j), a code never straddles the words, and the word a SAMPLE or
    // a CHECK reads is a select on bit 5 of its bit offset.
    template <class LowWord>
    static __device__ __forceinline__ void step_rec(const Shared &sh, const RecTab &tab, S &s, uint32_t a, uint32_t H, S fresh,
                                                    uint32_t &rec, LowWord lo)
    {
        const uint32_t s_lo = (uint32_t)s, s_hi = W == 2 ? (uint32_t)((uint64_t)s >> 32) : 0u;
        const uint32_t e = tab.e[a][s_lo & 0xFFu];
        // SAMPLE (rock.py:160-169): the cell's rock code, read at the entry's offset (entries of the other classes: the
        // sign bit is clear, so whatever this reads is never used)
        const bool up_s = W == 2 && (e & 32u) != 0u;                            // the code lies in the upper word
        const uint32_t code = __builtin_amdgcn_ubfe(up_s ? s_hi : s_lo, e, 2u);
        const bool ok = ((int32_t)e < 0) & (code != 1u);                        // an uncollected rock with an id < K is underfoot
        const uint32_t collect = (code ^ 1u) << (e & 31u);                      // its code -> 1
        // moves (rock.py:134-158): the position delta, for that class only
        const uint32_t move = (a < 4u ? e : 0u) & 0xFFu;
        const uint32_t d_lo = (ok & !up_s) ? collect : move, d_hi = (ok & up_s) ? collect : 0u;
        const uint32_t oc = ok ? code : __builtin_amdgcn_ubfe(e, 28u, 3u);      // outcome code
        const uint32_t rbyte = __builtin_amdgcn_perm(REC_LUT_HI, REC_LUT_LO, (oc << 16) | 0x0C000C0Cu);   // reward byte << 16
        const uint32_t done = oc & 1u;
        // CHECK rock a - 5 (rock.py:171-175, 401-407): its code sits at bits 2a-2, 2a-1 of the state; good = the upper one
        const uint32_t gi = 2u * a - 1u;
        const bool good = __builtin_amdgcn_ubfe((W == 2 && (gi & 32u)) ? s_hi : s_lo, gi, 1u) != 0u;
        const uint32_t kh = __builtin_amdgcn_alignbit(12u, H, 5u);              // (H >> 5) | 6 << 28: compares with the entry itself
        bool correct = kh < e;
        if (a > 4u && kh == e) correct = (lo() >> 6) <= thr_lo_of(sh, State{s}, (int)a - 5);   // probability 2^-27
        const uint32_t ob = a > 4u ? ((good == correct) ? 2u << 8 : 1u << 8) : 0u;
        rec = rbyte | (done << 24) | ob | a;
        if constexpr (W == 2) s = done ? fresh : (S)(((uint64_t)(s_hi ^ d_hi) << 32) | (s_lo ^ d_lo));
        else s = done ? fresh : (S)(s_lo ^ d_lo);
    }

    // observation of a CHECK from the sensor's high word (rock.py:404-407); `lo` yields the low word on a tie
    template <class LowWord>
    static __device__ __forceinline__ int sensor_ob(const Shared &sh, const State &st, const Aux &aux, uint32_t H, LowWord lo)
    {
        const uint32_t kh = H >> 5;
        bool correct = kh < aux.th;
        if (kh == aux.th) correct = (lo() >> 6) <= thr_lo_of(sh, st, aux.r);  // probability 2^-27
        return aux.want ? ((aux.good == correct) ? 2 : 1) : 0;
    }

    // The step of a lane that was handed its sensor high word H (element lane & 3 of the quad's STEP block): the fused
    // rollout kernel computes one such block per lane every four steps and passes the words around the quad.
    static constexpr bool QUAD_SENSOR = !STOCH;
    template <class RT>
    static __device__ __forceinline__ void step_with_H(const Shared &sh, const Params &p, State &st, int a, const RngKey &key,
                                                       uint32_t lane, uint32_t H, int &ob, RT &rew, int &done)
    {
        Aux aux;
        step_pre(sh, p, st, a, rew, done, aux);
        ob = sensor_ob(sh, st, aux, H, [&]() { return elem(quad_block(key, lane, 1u), lane & 3u); });
    }

    // The whole step for one lane (launches that do not pool the quad's sensor block: one lane per thread, rollouts).
    template <class RT>
    static __device__ __forceinline__ void step(const Shared &sh, const Params &p, State &st, int a,
                                                const RngKey &key, uint32_t lane, int &ob, RT &rew, int &done)
    {
        const uint32_t e = lane & 3u;
        if (STOCH) {
            // the first double of the step gates the whole action (rock.py:443), the sensor draw is the second
            const uint4 g = quad_block(key, lane, 0u);
            const bool act = k53_le(elem(g, e), (uint32_t)(p.act_thr >> 26), (uint32_t)p.act_thr & LO_MASK,
                                    [&]() { return elem(quad_block(key, lane, 1u), e); }) != (p.act_gt != 0);
            State nx = st;
            Aux aux; RT r2; int d2;
            step_pre(sh, p, nx, a, r2, d2, aux);
            const uint4 h = quad_block(key, lane, 2u);
            const int o2 = sensor_ob(sh, nx, aux, elem(h, e), [&]() { return elem(quad_block(key, lane, 3u), e); });
            if (act) { st = nx; rew = r2; done = d2; ob = o2; }
            else { rew = 0; done = 0; ob = 0; }
            return;
        }
        Aux aux;
        step_pre(sh, p, st, a, rew, done, aux);
        const uint4 h = quad_block(key, lane, 0u);
        ob = sensor_ob(sh, st, aux, elem(h, e), [&]() { return elem(quad_block(key, lane, 1u), e); });
    }
};

} // namespace pomdp
