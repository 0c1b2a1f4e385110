// envs.hip.h — per-lane transition / observation / reward functions of the five envs,
// written against the packed int32 lane state documented in include/pomdp_hip.h.
// Each Env type plugs into the generic kernels in pomdp_kernels.hip:
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

// index of the n-th (0-based) set bit of m, branch-free: a binary search on popcounts (n < popc(m))
__device__ __forceinline__ int nth_set_bit(uint32_t m, int n)
{
    int pos = 0;
#pragma unroll
    for (int w = 16; w >= 1; w >>= 1) {
        const int c = __popc(m & ((1u << w) - 1u));
        const bool up = n >= c;
        n -= up ? c : 0;
        pos += up ? w : 0;
        m = up ? (m >> w) : m;
    }
    return pos;
}

// ===========================================================================
// RockSample
// ===========================================================================
// Word contract of the RockSample envs ("split layout").  Every draw RockSample makes is a numpy double
// = (high word H, low word L) -> k53 = (H >> 5) * 2^26 + (L >> 6).  H and L live in DIFFERENT Philox blocks:
//   reset  (stream RESET, counter word 0 = lane):      double j = rock j:  H = block 2 (j >> 2), L = block 2 (j >> 2) + 1,
//                                                       element j & 3;
//   step   (stream STEP,  counter word 0 = lane >> 2): double j (RockEnv: j = 0 the sensor; StochasticRockEnv:
//                                                       j = 0 the action gate, j = 1 the sensor): H = block 2 j,
//                                                       L = block 2 j + 1, element lane & 3 — one block serves the four
//                                                       lanes of a quad.
// A comparison k53 <= thr is decided by H alone unless (H >> 5) == (thr >> 26), which happens with probability 2^-27
// per draw; only then is the L block generated.  So a reset costs ceil(K / 4) blocks instead of ceil(K / 2), a
// quad's sensor draws cost one block instead of four, and a wave's whole step fits one pooled Philox pass.
//
// ABLATE is a profiling aid (tools/microbench.hip): bit 0 drops the sensor Philox block, bit 1 the auto-reset,
// bit 2 the LDS table lookups.  The product only instantiates ABLATE = 0.  STOCH selects StochasticRockEnv
// (rock.py:428-504).
template <int W, int ABLATE = 0, bool STOCH = false> // W = state words per lane: 1 (K <= 12) or 2
struct RockEnv {
    using Params = pomdp_rock_params;
    using Reward = int32_t;
    using S = typename std::conditional<W == 1, uint32_t, uint64_t>::type; // 32-bit ALU when one word is enough
    static constexpr int WORDS = W;
    static constexpr bool POOLED_LPT2 = !STOCH;   // pomdp_kernels.hip: Finisher<RockEnv, 2, .>
    static constexpr int ABL = ABLATE;            // experiment switches (tools/microbench.hip); 0 in the product
    struct Shared {
        uint32_t thr_hi[32];   // sensor threshold by L1 distance, thr >> 26  (compared with H >> 5)
        uint32_t thr_lo[32];   // thr & (2^26 - 1)                            (compared with L >> 6 on a tie)
        int8_t grid[256];      // rock id stamped at [x * 16 + y], -1 = none
        uint8_t rxy[16];       // rock j position, x | y << 4
    };
    struct State { S s; };
    // what the lane step leaves for the deferred sensor draw (pooled launches fetch H from the wave's task pass)
    struct Aux { uint32_t th, tl; bool good, want; };

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
        sh.thr_hi[tid & 31] = (uint32_t)(r.t >> 26);
        sh.thr_lo[tid & 31] = (uint32_t)r.t & LO_MASK;
        sh.rxy[tid & 15] = (uint8_t)((r.rx & 15) | (r.ry << 4));
    }
    static __device__ __forceinline__ void stage(Shared &sh, const Params &p, int tid) { stage_store(sh, stage_load(p, tid), tid); }
    static __device__ __forceinline__ int n_actions(const Params &p) { return 5 + p.num_rocks; }

    static constexpr bool NT = !(ABLATE & 16);
    static __device__ __forceinline__ void load(State &st, const uint32_t *state, int64_t n, uint32_t i)
    {
        st.s = ld_stream<NT>(state + i);
        if (W == 2) st.s |= (S)((uint64_t)ld_stream<NT>(state + n + i) << 32);
    }
    static __device__ __forceinline__ void store(const State &st, uint32_t *state, int64_t n, uint32_t i, bool)
    {
        st_stream<NT>(state + i, (uint32_t)st.s);
        if (W == 2) st_stream<NT>(state + n + i, (uint32_t)((uint64_t)st.s >> 32));
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
    // the 2-bit codes of rocks 4 g .. 4 g + 3 (8 bits) of lane `lane`'s fresh episode: one high block, low block on a tie
    static __device__ __forceinline__ uint32_t reset_group(const RngKey &key, uint32_t lane, int g, int K)
    {
        const uint4 h = stream_block(key, lane, POMDP_STREAM_RESET, 2u * (uint32_t)g);
        return reset_group_codes(h, key, lane, g, K);
    }
    static __device__ __forceinline__ uint32_t reset_group_codes(const uint4 &h, const RngKey &key, uint32_t lane, int g, int K)
    {
        uint32_t c0 = rock_code_hi(h.x), c1 = rock_code_hi(h.y), c2 = rock_code_hi(h.z), c3 = rock_code_hi(h.w);
        if (c0 == 3u || c1 == 3u || c2 == 3u || c3 == 3u) {                    // some rock undecided: 2^-27 per rock
            const uint4 l = stream_block(key, lane, POMDP_STREAM_RESET, 2u * (uint32_t)g + 1u);
            if (c0 == 3u) c0 = rock_code_lo(l.x);
            if (c1 == 3u) c1 = rock_code_lo(l.y);
            if (c2 == 3u) c2 = rock_code_lo(l.z);
            if (c3 == 3u) c3 = rock_code_lo(l.w);
        }
        const int j = 4 * g;
        return (j < K ? c0 : 0u) | (j + 1 < K ? c1 << 2 : 0u) | (j + 2 < K ? c2 << 4 : 0u) | (j + 3 < K ? c3 << 6 : 0u);
    }
    // block `j2` (0 = sensor / gate high words, 1 = their low words, 2 / 3 = StochasticRock's sensor) of lane's quad
    static __device__ __forceinline__ uint4 quad_block(const RngKey &key, uint32_t lane, uint32_t j2)
    {
        return philox4x32_10(lane >> 2, key.t_lo, key.t_hi, ((uint32_t)POMDP_STREAM_STEP << 24) | j2, key.k0, key.k1);
    }

    // rock.py:236-241 reset -> 266-271 _get_init_state -> 78-86 Rock.__init__:
    // status_j = sign(U_j - .5), rocks in index order, one double each.
    static __device__ __forceinline__ int reset(const Shared &, const Params &p, State &st, const RngKey &key,
                                                uint32_t lane)
    {
        uint64_t s = (uint32_t)p.start_x | ((uint32_t)p.start_y << 4);
        const int K = p.num_rocks;
        for (int g = 0; 4 * g < K; ++g) s |= (uint64_t)reset_group(key, lane, g, K) << (8 + 8 * g);
        st.s = (S)s;
        return 0; // Obs.NULL
    }
    static __device__ __forceinline__ int reset_ob(const Params &, const State &) { return 0; }   // what reset() returned

    // Wave-cooperative reset (one lane per thread launches).  A fresh episode needs NG = ceil(K/4) high blocks, only
    // ~1/8 of a wave's lanes reset in a given step while nearly every wave has at least one: done per lane, the whole
    // wave would pay all NG blocks.  Instead the (resetting lane, block) tasks are dealt out across the 64 lanes — one
    // Philox block per lane per pass — and the rock codes travel back through ds_bpermute.
    static __device__ __forceinline__ void reset_where(const Shared &sh, const Params &p, State &st, bool fresh,
                                                       const RngKey &key, uint32_t lane)
    {
        int unused;
        reset_core<false>(sh, p, st, fresh, key, lane, key, 1u, unused);
    }
    // Same pass, plus the synthetic policy's actions for the NEXT call counter (C-side rollout driver, policy and
    // env sharing the Philox key): the wave's 16 action blocks ride in lanes 0-15 of the first pass.
    static __device__ __forceinline__ void reset_where_chain(const Shared &sh, const Params &p, State &st, bool fresh,
                                                             const RngKey &key, uint32_t lane, const RngKey &akey,
                                                             uint32_t n_actions, int &next_action)
    {
        reset_core<true>(sh, p, st, fresh, key, lane, akey, n_actions, next_action);
    }

    template <bool CHAIN>
    static __device__ __forceinline__ void reset_core(const Shared &, const Params &p, State &st, bool fresh,
                                                      const RngKey &key, uint32_t lane, const RngKey &akey,
                                                      uint32_t n_actions, int &next_action)
    {
        if (ABLATE & 2) { if (CHAIN) next_action = synthetic_action(akey, lane, n_actions); return; }
        const uint64_t mask = __ballot(fresh);
        if (!CHAIN && mask == 0ull) return;                            // wave-uniform
        const int K = p.num_rocks;
        const int NG = (K + 3) >> 2;                                   // high blocks per reset (wave-uniform, 1..4)
        const int lid = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
        const int me = (int)(threadIdx.x & 63u);
        const int nreset = __popcll(mask);
        // stable partition: resetting lanes first; lane r (< nreset) learns who the r-th resetting lane is
        const int dst = fresh ? lid : nreset + (me - lid);
        const int src_of_rank = __builtin_amdgcn_ds_permute(dst << 2, me);
        constexpr int NA = CHAIN ? 16 : 0;                             // task list: [16 policy blocks] ++ [NG per reset]
        const int ntask = NA + nreset * NG;
        const uint32_t inv = (65536u + (uint32_t)NG - 1u) / (uint32_t)NG; // t / NG == (t * inv) >> 16 for t < 512
        uint64_t bits = 0;
        uint4 aw = make_uint4(0, 0, 0, 0);
        for (int base = 0; base < ntask; base += 64) {
            const int tid = base + me;
            const bool is_act = CHAIN && tid < NA;
            const int rt = tid < NA ? 0 : tid - NA;
            const int r = (int)(((uint32_t)rt * inv) >> 16), g = rt - r * NG;
            const int srcl = __shfl(src_of_rank, r & 63, 64);
            uint32_t codes = 0;
            if (tid < ntask) {
                // ONE Philox instance for both task kinds: the counter words are per-lane selects
                const uint32_t src_lane = lane - (uint32_t)me + (uint32_t)srcl;
                const uint32_t c0 = is_act ? ((lane - (uint32_t)me) >> 2) + (uint32_t)tid : src_lane;
                const uint32_t c1 = is_act ? akey.t_lo : key.t_lo, c2 = is_act ? akey.t_hi : key.t_hi;
                const uint32_t c3 = is_act ? ((uint32_t)POMDP_STREAM_ACTION << 24) : (((uint32_t)POMDP_STREAM_RESET << 24) | (2u * (uint32_t)g));
                const uint4 w = philox4x32_10(c0, c1, c2, c3, key.k0, key.k1);
                if (CHAIN && base == 0) aw = w;                        // lanes >= 16 hold words nobody reads
                if (!is_act) codes = reset_group_codes(w, key, src_lane, g, K);
            }
            for (int gg = 0; gg < NG; ++gg) {                          // wave-uniform trip count
                const int t = NA + lid * NG + gg - base;
                const uint32_t got = (uint32_t)__shfl((int)codes, t & 63, 64);
                if (t >= 0 && t < 64) bits |= (uint64_t)got << (8 + 8 * gg);
            }
            if (CHAIN && base == 0) {
                // lane l takes word (l & 3) of the policy block computed by lane l >> 2
                const int q = me >> 2;
                const uint32_t x = (uint32_t)__shfl((int)aw.x, q, 64), y = (uint32_t)__shfl((int)aw.y, q, 64);
                const uint32_t z = (uint32_t)__shfl((int)aw.z, q, 64), ww = (uint32_t)__shfl((int)aw.w, q, 64);
                const int b = me & 3;
                next_action = (int)__umulhi(b == 0 ? x : b == 1 ? y : b == 2 ? z : ww, n_actions);
            }
        }
        if (fresh) st.s = (S)((uint64_t)((uint32_t)p.start_x | ((uint32_t)p.start_y << 4)) | bits);
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
    static __device__ __forceinline__ int legal_nth(const Shared &sh, const Params &p, const State &st, int idx)
    {
        uint32_t pre, alive; int n_pre;
        legal_count(sh, p, st, pre, n_pre, alive);
        if (idx < n_pre) return (int)((pre >> (3 * idx)) & 7u);
        // rock j sits at bit 2 j of `alive`: the (idx - n_pre)-th set bit, without a data-dependent loop
        const int j = nth_set_bit(alive, idx - n_pre) >> 1;
        const uint32_t rxy = sh.rxy[j & 15];
        return 5 + sh.grid[(rxy & 15u) * 16 + (rxy >> 4)];
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
    // stored position is the one the reading was taken from); keeps the rock's bit of b.check_ok current
    static __device__ __forceinline__ void belief_update(const Shared &sh, const Params &p, const State &st, int a, int ob,
                                                         const pomdp_rock_belief &b, int64_t n, uint32_t i)
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
        const uint32_t bit = 1u << r, m = b.check_ok[i];
        b.check_ok[i] = check_ok(measured, count, pv) ? (m | bit) : (m & ~bit);
    }

    // rock.py:293-374 _generate_preferred with use_heuristic=True, as a bitmask over actions: every list the
    // heuristic builds is in ascending action order ([SAMPLE], [EAST], or N/E/S/W then the CHECKs by rock index);
    // 0 = the heuristic produced nothing and the caller falls back to _generate_legal() (rock.py:374-375).
    // Per-rock tests come from the two derived words b.check_ok / h.move_ok: 16 bytes per lane, whatever K is.
    static __device__ __forceinline__ uint32_t preferred_mask(const Shared &sh, const Params &p, const State &st,
                                                              const pomdp_rock_belief &b, const pomdp_history &h,
                                                              int64_t n, uint32_t i)
    {
        return preferred_mask(sh, p, st, h, n, i, ld_stream(b.check_ok + i), ld_stream(h.move_ok + i), ld_stream(h.size + i));
    }
    // the same with the lane's three per-lane words already loaded (the fused kernel issues those loads up front)
    static __device__ __forceinline__ uint32_t preferred_mask(const Shared &sh, const Params &p, const State &st,
                                                              const pomdp_history &h, int64_t n, uint32_t i, uint32_t ck,
                                                              uint32_t mv, int hsize)
    {
        const S s = st.s;
        const int x = (int)(s & 15u), y = (int)((s >> 4) & 15u), K = p.num_rocks;
        const int id = sh.grid[x * 16 + y];
        if (id >= 0 && id < K && ((uint32_t)(s >> (8 + 2 * (id & 15))) & 3u) != 1u && hsize != 0)
            if (h.total_sample[(int64_t)id * n + i] > 0) return 1u << 4;                          // rock.py:300-313
        uint32_t alive = 0;                                                                       // uncollected rocks
        for (int j = 0; j < K; ++j) alive |= (uint32_t)(((uint32_t)(s >> (8 + 2 * j)) & 3u) != 1u) << j;
        uint32_t am = alive & mv;                                                                 // rock.py:335: total >= 0
        if (!am) return 1u << 1;                                                                  // all_bad: rock.py:347-349
        bool north = false, south = false, west = false, east = false;
        while (am) {                                                                              // rock.py:338-345
            const int j = __ffs((int)am) - 1;
            am &= am - 1u;
            const uint32_t rxy = sh.rxy[j];
            const int rx = (int)(rxy & 15u), ry = (int)(rxy >> 4);
            if (ry > y) north = true;
            else if (ry < y) south = true;
            else if (rx < x) west = true;
            else if (rx > x) east = true;
        }
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
        const int x = (int)(s & 15u), y = (int)((s >> 4) & 15u);
        const int size = p.size, K = p.num_rocks;
        // CHECK rock a-5 (rock.py:171-175, 401-407, 383-387; coord.py:133-135: L1 distance)
        const int r = (a - 5) & 15;
        const uint32_t rxy = (ABLATE & 4) ? (uint32_t)(r * 17) : sh.rxy[r];
        const int d = abs(x - (int)(rxy & 15u)) + abs(y - (int)(rxy >> 4));
        aux.th = (ABLATE & 4) ? (uint32_t)d << 22 : sh.thr_hi[d];
        aux.tl = (ABLATE & 4) ? 0u : sh.thr_lo[d];
        aux.good = ((uint32_t)(s >> (8 + 2 * r)) & 3u) == 2u;
        aux.want = a > 4;
        const int penalty = STOCH ? 0 : -100;                                  // rock.py:117 / rock.py:432
        // SAMPLE (rock.py:160-169); ids >= K raise IndexError in the reference, "no rock" here
        const int id = (ABLATE & 4) ? ((x ^ y) & 7) - (x & 1) : sh.grid[x * 16 + y];
        const int sh_ = 8 + 2 * (id & 15);
        const uint32_t code = (uint32_t)(s >> sh_) & 3u;
        const bool sample_ok = (id >= 0) & (id < K) & (code != 1u);
        const int rew_sample = sample_ok ? (code == 2u ? 10 : -10) : penalty;
        const S s_sample = sample_ok ? (S)((s & ~((S)3 << sh_)) | ((S)1 << sh_)) : s;
        // move: 0 N (0,+1)  1 E (+1,0)  2 S (0,-1)  3 W (-1,0)   (coord.py:155-160, rock.py:134-158)
        const int nx = x + (a == 1) - (a == 3), ny = y + (a == 0) - (a == 2);
        const bool inside = ((unsigned)nx < (unsigned)size) & ((unsigned)ny < (unsigned)size);
        const S s_move = inside ? (S)((s & ~(S)0xFF) | (S)(uint32_t)(nx | (ny << 4))) : s;
        const int rew_move = inside ? 0 : (a == 1 ? 10 : penalty);             // east exit / off-grid
        const bool is_move = a < 4, is_sample = a == 4;
        st.s = is_move ? s_move : (is_sample ? s_sample : s);
        rew = is_move ? rew_move : (is_sample ? rew_sample : 0);
        if (STOCH) done = is_move && !inside && a == 1;                        // penalties never terminate (rock.py:503)
        else done = is_move ? !inside : (rew == -100);                         // rock.py:139-141, 193
    }
    // observation of a CHECK from the sensor's high word (rock.py:404-407); `lo` yields the low word on a tie
    template <class LowWord>
    static __device__ __forceinline__ int sensor_ob(const Aux &aux, uint32_t H, LowWord lo)
    {
        const bool correct = k53_le(H, aux.th, aux.tl, lo);
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
        ob = sensor_ob(aux, H, [&]() { return elem(quad_block(key, lane, 1u), lane & 3u); });
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
                                    [&]() { return elem(quad_block(key, lane, 1u), e); });
            State nx = st;
            Aux aux; RT r2; int d2;
            step_pre(sh, p, nx, a, r2, d2, aux);
            const uint4 h = quad_block(key, lane, 2u);
            const int o2 = sensor_ob(aux, elem(h, e), [&]() { return elem(quad_block(key, lane, 3u), e); });
            if (act) { st = nx; rew = r2; done = d2; ob = o2; }
            else { rew = 0; done = 0; ob = 0; }
            return;
        }
        Aux aux;
        step_pre(sh, p, st, a, rew, done, aux);
        const uint4 h = (ABLATE & 1) ? make_uint4(lane * 2654435761u, lane, 0, 0) : quad_block(key, lane, 0u);
        ob = sensor_ob(aux, elem(h, e), [&]() { return elem(quad_block(key, lane, 1u), e); });
    }
};

// ===========================================================================
// Tag
// ===========================================================================
struct TagEnv {
    using Params = pomdp_tag_params;
    using Reward = float;
    static constexpr int WORDS = 1;
    static constexpr bool POOLED_LPT2 = true;     // pomdp_kernels.hip: Finisher<TagEnv, 2, .>
    static constexpr bool QUAD_SENSOR = false;
    static constexpr int ABL = 0;
    // The T-shaped board never changes (tag.py:36-78): two small LDS tables replace the coordinate arithmetic of the
    // hot step — cell -> x | y << 4, and (cell, move N0 E1 S2 W3) -> the cell the move leads to, or the cell itself
    // when that square does not exist.  Every workgroup computes them once (threads 0-127, one entry each).
    struct Shared { uint8_t xy[32]; uint8_t mv[32 * 4]; };
    struct State { uint32_t w; };

    static __device__ __forceinline__ int n_actions(const Params &) { return 5; }
    static __device__ __forceinline__ void load(State &st, const uint32_t *state, int64_t, uint32_t i) { st.w = ld_stream(state + i); }
    static __device__ __forceinline__ void store(const State &st, uint32_t *state, int64_t, uint32_t i, bool) { st_stream(state + i, st.w); }

    // tag.py:52-57 get_tag_coord, 59-66 get_index, 46-50 is_inside
    static __device__ __forceinline__ void coord(int idx, int &x, int &y)
    {
        if (idx < 20) { x = idx % 10; y = idx / 10; }
        else { idx -= 20; x = idx % 3 + 5; y = idx / 3 + 2; }
    }
    static __device__ __forceinline__ int index(int x, int y) { return y < 2 ? y * 10 + x : 20 + (y - 2) * 3 + x - 5; }
    static __device__ __forceinline__ bool inside(int x, int y)
    {
        return y >= 2 ? (x >= 5 && x < 8 && y < 5) : (x >= 0 && x < 10 && y >= 0);
    }
    static __device__ __forceinline__ void stage(Shared &sh, const Params &, int tid)
    {
        if (tid < 128) {
            const int cell = min(tid >> 2, 28), d = tid & 3;
            int x, y;
            coord(cell, x, y);
            const int nx = x + (d == 1) - (d == 3), ny = y + (d == 0) - (d == 2);
            sh.mv[tid] = (uint8_t)(inside(nx, ny) ? index(nx, ny) : cell);
            if (d == 0) sh.xy[tid >> 2] = (uint8_t)(x | (y << 4));
        }
    }
    static __device__ __forceinline__ int num_opp(uint32_t w) { return (int)w >> 25; } // sign-extending
    static __device__ __forceinline__ uint32_t with_num_opp(uint32_t w, int no)
    {
        no = no < -64 ? -64 : no;
        return (w & 0x01FFFFFFu) | ((uint32_t)no << 25);
    }
    // tag.py:219-226
    static __device__ __forceinline__ int sample_ob(const Params &p, uint32_t w, int a)
    {
        const uint32_t agent = w & 31u;
        int ob = (int)agent;
        if (a < 4)
            for (int j = 0; j < p.num_opponents; ++j)
                if (((w >> (5 + 5 * j)) & 31u) == agent) ob = p.obs_cells;
        return ob;
    }

    // tag.py:97-102 reset, 181-193 _get_init_state, 43-44 sample = randint(0, 29)
    static __device__ __forceinline__ int reset(const Shared &, const Params &p, State &st, const RngKey &key,
                                                uint32_t lane)
    {
        WordStream ws(key, lane, POMDP_STREAM_RESET);
        uint32_t w = ws.randint(29u);
        for (int j = 0; j < p.num_opponents; ++j) w |= ws.randint(29u) << (5 + 5 * j);
        st.w = with_num_opp(w, p.num_opponents);
        return sample_ob(p, st.w, 0);
    }
    static __device__ __forceinline__ int reset_ob(const Params &p, const State &st) { return sample_ob(p, st.w, 0); }

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
        reset_where_chain_default<TagEnv>(sh, p, st, fresh, key, lane, akey, n_actions, next_action);
    }

    // tag.py:228-229: every action is legal
    static __device__ __forceinline__ int legal_count(const Shared &, const Params &p, const State &) { return n_actions(p); }
    static __device__ __forceinline__ int legal_nth(const Shared &, const Params &, const State &, int idx) { return idx; }

    // tag.py:231-243 _generate_preferred as a bitmask (ascending order is the reference's list order); tag.py:68-74
    // is_corner, coord.py:75-77 opposite.  history.size == 0 gives the legal list (all five actions).
    static __device__ __forceinline__ uint32_t preferred_mask(const Shared &sh, const Params &p, const State &st,
                                                              const pomdp_rock_belief &, const pomdp_history &h,
                                                              int64_t n, uint32_t i)
    {
        return preferred_mask(sh, p, st, h, n, i, 0u, 0u, ld_stream(h.size + i));
    }
    static __device__ __forceinline__ uint32_t preferred_mask(const Shared &, const Params &, const State &st,
                                                              const pomdp_history &h, int64_t, uint32_t i, uint32_t,
                                                              uint32_t, int hsize)
    {
        if (hsize == 0) return 0x1Fu;
        const int agent = (int)(st.w & 31u);
        int x, y;
        coord(agent, x, y);
        const bool corner = y < 2 ? (x == 0 || x == 9) : (y == 4 && (x == 5 || x == 7));
        if (h.last_ob[i] == 29 && corner) return 1u << 4;          // grid.n_tiles, whatever obs_cells was set to
        const int la = h.last_action[i];
        uint32_t m = 0;
        const int dx[4] = {0, 1, 0, -1}, dy[4] = {1, 0, -1, 0};
#pragma unroll
        for (int d = 0; d < 4; ++d)
            if (la != ((d + 2) & 3) && inside(x + dx[d], y + dy[d])) m |= 1u << d;
        return m;
    }

    // tag.py:209-217 _compute_prob
    static __device__ __forceinline__ double compute_prob(const Shared &, const Params &p, const State &st, int, int ob)
    {
        const uint32_t w = st.w, agent = w & 31u;
        if (ob == p.obs_cells)
            for (int j = 0; j < p.num_opponents; ++j)
                if (((w >> (5 + 5 * j)) & 31u) == agent) return 1.0;
        return ob == (int)agent ? 1.0 : 0.0;
    }

    // tag.py:108-143 with one opponent (the default and the benchmark configuration), branch-free: under a random
    // policy every wave holds both moves and TAGs, so both outcomes are evaluated and selected.  Only a failed TAG
    // on a live opponent draws random numbers — words 0-2 of block 0 of the lane's STEP stream: binomial(1,
    // move_prob) on (w0, w1), then np.random.choice over a list whose length is 2 or 4, i.e. randint with an exact
    // mask (one word, no rejection).  The step is therefore split: `pre` does everything but the opponent's flight
    // and says whether the draw is needed, `flee` applies it; launches that pool Philox work call them separately.
    struct Flight { uint32_t list; int oi; bool need; };
    // tag.py:260-280 `_admissable_actions`: the list the eight appends build depends only on the signs of
    // (opponent - agent) in x and y.  Entry k = 3 (sign dx + 1) + (sign dy + 1), four 2-bit moves each (N0 E1 S2 W3,
    // the reference's order); the two-element lists of the diagonal cases are stored twice over, so that
    // `word & 3` picks from them exactly as randint(2)'s `word & 1` does.  k = 4 (same cell) never draws.
    static constexpr uint64_t ADMISSIBLE_LO = 0x61993100adccecbbull;   // k = 0..7
    static constexpr uint32_t ADMISSIBLE_8 = 0x11u;                     // k = 8
    template <class RT>
    static __device__ __forceinline__ void step_one_opponent_pre(const Shared &sh, const Params &p, State &st, int a,
                                                                 int &ob, RT &rew, int &done, Flight &f)
    {
        const uint32_t w = st.w;
        const int agent = (int)(w & 31u), oi = (int)((w >> 5) & 31u), no = num_opp(w);
        const int axy = sh.xy[agent], oxy = sh.xy[oi];
        // a < 4: the agent moves if the target cell exists (tag.py:112-117)
        const uint32_t agent_m = sh.mv[4 * agent + (a & 3)];
        // a == 4 (tag.py:119-134): tagged iff co-located; otherwise the opponent may flee (tag.py:201-207, 260-280)
        const bool colocated = oi == agent;
        const int dx = (oxy & 15) - (axy & 15), dy = (oxy >> 4) - (axy >> 4);
        const int sx1 = min(max(dx, -1), 1) + 1, sy1 = min(max(dy, -1), 1) + 1;              // v_med3_i32
        const int k = 3 * sx1 + sy1;
        const uint32_t list = k == 8 ? ADMISSIBLE_8 : (uint32_t)(ADMISSIBLE_LO >> (8 * (k & 7))) & 0xFFu;
        const bool tag = a == 4;
        const uint32_t w_tag = with_num_opp(w, no - (int)colocated);
        const uint32_t wn = tag ? w_tag : ((w & ~31u) | agent_m);
        rew = tag ? (colocated ? 10.f : -10.f) : -1.f;
        ob = (!tag && ((wn >> 5) & 31u) == (wn & 31u)) ? p.obs_cells : (int)(wn & 31u);   // tag.py:219-226
        done = num_opp(wn) == 0;
        st.w = wn;
        f.list = list; f.oi = oi;
        f.need = tag && !colocated && no > 0;
    }
    // the opponent's flight from words 0-2 of the lane's STEP block (tag.py:201-207)
    static __device__ __forceinline__ void flee(const Shared &sh, const Params &p, State &st, const Flight &f, uint32_t w0,
                                                uint32_t w1, uint32_t w2)
    {
        const uint32_t pick = (f.list >> (2 * (w2 & 3u))) & 3u;       // np.random.choice: randint(2 or 4), exact mask
        const uint32_t to = sh.mv[4 * f.oi + (int)pick];               // the cell itself if the square does not exist
        if (f.need && k53(w0, w1) <= p.move_thr) st.w = (st.w & ~(31u << 5)) | (to << 5);
    }
    template <class RT>
    static __device__ __forceinline__ void step_one_opponent(const Shared &sh, const Params &p, State &st, int a,
                                                             const RngKey &key, uint32_t lane, int &ob, RT &rew, int &done)
    {
        Flight f;
        step_one_opponent_pre(sh, p, st, a, ob, rew, done, f);
        const uint4 blk = stream_block(key, lane, POMDP_STREAM_STEP, 0u);
        flee(sh, p, st, f, blk.x, blk.y, blk.z);
    }
    // reset() from the four words of block 0 of the lane's RESET stream (tag.py:181-193: randint(29) per cell, each a
    // masked-rejection loop); false when the rejections ran past the block (probability < 1e-3) — the caller then
    // takes the general path
    static __device__ __forceinline__ bool reset_from_block(const Params &p, State &st, const uint4 &b)
    {
        const uint32_t wd[4] = {b.x, b.y, b.z, b.w};
        uint32_t w = 0; int have = 0;
        const int want = 1 + p.num_opponents;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t v = wd[j] & 31u;
            if (have < want && v <= 28u) { w |= v << (5 * have); ++have; }
        }
        if (have < want) return false;
        st.w = with_num_opp(w, p.num_opponents);
        return true;
    }

    // tag.py:108-143 step, 201-207 move_opponent, 260-280 _admissable_actions
    template <class RT>
    static __device__ __forceinline__ void step(const Shared &sh, const Params &p, State &st, int a,
                                                const RngKey &key, uint32_t lane, int &ob, RT &rew, int &done)
    {
        if (p.num_opponents == 1) { step_one_opponent(sh, p, st, a, key, lane, ob, rew, done); return; }   // wave-uniform
        uint32_t w = st.w;
        const int agent = (int)(w & 31u);
        int ax, ay;
        coord(agent, ax, ay);
        if (a == 4) {
            WordStream ws(key, lane, POMDP_STREAM_STEP);
            bool tagged = false;
            int no = num_opp(w);
            for (int j = 0; j < p.num_opponents; ++j) {
                const int sh = 5 + 5 * j;
                const int oi = (int)((w >> sh) & 31u);
                if (oi == agent) { tagged = true; no -= 1; }
                else if (no > 0) {
                    int ox, oy;
                    coord(oi, ox, oy);
                    // admissible moves, 2 bits each (index into N0 E1 S2 W3), in the reference's list order
                    uint32_t list = 0; int cnt = 0;
                    if (ox >= ax) { list |= 1u << (2 * cnt); ++cnt; }
                    if (oy >= ay) { list |= 0u << (2 * cnt); ++cnt; }
                    if (ox <= ax) { list |= 3u << (2 * cnt); ++cnt; }
                    if (oy <= ay) { list |= 2u << (2 * cnt); ++cnt; }
                    if (ox == ax && oy > ay) { list |= 0u << (2 * cnt); ++cnt; }
                    if (oy == ay && ox > ax) { list |= 1u << (2 * cnt); ++cnt; }
                    if (ox == ax && oy < ay) { list |= 2u << (2 * cnt); ++cnt; }
                    if (oy == ay && ox < ax) { list |= 3u << (2 * cnt); ++cnt; }
                    if (ws.next_k53() <= p.move_thr) {                    // binomial(1, move_prob)
                        const uint32_t pick = (list >> (2 * ws.randint((uint32_t)cnt))) & 3u; // np.random.choice
                        const int nx = ox + (pick == 1u) - (pick == 3u), ny = oy + (pick == 0u) - (pick == 2u);
                        if (inside(nx, ny)) w = (w & ~(31u << sh)) | ((uint32_t)index(nx, ny) << sh);
                    }
                }
            }
            rew = tagged ? 10.f : -10.f;
            w = with_num_opp(w, no);
        } else {
            rew = -1.f;
            const int nx = ax + (a == 1) - (a == 3), ny = ay + (a == 0) - (a == 2);
            if (inside(nx, ny)) w = (w & ~31u) | (uint32_t)index(nx, ny);
        }
        ob = sample_ob(p, w, a);
        done = num_opp(w) == 0;
        st.w = w;
    }
};

// ===========================================================================
// BattleShip
// ===========================================================================
template <int MW> // mask words: ceil((cells + 6) / 32)
struct BattleShipEnv {
    using Params = pomdp_battleship_params;
    using Reward = int32_t;
    static constexpr int WORDS = 2 * MW;
    static constexpr bool POOLED_LPT2 = false;
    static constexpr bool QUAD_SENSOR = false;
    static constexpr int ABL = 0;
    struct Shared { int unused; };
    // Each 128-bit mask is two 64-bit registers (never an addressable array or vector: a dynamically indexed
    // one is lowered through LDS by the compiler); bit tests are a 64-bit select and one variable shift.
    struct Mask {
        uint64_t lo, hi;
        __device__ __forceinline__ uint32_t word(int j) const { return (uint32_t)((j < 2 ? lo : hi) >> (32 * (j & 1))); }
        __device__ __forceinline__ void set_word(int j, uint32_t w)
        {
            const uint64_t m = 0xFFFFFFFFull << (32 * (j & 1)), v = (uint64_t)w << (32 * (j & 1));
            if (j < 2) lo = (lo & ~m) | v; else hi = (hi & ~m) | v;
        }
    };
    struct State { Mask occ, vis; };

    static __device__ __forceinline__ void stage(Shared &, const Params &, int) {}
    static __device__ __forceinline__ int n_actions(const Params &p) { return p.x_size * p.y_size; }
    static __device__ __forceinline__ int reset_ob(const Params &, const State &) { return 0; }   // battleship.py:131-137
    static __device__ __forceinline__ void load(State &st, const uint32_t *state, int64_t n, uint32_t i)
    {
        uint32_t o[4] = {0, 0, 0, 0}, v[4] = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < MW; ++j) { o[j] = ld_stream(state + (int64_t)j * n + i); v[j] = ld_stream(state + (int64_t)(MW + j) * n + i); }
        st.occ.lo = o[0] | ((uint64_t)o[1] << 32); st.occ.hi = o[2] | ((uint64_t)o[3] << 32);
        st.vis.lo = v[0] | ((uint64_t)v[1] << 32); st.vis.hi = v[2] | ((uint64_t)v[3] << 32);
    }
    // a step only changes the visited half; the occupied half is rewritten on reset
    static __device__ __forceinline__ void store(const State &st, uint32_t *state, int64_t n, uint32_t i, bool was_reset)
    {
#pragma unroll
        for (int j = 0; j < MW; ++j) st_stream(state + (int64_t)(MW + j) * n + i, (uint32_t)st.vis.word(j));
        if (was_reset) {
#pragma unroll
            for (int j = 0; j < MW; ++j) st_stream(state + (int64_t)j * n + i, (uint32_t)st.occ.word(j));
        }
    }
    static __device__ __forceinline__ bool bit(const Mask &m, int a) { return ((a < 64 ? m.lo : m.hi) >> (a & 63)) & 1ull; }
    static __device__ __forceinline__ void set_bit(Mask &m, int a)
    {
        const uint64_t b = 1ull << (a & 63);
        m.lo |= a < 64 ? b : 0ull;
        m.hi |= a < 64 ? 0ull : b;
    }
    static __device__ __forceinline__ bool occupied(const Params &p, const State &st, int x, int y)
    {
        return (unsigned)x < (unsigned)p.x_size && (unsigned)y < (unsigned)p.y_size && bit(st.occ, y * p.x_size + x);
    }

    // battleship.py:131-137 reset, 167-180 _get_init_state, 195-211 collision, 182-193 mark_ship,
    // coord.py:122-123 Grid.sample, battleship.py:33-37 Ship.__init__ (position word(s) before direction word).
    //
    // The reference's collision() walks L+1 cells from pos and, for each, looks at the cell itself and
    // its N, E, S, W, NE, SE, SW neighbours (Compass[0..7]; NW is never looked at).  Here that is one
    // AND of two 128-bit masks: `blocked` = every cell that has an occupied cell in that 8-neighbourhood
    // (7 shifted copies of the occupancy mask, column-wrap guarded), against the L+1 ship cells; the
    // "pos + dir stays inside for i = 0..L" test reduces to the far end pos + (L+1) dir being inside.
    typedef unsigned __int128 u128;
    static __device__ __forceinline__ int reset(const Shared &, const Params &p, State &st, const RngKey &key,
                                                uint32_t lane)
    {
        WordStream ws(key, lane, POMDP_STREAM_RESET);
        const int X = p.x_size, Y = p.y_size;
        u128 col0 = 0;                                     // cells with x == 0
        for (int y = 0; y < Y; ++y) col0 |= (u128)1 << (y * X);
        const u128 colL = col0 << (X - 1);                 // cells with x == X - 1
        u128 occ = 0;
        int remaining = 0;
        for (int len = p.max_len; len >= 2; --len) {
            const u128 e = occ & ~col0, w = occ & ~colL;   // sources that may shift one column west / east
            const u128 blocked = occ | (occ >> X) | (occ << X) | (e >> 1) | (w << 1) | (e >> (X + 1)) | (e << (X - 1)) |
                                 (w << (X + 1));
            int a0, dx, dy;
            for (;;) {
                a0 = (int)ws.randint((uint32_t)(X * Y));
                const uint32_t dir = ws.randint(4u);
                dx = (dir == 1u) - (dir == 3u); dy = (dir == 0u) - (dir == 2u);   // Compass N E S W
                const int px = a0 % X, py = a0 / X;
                const int ex = px + (len + 1) * dx, ey = py + (len + 1) * dy;
                if (!((unsigned)ex < (unsigned)X && (unsigned)ey < (unsigned)Y)) continue;
                const int stride = dy * X + dx;                                   // bit distance between ship cells
                const int lo = stride > 0 ? a0 : a0 + len * stride;               // lowest bit of the L+1 checked cells
                const int gap = stride > 0 ? stride : -stride;
                u128 cells = 0;
                for (int i = 0; i <= len; ++i) cells |= (u128)1 << (lo + i * gap);
                if ((cells & blocked) == 0) break;
            }
            const int stride = dy * X + dx;
            for (int i = 0; i < len; ++i) occ |= (u128)1 << (a0 + i * stride);     // mark_ship: L cells from pos
            remaining += len;
        }
        st.occ.lo = (uint64_t)occ; st.occ.hi = (uint64_t)(occ >> 64);
        st.vis.lo = 0; st.vis.hi = 0;
        st.vis.set_word(MW - 1, (uint32_t)remaining << 26);
        return 0;
    }

    // Wave-cooperative reset.  A BattleShip reset is a long sequential rejection loop (about 20 attempts on
    // 10x10, 42 on 5x5) and roughly one wave in five holds a lane that needs one; run per lane it stalls 63
    // other lanes behind ~2000 divergent instructions.  Here the whole wave serves one resetting lane at a
    // time and evaluates up to 64 candidate placements at once:
    //   1. one Philox pass gives a 64-word window of that lane's RESET stream (lane l holds word c + l);
    //   2. the reference consumes the stream as [position words until one is < n_tiles][direction word],
    //      repeated.  With A = ballot(word is an acceptable position), word l is a *direction* word iff
    //      word l-1 is an accepted position word, i.e. D[l] = A[l-1] & ~D[l-1]: inside every run of ones of
    //      A the roles alternate, which is the "escaped character" recurrence and has a branch-free 64-bit
    //      solution (add-with-carry over the run starts; Langdale & Lemire, "Parsing gigabytes of JSON per
    //      second", §3.1.1).  Lane l is a candidate iff A[l] & ~D[l]; its direction word is lane l+1's;
    //   3. every candidate lane tests its placement with 128-bit mask arithmetic against `blocked`; the
    //      lowest successful lane is the ship the reference would have placed, and the cursor moves just
    //      past its direction word.  No success: the cursor moves past the last fully parsed word.
    // Same words in the same order as reset() above, hence the same boards.
    static __device__ __forceinline__ u128 u128_of(const uint32_t (&w)[4])
    {
        return (u128)(w[0] | ((uint64_t)w[1] << 32)) | ((u128)(w[2] | ((uint64_t)w[3] << 32)) << 64);
    }
    static __device__ __forceinline__ uint64_t direction_words(uint64_t a)
    {
        const uint64_t even = 0x5555555555555555ull;
        const uint64_t follows = a << 1;                       // words preceded by an acceptable word
        const uint64_t odd_starts = a & ~even & ~follows;      // runs of A that start on an odd bit
        const uint64_t even_start_runs = odd_starts + a;       // carry ripples through those runs
        return (even ^ (even_start_runs << 1)) & follows;
    }
    static __device__ __forceinline__ void reset_where(const Shared &, const Params &p, State &st, bool fresh,
                                                       const RngKey &key, uint32_t lane)
    {
        uint64_t todo = __ballot(fresh);
        if (todo == 0ull) return;                                            // wave-uniform
        const int me = (int)(threadIdx.x & 63u);
        const int X = p.x_size, Y = p.y_size, cells = X * Y;
        const uint32_t rmask = 0xFFFFFFFFu >> __clz((uint32_t)(cells - 1) | 1u); // randint(cells) bit-smear mask
        const u128 col0 = u128_of(p.col0), colL = col0 << (X - 1);
        const uint32_t inv_x = (65536u + (uint32_t)X - 1u) / (uint32_t)X;   // a / X == (a * inv_x) >> 16 for a < 128, X <= 16
        while (todo != 0ull) {
            const int src = __ffsll((long long)todo) - 1;
            todo &= todo - 1ull;
            const uint32_t glane = (uint32_t)__builtin_amdgcn_readlane((int)lane, src);
            int c = 0;                                                        // next unread word of the stream
            u128 occ = 0;
            int remaining = 0;
            for (int len = p.max_len; len >= 2; --len) {
                // blocked = occ and its N, E, S, W, NE, SE, SW shifts (NW excluded) in four 128-bit shifts:
                // h = {self, E, W}; south side = h << X (S, SE, SW); north side = {self, E} >> X (N, NE)
                const u128 e1 = (occ & ~col0) >> 1, h = occ | e1 | ((occ & ~colL) << 1);
                const u128 blocked = h | ((occ | e1) >> X) | (h << X);
                const u128 hpat = ((u128)1 << (len + 1)) - 1;                 // the L+1 checked cells, from bit 0
                const u128 vpat = u128_of(p.vpat[len + 1]);
                for (;;) {
                    const uint32_t wi = (uint32_t)(c + me);
                    const uint4 blk = stream_block(key, glane, POMDP_STREAM_RESET, (wi >> 2) & 0xFFFFFFu);
                    const uint32_t sel = wi & 3u;
                    const uint32_t word = sel == 0 ? blk.x : sel == 1 ? blk.y : sel == 2 ? blk.z : blk.w;
                    const uint64_t A = __ballot((word & rmask) <= (uint32_t)(cells - 1));
                    const uint64_t D = direction_words(A);
                    const uint64_t cand = A & ~D & 0x7FFFFFFFFFFFFFFFull;     // position word with its direction word in the window
                    const uint32_t dirword = (uint32_t)__shfl((int)word, (me + 1) & 63, 64);
                    const int a0 = (int)(word & rmask);
                    const uint32_t dir = dirword & 3u;
                    const int dx = (dir == 1u) - (dir == 3u), dy = (dir == 0u) - (dir == 2u);   // Compass N E S W
                    const int py = (int)(((uint32_t)a0 * inv_x) >> 16), px = a0 - py * X;
                    const int ex = px + (len + 1) * dx, ey = py + (len + 1) * dy;
                    const int stride = dy * X + dx;
                    const bool inside = (unsigned)ex < (unsigned)X && (unsigned)ey < (unsigned)Y;
                    const int lo = stride > 0 ? a0 : a0 + len * stride;
                    const u128 cellsm = (dx != 0 ? hpat : vpat) << (lo & 127);
                    const bool ok = ((cand >> me) & 1ull) && inside && (cellsm & blocked) == 0;
                    const uint64_t succ = __ballot(ok);
                    if (succ != 0ull) {
                        const int r = __ffsll((long long)succ) - 1;
                        const int a0w = __builtin_amdgcn_readlane(a0, r), sw = __builtin_amdgcn_readlane(stride, r);
                        // mark_ship: L cells from pos = the L-cell pattern shifted to its lowest cell
                        const int low = sw > 0 ? a0w : a0w + (len - 1) * sw;
                        occ |= ((sw == 1 || sw == -1) ? (((u128)1 << len) - 1) : u128_of(p.vpat[len])) << low;
                        remaining += len;
                        c += r + 2;
                        break;
                    }
                    // no placement here: word 63 is unread only if it is an accepted position word (its direction
                    // word lies in the next window)
                    c += ((A & ~D) >> 63) ? 63 : 64;
                }
            }
            if (me == src) {
                st.occ.lo = (uint64_t)occ; st.occ.hi = (uint64_t)(occ >> 64);
                st.vis.lo = 0; st.vis.hi = 0;
                st.vis.set_word(MW - 1, (uint32_t)remaining << 26);
            }
        }
    }
    static __device__ __forceinline__ void reset_where_chain(const Shared &sh, const Params &p, State &st, bool fresh,
                                                             const RngKey &key, uint32_t lane, const RngKey &akey,
                                                             uint32_t n_actions, int &next_action)
    {
        reset_where_chain_default<BattleShipEnv<MW>>(sh, p, st, fresh, key, lane, akey, n_actions, next_action);
    }

    // battleship.py:157-165 _generate_legal: the unvisited cells, ascending
    static __device__ __forceinline__ uint32_t unvisited(const Params &p, const State &st, int j)
    {
        const int cells = p.x_size * p.y_size, lo = 32 * j;
        const uint32_t valid = cells - lo >= 32 ? 0xFFFFFFFFu : (cells > lo ? (1u << (cells - lo)) - 1u : 0u);
        return ~st.vis.word(j) & valid;
    }
    static __device__ __forceinline__ int legal_count(const Shared &, const Params &p, const State &st)
    {
        int c = 0;
#pragma unroll
        for (int j = 0; j < MW; ++j) c += __popc(unvisited(p, st, j));
        return c;
    }
    static __device__ __forceinline__ int legal_nth(const Shared &, const Params &p, const State &st, int idx)
    {
        int a = 0;
#pragma unroll
        for (int j = 0; j < MW; ++j) {
            uint32_t z = unvisited(p, st, j);
            const int c = __popc(z);
            if (idx >= 0 && idx < c) {
                for (int k = idx; k > 0; --k) z &= z - 1u;
                a = 32 * j + __ffs((int)z) - 1;
            }
            idx -= c;
        }
        return a;
    }

    // no heuristic: _generate_preferred is _generate_legal (0 = fall back to the legal list)
    static __device__ __forceinline__ uint32_t preferred_mask(const Shared &, const Params &, const State &,
                                                              const pomdp_rock_belief &, const pomdp_history &, int64_t,
                                                              uint32_t) { return 0u; }
    static __device__ __forceinline__ uint32_t preferred_mask(const Shared &, const Params &, const State &,
                                                              const pomdp_history &, int64_t, uint32_t, uint32_t, uint32_t,
                                                              int) { return 0u; }
    // battleship.py:80-89 _compute_prob (reads the grid as it is after the shot)
    static __device__ __forceinline__ double compute_prob(const Shared &, const Params &, const State &st, int a, int ob)
    {
        if (ob == 0 && bit(st.vis, a)) return 1.0;
        if (ob == 1 && bit(st.occ, a)) return 1.0;
        return ob == 0 ? 1.0 : 0.0;
    }

    // battleship.py:91-122
    template <class RT>
    static __device__ __forceinline__ void step(const Shared &, const Params &p, State &st, int a,
                                                const RngKey &, uint32_t, int &ob, RT &rew, int &done)
    {
        int remaining = (int)(st.vis.word(MW - 1) >> 26);
        ob = 0; done = 0;
        if (bit(st.vis, a)) rew = -10;
        else {
            rew = -1;
            if (bit(st.occ, a)) { ob = 1; remaining -= 1; }
            set_bit(st.vis, a);
        }
        if (remaining == 0) { rew += p.x_size * p.y_size; done = 1; }
        st.vis.set_word(MW - 1, (st.vis.word(MW - 1) & 0x03FFFFFFu) | ((uint32_t)remaining << 26));
    }
};

// ===========================================================================
// Tiger
// ===========================================================================
struct TigerEnv {
    using Params = pomdp_tiger_params;
    using Reward = int32_t;
    static constexpr int WORDS = 1;
    static constexpr bool POOLED_LPT2 = false;
    static constexpr bool QUAD_SENSOR = false;
    static constexpr int ABL = 0;
    struct Shared { int unused; };
    struct State { uint32_t w; };

    static __device__ __forceinline__ void stage(Shared &, const Params &, int) {}
    static __device__ __forceinline__ int n_actions(const Params &) { return 3; }
    static __device__ __forceinline__ void load(State &st, const uint32_t *state, int64_t, uint32_t i) { st.w = ld_stream(state + i); }
    static __device__ __forceinline__ void store(const State &st, uint32_t *state, int64_t, uint32_t i, bool) { st_stream(state + i, st.w); }

    // tiger.py:60-66: state = state_space.sample() (gym-space RNG -> stream RESET_SPACE); ob = NULL
    static __device__ __forceinline__ int reset(const Shared &, const Params &, State &st, const RngKey &key,
                                                uint32_t lane)
    {
        st.w = stream_block(key, lane, POMDP_STREAM_RESET_SPACE, 0u).x & 1u; // randint(2): mask 1, never rejects
        return 2;
    }
    static __device__ __forceinline__ int reset_ob(const Params &, const State &) { return 2; }

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
    // tiger.py:125-138 _compute_prob
    static __device__ __forceinline__ double compute_prob(const Shared &, const Params &, const State &st, int a, int ob)
    {
        if (a == 2 && ob != 2) return ((int)(st.w & 1u) == ob) ? .85 : 1 - .85;
        if (a != 2 && ob == 2) return 1.0;
        return 0.0;
    }

    // tiger.py:72-88 step, 117-119 _sample_state, 140-149 _sample_ob, 155-172
    template <class RT>
    static __device__ __forceinline__ void step(const Shared &, const Params &p, State &st, int a,
                                                const RngKey &key, uint32_t lane, int &ob, RT &rew, int &done)
    {
        const int tiger = (int)(st.w & 1u);
        if (a != 2 && a == tiger) { ob = tiger; rew = -20; done = 1; return; } // terminal: ob is the state
        done = 0;
        if (a == 2) {
            rew = -1;
            const uint4 w = stream_block(key, lane, POMDP_STREAM_STEP, 0u);
            const bool flip = k53(w.x, w.y) > p.listen_thr;                     // p > .85
            ob = tiger ^ (int)flip;
        } else {
            rew = 10;
            st.w = stream_block(key, lane, POMDP_STREAM_STEP_SPACE, 0u).x & 1u; // state resampled
            ob = 2; // the uniform() the reference draws here has no effect on anything returned
        }
    }
};

// ===========================================================================
// Network
// ===========================================================================
struct NetworkEnv {
    using Params = pomdp_network_params;
    using Reward = float;
    static constexpr int WORDS = 1;
    static constexpr bool POOLED_LPT2 = false;
    static constexpr bool QUAD_SENSOR = false;
    static constexpr int ABL = 0;
    struct Shared { int unused; };
    struct State { uint32_t w; };

    static __device__ __forceinline__ void stage(Shared &, const Params &, int) {}
    static __device__ __forceinline__ int n_actions(const Params &p) { return 2 * p.n_machines + 1; }
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
    // network.py:43-55 _compute_prob
    static __device__ __forceinline__ double compute_prob(const Shared &, const Params &p, const State &st, int a, int ob)
    {
        if (a < 2 * p.n_machines) return ((int)((st.w >> (a >> 1)) & 1u) == ob) ? .95 : 1 - .95;
        return ob == 2 ? 1.0 : 0.0;
    }

    // network.py:71-114.  The reference draws one double per *up* machine in index order, then one for
    // the action.  Lanes iterate over the draws (j = 0, 1, ...), not over the machines: j is
    // wave-uniform, so the Philox block that feeds doubles 2q and 2q+1 is generated under a uniform
    // condition and the only divergence left is the per-lane number of up machines.
    template <class RT>
    static __device__ __forceinline__ void step(const Shared &, const Params &p, State &st, int a,
                                                const RngKey &key, uint32_t lane, int &ob, RT &rew, int &done)
    {
        const uint32_t s0 = st.w;
        uint32_t s = s0;
        const int M = p.n_machines;
        // reward: 2 per up machine with > 2 neighbours, 1 per other up machine   network.py:87-92
        double r = (double)(__popc(s0) + __popc(s0 & p.deg_gt2_mask));
        // machines whose neighbourhood has a failure, from the pre-update state    network.py:82-85
        uint32_t nb_failed = 0;
        for (int i = 0; i < M; ++i) nb_failed |= ((~s0 & p.nb_mask[i]) != 0u ? 1u : 0u) << i;
        const bool has_action = a < 2 * M;
        const int n_draws = __popc(s0) + (has_action ? 1 : 0);
        // Split word layout: double j compares by its high word — element j & 3 of block 2 (j >> 2) —
        // and needs its low word (same element of the next block) only on a tie, probability 2^-27 per draw.  One
        // Philox block therefore serves four draws instead of two.  Thresholds as (high 27 bits, low 26 bits).
        constexpr uint32_t LO = (1u << 26) - 1u;
        const uint32_t th_fail = (uint32_t)(p.fail_thr >> 26), tl_fail = (uint32_t)p.fail_thr & LO;
        const uint32_t th_nb = (uint32_t)(p.fail_nb_thr >> 26), tl_nb = (uint32_t)p.fail_nb_thr & LO;
        const uint32_t th_obs = (uint32_t)(p.obs_thr >> 26), tl_obs = (uint32_t)p.obs_thr & LO;
        uint32_t todo = s0;
        uint4 blk = make_uint4(0, 0, 0, 0);
        bool truthful = false;
        for (int j = 0; __any(j < n_draws); ++j) {
            if ((j & 3) == 0) blk = stream_block(key, lane, POMDP_STREAM_STEP, 2u * (uint32_t)(j >> 2));
            const uint32_t H = (j & 3) == 0 ? blk.x : (j & 3) == 1 ? blk.y : (j & 3) == 2 ? blk.z : blk.w;
            const bool machine_draw = todo != 0u;                                // network.py:94-99, else the action's draw
            const int i = __ffs((int)todo) - 1;
            const bool nbf = machine_draw && ((nb_failed >> (i & 31)) & 1u);
            const uint32_t th = machine_draw ? (nbf ? th_nb : th_fail) : th_obs;
            const uint32_t tl = machine_draw ? (nbf ? tl_nb : tl_fail) : tl_obs;
            const uint32_t kh = H >> 5;
            bool le = kh < th;                                                   // k53 <= thr, decided by the high word
            if (kh == th) {                                                      // tie: fetch the low word
                const uint4 lo = stream_block(key, lane, POMDP_STREAM_STEP, 2u * (uint32_t)(j >> 2) + 1u);
                const uint32_t L = (j & 3) == 0 ? lo.x : (j & 3) == 1 ? lo.y : (j & 3) == 2 ? lo.z : lo.w;
                le = (L >> 6) <= tl;
            }
            if (machine_draw) {
                if (!le) s &= ~(1u << i);                                        // fails iff k > thr
                todo &= todo - 1u;
            } else if (j < n_draws) {
                truthful = le;
            }
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
