#pragma once
// kernels_common.hip.h — what the translation units of libpomdp_hip.so share (gfx950 kernels + the C ABI of include/pomdp_hip.h).
//
// One wavefront lane advances one env instance.  State, action, ob, reward and done
// are struct-of-arrays columns in HBM, so every access of a wave is one coalesced
// 256-byte (int32) or 64-byte (done) segment.  Lookup tables (RockSample's rock-id
// grid, rock coordinates and sensor thresholds) are staged from the kernarg segment
// into LDS once per workgroup.  No MFMA: the path is integer / branch work.  What bounds
// a launch depends on what it writes: one step per launch moves 21 B per RockSample
// lane (HBM latency / bandwidth); a fused launch writing int32 columns 13 B per
// lane-step (the store stream); one writing 4-byte records or only per-lane returns
// is bound by VALU issue — Philox and the lane step.
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC (see __graft_entry__.build()).
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>

#include "../../include/pomdp_hip.h"
#include "envs.hip.h"
#include "philox.hip.h"
#include "traj_out.hip.h"
#include <cstdio>

namespace pomdp {

constexpr int BLOCK = 256;        // 4 waves: one per SIMD
// launcher -> fused kernels only (never part of the ABI's flags): the launch derives the actions of its first step from
// the synthetic policy itself and writes them to row 0 of `action`, instead of reading what a policy launch left there
constexpr int FLAG_GEN_FIRST = 1 << 8;
constexpr int MAX_BLOCKS = 256 * 8; // helper kernels: 256 CUs x 8 resident workgroups, grid-stride beyond

static inline int grid_for(int64_t n)
{
    const int64_t b = (n + BLOCK - 1) / BLOCK;
    return (int)(b < 1 ? 1 : (b > MAX_BLOCKS ? MAX_BLOCKS : b));
}
static inline unsigned blocks_for(int64_t n) { return (unsigned)((n + BLOCK - 1) / BLOCK); }
// below this many lanes one lane per thread is as fast or faster (measured: tools/microbench.hip at 2^16 .. 2^19)
#ifdef POMDP_LPT2_MIN_LANES                                   // same-box A/B builds (tools/ab_build.sh)
constexpr int64_t LPT2_MIN_LANES = POMDP_LPT2_MIN_LANES;
#else
constexpr int64_t LPT2_MIN_LANES = 1 << 18;
#endif
#ifdef POMDP_STEP_QUAD_MIN_LANES                              // same-box A/B builds (tools/ab_build.sh)
constexpr int64_t STEP_QUAD_MIN_LANES = POMDP_STEP_QUAD_MIN_LANES;
#else
constexpr int64_t STEP_QUAD_MIN_LANES = 1 << 19;
#endif

#ifdef POMDP_STEP_TILES                                       // same-box A/B builds (tools/ab_build.sh): tiles per thread of the one-step quad kernels
constexpr int STEP_TILES = POMDP_STEP_TILES;
#else
constexpr int STEP_TILES = 1;                                 // measured (round 6): two tiles per thread 7.6 against 6.5 us per call — DESIGN.md §5.2
#endif

// envs whose lanes carry the board of their next episode (BattleShip): `next` is loaded only where a lane may need it
template <class Env, class = void> struct has_next : std::false_type {};
template <class Env> struct has_next<Env, std::enable_if_t<Env::HAS_NEXT>> : std::true_type {};

// ---------------------------------------------------------------------------
// step: transition + observation + reward (+ same-call auto-reset of done lanes)
//
// One workgroup = 256 threads = LPT x 256 consecutive lanes; thread `tid` owns lanes
// base + tid + 256 * j (j < LPT), so every wave access is still one coalesced segment.
// All HBM loads of all of a thread's lanes (action, state words, done flag) are issued
// unconditionally before the table staging and its barrier: a wave pays one memory latency for
// LPT x 64 lanes, and the independent per-lane chains (LDS lookups, Philox, cross-lane reset)
// overlap.  Lanes past n read lane n-1 and have their stores predicated off.  Every lane of a
// wave reaches Env::reset_where (wave-cooperative reset).
// ---------------------------------------------------------------------------
// ---------------------------------------------------------------------------
// Finisher: the lane step of a thread's lanes and what follows it — auto-reset of the done lanes and, for
// CHAIN launches, the synthetic policy's actions of the next call counter.  Generic form: Env::step per lane,
// one Env::reset_where[_chain] per 64-lane sub-batch.
// ---------------------------------------------------------------------------
template <class Env, int LPT, bool CHAIN, class = void>
struct Finisher {
    struct Aux {};
    static constexpr bool HAS_PREPASS = false;
    static constexpr bool LOOP_BARRIER = CHAIN;      // run() shares `pol` across waves: a fused multi-step loop must fence its reuse
    // The fused multi-step loop does not use `pol` at one lane per thread: lane e of a quad computes the quad's policy
    // block of step s + e once per four steps and the words travel by ds_bpermute (steps_kernel), so the loop has no
    // barrier at all and a wave that runs a long cooperative reset (BattleShip) no longer stalls the other three.
    static constexpr bool QUAD_POLICY = CHAIN && LPT == 1;
    static __device__ __forceinline__ void resets_only(const typename Env::Shared &sh, const typename Env::Params &p,
                                                       typename Env::State (&st)[LPT], const bool (&fresh)[LPT],
                                                       const RngKey &key, const uint32_t (&lane)[LPT])
    {
#pragma unroll
        for (int j = 0; j < LPT; ++j) Env::reset_where(sh, p, st[j], fresh[j], key, lane[j]);
    }
    template <class RT>
    static __device__ __forceinline__ void lane_step(const typename Env::Shared &sh, const typename Env::Params &p,
                                                     typename Env::State &st, int a, const RngKey &key, uint32_t lane,
                                                     int &ob, RT &rew, int &done, Aux &)
    {
        Env::step(sh, p, st, a, key, lane, ob, rew, done);
    }
    static __device__ __forceinline__ void run(const typename Env::Shared &sh, const typename Env::Params &p,
                                               typename Env::State (&st)[LPT], const bool (&fresh)[LPT],
                                               const RngKey &key, const uint32_t (&lane)[LPT], const RngKey &akey,
                                               uint32_t n_act, int (&a_next)[LPT], const Aux (&)[LPT], int (&)[LPT])
    {
        // CHAIN: the workgroup's BLOCK * LPT consecutive lanes share BLOCK * LPT / 4 policy blocks (one per quad); its
        // first wave(s) compute each once, the others pick their word up from LDS, instead of every lane computing
        // its quad's block itself
        constexpr int NQ = BLOCK * LPT / 4;
        __shared__ uint32_t pol[CHAIN ? NQ : 1][4];
        if (CHAIN && (int)threadIdx.x < NQ) {                                  // whole waves: NQ is a multiple of 64
            const uint32_t quad = ((lane[0] - threadIdx.x) >> 2) + threadIdx.x;
            const uint4 w = philox4x32_10(quad, akey.t_lo, akey.t_hi, (uint32_t)POMDP_STREAM_ACTION << 24, akey.k0, akey.k1);
            pol[threadIdx.x][0] = w.x; pol[threadIdx.x][1] = w.y; pol[threadIdx.x][2] = w.z; pol[threadIdx.x][3] = w.w;
        }
#pragma unroll
        for (int j = 0; j < LPT; ++j) Env::reset_where(sh, p, st[j], fresh[j], key, lane[j]);
        if (CHAIN) {
            __syncthreads();
#pragma unroll
            for (int j = 0; j < LPT; ++j) {
                const uint32_t rel = threadIdx.x + (uint32_t)(j * BLOCK);
                a_next[j] = (int)__umulhi(pol[rel >> 2][rel & 3], n_act);
            }
        }
    }
};

// RockSample with two or more lanes per thread: every random word of the step comes from a quad-shared block (DESIGN.md
// §2) that depends on lane ids and the call counter only, so ONE task list per wave (64 * LPT lanes) holds
//   - the 16 * LPT sensor blocks of its quads (stream STEP) — a done lane's fresh episode starts from the same word its
//     sensor draw would have come from (a step never makes both: rock.hip.h, auto-reset),
//   - for CHAIN launches the 16 * LPT policy blocks of the next call counter,
// i.e. 32 (CHAIN: 64) Philox blocks for 128 lane-steps, dealt out 64 per pass BEFORE the lane step: the kernel runs the
// passes right after issuing its HBM loads.  The lane step therefore runs WITHOUT its sensor draw (Env::step_pre) and
// the observation and the fresh episodes are completed here from the pooled words.  Tasks and results are exchanged
// through a wave-private LDS scratch; LDS operations of one wave complete in order, so no barrier is involved.  Low
// words (needed with probability 2^-27 per draw) are generated per lane on demand.
template <int W, int LPT, bool CHAIN>
struct Finisher<RockEnv<W, false>, LPT, CHAIN, typename std::enable_if<(LPT >= 2)>::type> {
    using Env = RockEnv<W, false>;
    using Aux = typename Env::Aux;
    static constexpr bool LOOP_BARRIER = false;              // every scratch array is wave-private
    static constexpr int NQ = 16 * LPT;                      // quads of the wave's 64 * LPT lanes: sensor blocks
    static constexpr int NA = CHAIN ? 16 * LPT : 0;          // policy blocks of the next call counter
    static constexpr int NT = NQ + NA;
    template <class RT>
    static __device__ __forceinline__ void lane_step(const typename Env::Shared &sh, const typename Env::Params &p,
                                                     typename Env::State &st, int a, const RngKey &, uint32_t, int &ob,
                                                     RT &rew, int &done, Aux &aux)
    {
        Env::step_pre(sh, p, st, a, rew, done, aux);
        ob = 0;
    }
    // wave-private LDS scratch (one instance: function-local static of this accessor)
    static __device__ __forceinline__ uint32_t (&blk_lds())[BLOCK / 64][NT][4]
    {
        __shared__ uint32_t a[BLOCK / 64][NT][4];            // [0, NQ): sensor, [NQ, NT): policy; (sub-batch, quad)
        return a;
    }
    static constexpr bool HAS_PREPASS = true;
    // Sub-batch j of a thread is 256 j lanes further on.
    static __device__ __forceinline__ void prepass(const RngKey &key, const uint32_t (&lane)[LPT], const RngKey &akey)
    {
        const int wv = (int)(threadIdx.x >> 6), me = (int)(threadIdx.x & 63u);
        const uint32_t first0 = lane[0] - (uint32_t)me;                        // first lane of the wave's sub-batch 0
#pragma unroll
        for (int base = 0; base < NT; base += 64) {
            const int tid = base + me;
            if (tid < NT) {
                // ONE Philox instance for the two task kinds: the counter words are per-lane selects
                const bool pol = tid >= NQ;
                const int qt = pol ? tid - NQ : tid;                           // (sub-batch, quad) index
                const uint32_t quad = ((first0 + (uint32_t)(qt >> 4) * BLOCK) >> 2) + (uint32_t)(qt & 15);
                const uint32_t c1 = pol ? akey.t_lo : key.t_lo, c2 = pol ? akey.t_hi : key.t_hi;
                const uint32_t c3 = (uint32_t)(pol ? POMDP_STREAM_ACTION : POMDP_STREAM_STEP) << 24;
                const uint4 w = philox4x32_10(quad, c1, c2, c3, key.k0, key.k1);
                uint32_t *dst = blk_lds()[wv][tid];
                dst[0] = w.x; dst[1] = w.y; dst[2] = w.z; dst[3] = w.w;
            }
        }
    }
    static __device__ __forceinline__ void run(const typename Env::Shared &sh, const typename Env::Params &p,
                                               typename Env::State (&st)[LPT], const bool (&fresh)[LPT], const RngKey &key,
                                               const uint32_t (&lane)[LPT], const RngKey &, uint32_t n_act,
                                               int (&a_next)[LPT], const Aux (&aux)[LPT], int (&ob)[LPT])
    {
        const int wv = (int)(threadIdx.x >> 6), me = (int)(threadIdx.x & 63u);
        uint32_t H[LPT], Rw[LPT], P[LPT];
#pragma unroll
        for (int j = 0; j < LPT; ++j) {                                         // every read in flight together, one wait
            H[j] = blk_lds()[wv][16 * j + (me >> 2)][me & 3];
            Rw[j] = H[j];                                                       // auto-reset: the sensor block's word
            P[j] = CHAIN ? blk_lds()[wv][NQ + 16 * j + (me >> 2)][me & 3] : 0u;
        }
#pragma unroll
        for (int j = 0; j < LPT; ++j) {
            st[j].s = fresh[j] ? Env::fresh_state(p, Rw[j], key, lane[j]) : st[j].s;
            ob[j] = Env::sensor_ob(sh, st[j], aux[j], H[j], [&]() { return Env::elem(Env::quad_block(key, lane[j], 1u), lane[j] & 3u); });
            if (CHAIN) a_next[j] = (int)__umulhi(P[j], n_act);
        }
    }
};

// Tag with two lanes per thread: only a failed TAG on a live opponent draws (about a fifth of the lanes under a random
// policy) and resets are rare (episodes last hundreds of steps), so per-lane Philox blocks would be mostly wasted.
// ONE task list per wave (128 lanes): for CHAIN launches the 32 policy blocks of the next call counter, and the quad's STEP
// block for every lane whose opponent may flee or whose episode ended (ABI 13: both read the lane's element of it) — ~58
// blocks for 128 lane-steps instead of 256 (512 chained), dealt out 64 per pass through a wave-private LDS scratch like
// RockSample's.  The lane step runs
// without the flight (TagEnv::step_one_opponent_pre) and TagEnv::flee completes it from the pooled words.
// More than one opponent (wave-uniform, from the params): the general per-lane path.
template <bool CHAIN>
struct Finisher<TagEnv, 2, CHAIN, void> {
    using Env = TagEnv;
    using Aux = typename Env::Flight;
    static constexpr bool HAS_PREPASS = false;
    static constexpr bool LOOP_BARRIER = false;
    template <class RT>
    static __device__ __forceinline__ void lane_step(const typename Env::Shared &sh, const typename Env::Params &p,
                                                     typename Env::State &st, int a, const RngKey &key, uint32_t lane,
                                                     int &ob, RT &rew, int &done, Aux &aux)
    {
        if (p.num_opponents == 1) Env::step_one_opponent_pre(sh, p, st, a, ob, rew, done, aux);
        else { aux.need = false; Env::step(sh, p, st, a, key, lane, ob, rew, done); }
    }
    static __device__ __forceinline__ void run(const typename Env::Shared &sh, const typename Env::Params &p,
                                               typename Env::State (&st)[2], const bool (&fresh)[2], const RngKey &key,
                                               const uint32_t (&lane)[2], const RngKey &akey, uint32_t n_act,
                                               int (&a_next)[2], const Aux (&aux)[2], int (&)[2])
    {
        if (p.num_opponents != 1) {                                            // wave-uniform
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (CHAIN) Env::reset_where_chain(sh, p, st[j], fresh[j], key, lane[j], akey, n_act, a_next[j]);
                else Env::reset_where(sh, p, st[j], fresh[j], key, lane[j]);
            }
            return;
        }
        __shared__ uint8_t src_lds[BLOCK / 64][128];         // task rank -> virtual lane (me + 64 * sub-batch)
        __shared__ uint32_t res_lds[BLOCK / 64][32 + 128][4];   // [0,32): policy blocks (sub-batch, quad); then task results
        const int wv = (int)(threadIdx.x >> 6), me = (int)(threadIdx.x & 63u);
        // flights first, resets after them: a lane is never both (a failed TAG does not end the episode)
        const uint64_t f0 = __ballot(aux[0].need), f1 = __ballot(aux[1].need);
        const uint64_t r0 = __ballot(fresh[0]), r1 = __ballot(fresh[1]);
        const int nf0 = __popcll(f0), nfl = nf0 + __popcll(f1), nr0 = __popcll(r0), ntsk = nfl + nr0 + __popcll(r1);
        auto below = [&](uint64_t m) {
            return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        };
        const int rank[2] = {aux[0].need ? below(f0) : nfl + below(r0),
                             aux[1].need ? nf0 + below(f1) : nfl + nr0 + below(r1)};
        if (aux[0].need || fresh[0]) src_lds[wv][rank[0]] = (uint8_t)me;
        if (aux[1].need || fresh[1]) src_lds[wv][rank[1]] = (uint8_t)(me + 64);
        constexpr int NA = CHAIN ? 32 : 0;
        const int ntask = NA + ntsk;
        const uint32_t first0 = lane[0] - (uint32_t)me, first1 = lane[1] - (uint32_t)me;   // first lane of each sub-batch
        for (int base = 0; base < ntask; base += 64) {
            const int tid = base + me;
            if (tid < ntask) {
                const bool is_act = tid < NA;
                const int r = is_act ? 0 : tid - NA;
                const int v = (int)src_lds[wv][r & 127];
                const uint32_t src_lane = ((v >> 6) ? first1 : first0) + (uint32_t)(v & 63);
                const uint32_t quad = (((tid >> 4) ? first1 : first0) >> 2) + (uint32_t)(tid & 15);
                const uint32_t c0 = is_act ? quad : (src_lane >> 2);           // flights and auto-resets: the QUAD's STEP block (tag.hip.h)
                const uint32_t c1 = is_act ? akey.t_lo : key.t_lo, c2 = is_act ? akey.t_hi : key.t_hi;
                const uint32_t strm = is_act ? POMDP_STREAM_ACTION : POMDP_STREAM_STEP;
                const uint4 w = philox4x32_10(c0, c1, c2, strm << 24, key.k0, key.k1);
                uint32_t *dst = res_lds[wv][is_act ? tid : 32 + (r & 127)];
                dst[0] = w.x; dst[1] = w.y; dst[2] = w.z; dst[3] = w.w;
            }
        }
        uint4 rb[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {                                          // both blocks in flight, one wait
            const uint32_t *res = res_lds[wv][32 + (rank[j] & 127)];
            rb[j] = make_uint4(res[0], res[1], res[2], res[3]);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const uint32_t W = Env::elem(rb[j], lane[j] & 3u);
            Env::flee_word(sh, p, st[j], aux[j], W, [&]() { return Env::elem(Env::quad_block(key, lane[j], 1u), lane[j] & 3u); });
            if (fresh[j]) Env::auto_reset_word(p, st[j], W, key, lane[j]);
            if (CHAIN) a_next[j] = (int)__umulhi(res_lds[wv][16 * j + (me >> 2)][me & 3], n_act);
        }
    }
};

// Finishers that take the fused loop's policy words from quad-multiplexed blocks (generic form, one lane per thread)
template <int J>
static __device__ __forceinline__ uint32_t quad_bcast(uint32_t v)     // v of lane J of the caller's quad
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, J * 0x55, 0xF, 0xF, false);
}
// 4 x 4 transpose within a quad: lane e of the quad passes the four words of ITS block and gets word e of the blocks of
// lanes 0, 1, 2, 3 (t.x .. t.w).  The fused rollout and heuristic loops let lane e of a quad compute the quad-shared block
// of step base + e; step base + J then reads component J of the result — compile-time — where four broadcasts and a
// per-lane select per step cost twice as much.  Two butterfly stages (partner e ^ 1, then e ^ 2): each lane first
// selects the two words its partner lacks, so a stage is 2 selects + 2 DPP moves + 4 selects.
static __device__ __forceinline__ uint4 quad_transpose4(const uint4 &v, uint32_t e)
{
    const bool b0 = e & 1u, b1 = e & 2u;
    const uint32_t s0 = b0 ? v.x : v.y, s1 = b0 ? v.z : v.w;
    const uint32_t r0 = (uint32_t)__builtin_amdgcn_update_dpp((int)s0, (int)s0, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]
    const uint32_t r1 = (uint32_t)__builtin_amdgcn_update_dpp((int)s1, (int)s1, 0xB1, 0xF, 0xF, false);
    // column (e & 1) / 2 + (e & 1) of the rows (e & ~1, e | 1)
    const uint32_t p0 = b0 ? r0 : v.x, p1 = b0 ? v.y : r0, q0 = b0 ? r1 : v.z, q1 = b0 ? v.w : r1;
    const uint32_t u0 = b1 ? p0 : q0, u1 = b1 ? p1 : q1;
    const uint32_t w0 = (uint32_t)__builtin_amdgcn_update_dpp((int)u0, (int)u0, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
    const uint32_t w1 = (uint32_t)__builtin_amdgcn_update_dpp((int)u1, (int)u1, 0x4E, 0xF, 0xF, false);
    return make_uint4(b1 ? w0 : p0, b1 ? w1 : p1, b1 ? q0 : w0, b1 ? q1 : w1);
}
// Every load issued so far has landed.  Placed in front of a storing loop: gfx9 counts loads and stores on one counter, and
// a compiler that cannot prove the pre-loop loads settled waits on vmcnt(0) INSIDE the loop — i.e. for the previous step's
// stores, every step.  (It settles them in the pre-header by itself only when the loop has a single entry; the priority
// ladder below gives it two.)  s_waitcnt vmcnt(0) expcnt(7) lgkmcnt(15) on gfx9.
static __device__ __forceinline__ void wait_loads() { __builtin_amdgcn_s_waitcnt(0x0F70); }

// Issue priority of a wave inside a fused step loop (round 3, profiles/r03_prio_timeline.txt).  A SIMD's arbiter serves its
// OLDEST wave first: of the four workgroups a CU holds, the first ran its 64 steps almost as if alone (1.3 us per step, done
// after 83 us) and the last one finished at 191 us; once the early workgroups are gone the SIMDs run with three, two, one
// wave and lose throughput — the launch took 159 us where the median workgroup needed 130.  s_setprio outranks age, so a
// wave lowers its priority as it nears the end of the launch: with `rem` steps to go it runs at priority 3 above 7k/16
// steps, 2 above 3k/16, 1 above k/16, else 0.  The waves that are ahead wait at each threshold (still filling the issue
// slots the others leave) until the others are as close to the end, and a SIMD's waves finish within a step or two of
// each other: RockSample 159 -> 141 us per 64-step launch, 59 -> 54 per 20-step launch (thresholds measured at k = 16 ..
// 64: the geometric ladder beats quartiles, and it scales with k).
#ifndef POMDP_PRIO_MIN_WGS                                     // same-box A/B builds (tools/ab_build.sh)
#define POMDP_PRIO_MIN_WGS (2 * 256)
#endif
struct LoopPrio {
    int k, t0, t1, t2;
    bool on;
    // Only where a SIMD holds two or more of the launch's waves (256-thread workgroups: from two per CU; measured at 2^17
    // lanes with one lane per thread and at 2^19 with a quad: 3-8 % either way); a lone wave has nobody to yield to.
    __device__ __forceinline__ explicit LoopPrio(int k_steps, bool enable = true)
        : k(k_steps), t0(k_steps >> 4 > 0 ? k_steps >> 4 : 1), t1((3 * k_steps) >> 4), t2((7 * k_steps) >> 4)
    {
#ifdef POMDP_NO_LOOP_PRIO                                     // same-box A/B builds (tools/ab_build.sh)
        on = false;
#else
        on = enable && gridDim.x >= (unsigned)POMDP_PRIO_MIN_WGS;
#endif
    }
    // The step loop runs as four consecutive segments, one per priority: segment `seg` sets its priority and returns the
    // step at which it ends (`unit`: the loop's stride — the heuristic loop advances four steps at a time).  The ladder
    // lives OUTSIDE the step loop on purpose: as a compare-and-branch chain inside it, it cost a lone wave five taken
    // branches per step (Tiger, 2^14 lanes: 0.26 -> 0.34 us per step) and gave the loop a second entry, after which the
    // compiler waited for the pre-loop loads inside the loop (wait_loads above).
    template <int UNIT = 1>
    __device__ __forceinline__ int segment(int seg) const
    {
        if (!on) return k;                                 // one segment, the dispatch priority
        int end;
        switch (seg) {
        case 0: __builtin_amdgcn_s_setprio(3); end = k - t2; break;
        case 1: __builtin_amdgcn_s_setprio(2); end = k - t1; break;
        case 2: __builtin_amdgcn_s_setprio(1); end = k - t0; break;
        default: __builtin_amdgcn_s_setprio(0); return k;
        }
        if (UNIT == 1) return end;
        const int up = (end + UNIT - 1) / UNIT * UNIT;     // a stride's boundary, but never past the launch's last step
        return up < k ? up : k;
    }
};

template <int J> static __device__ __forceinline__ uint32_t comp(const uint4 &v) { return J == 0 ? v.x : J == 1 ? v.y : J == 2 ? v.z : v.w; }

template <class Fin, class = void> struct quad_policy_of : std::false_type {};
template <class Fin> struct quad_policy_of<Fin, std::enable_if_t<Fin::QUAD_POLICY>> : std::true_type {};

// envs whose step takes its words from QUAD_WORDS quad-shared blocks that a one-lane-per-thread loop can time-share (Network)
template <class Env, class = void> struct quad_words_of { static constexpr int value = 0; };
template <class Env> struct quad_words_of<Env, std::enable_if_t<(Env::QUAD_WORDS > 0)>> { static constexpr int value = Env::QUAD_WORDS; };

// envs whose step (and auto-reset) reads ONE word of a quad-shared STEP block (Tiger, Tag: Env::quad_block, step_w, fresh_w)
template <class Env, class = void> struct quad_word_env : std::false_type {};
template <class Env> struct quad_word_env<Env, std::enable_if_t<(Env::QUAD_WORD > 0)>> : std::true_type {};

struct NoTab {};
// the (position, action) table of a table-driven launch: RockEnv::RecTab, whose lane step (step_rec) yields the lane's
// packed record and its new state in one go
template <class Env, bool ON> struct step_tab_of { using type = NoTab; };
template <class Env> struct step_tab_of<Env, true> { using type = typename Env::RecTab; };

static inline RngKey make_key(uint64_t seed, uint64_t t)
{
    RngKey k;
    k.k0 = (uint32_t)seed; k.k1 = (uint32_t)(seed >> 32);
    k.t_lo = (uint32_t)t; k.t_hi = (uint32_t)(t >> 32);
    return k;
}

static inline bool bad_range(int64_t n, uint32_t lane0) { return n < 0 || (uint64_t)lane0 + (uint64_t)n > (1ull << 32); }

// pomdp_step_sync / pomdp_reset_sync (scalar mode): the flag the next one-lane launch publishes its outputs through;
// the launcher that takes it sets the pointer back to null (defined in api.hip)
extern thread_local uint32_t *tl_host_flag;
extern thread_local uint32_t tl_flag_value;

// Smallest batch each quad-per-thread loop takes (1024 lanes per workgroup).  Measured on MI355X, us per fused step at
// 2^17 / 2^18 / 2^19 / 2^20 lanes (profiles/r02_small_shards_gates.txt, r02k_small_shards.txt): RockSample(7,8) quad 1.50 /
// 1.52 / 1.82 / 2.89 against 1.25 / 1.39 / 1.99 with one or two lanes per thread; Tag (table-driven) 1.59 / 1.61 / 1.83 /
// 2.67 against 0.96 / 1.43 / 2.00; Tiger 0.67 / 0.67 / 1.21 / 2.51 against 0.44 / 0.85 / 1.41 / 2.66; Network 2.30 / 2.30 / 2.91 /
// 4.72 against 1.45 / 1.94 / 3.32 / 5.99 — below these sizes every kernel is bound by the latency of one wave's step
// (1.1-2.3 us), and more, lighter waves hide it better than fewer, heavier ones.
// Re-measured at the end of round 3 (priority ladder, RockSample's auto-reset from the sensor block; profiles/
// r03b_gates.txt): RockSample between 2^19 and 3 * 2^18 lanes is faster with the pooled two-lanes-per-thread loop (four waves
// per SIMD instead of two or two and a half; (15,15) at 2^19 lanes 1.25 against 1.46 us per step, (7,8) at 5 * 2^17 lanes
// 1.50 against 1.65) and from 3 * 2^18 lanes — three workgroups per CU — with the quad loop; Tiger at 2^18 lanes with one
// lane per thread (0.66 against 0.72).  StochasticRock has no pooled loop and keeps 2^19.
// Round 4: BattleShip's quad loop from 2^16 lanes (feed2 halved what its board pool costs, the one-lane loop builds boards on
// the spot with the wave-cooperative builder: 2^16 / 2^17 lanes 0.995 / 1.036 against 1.074 / 1.209 us per step).
// Round 5 (ABI 13: Tiger's and Tag's steps read one word of the quad's STEP block): the one-lane loops time-share that block
// like the policy's, the quad loops have it thread-local — Tiger 2^18 / 2^19 lanes 0.382 / 0.811 us per step with one lane
// per thread against 0.522 / 0.650 with a quad, Tag 2^18 0.765 against 0.919 (2^19: the pooled two-lane loop 1.74, the quad
// loop 1.08): the gates stay.
// POMDP_QUAD_MIN_LANES overrides all of them at build time for same-box A/B runs (tools/ab_build.sh lib ... -D...).
#ifdef POMDP_QUAD_MIN_LANES
constexpr int64_t QUAD_MIN_ROCK = POMDP_QUAD_MIN_LANES, QUAD_MIN_STOCHROCK = POMDP_QUAD_MIN_LANES, QUAD_MIN_TAG = POMDP_QUAD_MIN_LANES,
                  QUAD_MIN_GENERIC = POMDP_QUAD_MIN_LANES, QUAD_MIN_NETWORK = POMDP_QUAD_MIN_LANES, QUAD_MIN_BATTLESHIP = POMDP_QUAD_MIN_LANES;
#else
constexpr int64_t QUAD_MIN_ROCK = 3 << 18, QUAD_MIN_STOCHROCK = 1 << 19, QUAD_MIN_TAG = 1 << 19, QUAD_MIN_GENERIC = 1 << 19,
                  QUAD_MIN_NETWORK = 1 << 19, QUAD_MIN_BATTLESHIP = 1 << 16;
#endif

// BattleShip with half a quad per thread (battleship_steps_quad_kernel<.., 2>): the shards the quad loop gives two waves per
// SIMD or fewer.  -DPOMDP_BS_PAIR_MAX_LANES=0 builds the A arm (quad loop everywhere).
#ifdef POMDP_BS_PAIR_MAX_LANES
constexpr int64_t BS_PAIR_MAX_LANES = POMDP_BS_PAIR_MAX_LANES;
#else
constexpr int64_t BS_PAIR_MAX_LANES = 1 << 19;
#endif
// RockSample / StochasticRock with half a quad per thread (steps_quad_kernel<.., 2>), records and typed planes only.  Measured
// against both neighbours (profiles/r06_rock_pair_ab.txt, us per step, Packed): RockSample(7,8) at 3 * 2^17 lanes level with
// the pooled kernel (0.80 / 0.81), 13 * 2^15 .. 5 * 2^17 ahead (2^19: 1.07 -> 0.87), at 3 * 2^18 level with the quad loop;
// StochasticRock ahead of the pooled kernel from 3 * 2^17 (1.47 -> 1.11; 7 * 2^16: 2.06 -> 1.35) and behind the quad loop from
// 2^19 (1.32 / 1.36).  The returns sink loses at 2^19 (1.01 / 1.05: its float64 chain wants the quad's four independent
// lanes) and is not built in this form.  -DPOMDP_ROCK_PAIR_MAX_LANES=0: the A arm; with _MIN_LANES: one gate for both envs.
#ifdef POMDP_ROCK_PAIR_MAX_LANES
constexpr int64_t ROCK_PAIR_MAX_LANES = POMDP_ROCK_PAIR_MAX_LANES, STOCHROCK_PAIR_MAX_LANES = POMDP_ROCK_PAIR_MAX_LANES;
#else
constexpr int64_t ROCK_PAIR_MAX_LANES = (3 << 18) - 1, STOCHROCK_PAIR_MAX_LANES = (1 << 19) - 1;
#endif
#ifdef POMDP_ROCK_PAIR_MIN_LANES
constexpr int64_t ROCK_PAIR_MIN_LANES = POMDP_ROCK_PAIR_MIN_LANES, STOCHROCK_PAIR_MIN_LANES = POMDP_ROCK_PAIR_MIN_LANES;
#else
constexpr int64_t ROCK_PAIR_MIN_LANES = (3 << 17) + 1, STOCHROCK_PAIR_MIN_LANES = 3 << 17;
#endif
#ifdef POMDP_NO_STEP_LOOP_UNROLL4                              // the A arm: the time-shared one-lane-per-thread loops step by step
constexpr bool STEP_LOOP_UNROLL4 = false;
#else
constexpr bool STEP_LOOP_UNROLL4 = true;
#endif
#ifdef POMDP_NO_STEP_LOOP_UNROLL4_TAIL                         // the A arm: every step of a group of four behind its own bound check
constexpr bool STEP_LOOP_UNROLL4_TAIL = false;
#else
constexpr bool STEP_LOOP_UNROLL4_TAIL = true;
#endif
#ifdef POMDP_NO_TAPE_TWO_STEPS_AHEAD                           // the A arm: the tape's row of step s + 1 asked for at the top of step s
constexpr bool TAPE_TWO_STEPS_AHEAD = false;
#else
constexpr bool TAPE_TWO_STEPS_AHEAD = true;
#endif
#ifdef POMDP_NO_TAPE_SMALL_SHARD_LOOPS                         // the A arm: a tape below the quad gates always takes the general loop
constexpr bool TAPE_SMALL_SHARD_LOOPS = false;
#else
constexpr bool TAPE_SMALL_SHARD_LOOPS = true;
#endif
#ifdef POMDP_POLICY_AFTER_STEP                                 // the A arm: the policy's block drawn after the lane step (until round 6)
constexpr bool POLICY_WITH_STEP_BLOCKS = false;
#else
constexpr bool POLICY_WITH_STEP_BLOCKS = true;
#endif
// Tag (one opponent, table-driven) the same way (tag_steps_quad_kernel<.., 2>), every pair sink: ahead of the one-lane-per-thread
// loop from 3 * 2^17 lanes (0.91 -> 0.80 us per step of records; 7 * 2^16: 1.21 -> 0.92; 2^19, against the quad loop: 1.08 ->
// 0.92; returns 1.18 -> 1.03), level with the quad loop at 3 * 2^18 (profiles/r06_tag_pair_ab.txt).  Tiger's loop in this form
// is level with its quad loop at 2^19 (0.63) and behind below: not built.
#ifdef POMDP_TAG_PAIR_MAX_LANES
constexpr int64_t TAG_PAIR_MAX_LANES = POMDP_TAG_PAIR_MAX_LANES;
#else
constexpr int64_t TAG_PAIR_MAX_LANES = (3 << 18) - 1;
#endif
#ifdef POMDP_TAG_PAIR_MIN_LANES
constexpr int64_t TAG_PAIR_MIN_LANES = POMDP_TAG_PAIR_MIN_LANES;
#else
constexpr int64_t TAG_PAIR_MIN_LANES = 3 << 17;
#endif
#ifdef POMDP_BS_VIS_LDS_MAX_LANES                             // the visited mask in LDS up to this many lanes (48 B of LDS per lane)
constexpr int64_t BS_VIS_LDS_MAX_LANES = POMDP_BS_VIS_LDS_MAX_LANES;
#else
constexpr int64_t BS_VIS_LDS_MAX_LANES = 1 << 19;
#endif
#ifdef POMDP_BS_PAIR_MIN_LANES
constexpr int64_t BS_PAIR_MIN_LANES = POMDP_BS_PAIR_MIN_LANES;
#else
constexpr int64_t BS_PAIR_MIN_LANES = 1 << 17;
#endif

// steps per fused launch of the C-side drivers (pomdp_fuse_max; defined in api.hip).  One place instead of a constant per
// driver; a launch's fixed cost is paid once per this many steps, results never depend on it
extern std::atomic<int> g_fuse_max;
constexpr int FUSE_MAX_LIMIT = 256;                            // BattleShip's board pool keeps a lane's deal step in a byte
static inline int64_t fuse_max() { return (int64_t)g_fuse_max.load(std::memory_order_relaxed); }

// which kernel the calling thread's most recent fused launch picked (pomdp_last_fused_kernel: bench.py names the kernel
// it timed from this instead of guessing the launcher's choice; defined in api.hip)
extern thread_local char g_last_fused[96];
static inline void note_fused(const char *kernel, const char *env, const char *variant)
{
    snprintf(g_last_fused, sizeof g_last_fused, "%s<%s%s>", kernel, env, variant);
}

using StochRock1 = RockEnv<1, true>;   // StochasticRockEnv, one / two state words
using StochRock2 = RockEnv<2, true>;

static inline bool rock_ok(const pomdp_rock_params *p)
{
    if (!(p && p->size >= 1 && p->size <= 15 && p->num_rocks >= 1 && p->num_rocks <= 16 &&
          (unsigned)p->start_x < (unsigned)p->size && (unsigned)p->start_y < (unsigned)p->size))
        return false;
    for (int i = 0; i < p->num_rocks; ++i)     // rock coordinates index the LDS tables: keep them on the board
        if ((unsigned)p->rock_x[i] >= (unsigned)p->size || (unsigned)p->rock_y[i] >= (unsigned)p->size) return false;
    for (int i = 0; i < 256; ++i)
        if (p->grid[i] < -1 || p->grid[i] > 15) return false;
    return true;
}
static inline int bs_mask_words(const pomdp_battleship_params *p)
{
    if (!p || p->x_size < 1 || p->y_size < 1 || p->x_size > 16 || p->y_size > 16) return 0;
    const int cells = p->x_size * p->y_size;
    if (cells > 122 || p->max_len < 2 || p->max_len > 10) return 0;
    // a ship of length L needs L + 2 cells in a line (battleship.py:199-201): on a board where the longest ship
    // cannot be placed the reference's rejection loop never ends, and neither would the kernel's
    const int longest = p->x_size > p->y_size ? p->x_size : p->y_size;
    if (longest < p->max_len + 2) return 0;
    return (cells + 6 + 31) / 32;
}

// ---- the launchers: one function template per kind of launch, defined in step_impl.hip.h / fused_impl.hip.h and
// instantiated for its env types by exactly one translation unit each (the family files); everybody else sees the
// declarations below and links against them
template <class Env>
int launch_reset(const typename Env::Params &p, uint32_t *state, int32_t *ob, int64_t n, uint64_t seed, uint32_t lane0, uint64_t t,
                 void *stream);
template <class Env>
int launch_step(const typename Env::Params &p, uint32_t *state, const int32_t *action, int32_t *ob, typename Env::Reward *reward,
                uint8_t *done, uint32_t *err, int64_t n, uint64_t seed, uint32_t lane0, uint64_t t, int flags, void *stream);
template <class Env>
int launch_step_chain(const typename Env::Params &p, uint32_t *state, int32_t *action, int32_t *ob, typename Env::Reward *reward,
                      uint8_t *done, uint32_t *err, int64_t n, uint64_t seed, uint64_t action_seed, uint32_t lane0, uint64_t t,
                      int flags, void *stream);
template <class Env>
int launch_steps_fused(const typename Env::Params &p, uint32_t *state, int32_t *action, int32_t *ob, typename Env::Reward *reward,
                       uint8_t *done, uint32_t *err, int64_t n, uint64_t seed, uint64_t action_seed, uint32_t lane0, uint64_t t,
                       int k, int flags, int64_t rec, bool gen_first, int layout, TapeRef tape, void *stream);
constexpr TapeRef NO_TAPE = {nullptr, 0, nullptr};         // the synthetic policy

using Rock1 = RockEnv<1>;
using Rock2 = RockEnv<2>;
using BattleShip1 = BattleShipEnv<1>;
using BattleShip2 = BattleShipEnv<2>;
using BattleShip3 = BattleShipEnv<3>;
using BattleShip4 = BattleShipEnv<4>;
#define POMDP_STEP_LAUNCHERS(X, E)                                                                                          \
    X template int launch_reset<E>(const E::Params &, uint32_t *, int32_t *, int64_t, uint64_t, uint32_t, uint64_t, void *);  \
    X template int launch_step<E>(const E::Params &, uint32_t *, const int32_t *, int32_t *, E::Reward *, uint8_t *,          \
                                  uint32_t *, int64_t, uint64_t, uint32_t, uint64_t, int, void *);                           \
    X template int launch_step_chain<E>(const E::Params &, uint32_t *, int32_t *, int32_t *, E::Reward *, uint8_t *,          \
                                        uint32_t *, int64_t, uint64_t, uint64_t, uint32_t, uint64_t, int, void *);
#define POMDP_FUSED_LAUNCHER(X, E)                                                                                            \
    X template int launch_steps_fused<E>(const E::Params &, uint32_t *, int32_t *, int32_t *, E::Reward *, uint8_t *,          \
                                         uint32_t *, int64_t, uint64_t, uint64_t, uint32_t, uint64_t, int, int, int64_t,     \
                                         bool, int, TapeRef, void *);
#define POMDP_EACH_ENV(M, X)                                                                                                  \
    M(X, Rock1) M(X, Rock2) M(X, StochRock1) M(X, StochRock2) M(X, TagEnv) M(X, BattleShip1) M(X, BattleShip2)                  \
    M(X, BattleShip3) M(X, BattleShip4) M(X, TigerEnv) M(X, NetworkEnv)
#ifndef POMDP_NO_EXTERN_LAUNCHERS
POMDP_EACH_ENV(POMDP_STEP_LAUNCHERS, extern)
POMDP_EACH_ENV(POMDP_FUSED_LAUNCHER, extern)
#endif

} // namespace pomdp

using namespace pomdp;

// the env's action count from its params; 0 = unknown env
static inline uint32_t env_action_count(int env, const void *params)
{
    switch (env) {
    case POMDP_ENV_ROCK: return 5u + (uint32_t)((const pomdp_rock_params *)params)->num_rocks;
    case POMDP_ENV_TAG: return 5u;
    case POMDP_ENV_BATTLESHIP: {
        const pomdp_battleship_params *p = (const pomdp_battleship_params *)params;
        return (uint32_t)(p->x_size * p->y_size);
    }
    case POMDP_ENV_TIGER: return 3u;
    case POMDP_ENV_NETWORK: return 2u * (uint32_t)((const pomdp_network_params *)params)->n_machines + 1u;
    default: return 0u;
    }
}

// Resolve (env kind, params) to the env type the kernels are instantiated for, validate the params against what the
// packed layouts support, and call f(EnvTag<Env>{}, typed params).
template <class E> struct EnvTag { using Env = E; };
template <class F>
static inline int dispatch_env(int env, const void *params, F &&f)
{
    switch (env) {
    case POMDP_ENV_ROCK: {
        const pomdp_rock_params *p = (const pomdp_rock_params *)params;
        if (!rock_ok(p)) return POMDP_E_BADPARAMS;
        if (p->stochastic) return p->num_rocks <= 12 ? f(EnvTag<StochRock1>{}, *p) : f(EnvTag<StochRock2>{}, *p);
        return p->num_rocks <= 12 ? f(EnvTag<RockEnv<1>>{}, *p) : f(EnvTag<RockEnv<2>>{}, *p);
    }
    case POMDP_ENV_TAG: {
        const pomdp_tag_params *p = (const pomdp_tag_params *)params;
        if (p->num_opponents < 1 || p->num_opponents > 4) return POMDP_E_BADPARAMS;
        return f(EnvTag<TagEnv>{}, *p);
    }
    case POMDP_ENV_BATTLESHIP: {
        const pomdp_battleship_params *p = (const pomdp_battleship_params *)params;
        switch (bs_mask_words(p)) {
        case 1: return f(EnvTag<BattleShipEnv<1>>{}, *p);
        case 2: return f(EnvTag<BattleShipEnv<2>>{}, *p);
        case 3: return f(EnvTag<BattleShipEnv<3>>{}, *p);
        case 4: return f(EnvTag<BattleShipEnv<4>>{}, *p);
        default: return POMDP_E_BADPARAMS;
        }
    }
    case POMDP_ENV_TIGER: return f(EnvTag<TigerEnv>{}, *(const pomdp_tiger_params *)params);
    case POMDP_ENV_NETWORK: {
        const pomdp_network_params *p = (const pomdp_network_params *)params;
        if (p->n_machines < 1 || p->n_machines > 32) return POMDP_E_BADPARAMS;
        // Bernoulli thresholds are numerators of numpy's 53-bit doubles: k53 <= thr.  The fast step compares 32-bit high
        // words against (thr >> 26) << 5, which wraps for thr >= 2^53 (a probability of 1.0)
        if ((p->fail_thr | p->fail_nb_thr | p->obs_thr) >> 53) return POMDP_E_BADPARAMS;
        return f(EnvTag<NetworkEnv>{}, *p);
    }
    default: return POMDP_E_BADARG;
    }
}

