// pomdp_kernels.hip — gfx950 kernels + the C ABI of include/pomdp_hip.h.
//
// One wavefront lane advances one env instance.  State, action, ob, reward and done
// are struct-of-arrays columns in HBM, so every access of a wave is one coalesced
// 256-byte (int32) or 64-byte (done) segment.  Lookup tables (RockSample's rock-id
// grid, rock coordinates and sensor thresholds) are staged from the kernarg segment
// into LDS once per workgroup.  No MFMA: the path is integer / branch work, bounded
// by HBM traffic (21 B per RockSample step) and by Philox ALU throughput.
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC (see __graft_entry__.build()).
#include <hip/hip_runtime.h>
#include <chrono>

#include "../../include/pomdp_hip.h"
#include "envs.hip.h"
#include "philox.hip.h"
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
#ifdef POMDP_DEV_TIMELINE                                      // dev builds only (tools/ab_build.sh): per-workgroup phase stamps
__device__ uint64_t *g_timeline = nullptr;
#define TL(k) do { if (threadIdx.x == 0 && g_timeline) g_timeline[blockIdx.x * 8 + (k)] = wall_clock64(); } while (0)
#else
#define TL(k) do { } while (0)
#endif
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

// envs whose lanes carry the board of their next episode (BattleShip): `next` is loaded only where a lane may need it
template <class Env, class = void> struct has_next : std::false_type {};
template <class Env> struct has_next<Env, std::enable_if_t<Env::HAS_NEXT>> : std::true_type {};

// ---------------------------------------------------------------------------
// reset: every lane starts a fresh episode from stream RESET of (seed, lane, t)
// ---------------------------------------------------------------------------
template <class Env>
__global__ __launch_bounds__(BLOCK) void reset_kernel(const typename Env::Params p, uint32_t *__restrict__ state,
                                                      int32_t *__restrict__ ob, int64_t n, RngKey key, uint32_t lane0,
                                                      uint32_t *__restrict__ host_flag = nullptr, uint32_t flag_value = 0)
{
    __shared__ typename Env::Shared sh;
    Env::stage(sh, p, (int)threadIdx.x);
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) {
        typename Env::State st;
        const int o = Env::reset(sh, p, st, key, lane0 + (uint32_t)i);
        Env::store(st, state, n, i, true);
        if (ob) ob[i] = o;
        // scalar mode (n == 1, `ob` in pinned host memory): see step_kernel
        if (host_flag && i == 0) __hip_atomic_store(host_flag, flag_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

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
//   - the 16 * LPT sensor blocks of its quads (stream STEP),
//   - their 16 * LPT RESET blocks (a quad's four lanes take one word each: the statuses of all rocks of a fresh episode),
//   - for CHAIN launches the 16 * LPT policy blocks of the next call counter,
// i.e. 64 (CHAIN: 96) Philox blocks for 128 lane-steps, dealt out 64 per pass BEFORE the lane step: the kernel runs the
// passes right after issuing its HBM loads.  The lane step therefore runs WITHOUT its sensor draw (Env::step_pre) and
// the observation and the fresh episodes are completed here from the pooled words.  Tasks and results are exchanged
// through a wave-private LDS scratch; LDS operations of one wave complete in order, so no barrier is involved.  Low
// words (needed with probability 2^-27 per draw) are generated per lane on demand.
template <int W, int LPT, bool CHAIN>
struct Finisher<RockEnv<W, false>, LPT, CHAIN, typename std::enable_if<(LPT >= 2)>::type> {
    using Env = RockEnv<W, false>;
    using Aux = typename Env::Aux;
    static constexpr bool LOOP_BARRIER = false;              // every scratch array is wave-private
    static constexpr int NQ = 16 * LPT;                      // quads of the wave's 64 * LPT lanes: sensor blocks, reset blocks
    static constexpr int NA = CHAIN ? 16 * LPT : 0;          // policy blocks of the next call counter
    static constexpr int NT = 2 * NQ + NA;
    template <class RT>
    static __device__ __forceinline__ void lane_step(const typename Env::Shared &sh, const typename Env::Params &p,
                                                     typename Env::State &st, int a, const RngKey &, uint32_t, int &ob,
                                                     RT &rew, int &done, Aux &aux)
    {
        Env::step_pre(sh, p, st, a, rew, done, aux);
        ob = 0;
    }
    // the same from the (position, action) table of a multi-step launch (RockEnv::StepTab)
    template <class Tab, class RT>
    static __device__ __forceinline__ void lane_step_tab(const Tab &tab, typename Env::State &st, int a, int &ob, RT &rew,
                                                         int &done, Aux &aux)
    {
        Env::step_tab(tab, st, a, rew, done, aux);
        ob = 0;
    }
    // wave-private LDS scratch (one instance: function-local static of this accessor)
    static __device__ __forceinline__ uint32_t (&blk_lds())[BLOCK / 64][NT][4]
    {
        __shared__ uint32_t a[BLOCK / 64][NT][4];            // [0, NQ): sensor, [NQ, 2 NQ): reset, [2 NQ, NT): policy; (sub-batch, quad)
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
                // ONE Philox instance for the three task kinds: the counter words are per-lane selects
                const int kind = tid < NQ ? 0 : (tid < 2 * NQ ? 1 : 2);
                const int qt = tid - kind * NQ;                                // (sub-batch, quad) index
                const uint32_t quad = ((first0 + (uint32_t)(qt >> 4) * BLOCK) >> 2) + (uint32_t)(qt & 15);
                const uint32_t c1 = kind == 2 ? akey.t_lo : key.t_lo, c2 = kind == 2 ? akey.t_hi : key.t_hi;
                const uint32_t c3 = (uint32_t)(kind == 0 ? POMDP_STREAM_STEP : kind == 1 ? POMDP_STREAM_RESET : POMDP_STREAM_ACTION) << 24;
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
            Rw[j] = blk_lds()[wv][NQ + 16 * j + (me >> 2)][me & 3];
            P[j] = CHAIN ? blk_lds()[wv][2 * NQ + 16 * j + (me >> 2)][me & 3] : 0u;
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
// ONE task list per wave (128 lanes): for CHAIN launches the 32 policy blocks of the next call counter, one STEP block
// per lane whose opponent may flee, one RESET block per resetting lane — ~58 blocks for 128 lane-steps instead of 256
// (512 chained), dealt out 64 per pass through a wave-private LDS scratch like RockSample's.  The lane step runs
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
                const uint32_t c0 = is_act ? quad : src_lane;
                const uint32_t c1 = is_act ? akey.t_lo : key.t_lo, c2 = is_act ? akey.t_hi : key.t_hi;
                const uint32_t strm = is_act ? POMDP_STREAM_ACTION : (r < nfl ? POMDP_STREAM_STEP : POMDP_STREAM_RESET);
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
            if (aux[j].need) Env::flee(sh, p, st[j], aux[j], rb[j].x, rb[j].y, rb[j].z);
            if (fresh[j]) {
                if (!Env::reset_from_block(p, st[j], rb[j])) Env::reset(sh, p, st[j], key, lane[j]);   // rejections ran past the block
            }
            if (CHAIN) a_next[j] = (int)__umulhi(res_lds[wv][16 * j + (me >> 2)][me & 3], n_act);
        }
    }
};

// CHAIN (C-side rollout driver only): after stepping, action[i] is overwritten with the synthetic
// policy's action for call counter t + 1 (key `akey`), so the next launch finds its input ready and
// no separate policy kernel runs.
template <class Env, int LPT, bool CHAIN = false>
__global__ __launch_bounds__(BLOCK) void step_kernel(uint32_t *__restrict__ state,
                                                     typename std::conditional<CHAIN, int32_t, const int32_t>::type *__restrict__ action,
                                                     int32_t *__restrict__ ob, typename Env::Reward *__restrict__ reward,
                                                     uint8_t *__restrict__ done, uint32_t *__restrict__ err,
                                                     int64_t n, RngKey key, uint32_t lane0, int flags, RngKey akey,
                                                     const typename Env::Params p,   // pointers first: what a wave needs first
                                                     uint32_t *__restrict__ host_flag = nullptr, uint32_t flag_value = 0)
{
    __shared__ typename Env::Shared sh;
    TL(0);
    const bool auto_reset = flags & POMDP_AUTO_RESET;
    // Addressing: the workgroup's first lane is wave-uniform, so every column gets a per-workgroup base pointer in
    // SGPRs and a thread only ever adds a small 32-bit offset (rel < BLOCK * LPT) — `global_load/store v_off, s[base]`
    // with no per-access 64-bit VALU arithmetic, for any n up to the ABI's 2^32 lanes.
    const uint32_t wg0 = blockIdx.x * (uint32_t)(BLOCK * LPT);
    const uint32_t last = (uint32_t)((uint64_t)(n - 1) - wg0);        // offset of lane n-1 (the grid has no empty workgroup)
    auto *const action_w = action + wg0;
    uint32_t *const state_w = state + wg0;
    int32_t *const ob_w = ob + wg0;
    typename Env::Reward *const reward_w = reward + wg0;
    uint8_t *const done_w = done + wg0;
    uint32_t idx[LPT], rel[LPT];
    bool in_range[LPT], was_done[LPT];
    int a_raw[LPT];
    typename Env::State st[LPT];
#pragma unroll
    for (int j = 0; j < LPT; ++j) {
        rel[j] = threadIdx.x + (uint32_t)(j * BLOCK);
        idx[j] = wg0 + rel[j];
        in_range[j] = rel[j] <= last;
        const uint32_t rc = in_range[j] ? rel[j] : last;               // out-of-range threads read lane n-1
        __builtin_assume(rc < (uint32_t)(BLOCK * LPT));
        a_raw[j] = ld_stream(action_w + rc);
        Env::load(st[j], state_w, n, rc);
        was_done[j] = auto_reset ? false : (ld_stream(done_w + rc) != 0);   // frozen lane (the reference would assert)
    }
    using Fin = Finisher<Env, LPT, CHAIN>;
    if constexpr (Fin::HAS_PREPASS) {
        // table loads, then the Philox blocks that depend on lane ids only, then the first use of any load
        uint32_t gl[LPT];
#pragma unroll
        for (int j = 0; j < LPT; ++j) gl[j] = lane0 + idx[j];
        const auto staged = Env::stage_load(p, (int)threadIdx.x);
        Fin::prepass(key, gl, akey);
        Env::stage_store(sh, staged, (int)threadIdx.x);
    } else {
        Env::stage(sh, p, (int)threadIdx.x);
    }
    __syncthreads();
    TL(1);

    const int n_act = Env::n_actions(p);
    int o[LPT], d[LPT];
    typename Env::Reward r[LPT];
    typename Fin::Aux aux[LPT];
    bool live[LPT], valid[LPT];
#pragma unroll
    for (int j = 0; j < LPT; ++j) {
        valid[j] = (unsigned)a_raw[j] < (unsigned)n_act;
        live[j] = in_range[j] && valid[j] && !was_done[j];
        Fin::lane_step(sh, p, st[j], valid[j] ? a_raw[j] : 0, key, lane0 + idx[j], o[j], r[j], d[j], aux[j]);
        if (!live[j]) { r[j] = 0; d[j] = was_done[j]; }               // step result discarded unless live
    }
#ifdef POMDP_DEV_TIMELINE
    { int x = 0; for (int j = 0; j < LPT; ++j) x += d[j] + (int)r[j]; asm volatile("" :: "v"(x)); }
    TL(2);
#endif
    bool fresh[LPT];
    uint32_t glane[LPT];
    int a_next[LPT];
#pragma unroll
    for (int j = 0; j < LPT; ++j) {
        fresh[j] = live[j] && d[j] && auto_reset; glane[j] = lane0 + idx[j]; a_next[j] = 0;
        if constexpr (has_next<Env>::value) { if (fresh[j]) Env::load_next(st[j], state_w, n, rel[j]); }   // the cached board moves in
    }
    Fin::run(sh, p, st, fresh, key, glane, akey, (uint32_t)n_act, a_next, aux, o);
#ifdef POMDP_DEV_TIMELINE
    { int x = 0; for (int j = 0; j < LPT; ++j) x += o[j]; asm volatile("" :: "v"(x)); }
    TL(3);
#endif
#pragma unroll
    for (int j = 0; j < LPT; ++j) {
        if (!live[j]) o[j] = 0;
        if (CHAIN) { if (in_range[j]) st_stream(const_cast<int32_t *>(action_w) + rel[j], (int32_t)a_next[j]); }
        if (live[j]) Env::store(st[j], state_w, n, rel[j], fresh[j]);
        if (in_range[j]) {
            st_stream(ob_w + rel[j], (int32_t)o[j]);
            st_stream(reward_w + rel[j], r[j]);
            st_stream(done_w + rel[j], (uint8_t)d[j]);
            // the reference asserts on an out-of-range action; here the lane is left untouched and counted
            if (!valid[j] && !was_done[j] && err) atomicAdd(err, 1u);
        }
    }
    // scalar mode (n == 1, outputs in pinned host memory): lane 0 wrote everything the host reads; publish it with a
    // system-scope release so that the host can poll `host_flag` instead of waiting for the end-of-kernel signal
    if (host_flag && blockIdx.x == 0 && threadIdx.x == 0)
        __hip_atomic_store(host_flag, flag_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
#ifdef POMDP_DEV_TIMELINE
    TL(4);
    __builtin_amdgcn_s_waitcnt(0);
    TL(5);
#endif
}

// k consecutive chained steps in ONE launch: exactly the memory state k launches of step_kernel<Env, LPT, true> leave —
// every step's ob / reward / done / state / next action is computed and written — but a lane's state and action stay in
// registers from one step to the next (nothing is re-read) and there is one launch ramp per k steps instead of per
// step.  Possible because a lane's step t+1 depends only on its own step t and all cooperation (pooled Philox passes,
// cooperative resets) is wave- or workgroup-local: no grid-wide synchronisation is involved.
// SIMPLE: every thread's lanes exist (n is a multiple of the workgroup's BLOCK * LPT lanes) and done lanes auto-reset, so
// no lane is ever out of range or frozen, and the actions are the driver's own (always valid): the bookkeeping for those
// cases is compiled out.
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
    const uint32_t r0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s0, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]
    const uint32_t r1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s1, 0xB1, 0xF, 0xF, false);
    // column (e & 1) / 2 + (e & 1) of the rows (e & ~1, e | 1)
    const uint32_t p0 = b0 ? r0 : v.x, p1 = b0 ? v.y : r0, q0 = b0 ? r1 : v.z, q1 = b0 ? v.w : r1;
    const uint32_t u0 = b1 ? p0 : q0, u1 = b1 ? p1 : q1;
    const uint32_t w0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)u0, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
    const uint32_t w1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)u1, 0x4E, 0xF, 0xF, false);
    return make_uint4(b1 ? w0 : p0, b1 ? w1 : p1, b1 ? q0 : w0, b1 ? q1 : w1);
}
template <int J> static __device__ __forceinline__ uint32_t comp(const uint4 &v) { return J == 0 ? v.x : J == 1 ? v.y : J == 2 ? v.z : v.w; }

template <class Fin, class = void> struct quad_policy_of : std::false_type {};
template <class Fin> struct quad_policy_of<Fin, std::enable_if_t<Fin::QUAD_POLICY>> : std::true_type {};

struct NoTab {};
template <class Env, bool ON> struct step_tab_of { using type = NoTab; };
template <class Env> struct step_tab_of<Env, true> { using type = typename Env::StepTab; };

template <class Env, int LPT, bool SIMPLE, bool TAB = false>
__global__ __launch_bounds__(BLOCK) void steps_kernel(uint32_t *__restrict__ state, int32_t *__restrict__ action,
                                                      int32_t *__restrict__ ob, typename Env::Reward *__restrict__ reward,
                                                      uint8_t *__restrict__ done, uint32_t *__restrict__ err, int64_t n,
                                                      RngKey key0, uint32_t lane0, int flags, RngKey akey0, int k_steps,
                                                      int64_t rec, const typename Env::Params p)
{
    __shared__ typename Env::Shared sh;
    __shared__ typename step_tab_of<Env, TAB>::type tab;   // TAB: the lane step reads a (position, action) table built below
    static_assert(!TAB || SIMPLE, "the table-driven step serves the SIMPLE instantiation");
    const bool auto_reset = SIMPLE || (flags & POMDP_AUTO_RESET);
    const uint32_t wg0 = blockIdx.x * (uint32_t)(BLOCK * LPT);
    const uint32_t last = SIMPLE ? (uint32_t)(BLOCK * LPT - 1) : (uint32_t)((uint64_t)(n - 1) - wg0);
    // rec = 0: every step overwrites the same n-element outputs (what the per-step launches do); rec = row pitch in
    // elements: step s writes row s of ob / reward / done and row s + 1 of action (row s being the actions it took)
    int32_t *action_w = action + wg0;
    uint32_t *const state_w = state + wg0;
    int32_t *ob_w = ob + wg0;
    typename Env::Reward *reward_w = reward + wg0;
    uint8_t *done_w = done + wg0;
    uint32_t rel[LPT], glane[LPT];
    bool in_range[LPT], was_done[LPT], ever_fresh[LPT];
    int a_cur[LPT];
    typename Env::State st[LPT];
#pragma unroll
    for (int j = 0; j < LPT; ++j) {
        rel[j] = threadIdx.x + (uint32_t)(j * BLOCK);
        glane[j] = lane0 + wg0 + rel[j];
        ever_fresh[j] = false;
        in_range[j] = SIMPLE || rel[j] <= last;
        const uint32_t rc = in_range[j] ? rel[j] : last;
        __builtin_assume(rc < (uint32_t)(BLOCK * LPT));
        a_cur[j] = (flags & FLAG_GEN_FIRST) ? 0 : ld_stream(action_w + rc);
        Env::load(st[j], state_w, n, rc);
        if constexpr (has_next<Env>::value) Env::load_next(st[j], state_w, n, rc);   // once per launch, with the other words
        was_done[j] = auto_reset ? false : (ld_stream(done_w + rc) != 0);
    }
    using Fin = Finisher<Env, LPT, true>;
    constexpr bool quad_policy = quad_policy_of<Fin>::value;
    uint4 aq = make_uint4(0, 0, 0, 0), sq = make_uint4(0, 0, 0, 0), rq = make_uint4(0, 0, 0, 0);
    const int n_act = Env::n_actions(p);
    const uint64_t t0 = ((uint64_t)key0.t_hi << 32) | key0.t_lo, ta0 = ((uint64_t)akey0.t_hi << 32) | akey0.t_lo;
    if (flags & FLAG_GEN_FIRST) {                        // wave-uniform: the policy's actions of the first call counter
        RngKey fkey = akey0;
        fkey.t_lo = (uint32_t)(ta0 - 1ull); fkey.t_hi = (uint32_t)((ta0 - 1ull) >> 32);
#pragma unroll
        for (int j = 0; j < LPT; ++j) {
            a_cur[j] = synthetic_action(fkey, glane[j], (uint32_t)n_act);
            if (in_range[j]) st_stream(action_w + rel[j], (int32_t)a_cur[j]);
        }
    }
    action_w += rec;
    // Tables once, BEFORE the loop; the first pre-pass rides under the load latency.  The staging reads the kernarg-resident
    // tables with vector loads, and a loop that contains any load keeps the compiler from settling the loads above in the
    // loop's pre-header: it then waits on vmcnt(0) in EVERY iteration for a register that arrived long ago — and stores
    // count on that counter too (gfx9), so every step waited for the acknowledgement of the previous step's stores
    // (round 3: 0.70 -> 0.45 us per step of a lone wave).  The loop below has no load.
    if constexpr (Fin::HAS_PREPASS) {
        const auto staged = Env::stage_load(p, (int)threadIdx.x);
        Fin::prepass(key0, glane, akey0);
        Env::stage_store(sh, staged, (int)threadIdx.x);
    } else {
        Env::stage(sh, p, (int)threadIdx.x);
    }
    __syncthreads();
    if constexpr (TAB) {                                 // BLOCK threads = the 256 position bytes
        Env::build_tab(tab, sh, p, (int)threadIdx.x);
        __syncthreads();
    }
    for (int s = 0; s < k_steps; ++s) {
        RngKey key = key0, akey = akey0;
        key.t_lo = (uint32_t)(t0 + (uint64_t)s); key.t_hi = (uint32_t)((t0 + (uint64_t)s) >> 32);
        akey.t_lo = (uint32_t)(ta0 + (uint64_t)s); akey.t_hi = (uint32_t)((ta0 + (uint64_t)s) >> 32);
        if constexpr (Fin::HAS_PREPASS) { if (s > 0) Fin::prepass(key, glane, akey); }
        int o[LPT], d[LPT];
        typename Env::Reward r[LPT];
        typename Fin::Aux aux[LPT];
        bool live[LPT], valid[LPT], fresh[LPT];
        int a_next[LPT];
        typename Env::State before[LPT];
#pragma unroll
        for (int j = 0; j < LPT; ++j) {
            before[j] = st[j];
            valid[j] = SIMPLE || (unsigned)a_cur[j] < (unsigned)n_act;
            live[j] = SIMPLE || (in_range[j] && valid[j] && !was_done[j]);
            if constexpr (quad_policy && Env::QUAD_SENSOR) {
                // one lane per thread, the sensor block shared by the quad (RockSample shards below the pooled kernels' gates):
                // lane e computes the block of step s + e once per four steps and the words reach their lanes by the same
                // transpose as the policy's — one Philox block per lane per four steps instead of one per step
                if ((s & 3) == 0) {
                    const uint64_t te = t0 + (uint64_t)s + (uint64_t)(glane[0] & 3u);
                    RngKey ke = key0;
                    ke.t_lo = (uint32_t)te; ke.t_hi = (uint32_t)(te >> 32);
                    sq = quad_transpose4(Env::quad_block(ke, glane[0], 0u), glane[0] & 3u);
                    rq = quad_transpose4(Env::reset_block(ke, glane[0], 0u), glane[0] & 3u);   // the quad's RESET words likewise
                }
                const int sj = s & 3;                                            // wave-uniform selects
                const uint32_t H = sj == 0 ? sq.x : sj == 1 ? sq.y : sj == 2 ? sq.z : sq.w;
                if constexpr (TAB) Env::step_with_H_tab(sh, tab, st[j], a_cur[j], key, glane[j], H, o[j], r[j], d[j]);
                else Env::step_with_H(sh, p, st[j], valid[j] ? a_cur[j] : 0, key, glane[j], H, o[j], r[j], d[j]);
            }
            else if constexpr (TAB) Fin::lane_step_tab(tab, st[j], a_cur[j], o[j], r[j], d[j], aux[j]);
            else Fin::lane_step(sh, p, st[j], valid[j] ? a_cur[j] : 0, key, glane[j], o[j], r[j], d[j], aux[j]);
            if (!live[j]) { r[j] = 0; d[j] = was_done[j]; }
            fresh[j] = live[j] && d[j] && auto_reset;
            a_next[j] = 0;
        }
        if constexpr (quad_policy) {
            const uint32_t e = glane[0] & 3u;
            if ((s & 3) == 0) {                                              // this lane's block: the policy of step s + e ...
                const uint64_t te = ta0 + (uint64_t)s + (uint64_t)e;
                // ... transposed within the quad: component J is then THIS lane's word of step s + J
                aq = quad_transpose4(philox4x32_10(glane[0] >> 2, (uint32_t)te, (uint32_t)(te >> 32),
                                                   (uint32_t)POMDP_STREAM_ACTION << 24, akey0.k0, akey0.k1), e);
            }
            const int sj = s & 3;                                            // wave-uniform selects
            if constexpr (Env::QUAD_SENSOR) {                                // RockSample: this lane's RESET word of step s
                const uint32_t rword = sj == 0 ? rq.x : sj == 1 ? rq.y : sj == 2 ? rq.z : rq.w;
                st[0].s = fresh[0] ? Env::fresh_state(p, rword, key, glane[0]) : st[0].s;
            } else {
                Fin::resets_only(sh, p, st, fresh, key, glane);
            }
            const uint32_t word = sj == 0 ? aq.x : sj == 1 ? aq.y : sj == 2 ? aq.z : aq.w;
            a_next[0] = (int)__umulhi(word, (uint32_t)n_act);
        } else {
            Fin::run(sh, p, st, fresh, key, glane, akey, (uint32_t)n_act, a_next, aux, o);
        }
#pragma unroll
        for (int j = 0; j < LPT; ++j) {
            if (!live[j]) { o[j] = 0; st[j] = before[j]; }                  // a lane that did not step keeps its state
            if (in_range[j]) st_stream(action_w + rel[j], (int32_t)a_next[j]);
            ever_fresh[j] |= fresh[j];
            if (in_range[j]) {
                st_stream(ob_w + rel[j], (int32_t)o[j]);
                st_stream(reward_w + rel[j], r[j]);
                st_stream(done_w + rel[j], (uint8_t)d[j]);
                if (!valid[j] && !was_done[j] && err) atomicAdd(err, 1u);
            }
            a_cur[j] = a_next[j];
            was_done[j] = auto_reset ? false : (d[j] != 0);
        }
        action_w += rec; ob_w += rec; reward_w += rec; done_w += rec;
        if constexpr (Fin::LOOP_BARRIER && !quad_policy) __syncthreads();
    }
    // the state is the loop's carry: it lived in registers and reaches memory once (a lane that never stepped writes back
    // what it read; BattleShip's ship words only if some step of the launch dealt a new board)
#pragma unroll
    for (int j = 0; j < LPT; ++j)
        if (in_range[j]) Env::store(st[j], state_w, n, rel[j], ever_fresh[j]);
}

// ---------------------------------------------------------------------------
// planner hooks: _generate_legal and random rollouts
// ---------------------------------------------------------------------------
template <class Env>
__global__ __launch_bounds__(BLOCK) void legal_kernel(const typename Env::Params p, const uint32_t *__restrict__ state,
                                                      int32_t *__restrict__ list, int32_t *__restrict__ len, int64_t n,
                                                      int stride)
{
    __shared__ typename Env::Shared sh;
    Env::stage(sh, p, (int)threadIdx.x);
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    typename Env::State st;
    Env::load(st, state, n, i);
    const int c = Env::legal_count(sh, p, st);
    len[i] = c;
    for (int k = 0; k < stride; ++k) list[i * stride + k] = k < c ? Env::legal_nth(sh, p, st, k) : -1;
}

template <class Env>
__global__ __launch_bounds__(BLOCK) void prob_kernel(const typename Env::Params p, const uint32_t *__restrict__ state,
                                                     const int32_t *__restrict__ action, const int32_t *__restrict__ ob,
                                                     double *__restrict__ out, int64_t n)
{
    __shared__ typename Env::Shared sh;
    Env::stage(sh, p, (int)threadIdx.x);
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    typename Env::State st;
    Env::load(st, state, n, i);
    const int a = action[i];
    out[i] = (unsigned)a < (unsigned)Env::n_actions(p) ? Env::compute_prob(sh, p, st, a, ob[i]) : 0.0;
}

// ---------------------------------------------------------------------------
// heuristic-policy support: side statistics, history sums, _generate_preferred
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void belief_reset_kernel(pomdp_rock_belief b, int K, const uint8_t *__restrict__ where,
                                                             int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n || (where && !where[i])) return;
    for (int j = 0; j < K; ++j) {                                              // rock.py:81-86
        const int64_t k = (int64_t)j * n + i;
        b.count[k] = 0; b.measured[k] = 0; b.lkv[k] = 1.; b.lkw[k] = 1.; b.prob_valuable[k] = .5;
    }
    b.check_ok[i] = (1u << K) - 1u;                                            // a fresh rock passes the test of rock.py:371
}

__global__ __launch_bounds__(BLOCK) void belief_refresh_kernel(pomdp_rock_belief b, int K, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    uint32_t m = 0;
    for (int j = 0; j < K; ++j) {
        const int64_t k = (int64_t)j * n + i;
        m |= (uint32_t)RockEnv<1>::check_ok(b.measured[k], b.count[k], b.prob_valuable[k]) << j;
    }
    b.check_ok[i] = m;
}

template <class Env>
__global__ __launch_bounds__(BLOCK) void belief_update_kernel(const typename Env::Params p, const uint32_t *__restrict__ state,
                                                              const int32_t *__restrict__ action, const int32_t *__restrict__ ob,
                                                              const uint8_t *__restrict__ done, pomdp_rock_belief b, int64_t n,
                                                              int auto_reset)
{
    __shared__ typename Env::Shared sh;
    Env::stage(sh, p, (int)threadIdx.x);
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    if (done[i]) {
        if (auto_reset) {
            for (int j = 0; j < p.num_rocks; ++j) {
                const int64_t k = (int64_t)j * n + i;
                b.count[k] = 0; b.measured[k] = 0; b.lkv[k] = 1.; b.lkw[k] = 1.; b.prob_valuable[k] = .5;
            }
            b.check_ok[i] = (1u << p.num_rocks) - 1u;
        }
        return;
    }
    const int a = action[i], o = ob[i];
    if (a <= 4 || a >= 5 + p.num_rocks || o == 0) return;                      // not an executed CHECK
    typename Env::State st;
    Env::load(st, state, n, (uint32_t)i);
    uint32_t ck = b.check_ok[i];
    Env::belief_update(sh, p, st, a, o, b, n, (uint32_t)i, ck);
    b.check_ok[i] = ck;
}

template <class Env>
__global__ __launch_bounds__(BLOCK) void select_target_kernel(const typename Env::Params p, const uint32_t *__restrict__ state,
                                                              pomdp_rock_belief b, int32_t *__restrict__ target, int64_t n)
{
    __shared__ typename Env::Shared sh;
    Env::stage(sh, p, (int)threadIdx.x);
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    typename Env::State st;
    Env::load(st, state, n, (uint32_t)i);
    target[i] = Env::select_target(sh, p, st, b, n, (uint32_t)i);
}

// the two sums over CHECK-j transitions (rock.py:303-310, 327-334) and the derived bits j and 16 + j of move_ok: the contribution of
// one transition (action CHECK j, next observation, observation before it) is added (sign = 1) or, when a bounded history
// drops the transition, taken out again (sign = -1)
static __device__ __forceinline__ void history_check_sums(const pomdp_history &h, int j, int next_ob, bool prev_bad, int64_t n,
                                                          uint32_t i, uint32_t &mv, int sign = 1)   // mv: the caller's copy of h.move_ok[i]
{
    const int64_t k = (int64_t)j * n + i;
    const int ds = sign * ((next_ob == 2) - (next_ob == 1));
    const int dm = sign * (next_ob == 2 ? 1 : (prev_bad ? -1 : 0));
    if (ds) {                                                                  // bit 16 + j: total_sample[j] > 0 (rock.py:311)
        const int ts = h.total_sample[k] + ds;
        h.total_sample[k] = ts;
        const uint32_t sbit = 0x10000u << j;
        mv = ts > 0 ? (mv | sbit) : (mv & ~sbit);
    }
    if (dm) {
        const int tm = h.total_move[k] + dm;
        h.total_move[k] = tm;
        const uint32_t bit = 1u << j;
        mv = tm >= 0 ? (mv | bit) : (mv & ~bit);
    }
}

// history.append(transition) of rock.py:541-544 on the lane's words: `size` (the list length), the window of a bounded
// history (max_size >= 0: one byte per kept transition — action | next_ob << 5 | (observation == BAD) << 7 — in a ring of
// max_size + 1 rows, `head` = the row the next transition goes to, which holds the OLDEST one once the ring is full:
// the reference pops element 0 when size > max_size and then appends, so the list settles at max_size + 1 records) and,
// for RockSample (K > 0), the two per-rock sums kept current as transitions enter and leave the window.
// RING = false: the caller knows there is no window to keep (an unbounded history, or an env without rocks) — the
// heuristic loop is instantiated both ways so that the unbounded history does not carry the window's registers and branches.
template <bool RING = true>
static __device__ __forceinline__ void history_push(const pomdp_history &h, int K, int a, int next_ob, int prev_ob, int64_t n,
                                                    uint32_t i, int &hsize, int &head, uint32_t &mv)
{
    const int W = h.max_size + 1;                                              // 0: unbounded
    if (RING && W > 0 && K > 0) {
        uint8_t *slot = h.ring + (int64_t)head * n + i;
        if (hsize == W) {                                                      // self._history.pop(0)
            const uint32_t old = *slot;
            const int oa = (int)(old & 31u);
            if (oa >= 5 && oa < 5 + K) history_check_sums(h, oa - 5, (int)((old >> 5) & 3u), (old >> 7) != 0u, n, i, mv, -1);
        }
        *slot = (uint8_t)((uint32_t)a | ((uint32_t)next_ob << 5) | ((prev_ob == 1) ? 128u : 0u));
        head = head + 1 == W ? 0 : head + 1;
    }
    hsize = (W > 0 && hsize == W) ? W : hsize + 1;
    if (a >= 5 && a < 5 + K) history_check_sums(h, a - 5, next_ob, prev_ob == 1, n, i, mv);
}

__global__ __launch_bounds__(BLOCK) void history_clear_kernel(pomdp_history h, int K, const uint8_t *__restrict__ where, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n || (where && !where[i])) return;
    h.size[i] = 0; h.last_action[i] = -1; h.last_ob[i] = -1;
    for (int j = 0; j < K; ++j) { h.total_sample[(int64_t)j * n + i] = 0; h.total_move[(int64_t)j * n + i] = 0; }
    if (K) h.move_ok[i] = (1u << K) - 1u;
    if (h.head) h.head[i] = 0;
}

// rock.py:541-544 History.append + the sums _generate_preferred takes over the records (rock.py:303-310, 327-334)
__global__ __launch_bounds__(BLOCK) void history_append_kernel(pomdp_history h, int K, const int32_t *__restrict__ observation,
                                                               const int32_t *__restrict__ action,
                                                               const int32_t *__restrict__ next_observation,
                                                               const uint8_t *__restrict__ done, int64_t n, int auto_reset)
{
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    if (done[i] && auto_reset) {                                               // next episode: a new, empty History
        h.size[i] = 0; h.last_action[i] = -1; h.last_ob[i] = -1;
        for (int j = 0; j < K; ++j) { h.total_sample[(int64_t)j * n + i] = 0; h.total_move[(int64_t)j * n + i] = 0; }
        if (K) h.move_ok[i] = (1u << K) - 1u;
        if (h.head) h.head[i] = 0;
        return;
    }
    const int a = action[i], o = next_observation[i];
    int hsize = h.size[i], head = h.head ? h.head[i] : 0;
    uint32_t mv = K ? h.move_ok[i] : 0u;
    history_push(h, K, a, o, observation[i], n, (uint32_t)i, hsize, head, mv);
    h.size[i] = hsize; h.last_action[i] = a; h.last_ob[i] = o;
    if (K) h.move_ok[i] = mv;
    if (h.head) h.head[i] = head;
}

// envs whose _generate_preferred reads extra LDS tables fill them with Env::stage_policy
template <class Env, class = void> struct HasPolicyTables : std::false_type {};
template <class Env> struct HasPolicyTables<Env, std::void_t<decltype(&Env::stage_policy)>> : std::true_type {};
template <class Env>
static __device__ __forceinline__ void stage_policy_tables(typename Env::Shared &sh, const typename Env::Params &p)
{
    if constexpr (HasPolicyTables<Env>::value) Env::stage_policy(sh, p, (int)threadIdx.x);
}

template <class Env>
__global__ __launch_bounds__(BLOCK) void preferred_kernel(const typename Env::Params p, const uint32_t *__restrict__ state,
                                                          pomdp_rock_belief b, pomdp_history h, int32_t *__restrict__ list,
                                                          int32_t *__restrict__ len, int64_t n, int stride)
{
    __shared__ typename Env::Shared sh;
    Env::stage(sh, p, (int)threadIdx.x);
    stage_policy_tables<Env>(sh, p);
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    typename Env::State st;
    Env::load(st, state, n, (uint32_t)i);
    uint32_t m = Env::preferred_mask(sh, p, st, b, h, n, (uint32_t)i);
    if (m) {                                                                   // ascending action order
        const int c = __popc(m);
        len[i] = c;
        for (int k = 0; k < stride; ++k) {
            int a = -1;
            if (k < c) { a = __ffs((int)m) - 1; m &= m - 1u; }
            list[i * stride + k] = a;
        }
    } else {                                                                   // _generate_legal()
        const int c = Env::legal_count(sh, p, st);
        len[i] = c;
        for (int k = 0; k < stride; ++k) list[i * stride + k] = k < c ? Env::legal_nth(sh, p, st, k) : -1;
    }
}

// the caller's np.random.choice(list): the synthetic policy's word of the lane picks the element
__global__ __launch_bounds__(BLOCK) void pick_actions_kernel(const int32_t *__restrict__ list, const int32_t *__restrict__ len,
                                                             int stride, int32_t *__restrict__ action, int64_t n, RngKey key,
                                                             uint32_t lane0)
{
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    const uint32_t lane = lane0 + (uint32_t)i, e = lane & 3u;
    const uint4 w = stream_block(key, lane >> 2, POMDP_STREAM_ACTION, 0u);
    const uint32_t word = e == 0 ? w.x : e == 1 ? w.y : e == 2 ? w.z : w.w;
    const int c = len[i];
    action[i] = c > 0 ? list[i * stride + (int64_t)__umulhi(word, (uint32_t)c)] : -1;
}

// RockSample envs maintain side statistics; the other envs have none
template <class Env, class = void>
struct BeliefOps {
    static __device__ __forceinline__ void update(const typename Env::Shared &, const typename Env::Params &,
                                                  const typename Env::State &, int, int, const pomdp_rock_belief &, int64_t,
                                                  uint32_t, uint32_t &) {}
};
template <int W, bool STOCH>
struct BeliefOps<RockEnv<W, STOCH>, void> {
    using Env = RockEnv<W, STOCH>;
    static __device__ __forceinline__ void update(const typename Env::Shared &sh, const typename Env::Params &p,
                                                  const typename Env::State &st, int a, int o, const pomdp_rock_belief &b,
                                                  int64_t n, uint32_t i, uint32_t &ck) { Env::belief_update(sh, p, st, a, o, b, n, i, ck); }
};
template <class Env>
static __device__ __forceinline__ void heuristic_belief_update(const typename Env::Shared &sh, const typename Env::Params &p,
                                                               const typename Env::State &st, int a, int o,
                                                               const pomdp_rock_belief &b, int64_t n, uint32_t i, uint32_t &ck)
{
    BeliefOps<Env>::update(sh, p, st, a, o, b, n, i, ck);
}


// _generate_legal() as the rollout loop uses it — the list's length, then its idx-th entry: envs that derive both from one
// intermediate form (Env::Legal, Env::legal_set, Env::legal_pick) compute it once per step, the others go through
// legal_count / legal_nth
template <class Env, class = void> struct LegalOf {
    struct Set { int count; };
    static __device__ __forceinline__ Set make(const typename Env::Shared &sh, const typename Env::Params &p,
                                               const typename Env::State &st, bool skip)
    {
        return Set{skip ? 0 : Env::legal_count(sh, p, st)};
    }
    static __device__ __forceinline__ int pick(const typename Env::Shared &sh, const typename Env::Params &p,
                                               const typename Env::State &st, const Set &, int idx)
    {
        return Env::legal_nth(sh, p, st, idx);
    }
};
template <class Env> struct LegalOf<Env, std::void_t<typename Env::Legal>> {
    using Set = typename Env::Legal;
    static __device__ __forceinline__ Set make(const typename Env::Shared &sh, const typename Env::Params &p,
                                               const typename Env::State &st, bool skip)
    {
        if (skip) return Set{};
        return Env::legal_set(sh, p, st);
    }
    static __device__ __forceinline__ int pick(const typename Env::Shared &sh, const typename Env::Params &,
                                               const typename Env::State &, const Set &L, int idx)
    {
        return Env::legal_pick(sh, L, idx);
    }
};

// k heuristic-policy steps in one launch: per step choice(_generate_preferred(history)) -> step -> side statistics ->
// history.append, i.e. preferred_kernel + pick_actions_kernel + step_kernel + belief_update_kernel +
// history_append_kernel on the same call counter, with the lists never leaving registers.  Across the k steps a lane's
// state, its history words (size, last action / observation, prev_ob), the two derived words and the running return stay
// in registers and are written back once; the per-rock arrays are read and written in place when a CHECK touches them;
// every step's action / ob / reward / done (and state) is written as the single-step launches write them.
template <class Env, bool RING>   // RING: a bounded RockSample history (history_push keeps its window)
__global__ __launch_bounds__(BLOCK) void heuristic_steps_kernel(const typename Env::Params p, uint32_t *__restrict__ state,
                                                                pomdp_rock_belief b, pomdp_history h, int K, pomdp_returns R,
                                                                int32_t *__restrict__ prev_ob, int32_t *__restrict__ action,
                                                                int32_t *__restrict__ ob, typename Env::Reward *__restrict__ reward,
                                                                uint8_t *__restrict__ done, int64_t n, RngKey key0, uint32_t lane0,
                                                                int flags, int k_steps)
{
#pragma clang fp contract(off)
    __shared__ typename Env::Shared sh;
    const bool auto_reset = flags & POMDP_AUTO_RESET;
    const uint32_t idx = blockIdx.x * (uint32_t)BLOCK + threadIdx.x;
    const bool in_range = (uint64_t)idx < (uint64_t)n;
    const uint32_t i = in_range ? idx : (uint32_t)(n - 1);
    const uint32_t lane = lane0 + i;
    // every per-lane word first (one memory latency), then the tables
    typename Env::State st;
    Env::load(st, state, n, i);
    if constexpr (has_next<Env>::value) Env::load_next(st, state, n, i);
    int hsize = ld_stream(h.size + i), pob = ld_stream(prev_ob + i), head = RING ? ld_stream(h.head + i) : 0;
    int la = ld_stream(h.last_action + i), lo = ld_stream(h.last_ob + i);
    uint32_t ck = K ? ld_stream(b.check_ok + i) : 0u, mv = K ? ld_stream(h.move_ok + i) : 0u;
    bool was_done = auto_reset ? false : (ld_stream(done + i) != 0);
    double ret = R.ret ? R.ret[i] : 0.0, disc = R.ret ? R.disc[i] : 1.0;
    Env::stage(sh, p, (int)threadIdx.x);
    stage_policy_tables<Env>(sh, p);
    __syncthreads();
    bool ever_fresh = false;
    const int hcap = h.max_size >= 0 ? h.max_size + 1 : 0x7FFFFFFF;            // len(history) stops there (rock.py:541-544)
    const uint64_t t0 = ((uint64_t)key0.t_hi << 32) | key0.t_lo;
    const uint32_t e = lane & 3u;
    // Random words four steps at a time, as in the rollout kernel: the policy's ACTION block is shared by the four lanes
    // of a quad (and so is RockSample's STEP block), so lane e of a quad computes the block(s) of step base + e and the
    // words travel by DPP quad-broadcast — one block per lane per four steps instead of four.
    for (int base = 0; base < k_steps; base += 4) {
        const uint64_t te = t0 + (uint64_t)base + (uint64_t)e;
        RngKey ke = key0;
        ke.t_lo = (uint32_t)te; ke.t_hi = (uint32_t)(te >> 32);
        const uint4 aq = quad_transpose4(stream_block(ke, lane >> 2, POMDP_STREAM_ACTION, 0u), e);   // .J: this lane's word of step base + J
        uint4 sq = make_uint4(0, 0, 0, 0);
        if constexpr (Env::QUAD_SENSOR) sq = quad_transpose4(Env::quad_block(ke, lane, 0u), e);
        auto one_step = [&](auto jc) {
            constexpr int J = decltype(jc)::value;
            const int s = base + J;
            if (s >= k_steps) return;                                          // wave-uniform
            RngKey key = key0;
            key.t_lo = (uint32_t)(t0 + (uint64_t)s); key.t_hi = (uint32_t)((t0 + (uint64_t)s) >> 32);
            // the policy: a = list[(w * len(list)) >> 32] over the preferred list (ascending mask order) or the legal list
            const uint32_t word = comp<J>(aq);
            const uint32_t m = Env::preferred_mask(sh, p, st, h, n, i, ck, mv, hsize, la, lo);
            int a;
            if (m) a = nth_set_bit(m, (int)__umulhi(word, (uint32_t)__popc(m)));
            else {
                const auto L = LegalOf<Env>::make(sh, p, st, false);
                a = LegalOf<Env>::pick(sh, p, st, L, (int)__umulhi(word, (uint32_t)L.count));
            }
            const bool live = in_range && !was_done;
            const typename Env::State before = st;
            int o, d;
            typename Env::Reward r;
            if constexpr (Env::QUAD_SENSOR) {
                Env::step_with_H(sh, p, st, a, key, lane, comp<J>(sq), o, r, d);
            } else {
                Env::step(sh, p, st, a, key, lane, o, r, d);
            }
            if (!live) { o = 0; r = 0; d = was_done; st = before; }
            const bool fresh = live && d && auto_reset;
            Env::reset_where(sh, p, st, fresh, key, lane);                     // wave-cooperative: every lane calls it
            ever_fresh |= fresh;
            if (live) {
                if (R.ret) {                                                   // r += rw * discount; discount *= _discount
                    const double term = disc * (double)r;
                    const double acc = ret + term;
                    if (d) R.ret_done[i] = acc;
                    ret = fresh ? 0.0 : acc;
                    disc = fresh ? 1.0 : disc * R.discount;
                }
                if (fresh) {                                                   // new episode: fresh Rock objects, empty History
                    for (int j = 0; j < K; ++j) {
                        const int64_t k = (int64_t)j * n + i;
                        b.count[k] = 0; b.measured[k] = 0; b.lkv[k] = 1.; b.lkw[k] = 1.; b.prob_valuable[k] = .5;
                        h.total_sample[k] = 0; h.total_move[k] = 0;
                    }
                    ck = mv = K ? (1u << K) - 1u : 0u;
                    hsize = 0; la = -1; lo = -1; head = 0;
                    pob = Env::reset_ob(p, st);
                } else {
                    la = a; lo = o;                                            // a terminal transition is recorded too
                    if constexpr (RING) {
                        history_push<true>(h, K, a, o, pob, n, i, hsize, head, mv);
                        if (a >= 5 && a < 5 + K && o != 0 && !d) heuristic_belief_update<Env>(sh, p, st, a, o, b, n, i, ck);
                    } else {                                                   // no window: history_push<false>, one CHECK branch
                        hsize += (int)(hsize != hcap);
                        if (a >= 5 && a < 5 + K) {                             // K > 0: RockSample CHECK
                            history_check_sums(h, a - 5, o, pob == 1, n, i, mv);
                            if (o != 0 && !d) heuristic_belief_update<Env>(sh, p, st, a, o, b, n, i, ck);
                        }
                    }
                    pob = o;
                }
            }
            // the step's outputs leave LAST: the per-rock sums and side statistics above are read-modify-writes, and a load
            // waits for every store issued before it (one counter for both on gfx9) — behind these four it waited for their
            // acknowledgements every step
            if (in_range) {
                st_stream(action + i, (int32_t)(live ? a : -1));
                st_stream(ob + i, (int32_t)o);
                st_stream(reward + i, r);
                st_stream(done + i, (uint8_t)d);
            }
            if (live) was_done = auto_reset ? false : (d != 0);
        };
        one_step(std::integral_constant<int, 0>{});
        one_step(std::integral_constant<int, 1>{});
        one_step(std::integral_constant<int, 2>{});
        one_step(std::integral_constant<int, 3>{});
    }
    if (!in_range) return;
    Env::store(st, state, n, i, ever_fresh);                                   // the loop's carry, written once
    st_stream(h.size + i, (int32_t)hsize); st_stream(h.last_action + i, (int32_t)la); st_stream(h.last_ob + i, (int32_t)lo);
    st_stream(prev_ob + i, (int32_t)pob);
    if (K) { st_stream(b.check_ok + i, ck); st_stream(h.move_ok + i, mv); }
    if (RING) st_stream(h.head + i, (int32_t)head);                            // without a window `head` never moves
    if (R.ret) { R.ret[i] = ret; R.disc[i] = disc; }
}

// RockSample's rollouts read the lane step from the (position, action) table of the fused loops, built once per launch
// (2.61 -> 2.73e11 steps/s on (15,15), 2.72 -> 2.80e11 on (7,8))
template <class Env, class = void> struct ROLLOUT_TAB : std::false_type {};
template <class Env> struct ROLLOUT_TAB<Env, typename std::enable_if<Env::QUAD_SENSOR && Env::QUAD_TAB>::type> : std::true_type {};

// Lane i simulates from root state column i / sims_per_root for up to `depth` steps: the state lives in registers
// and nothing is written but the per-lane results.  Random words, four steps at a time:
//   - the policy pick of step k is word k of the lane's ROLLOUT stream at t0: one Philox block per four steps;
//   - the env draws come from stream STEP at t0 + k, as in step().  RockSample's STEP block is shared by the four
//     lanes of a quad, so lane e of a quad computes the block of step 4 g + e and the words travel by DPP
//     quad-broadcast: one block per lane per four steps instead of four.
// The discounted return accumulates in IEEE double with separate multiply and add (so a CPU restatement reproduces
// it bit-for-bit).
template <class Env>
__global__ __launch_bounds__(BLOCK) void rollout_kernel(const typename Env::Params p, const uint32_t *__restrict__ state,
                                                        int64_t n_roots, int64_t sims_per_root, int depth,
                                                        double discount, int all_actions, RngKey key0, uint32_t lane0,
                                                        double *__restrict__ ret, int32_t *__restrict__ n_steps,
                                                        int32_t *__restrict__ first_action, int32_t *__restrict__ last_ob,
                                                        uint8_t *__restrict__ terminated)
{
#pragma clang fp contract(off) // the discounted return must not be fused into FMAs (hipcc defaults to contract=fast)
    __shared__ typename Env::Shared sh;
    constexpr bool TAB = ROLLOUT_TAB<Env>::value;            // RockSample: the (position, action) table of the fused loops
    __shared__ typename step_tab_of<Env, TAB>::type tab;
    Env::stage(sh, p, (int)threadIdx.x);
    __syncthreads();
    if constexpr (TAB) {
        Env::build_tab(tab, sh, p, (int)threadIdx.x);
        __syncthreads();
    }
    const int64_t n = n_roots * sims_per_root;
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const bool in_range = i < n;
    const int64_t ic = in_range ? i : n - 1;
    typename Env::State st;
    Env::load(st, state, n_roots, (uint32_t)(ic / sims_per_root));
    const uint32_t lane = lane0 + (uint32_t)i;
    const int n_act = Env::n_actions(p);
    double acc = 0.0, disc = 1.0;
    int k = 0, d = 0, o = 0, first = -1;
    bool active = in_range, live_wave = true;
    const uint64_t t0 = ((uint64_t)key0.t_hi << 32) | key0.t_lo;
    for (int base = 0; base < depth && live_wave; base += 4) {
        const uint4 pw = stream_block(key0, lane, POMDP_STREAM_ROLLOUT, (uint32_t)(base >> 2));
        uint4 sq = make_uint4(0, 0, 0, 0);
        if constexpr (Env::QUAD_SENSOR) {            // this lane's share: the quad's STEP block of step base + (lane & 3)
            const uint64_t te = t0 + (uint64_t)base + (uint64_t)(lane & 3u);
            RngKey ke = key0;
            ke.t_lo = (uint32_t)te; ke.t_hi = (uint32_t)(te >> 32);
            sq = quad_transpose4(Env::quad_block(ke, lane, 0u), lane & 3u);   // .J: this lane's word of step base + J
        }
        auto one_step = [&](auto jc) {
            constexpr int J = decltype(jc)::value;
            const int step = base + J;
            if (step >= depth || !live_wave) return;
            const auto L = LegalOf<Env>::make(sh, p, st, all_actions != 0);
            const int count = all_actions ? n_act : L.count;
            active = active && !d && count > 0;
            if (!__any(active)) { live_wave = false; return; }           // wave-uniform exit
            const uint64_t t = t0 + (uint64_t)step;
            RngKey key = key0;
            key.t_lo = (uint32_t)t; key.t_hi = (uint32_t)(t >> 32);
            const uint32_t w = J == 0 ? pw.x : J == 1 ? pw.y : J == 2 ? pw.z : pw.w;
            const int idx = (int)__umulhi(w, (uint32_t)(count > 0 ? count : 1));
            const int a = all_actions ? idx : LegalOf<Env>::pick(sh, p, st, L, idx);
            typename Env::State nx = st;
            int o2, d2;
            double r;
            if constexpr (Env::QUAD_SENSOR) {      // every lane runs it (the broadcasts need the whole quad); inactive lanes discard
                if constexpr (TAB) Env::step_with_H_tab(sh, tab, nx, a, key, lane, comp<J>(sq), o2, r, d2);
                else Env::step_with_H(sh, p, nx, a, key, lane, comp<J>(sq), o2, r, d2);
            } else {
                Env::step(sh, p, nx, a, key, lane, o2, r, d2);
            }
            if (active) {
                st = nx; o = o2; d = d2;
                if (step == 0) first = a;
                const double term = disc * r;
                acc = acc + term;
                disc = disc * discount;
                k = step + 1;
            }
        };
        one_step(std::integral_constant<int, 0>{});
        one_step(std::integral_constant<int, 1>{});
        one_step(std::integral_constant<int, 2>{});
        one_step(std::integral_constant<int, 3>{});
    }
    if (in_range) {
        ret[i] = acc;
        n_steps[i] = k;
        first_action[i] = first;
        last_ob[i] = o;
        terminated[i] = (uint8_t)d;
    }
}

// ---------------------------------------------------------------------------
// helpers
// ---------------------------------------------------------------------------
// one thread = four consecutive lanes = one Philox block = one 16-byte store (the last quad of a ragged batch: scalar stores)
__global__ __launch_bounds__(BLOCK) void synthetic_actions_kernel(int32_t *__restrict__ action, int64_t n, RngKey key,
                                                                 uint32_t q0, uint32_t n_actions)
{
    const int64_t stride = (int64_t)gridDim.x * BLOCK, n4 = (n + 3) >> 2;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n4; i += stride) {
        const uint4 w = philox4x32_10(q0 + (uint32_t)i, key.t_lo, key.t_hi, (uint32_t)POMDP_STREAM_ACTION << 24,
                                      key.k0, key.k1);
        typedef int v4i __attribute__((ext_vector_type(4)));
        const v4i a = {(int)__umulhi(w.x, n_actions), (int)__umulhi(w.y, n_actions), (int)__umulhi(w.z, n_actions),
                       (int)__umulhi(w.w, n_actions)};
        if (4 * i + 4 <= n) __builtin_nontemporal_store(a, reinterpret_cast<v4i *>(action) + i);   // streamed, like every lane column
        else for (int64_t l = 4 * i; l < n; ++l) action[l] = a[(int)(l & 3)];
    }
}

__global__ void philox_blocks_kernel(const uint32_t *__restrict__ ck, uint32_t *__restrict__ out, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const uint4 w = philox4x32_10(ck[6 * i], ck[6 * i + 1], ck[6 * i + 2], ck[6 * i + 3], ck[6 * i + 4], ck[6 * i + 5]);
        out[4 * i] = w.x; out[4 * i + 1] = w.y; out[4 * i + 2] = w.z; out[4 * i + 3] = w.w;
    }
}

static inline RngKey make_key(uint64_t seed, uint64_t t)
{
    RngKey k;
    k.k0 = (uint32_t)seed; k.k1 = (uint32_t)(seed >> 32);
    k.t_lo = (uint32_t)t; k.t_hi = (uint32_t)(t >> 32);
    return k;
}

static inline bool bad_range(int64_t n, uint32_t lane0) { return n < 0 || (uint64_t)lane0 + (uint64_t)n > (1ull << 32); }

// pomdp_step_sync / pomdp_reset_sync (scalar mode): the flag the next one-lane launch publishes its outputs through;
// the launcher that takes it sets the pointer back to null
static thread_local uint32_t *tl_host_flag = nullptr;
static thread_local uint32_t tl_flag_value = 0;

template <class Env>
static int launch_reset(const typename Env::Params &p, uint32_t *state, int32_t *ob, int64_t n, uint64_t seed,
                        uint32_t lane0, uint64_t t, void *stream)
{
    if (!state || bad_range(n, lane0)) return POMDP_E_BADARG;
    if (n == 0) return 0;
    uint32_t *flag = nullptr;
    if (n == 1 && ob && tl_host_flag) { flag = tl_host_flag; tl_host_flag = nullptr; }
    hipLaunchKernelGGL(reset_kernel<Env>, dim3(blocks_for(n)), dim3(BLOCK), 0, (hipStream_t)stream, p, state, ob, n,
                       make_key(seed, t), lane0, flag, tl_flag_value);
    return (int)hipGetLastError();
}

template <class Env>   // defined with the other quad-per-thread kernels below
__global__ void step_quad_kernel(uint32_t *__restrict__ state, const int32_t *__restrict__ action, int32_t *__restrict__ ob,
                                 int32_t *__restrict__ reward, uint8_t *__restrict__ done, uint32_t *__restrict__ err, int64_t n,
                                 RngKey key, uint32_t lane0, int flags, const typename Env::Params p);

template <class Env>
static int launch_step(const typename Env::Params &p, uint32_t *state, const int32_t *action, int32_t *ob,
                       typename Env::Reward *reward, uint8_t *done, uint32_t *err, int64_t n, uint64_t seed,
                       uint32_t lane0, uint64_t t, int flags, void *stream)
{
    if (!state || !action || !ob || !reward || !done || bad_range(n, lane0)) return POMDP_E_BADARG;
    if (n == 0) return 0;
    // Two lanes per thread where the env pools work across a wave's two 64-lane sub-batches (RockSample: the
    // Finisher specialisation above) and the batch still gives every CU several workgroups; one lane per thread
    // otherwise (measured equal within 2 % for the generic envs, tools/microbench.hip).
    if constexpr (Env::QUAD_STEP) {
        // a quad of lanes per thread (step_quad_kernel) once the batch fills the chip with such workgroups
        const bool cols16 = ((reinterpret_cast<uintptr_t>(state) | reinterpret_cast<uintptr_t>(action) | reinterpret_cast<uintptr_t>(ob) |
                              reinterpret_cast<uintptr_t>(reward)) & 15u) == 0 && (reinterpret_cast<uintptr_t>(done) & 3u) == 0;
        if (n >= STEP_QUAD_MIN_LANES && n % (4 * BLOCK) == 0 && (lane0 & 3u) == 0 && cols16) {
            hipLaunchKernelGGL(step_quad_kernel<Env>, dim3((unsigned)(n / (4 * BLOCK))), dim3(BLOCK), 0, (hipStream_t)stream, state,
                               action, ob, reward, done, err, n, make_key(seed, t), lane0, flags, p);
            return (int)hipGetLastError();
        }
    }
    if constexpr (Env::POOLED_LPT2) {
        if (n >= LPT2_MIN_LANES) {
            hipLaunchKernelGGL((step_kernel<Env, 2>), dim3((unsigned)((n + 2 * BLOCK - 1) / (2 * BLOCK))), dim3(BLOCK), 0,
                               (hipStream_t)stream, state, action, ob, reward, done, err, n, make_key(seed, t), lane0, flags, RngKey(), p);
            return (int)hipGetLastError();
        }
    }
    uint32_t *flag = nullptr;
    if (n == 1 && tl_host_flag) { flag = tl_host_flag; tl_host_flag = nullptr; }
    hipLaunchKernelGGL((step_kernel<Env, 1>), dim3(blocks_for(n)), dim3(BLOCK), 0, (hipStream_t)stream, state,
                       action, ob, reward, done, err, n, make_key(seed, t), lane0, flags, RngKey(), p, flag, tl_flag_value);
    return (int)hipGetLastError();
}

// step + policy for the next call counter in one launch (see step_kernel<.., CHAIN>)
template <class Env>
static int launch_step_chain(const typename Env::Params &p, uint32_t *state, int32_t *action, int32_t *ob,
                             typename Env::Reward *reward, uint8_t *done, uint32_t *err, int64_t n, uint64_t seed,
                             uint64_t action_seed, uint32_t lane0, uint64_t t, int flags, void *stream)
{
    if (!state || !action || !ob || !reward || !done || bad_range(n, lane0) || (lane0 & 3u)) return POMDP_E_BADARG;
    if (n == 0) return 0;
    if constexpr (Env::POOLED_LPT2) {
        if (n >= LPT2_MIN_LANES) {
            hipLaunchKernelGGL((step_kernel<Env, 2, true>), dim3((unsigned)((n + 2 * BLOCK - 1) / (2 * BLOCK))), dim3(BLOCK),
                               0, (hipStream_t)stream, state, action, ob, reward, done, err, n, make_key(seed, t), lane0,
                               flags, make_key(action_seed, t + 1), p);
            return (int)hipGetLastError();
        }
    }
    hipLaunchKernelGGL((step_kernel<Env, 1, true>), dim3(blocks_for(n)), dim3(BLOCK), 0, (hipStream_t)stream, state, action,
                       ob, reward, done, err, n, make_key(seed, t), lane0, flags, make_key(action_seed, t + 1), p);
    return (int)hipGetLastError();
}

// The actions of a quad-per-thread launch's first step: read from row 0 of `action`, or (gen_first, wave-uniform) the
// quad's block of the synthetic policy at the call counter before akey0's — computed here and written to that row.
static __device__ __forceinline__ u32x4 first_actions4(uint32_t *action_row0, int gen_first, uint32_t glane0, const RngKey &akey0,
                                                       uint32_t n_act)
{
    if (!gen_first) return ld_stream4(action_row0);
    const uint64_t tf = (((uint64_t)akey0.t_hi << 32) | akey0.t_lo) - 1ull;
    const uint4 w = philox4x32_10(glane0 >> 2, (uint32_t)tf, (uint32_t)(tf >> 32), (uint32_t)POMDP_STREAM_ACTION << 24, akey0.k0, akey0.k1);
    const u32x4 a = {__umulhi(w.x, n_act), __umulhi(w.y, n_act), __umulhi(w.z, n_act), __umulhi(w.w, n_act)};
    st_stream4(action_row0, a[0], a[1], a[2], a[3]);
    return a;
}

// The fused RockSample loop with a thread owning four CONSECUTIVE lanes — a quad.  RockSample's word contract shares the
// STEP block, the RESET block and the policy's ACTION block among the four lanes of a quad, so with this mapping all
// three are the thread's own: three Philox blocks per thread-step straight into registers, lane j taking element j — no
// exchange through LDS, no ballots, no task lists, and no dependence on the lane step: the compiler interleaves the three
// chains with the table lookups.  A thread's outputs are four consecutive elements of each column: one 16-byte store per
// int32 column and one 4-byte store of the packed done bytes per step instead of twenty scalar stores; state and first
// actions come in the same way.  The lane step is the table-driven one.  Full workgroups of 1024 lanes and auto-reset
// only (the launcher's SIMPLE conditions).  Same results as steps_kernel: the mapping of lanes to threads is invisible
// to a lane's random words.
template <class Env>
__global__ __launch_bounds__(BLOCK) void steps_quad_kernel(uint32_t *__restrict__ state, int32_t *__restrict__ action,
                                                           int32_t *__restrict__ ob, int32_t *__restrict__ reward,
                                                           uint8_t *__restrict__ done, int64_t n, RngKey key0, uint32_t lane0,
                                                           RngKey akey0, int k_steps, int64_t rec, int gen_first,
                                                           const typename Env::Params p)
{
    constexpr int W = Env::WORDS;
    using S = typename Env::S;
    __shared__ typename Env::Shared sh;
    __shared__ typename Env::StepTab tab;
    const uint32_t l0 = blockIdx.x * (uint32_t)(4 * BLOCK) + 4u * threadIdx.x;   // this thread's first lane within the shard
    const uint32_t glane0 = lane0 + l0;                                          // ... and its global lane id (a multiple of 4)
    uint32_t *action_w = reinterpret_cast<uint32_t *>(action) + l0, *ob_w = reinterpret_cast<uint32_t *>(ob) + l0;
    uint32_t *reward_w = reinterpret_cast<uint32_t *>(reward) + l0;
    uint32_t *done_w = reinterpret_cast<uint32_t *>(done + l0);
    typename Env::State st[4];
    int a_cur[4];
    const uint32_t n_act = (uint32_t)Env::n_actions(p);
    {
        const u32x4 s_lo = ld_stream4(state + l0);
        u32x4 s_hi = {0, 0, 0, 0};
        if (W == 2) s_hi = ld_stream4(state + n + l0);
        const u32x4 a4 = first_actions4(action_w, gen_first, glane0, akey0, n_act);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            a_cur[j] = (int)a4[j];
            st[j].s = (S)((uint64_t)s_lo[j] | ((uint64_t)s_hi[j] << 32));
        }
    }
    action_w += rec;
    Env::stage(sh, p, (int)threadIdx.x);
    __syncthreads();
    Env::build_tab(tab, sh, p, (int)threadIdx.x);
    __syncthreads();
    const int K = p.num_rocks;
    const uint32_t start = (uint32_t)p.start_x | ((uint32_t)p.start_y << 4);
    const uint64_t t0 = ((uint64_t)key0.t_hi << 32) | key0.t_lo, ta0 = ((uint64_t)akey0.t_hi << 32) | akey0.t_lo;
    for (int s = 0; s < k_steps; ++s) {
        RngKey key = key0;
        key.t_lo = (uint32_t)(t0 + (uint64_t)s); key.t_hi = (uint32_t)((t0 + (uint64_t)s) >> 32);
        const uint64_t ta = ta0 + (uint64_t)s;
        // the quad's sensor words of this step (StochasticRock: block 2 of the stream — block 0 gates the actions, rock.py:443),
        // the words its fresh episodes start from, and its policy words of the next call counter
        constexpr uint32_t SENSOR_BLOCK = Env::STOCHASTIC ? 2u : 0u;
        const uint4 sw = philox4x32_10(glane0 >> 2, key.t_lo, key.t_hi, ((uint32_t)POMDP_STREAM_STEP << 24) | SENSOR_BLOCK, key.k0, key.k1);
        const uint4 rw = Env::reset_block(key, glane0, 0u);
        const uint4 pw = philox4x32_10(glane0 >> 2, (uint32_t)ta, (uint32_t)(ta >> 32), (uint32_t)POMDP_STREAM_ACTION << 24, key.k0, key.k1);
        const uint32_t H[4] = {sw.x, sw.y, sw.z, sw.w}, P[4] = {pw.x, pw.y, pw.z, pw.w}, R[4] = {rw.x, rw.y, rw.z, rw.w};
        bool acts[4] = {true, true, true, true};
        if constexpr (Env::STOCHASTIC) {                                       // the action is applied iff binomial(1, p_move) says so
            const uint4 gw = Env::quad_block(key, glane0, 0u);
            const uint32_t G[4] = {gw.x, gw.y, gw.z, gw.w};
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acts[j] = Env::k53_le(G[j], (uint32_t)(p.act_thr >> 26), (uint32_t)p.act_thr & Env::LO_MASK,
                                      [&]() { return Env::elem(Env::quad_block(key, glane0, 1u), (uint32_t)j); });
        }
        int r[4], d[4];
        uint32_t o[4], a_next[4], codes[4];
        Env::reset_codes4(R, key, glane0, K, codes);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            typename Env::Aux aux;
            if constexpr (Env::STOCHASTIC) {
                typename Env::State nx = st[j];
                Env::step_tab(tab, nx, a_cur[j], r[j], d[j], aux);
                if (acts[j]) st[j] = nx; else { r[j] = 0; d[j] = 0; aux.want = false; }
            } else {
                Env::step_tab(tab, st[j], a_cur[j], r[j], d[j], aux);
            }
            const uint32_t lane = glane0 + (uint32_t)j;
            st[j].s = d[j] ? (S)((uint64_t)start | ((uint64_t)codes[j] << 8)) : st[j].s;   // done lanes start a new episode
            o[j] = (uint32_t)Env::sensor_ob(sh, st[j], aux, H[j], [&]() { return Env::elem(Env::quad_block(key, lane, SENSOR_BLOCK + 1u), (uint32_t)j); });
            a_next[j] = __umulhi(P[j], n_act);
            a_cur[j] = (int)a_next[j];
        }
        st_stream4(action_w, a_next[0], a_next[1], a_next[2], a_next[3]);
        st_stream4(ob_w, o[0], o[1], o[2], o[3]);
        st_stream4(reward_w, (uint32_t)r[0], (uint32_t)r[1], (uint32_t)r[2], (uint32_t)r[3]);
        st_stream(done_w, (uint32_t)d[0] | ((uint32_t)d[1] << 8) | ((uint32_t)d[2] << 16) | ((uint32_t)d[3] << 24));
        action_w += rec; ob_w += rec; reward_w += rec; done_w += rec / 4;
    }
    // the state is the loop's carry: it reaches memory once
    st_stream4(state + l0, (uint32_t)st[0].s, (uint32_t)st[1].s, (uint32_t)st[2].s, (uint32_t)st[3].s);
    if (W == 2)
        st_stream4(state + n + l0, (uint32_t)((uint64_t)st[0].s >> 32), (uint32_t)((uint64_t)st[1].s >> 32),
                   (uint32_t)((uint64_t)st[2].s >> 32), (uint32_t)((uint64_t)st[3].s >> 32));
}

// ONE step of RockSample with the caller's actions (env.step()) and a quad of consecutive lanes per thread: the quad's
// STEP block is the thread's own (one Philox block for four lane-steps, computed under the latency of the loads, no
// exchange through LDS), state / action / ob / reward move as 16-byte accesses and the four done bytes as one word —
// 105 VALU instructions per lane-step where step_kernel<Env, 2> issues 187; 7.8 against 8.3 us per step of 2^20 lanes for
// RockSample(7,8), 7.5 against 8.8 for StochasticRock (whose step_kernel runs one lane per thread), inside a python loop
// (DESIGN.md §5 has the timeline of such a launch).  Same contract as step_kernel: a lane whose action is out of range is left untouched and counted in
// *err, without auto-reset a done lane stays frozen.  Full workgroups of 1024 lanes on 16-byte column boundaries only
// (launch_step); anything else takes step_kernel.
template <class Env>
__global__ __launch_bounds__(BLOCK) void step_quad_kernel(uint32_t *__restrict__ state, const int32_t *__restrict__ action,
                                                          int32_t *__restrict__ ob, int32_t *__restrict__ reward,
                                                          uint8_t *__restrict__ done, uint32_t *__restrict__ err, int64_t n,
                                                          RngKey key, uint32_t lane0, int flags, const typename Env::Params p)
{
    constexpr int W = Env::WORDS;
    using S = typename Env::S;
    __shared__ typename Env::Shared sh;
    TL(0);
    const bool auto_reset = flags & POMDP_AUTO_RESET;
    const uint32_t l0 = blockIdx.x * (uint32_t)(4 * BLOCK) + 4u * threadIdx.x;
    const uint32_t glane0 = lane0 + l0;
    uint32_t *const done_w = reinterpret_cast<uint32_t *>(done + l0);
    const u32x4 s_lo = ld_stream4(state + l0);
    u32x4 s_hi = {0, 0, 0, 0};
    if (W == 2) s_hi = ld_stream4(state + n + l0);
    const u32x4 a4 = ld_stream4(reinterpret_cast<const uint32_t *>(action) + l0);
    const uint32_t dn = auto_reset ? 0u : ld_stream(done_w);                       // frozen lanes (the reference would assert)
    const auto staged = Env::stage_load(p, (int)threadIdx.x);
    // the quad's words depend on lane ids only: Philox under the load latency
    constexpr uint32_t SENSOR_BLOCK = Env::STOCHASTIC ? 2u : 0u;
    const uint4 sw = Env::quad_block(key, glane0, SENSOR_BLOCK), rw = Env::reset_block(key, glane0, 0u);
    const uint32_t H[4] = {sw.x, sw.y, sw.z, sw.w}, R[4] = {rw.x, rw.y, rw.z, rw.w};
    uint32_t G[4] = {0, 0, 0, 0};
    if constexpr (Env::STOCHASTIC) { const uint4 gw = Env::quad_block(key, glane0, 0u); G[0] = gw.x; G[1] = gw.y; G[2] = gw.z; G[3] = gw.w; }
    Env::stage_store(sh, staged, (int)threadIdx.x);
    __syncthreads();
#ifdef POMDP_DEV_TIMELINE
    TL(1);
    asm volatile("" :: "v"(s_lo[0] + a4[0]));
    TL(2);
#endif
    const uint32_t n_act = (uint32_t)Env::n_actions(p);
    typename Env::State st[4];
    typename Env::Aux aux[4];
    int r[4], d[4];
    bool live[4], fresh[4];
    uint32_t n_bad = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const bool valid = a4[j] < n_act, was_done = ((dn >> (8 * j)) & 0xFFu) != 0u;
        live[j] = valid && !was_done;
        n_bad += (uint32_t)(!valid && !was_done);
        st[j].s = (S)((uint64_t)s_lo[j] | ((uint64_t)s_hi[j] << 32));
        typename Env::State nx = st[j];
        Env::step_pre(sh, p, nx, valid ? (int)a4[j] : 0, r[j], d[j], aux[j]);
        bool acts = live[j];
        if constexpr (Env::STOCHASTIC)                                             // applied iff binomial(1, p_move) says so (rock.py:443)
            acts = acts && Env::k53_le(G[j], (uint32_t)(p.act_thr >> 26), (uint32_t)p.act_thr & Env::LO_MASK,
                                       [&]() { return Env::elem(Env::quad_block(key, glane0, 1u), (uint32_t)j); });
        if (acts) st[j] = nx; else { r[j] = 0; d[j] = live[j] ? 0 : (int)was_done; aux[j].want = false; }
        fresh[j] = acts && d[j] != 0 && auto_reset;                                // done lanes start a new episode
    }
    // (A CHECK neither moves the agent nor ends the episode: the sensor reads the state the step left.)
    uint32_t o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
        o[j] = (uint32_t)Env::sensor_ob(sh, st[j], aux[j], H[j], [&]() { return Env::elem(Env::quad_block(key, glane0, SENSOR_BLOCK + 1u), (uint32_t)j); });
    st_stream4(reinterpret_cast<uint32_t *>(ob) + l0, o[0], o[1], o[2], o[3]);
    st_stream4(reinterpret_cast<uint32_t *>(reward) + l0, (uint32_t)r[0], (uint32_t)r[1], (uint32_t)r[2], (uint32_t)r[3]);
    st_stream(done_w, (uint32_t)d[0] | ((uint32_t)d[1] << 8) | ((uint32_t)d[2] << 16) | ((uint32_t)d[3] << 24));
    TL(3);
#pragma unroll
    for (int j = 0; j < 4; ++j)
        st[j].s = fresh[j] ? Env::fresh_state(p, R[j], key, glane0 + (uint32_t)j) : st[j].s;
    st_stream4(state + l0, (uint32_t)st[0].s, (uint32_t)st[1].s, (uint32_t)st[2].s, (uint32_t)st[3].s);
    if (W == 2)
        st_stream4(state + n + l0, (uint32_t)((uint64_t)st[0].s >> 32), (uint32_t)((uint64_t)st[1].s >> 32),
                   (uint32_t)((uint64_t)st[2].s >> 32), (uint32_t)((uint64_t)st[3].s >> 32));
    if (n_bad && err) atomicAdd(err, n_bad);
#ifdef POMDP_DEV_TIMELINE
    TL(4);
    __builtin_amdgcn_s_waitcnt(0);
    TL(5);
#endif
}

// Tag (one opponent) with a quad per thread: the policy's ACTION block is the thread's own, the flights of failed TAGs
// (about a fifth of the lanes) and the rare resets are pooled per wave of 256 lanes — one Philox pass instead of the two per
// 256 lanes that Finisher<TagEnv, 2> needs with the policy blocks in its task list — and the outputs leave as 16-byte stores.
template <bool TAB>   // TAB: the lane step reads the (cells, action) table built when the launch starts (from 16 steps per launch)
__global__ __launch_bounds__(BLOCK) void tag_steps_quad_kernel(uint32_t *__restrict__ state, int32_t *__restrict__ action,
                                                               int32_t *__restrict__ ob, float *__restrict__ reward,
                                                               uint8_t *__restrict__ done, int64_t n, RngKey key0,
                                                               uint32_t lane0, RngKey akey0, int k_steps, int64_t rec,
                                                               int gen_first, const TagEnv::Params p)
{
    using Env = TagEnv;
    __shared__ Env::Shared sh;
    __shared__ typename std::conditional<TAB, Env::StepTab, NoTab>::type tab;
    __shared__ uint8_t src_lds[BLOCK / 64][256];             // task rank -> lane within the wave's 256
    __shared__ uint32_t res_lds[BLOCK / 64][256][4];         // task rank -> its Philox block
    const int wv = (int)(threadIdx.x >> 6), me = (int)(threadIdx.x & 63u);
    const uint32_t l0 = blockIdx.x * (uint32_t)(4 * BLOCK) + 4u * threadIdx.x;
    const uint32_t glane0 = lane0 + l0, wave0 = glane0 - 4u * (uint32_t)me;
    uint32_t *action_w = reinterpret_cast<uint32_t *>(action) + l0, *ob_w = reinterpret_cast<uint32_t *>(ob) + l0;
    uint32_t *reward_w = reinterpret_cast<uint32_t *>(reward) + l0;
    uint32_t *done_w = reinterpret_cast<uint32_t *>(done + l0);
    Env::State st[4];
    int a_cur[4];
    const uint32_t n_act = (uint32_t)Env::n_actions(p);
    {
        const u32x4 s4 = ld_stream4(state + l0);
        const u32x4 a4 = first_actions4(action_w, gen_first, glane0, akey0, n_act);
#pragma unroll
        for (int j = 0; j < 4; ++j) { a_cur[j] = (int)a4[j]; st[j].w = s4[j]; }
    }
    action_w += rec;
    Env::stage(sh, p, (int)threadIdx.x);
    __syncthreads();
    if constexpr (TAB) {
        Env::build_tab(tab, sh, p, (int)threadIdx.x);
        __syncthreads();
    }
    const uint64_t t0 = ((uint64_t)key0.t_hi << 32) | key0.t_lo, ta0 = ((uint64_t)akey0.t_hi << 32) | akey0.t_lo;
    for (int s = 0; s < k_steps; ++s) {
        RngKey key = key0;
        key.t_lo = (uint32_t)(t0 + (uint64_t)s); key.t_hi = (uint32_t)((t0 + (uint64_t)s) >> 32);
        const uint64_t ta = ta0 + (uint64_t)s;
        const uint4 pw = philox4x32_10(glane0 >> 2, (uint32_t)ta, (uint32_t)(ta >> 32), (uint32_t)POMDP_STREAM_ACTION << 24, key.k0, key.k1);
        const uint32_t P[4] = {pw.x, pw.y, pw.z, pw.w};
        int o[4], d[4];
        float r[4];
        Env::Flight f[4];
        uint64_t fm[4], rm[4];
        int nfl = 0, nrs = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if constexpr (TAB) Env::step_one_opponent_tab(tab, st[j], a_cur[j], o[j], r[j], d[j], f[j]);
            else Env::step_one_opponent_pre(sh, p, st[j], a_cur[j], o[j], r[j], d[j], f[j]);
            fm[j] = __ballot(f[j].need);
            rm[j] = __ballot(d[j] != 0);
            nfl += __popcll(fm[j]);
            nrs += __popcll(rm[j]);
        }
        // task list: the flights, then the resets (a lane is never both: a failed TAG does not end the episode)
        auto below = [&](uint64_t m) {
            return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        };
        int rank[4], cf = 0, cr = nfl;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            rank[j] = f[j].need ? cf + below(fm[j]) : cr + below(rm[j]);
            cf += __popcll(fm[j]);
            cr += __popcll(rm[j]);
            if (f[j].need || d[j]) src_lds[wv][rank[j] & 255] = (uint8_t)(4 * me + j);
        }
        const int ntask = nfl + nrs;
        for (int base = 0; base < ntask; base += 64) {
            const int q = base + me;
            if (q < ntask) {
                const uint32_t src_lane = wave0 + (uint32_t)src_lds[wv][q & 255];
                const uint32_t strm = q < nfl ? POMDP_STREAM_STEP : POMDP_STREAM_RESET;
                const uint4 w = philox4x32_10(src_lane, key.t_lo, key.t_hi, strm << 24, key.k0, key.k1);
                uint32_t *dst = res_lds[wv][q & 255];
                dst[0] = w.x; dst[1] = w.y; dst[2] = w.z; dst[3] = w.w;
            }
        }
        uint32_t a_next[4];
        uint4 rb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {                                          // all four blocks in flight, one wait; used by the lanes with a task
            const uint32_t *res = res_lds[wv][rank[j] & 255];
            rb[j] = make_uint4(res[0], res[1], res[2], res[3]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (f[j].need) Env::flee(sh, p, st[j], f[j], rb[j].x, rb[j].y, rb[j].z);
            if (d[j]) {
                if (!Env::reset_from_block(p, st[j], rb[j])) Env::reset(sh, p, st[j], key, glane0 + (uint32_t)j);   // rejections ran past the block
            }
            a_next[j] = __umulhi(P[j], n_act);
            a_cur[j] = (int)a_next[j];
        }
        st_stream4(action_w, a_next[0], a_next[1], a_next[2], a_next[3]);
        st_stream4(ob_w, (uint32_t)o[0], (uint32_t)o[1], (uint32_t)o[2], (uint32_t)o[3]);
        st_stream4(reward_w, __float_as_uint(r[0]), __float_as_uint(r[1]), __float_as_uint(r[2]), __float_as_uint(r[3]));
        st_stream(done_w, (uint32_t)d[0] | ((uint32_t)d[1] << 8) | ((uint32_t)d[2] << 16) | ((uint32_t)d[3] << 24));
        action_w += rec; ob_w += rec; reward_w += rec; done_w += rec / 4;
    }
    st_stream4(state + l0, st[0].w, st[1].w, st[2].w, st[3].w);
}

// Network with a quad per thread.  The reference draws one double per UP machine and one for the action (network.py:94-109),
// from the lane's own stream, four high words per Philox block (split layout).  Under a random policy a lane
// has 1.4 machines up on average (12 % of the lanes have three or more, 2 % four or more), so almost every lane-step is
// served by the FIRST block of its stream: each of the thread's four lanes computes that block and applies its first two
// words to its first two up machines straight-line (NetworkEnv::draws), the action's draw being the word after the last
// machine — no loop to the wave's largest draw count, which is what steps_kernel<NetworkEnv> pays for every lane.  Lanes
// with more draws to make (a third machine, or the action's draw behind three) hand (machines left, failed-neighbour set,
// the block's other two words) to a per-wave task list; one pooled pass per 64 such lanes continues their streams — the
// two words, then block by block — and returns the machines that fail and the action's draw.  The policy's ACTION block
// is the thread's own, the outputs leave as 16-byte stores, the reward comes from a table of the float32(float64) values
// the reference's arithmetic gives.  A draw decided by its low word (2^-27 per draw) sends the lane through
// NetworkEnv::step_exact, the exact per-lane form.  Network never terminates, so there is no reset.
template <int NB>   // bytes of the machine set: ceil(n_machines / 8)
__global__ __launch_bounds__(BLOCK) void network_steps_quad_kernel(uint32_t *__restrict__ state, int32_t *__restrict__ action,
                                                                   int32_t *__restrict__ ob, float *__restrict__ reward,
                                                                   uint8_t *__restrict__ done, int64_t n, RngKey key0,
                                                                   uint32_t lane0, RngKey akey0, int k_steps, int64_t rec,
                                                                   int gen_first, const NetworkEnv::Params p)
{
    using Env = NetworkEnv;
    __shared__ Env::Shared sh;                               // the nibble tables of the exact per-lane form (ties only)
    __shared__ uint32_t nbf8[NB][256];                       // nbf8[k][v]: machines that see a failed neighbour when the down
                                                             // machines among 8 k .. 8 k + 7 are the set v (network.py:82-85)
    __shared__ float rtab[3][68];                            // reward by (no action / ping / reboot, 2 per up machine with > 2
                                                             // neighbours + 1 per other up machine): network.py:87-92, 103, 110
    __shared__ uint32_t task_lds[BLOCK / 64][256][6];        // task rank -> {lane within the wave's 256 | has_action << 8, machines
                                                             // left, failed-neighbour set, words 2 and 3 of the lane's first
                                                             // block}; overwritten with {machines that fail, flags}
    const int wv = (int)(threadIdx.x >> 6), me = (int)(threadIdx.x & 63u);
    const uint32_t l0 = blockIdx.x * (uint32_t)(4 * BLOCK) + 4u * threadIdx.x;
    const uint32_t glane0 = lane0 + l0, wave0 = glane0 - 4u * (uint32_t)me;
    uint32_t *action_w = reinterpret_cast<uint32_t *>(action) + l0, *ob_w = reinterpret_cast<uint32_t *>(ob) + l0;
    uint32_t *reward_w = reinterpret_cast<uint32_t *>(reward) + l0;
    uint32_t *done_w = reinterpret_cast<uint32_t *>(done + l0);
    const uint32_t n_act = (uint32_t)Env::n_actions(p);
    uint32_t st[4];
    int a_cur[4];
    {
        const u32x4 s4 = ld_stream4(state + l0);
        const u32x4 a4 = first_actions4(action_w, gen_first, glane0, akey0, n_act);
#pragma unroll
        for (int j = 0; j < 4; ++j) { a_cur[j] = (int)a4[j]; st[j] = s4[j]; }
    }
    action_w += rec;
    Env::stage(sh, p, (int)threadIdx.x);
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {                        // BLOCK threads = the 256 values of a byte
        uint32_t m = 0;
        for (int i = 0; i < p.n_machines; ++i) m |= (((p.nb_mask[i] >> (8 * kb)) & threadIdx.x) != 0u ? 1u : 0u) << i;
        nbf8[kb][threadIdx.x] = m;
    }
    if (threadIdx.x < 3 * 68) {                              // r = float32(float64(base) - cost), as the reference computes it
        const int kind = (int)threadIdx.x / 68, b = (int)threadIdx.x % 68;
        double r = (double)b;
        if (kind == 1) r -= .1;
        if (kind == 2) r -= 2.5;
        rtab[kind][b] = (float)r;
    }
    __syncthreads();
    const Env::Thr T = Env::thresholds(p);
    const uint32_t all_up = p.n_machines >= 32 ? 0xFFFFFFFFu : ((1u << p.n_machines) - 1u);
    const int M2 = 2 * p.n_machines;
    const uint64_t t0 = ((uint64_t)key0.t_hi << 32) | key0.t_lo, ta0 = ((uint64_t)akey0.t_hi << 32) | akey0.t_lo;
    for (int s = 0; s < k_steps; ++s) {
        RngKey key = key0;
        key.t_lo = (uint32_t)(t0 + (uint64_t)s); key.t_hi = (uint32_t)((t0 + (uint64_t)s) >> 32);
        const uint64_t ta = ta0 + (uint64_t)s;
        const uint4 pw = philox4x32_10(glane0 >> 2, (uint32_t)ta, (uint32_t)(ta >> 32), (uint32_t)POMDP_STREAM_ACTION << 24, key.k0, key.k1);
        const uint32_t P[4] = {pw.x, pw.y, pw.z, pw.w};
        uint32_t kill[4], todo[4], nbf[4], near[4], hz[4], hw[4];
        int base[4];
        bool truthful[4], more[4], act_pending[4];
        uint64_t mm[4];
        int ntask = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t s0 = st[j];
            const uint4 h = stream_block(key, glane0 + (uint32_t)j, POMDP_STREAM_STEP, 0u);
            const int n_up = __popc(s0);
            base[j] = n_up + __popc(s0 & p.deg_gt2_mask);                      // network.py:87-92
            {
                const uint32_t down = ~s0 & all_up;
                uint32_t f = nbf8[0][down & 255u];
#pragma unroll
                for (int kb = 1; kb < NB; ++kb) f |= nbf8[kb][(down >> (8 * kb)) & 255u];
                nbf[j] = f;
            }
            todo[j] = s0;
            near[j] = 0xFFFFFFFFu;
            const uint32_t H2[2] = {h.x, h.y};
            kill[j] = Env::draws<2>(H2, todo[j], nbf[j], T, near[j]);
            hz[j] = h.z; hw[j] = h.w;
            const bool has_action = a_cur[j] < M2;
            uint32_t aw = n_up == 1 ? h.y : h.x;                                // word n_up of the block (n_up < 3), as selects
            aw = n_up >= 2 ? h.z : aw;
            uint32_t near_a = 0xFFFFFFFFu;
            const bool tr = Env::truthful_of(aw, T, near_a);
            const bool here = has_action && n_up < 3;                           // the action's draw is one of these three words
            truthful[j] = here && tr;
            near[j] = min(near[j], here ? near_a : 0xFFFFFFFFu);
            act_pending[j] = has_action && !here;
            more[j] = todo[j] != 0u || act_pending[j];
            mm[j] = __ballot(more[j]);
            ntask += __popcll(mm[j]);
        }
        if (ntask) {                                                           // wave-uniform
            int rank[4], c = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                rank[j] = c + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mm[j] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mm[j], 0u));
                c += __popcll(mm[j]);
                if (more[j]) {
                    uint32_t *t = task_lds[wv][rank[j] & 255];
                    t[0] = (uint32_t)(4 * me + j) | ((uint32_t)act_pending[j] << 8); t[1] = todo[j]; t[2] = nbf[j];
                    t[3] = hz[j]; t[4] = hw[j];
                }
            }
            for (int b0 = 0; b0 < ntask; b0 += 64) {
                const int q = b0 + me;
                if (q < ntask) {
                    uint32_t *t = task_lds[wv][q & 255];
                    const uint32_t w0 = t[0], src_lane = wave0 + (w0 & 255u), nb = t[2];
                    uint32_t td = t[1], nr = 0xFFFFFFFFu;
                    bool pend = (w0 >> 8) & 1u, tr = false;
                    // words 2 and 3 of the first block, then the stream's following blocks
                    const uint32_t H2[2] = {t[3], t[4]};
                    int left = __popc(td);
                    uint32_t kl = Env::draws<2>(H2, td, nb, T, nr);
                    if (pend && left < 2) { tr = Env::truthful_of(left == 0 ? H2[0] : H2[1], T, nr); pend = false; }
                    for (uint32_t blk = 1; td != 0u || pend; ++blk) {
                        const uint4 h = stream_block(key, src_lane, POMDP_STREAM_STEP, 2u * blk);
                        left = __popc(td);
                        kl |= Env::draw4(h, td, nb, T, nr);
                        if (pend && left < 4) {
                            uint32_t w = left == 1 ? h.y : h.x;
                            w = left == 2 ? h.z : w;
                            w = left == 3 ? h.w : w;
                            tr = Env::truthful_of(w, T, nr);
                            pend = false;
                        }
                    }
                    t[0] = kl; t[1] = (uint32_t)tr | (nr < 32u ? 2u : 0u);
                }
            }
            uint32_t tk[4], tf[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {                                      // all four reads in flight, one wait; used where more[j]
                const uint32_t *t = task_lds[wv][rank[j] & 255];
                tk[j] = t[0]; tf[j] = t[1];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                kill[j] |= more[j] ? tk[j] : 0u;
                truthful[j] = (more[j] && act_pending[j]) ? (tf[j] & 1u) != 0u : truthful[j];
                near[j] = (more[j] && (tf[j] & 2u)) ? 0u : near[j];
            }
        }
        uint32_t o4[4], r4[4], a_next[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int o;
            float r;
            if (near[j] < 32u) {                                               // a draw decided by its low word: the exact per-lane form
                Env::State e{st[j]};
                int d;
                Env::step_exact(sh, p, e, a_cur[j], key, glane0 + (uint32_t)j, o, r, d);
                st[j] = e.w;
            } else {                                                           // network.py:101-112
                const int a = a_cur[j], machine = (a >> 1) & 31;
                const bool has_action = a < M2, reboot = has_action && (a & 1);
                uint32_t sn = st[j] & ~kill[j];
                sn |= reboot ? 1u << machine : 0u;
                const int up = (int)((sn >> machine) & 1u);                    // a rebooted machine is up: ob = truthful either way
                o = has_action ? (truthful[j] ? up : 1 - up) : 2;
                r = rtab[has_action ? 1 + (a & 1) : 0][base[j]];
                st[j] = sn;
            }
            o4[j] = (uint32_t)o;
            r4[j] = __float_as_uint(r);
            a_next[j] = __umulhi(P[j], n_act);
            a_cur[j] = (int)a_next[j];
        }
        st_stream4(action_w, a_next[0], a_next[1], a_next[2], a_next[3]);
        st_stream4(ob_w, o4[0], o4[1], o4[2], o4[3]);
        st_stream4(reward_w, r4[0], r4[1], r4[2], r4[3]);
        st_stream(done_w, 0u);                                                 // network.py:113: never done
        action_w += rec; ob_w += rec; reward_w += rec; done_w += rec / 4;
    }
    st_stream4(state + l0, st[0], st[1], st[2], st[3]);
}

// BattleShip with a quad per thread.  A board is a long sequential rejection loop (battleship.py:167-180: about 23 words of a
// lane's stream on 10x10, 42 on 5x5) that one lane in ~285 needs per step; built when it comes up — by the whole wave, one
// lane at a time (BattleShipEnv::reset_where) — it is two thirds of all instructions steps_kernel<BattleShipEnv> issues.
// Under the board contract (include/pomdp_hip.h) a lane carries the board of its NEXT episode, drawn from
// stream NEXT at the call counter at which its current board was dealt; so inside the loop the end of an episode is a
// handful of selects (the cached board moves in, the lane remembers the step), and the boards the wave's lanes used up
// are built AFTER the loop, dealt out one per thread and 64 side by side (board_lockstep).  A lane that finishes a second
// episode before its next board exists triggers that pass early, for every lane of the wave that is waiting.  The state
// that reaches memory is the same whichever kernel ran: current board, visited mask, next board.
template <int MW>
__global__ __launch_bounds__(BLOCK) void battleship_steps_quad_kernel(uint32_t *__restrict__ state, int32_t *__restrict__ action,
                                                                      int32_t *__restrict__ ob, int32_t *__restrict__ reward,
                                                                      uint8_t *__restrict__ done, int64_t n, RngKey key0,
                                                                      uint32_t lane0, RngKey akey0, int k_steps, int64_t rec,
                                                                      int gen_first, const pomdp_battleship_params p)
{
    using Env = BattleShipEnv<MW>;
    __shared__ typename Env::Shared sh;
    __shared__ uint8_t task_lds[BLOCK / 64][256];            // task rank -> lane within the wave's 256
    __shared__ uint8_t ts_lds[BLOCK / 64][256];              // ... and the step at which that lane's current board was dealt
    __shared__ uint32_t res_lds[BLOCK / 64][256][MW];        // task rank -> the board built for it
    __shared__ typename Env::SeqTables seq;                  // the column patterns of the board builder
    Env::stage_seq(seq, p, (int)threadIdx.x);
    const int wv = (int)(threadIdx.x >> 6), me = (int)(threadIdx.x & 63u);
    const uint32_t l0 = blockIdx.x * (uint32_t)(4 * BLOCK) + 4u * threadIdx.x;
    const uint32_t glane0 = lane0 + l0, wave0 = glane0 - 4u * (uint32_t)me;
    uint32_t *action_w = reinterpret_cast<uint32_t *>(action) + l0, *ob_w = reinterpret_cast<uint32_t *>(ob) + l0;
    uint32_t *reward_w = reinterpret_cast<uint32_t *>(reward) + l0;
    uint32_t *done_w = reinterpret_cast<uint32_t *>(done + l0);
    const uint32_t n_act = (uint32_t)Env::n_actions(p);
    typename Env::State st[4];
    int a_cur[4];
    {
        uint32_t w[3 * MW][4];
#pragma unroll
        for (int q = 0; q < 3 * MW; ++q) {
            const u32x4 v = ld_stream4(state + (int64_t)q * n + l0);
#pragma unroll
            for (int j = 0; j < 4; ++j) w[q][j] = v[j];
        }
        const u32x4 a4 = first_actions4(action_w, gen_first, glane0, akey0, n_act);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            a_cur[j] = (int)a4[j];
            st[j].occ.lo = st[j].occ.hi = st[j].vis.lo = st[j].vis.hi = st[j].next.lo = st[j].next.hi = 0;
#pragma unroll
            for (int q = 0; q < MW; ++q) { st[j].occ.set_word(q, w[q][j]); st[j].vis.set_word(q, w[MW + q][j]); st[j].next.set_word(q, w[2 * MW + q][j]); }
        }
    }
    action_w += rec;
    __syncthreads();
    const uint64_t t0 = ((uint64_t)key0.t_hi << 32) | key0.t_lo, ta0 = ((uint64_t)akey0.t_hi << 32) | akey0.t_lo;
    int pend[4] = {-1, -1, -1, -1};                          // >= 0: the step at which the lane's board was dealt; its `next` is yet to be built
    // the boards of every waiting lane of the wave, 64 per pass (wave-uniform control flow; the scratch is wave-private)
    auto build_boards = [&]() {
        int rank[4], ntask = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint64_t m = __ballot(pend[j] >= 0);
            rank[j] = ntask + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            ntask += __popcll(m);
            if (pend[j] >= 0) { task_lds[wv][rank[j] & 255] = (uint8_t)(4 * me + j); ts_lds[wv][rank[j] & 255] = (uint8_t)pend[j]; }
        }
        if (ntask == 0) return;                                                // wave-uniform
        // A pool of 64 builders works the task list off: every lane feeds its board one word of its stream per iteration
        // (four iterations per Philox block, the blocks computed by all lanes at once, each with its own counter), and a
        // lane whose board is complete takes the next unclaimed task at the following block boundary instead of idling
        // until the slowest board of its batch is done (a 5x5 board takes 42 words on average and over a hundred at worst).
        const typename Env::BuildConsts bc = Env::build_consts(p);
        typename Env::Builder bld;
        int my = me < ntask ? me : -1, next_task = ntask < 64 ? ntask : 64;    // this lane's task; the first unclaimed one
        uint32_t blane = 0, bt_lo = 0, bt_hi = 0, blk = 0;
        auto take = [&](int q) {                                               // lanes with q >= 0 start on task q
            const int idx = q >= 0 ? (int)task_lds[wv][q & 255] : 0, s0 = q >= 0 ? (int)ts_lds[wv][q & 255] : 0;
            const uint64_t td = t0 + (uint64_t)s0;                             // battleship.py:131-137 on stream NEXT of that step's call counter
            if (q >= 0) { blane = wave0 + (uint32_t)idx; bt_lo = (uint32_t)td; bt_hi = (uint32_t)(td >> 32); blk = 0; bld.start(p.max_len); }
        };
        bld.idle();
        take(my);
        while (__any(my >= 0)) {
            const uint4 b4 = philox4x32_10(blane, bt_lo, bt_hi, ((uint32_t)POMDP_STREAM_NEXT << 24) | (blk & 0xFFFFFFu), key0.k0, key0.k1);
            ++blk;
            Env::feed(bld, seq, bc, b4.x);
            Env::feed(bld, seq, bc, b4.y);
            Env::feed(bld, seq, bc, b4.z);
            Env::feed(bld, seq, bc, b4.w);
            const bool fin = my >= 0 && !bld.busy();
            const uint64_t fm = __ballot(fin);
            if (fm != 0ull) {                                                  // wave-uniform
                if (fin) {
#pragma unroll
                    for (int w = 0; w < MW; ++w) res_lds[wv][my & 255][w] = (uint32_t)(bld.occ >> (32 * w));
                }
                const int r = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(fm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fm, 0u));
                if (fin) { my = next_task + r < ntask ? next_task + r : -1; bld.idle(); take(my); }
                next_task += __popcll(fm);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {                                          // four reads in flight, one wait
            const uint32_t *res = res_lds[wv][rank[j] & 255];
            if (pend[j] >= 0) {
                st[j].next.lo = st[j].next.hi = 0;
#pragma unroll
                for (int w = 0; w < MW; ++w) st[j].next.set_word(w, res[w]);
                pend[j] = -1;
            }
        }
    };
    for (int s = 0; s < k_steps; ++s) {
        const uint64_t ta = ta0 + (uint64_t)s;
        const uint4 pw = philox4x32_10(glane0 >> 2, (uint32_t)ta, (uint32_t)(ta >> 32), (uint32_t)POMDP_STREAM_ACTION << 24, akey0.k0, akey0.k1);
        const uint32_t P[4] = {pw.x, pw.y, pw.z, pw.w};
        uint32_t o4[4], r4[4], a_next[4], dpack = 0;
        int d[4];
        bool again = false;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int o, r;
            Env::step(sh, p, st[j], a_cur[j], key0, 0u, o, r, d[j]);              // battleship.py:91-122: draws nothing
            again |= d[j] && pend[j] >= 0;
            o4[j] = (uint32_t)o; r4[j] = (uint32_t)r;
            dpack |= (uint32_t)(d[j] != 0) << (8 * j);
            a_next[j] = __umulhi(P[j], n_act);
            a_cur[j] = (int)a_next[j];
        }
        st_stream4(action_w, a_next[0], a_next[1], a_next[2], a_next[3]);
        st_stream4(ob_w, o4[0], o4[1], o4[2], o4[3]);
        st_stream4(reward_w, r4[0], r4[1], r4[2], r4[3]);
        st_stream(done_w, dpack);
        action_w += rec; ob_w += rec; reward_w += rec; done_w += rec / 4;
        if (__any(again)) build_boards();                                      // a second episode ended before the lane's next board exists
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (d[j]) { Env::swap_in(st[j]); pend[j] = s; }
    }
    build_boards();
#pragma unroll
    for (int q = 0; q < MW; ++q) {
        st_stream4(state + (int64_t)q * n + l0, st[0].occ.word(q), st[1].occ.word(q), st[2].occ.word(q), st[3].occ.word(q));
        st_stream4(state + (int64_t)(MW + q) * n + l0, st[0].vis.word(q), st[1].vis.word(q), st[2].vis.word(q), st[3].vis.word(q));
        st_stream4(state + (int64_t)(2 * MW + q) * n + l0, st[0].next.word(q), st[1].next.word(q), st[2].next.word(q), st[3].next.word(q));
    }
}

// The generic fused loop with a quad per thread, for envs whose lane step is light enough that four of them fit a thread
// (Env::QUAD_FUSED; one state word): the policy's ACTION block is the thread's own, Env::step / Env::reset_where run per
// lane as in steps_kernel, the outputs leave as 16-byte stores.  Full workgroups of 1024 lanes, auto-reset.  Only for envs
// whose reset_where does not assume that a wave's 64 lanes are consecutive (RockSample's cooperative reset does).
template <class Env, class = void> struct quad_tab : std::false_type {};
template <class Env> struct quad_tab<Env, std::enable_if_t<Env::QUAD_TAB>> : std::true_type {};
template <class Env, class = void> struct quad_fused : std::false_type {};
template <class Env> struct quad_fused<Env, std::enable_if_t<Env::QUAD_FUSED>> : std::true_type {};

template <class Env>
__global__ __launch_bounds__(BLOCK) void steps_quad_generic_kernel(uint32_t *__restrict__ state, int32_t *__restrict__ action,
                                                                   int32_t *__restrict__ ob,
                                                                   typename Env::Reward *__restrict__ reward,
                                                                   uint8_t *__restrict__ done, int64_t n, RngKey key0,
                                                                   uint32_t lane0, RngKey akey0, int k_steps, int64_t rec,
                                                                   int gen_first, const typename Env::Params p)
{
    static_assert(Env::WORDS == 1 && sizeof(typename Env::Reward) == 4, "one state word, 4-byte rewards");
    __shared__ typename Env::Shared sh;
    const uint32_t l0 = blockIdx.x * (uint32_t)(4 * BLOCK) + 4u * threadIdx.x;
    const uint32_t glane0 = lane0 + l0;
    uint32_t *action_w = reinterpret_cast<uint32_t *>(action) + l0, *ob_w = reinterpret_cast<uint32_t *>(ob) + l0;
    uint32_t *reward_w = reinterpret_cast<uint32_t *>(reward) + l0;
    uint32_t *done_w = reinterpret_cast<uint32_t *>(done + l0);
    typename Env::State st[4];
    int a_cur[4];
    const uint32_t n_act = (uint32_t)Env::n_actions(p);
    {
#pragma unroll
        for (int j = 0; j < 4; ++j) Env::load(st[j], state, n, l0 + (uint32_t)j);
        const u32x4 a4 = first_actions4(action_w, gen_first, glane0, akey0, n_act);
#pragma unroll
        for (int j = 0; j < 4; ++j) a_cur[j] = (int)a4[j];
    }
    action_w += rec;
    Env::stage(sh, p, (int)threadIdx.x);
    __syncthreads();
    const uint64_t t0 = ((uint64_t)key0.t_hi << 32) | key0.t_lo, ta0 = ((uint64_t)akey0.t_hi << 32) | akey0.t_lo;
    for (int s = 0; s < k_steps; ++s) {
        RngKey key = key0;
        key.t_lo = (uint32_t)(t0 + (uint64_t)s); key.t_hi = (uint32_t)((t0 + (uint64_t)s) >> 32);
        const uint64_t ta = ta0 + (uint64_t)s;
        const uint4 pw = philox4x32_10(glane0 >> 2, (uint32_t)ta, (uint32_t)(ta >> 32), (uint32_t)POMDP_STREAM_ACTION << 24, key.k0, key.k1);
        const uint32_t P[4] = {pw.x, pw.y, pw.z, pw.w};
        uint32_t o4[4], r4[4], a_next[4], dpack = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int o, d;
            typename Env::Reward r;
            const uint32_t lane = glane0 + (uint32_t)j;
            Env::step(sh, p, st[j], a_cur[j], key, lane, o, r, d);
            Env::reset_where(sh, p, st[j], d != 0, key, lane);                 // wave-convergent: every lane calls it
            o4[j] = (uint32_t)o;
            __builtin_memcpy(&r4[j], &r, 4);
            dpack |= (uint32_t)(d != 0) << (8 * j);
            a_next[j] = __umulhi(P[j], n_act);
            a_cur[j] = (int)a_next[j];
        }
        st_stream4(action_w, a_next[0], a_next[1], a_next[2], a_next[3]);
        st_stream4(ob_w, o4[0], o4[1], o4[2], o4[3]);
        st_stream4(reward_w, r4[0], r4[1], r4[2], r4[3]);
        st_stream(done_w, dpack);
        action_w += rec; ob_w += rec; reward_w += rec; done_w += rec / 4;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) Env::store(st[j], state, n, l0 + (uint32_t)j, true);
}

// Smallest batch each quad-per-thread loop takes (1024 lanes per workgroup).  Measured on MI355X, us per fused step at
// 2^17 / 2^18 / 2^19 / 2^20 lanes (profiles/r02_small_shards_gates.txt, r02k_small_shards.txt): RockSample(7,8) quad 1.50 /
// 1.52 / 1.82 / 2.89 against 1.25 / 1.39 / 1.99 with one or two lanes per thread; Tag (table-driven) 1.59 / 1.61 / 1.83 /
// 2.67 against 0.96 / 1.43 / 2.00; Tiger 0.67 / 0.67 / 1.21 / 2.51 against 0.44 / 0.85 / 1.41 / 2.66; Network 2.30 / 2.30 / 2.91 /
// 4.72 against 1.45 / 1.94 / 3.32 / 5.99 — below these sizes every kernel is bound by the latency of one wave's step
// (1.1-2.3 us), and more, lighter waves hide it better than fewer, heavier ones.
// POMDP_QUAD_MIN_LANES overrides all of them at build time for same-box A/B runs (tools/ab_build.sh lib ... -D...).
#ifdef POMDP_QUAD_MIN_LANES
constexpr int64_t QUAD_MIN_ROCK = POMDP_QUAD_MIN_LANES, QUAD_MIN_TAG = POMDP_QUAD_MIN_LANES, QUAD_MIN_GENERIC = POMDP_QUAD_MIN_LANES,
                  QUAD_MIN_NETWORK = POMDP_QUAD_MIN_LANES, QUAD_MIN_BATTLESHIP = POMDP_QUAD_MIN_LANES;
#else
constexpr int64_t QUAD_MIN_ROCK = 1 << 19, QUAD_MIN_TAG = 1 << 19, QUAD_MIN_GENERIC = 1 << 18, QUAD_MIN_NETWORK = 1 << 19,
                  QUAD_MIN_BATTLESHIP = 1 << 18;
#endif

// which kernel the calling thread's most recent fused launch picked (pomdp_last_fused_kernel: bench.py names the kernel
// it timed from this instead of guessing the launcher's choice)
static thread_local char g_last_fused[96] = "";
static void note_fused(const char *kernel, const char *env, const char *variant)
{
    snprintf(g_last_fused, sizeof g_last_fused, "%s<%s%s>", kernel, env, variant);
}

// the same as k launch_step_chain calls at t, t + 1, ..., in one launch.  gen_first: the launch derives the actions of
// call counter t itself (and writes them to `action`) instead of reading them — the caller skips the policy launch.
template <class Env>
static int launch_steps_fused(const typename Env::Params &p, uint32_t *state, int32_t *action, int32_t *ob,
                              typename Env::Reward *reward, uint8_t *done, uint32_t *err, int64_t n, uint64_t seed,
                              uint64_t action_seed, uint32_t lane0, uint64_t t, int k, int flags, int64_t rec, bool gen_first,
                              void *stream)
{
    if (!state || !action || !ob || !reward || !done || bad_range(n, lane0) || (lane0 & 3u) || k < 1) return POMDP_E_BADARG;
    if (n == 0) return 0;
    // two lanes per thread from 2^19 lanes, where the batch does not qualify for a quad-per-thread loop: at 2^18 lanes the
    // one-lane-per-thread loops take 0.90 (RockSample; 1.09 with two) and 1.08 us per step (Tag)
    const bool lpt2 = Env::POOLED_LPT2 && n >= 2 * LPT2_MIN_LANES;
    const bool simple = (flags & POMDP_AUTO_RESET) && n % (lpt2 ? 2 * BLOCK : BLOCK) == 0;
    const dim3 grid(lpt2 ? (unsigned)((n + 2 * BLOCK - 1) / (2 * BLOCK)) : blocks_for(n));
    const int kflags = (flags & POMDP_AUTO_RESET) | (gen_first ? FLAG_GEN_FIRST : 0);
    const int gf = gen_first ? 1 : 0;
#define POMDP_LAUNCH_STEPS(LPT_, SIMPLE_, GRID_)                                                                       \
    do {                                                                                                                 \
        note_fused("steps_kernel", Env::NAME, ", " #LPT_ ", " #SIMPLE_);                                                 \
        hipLaunchKernelGGL((steps_kernel<Env, LPT_, SIMPLE_>), GRID_, dim3(BLOCK), 0, (hipStream_t)stream, state, action, \
                           ob, reward, done, err, n, make_key(seed, t), lane0, kflags, make_key(action_seed, t + 1), k,   \
                           rec, p);                                                                                      \
    } while (0)
    // RockSample's pooled passes exist for any number of lanes per thread; in the fused loop (no load latency to hide)
    // four per thread, with fuller passes, beat two by 5 % from 2^20 lanes up (3.97 vs 4.16 us per step) when the state
    // is one word; with two state words (K > 12) the extra registers cost more (5.04 vs 4.74 us).  Only the geometries
    // an env can take are instantiated.
    // the quad-per-thread loops move 16 bytes at a time (4 for the done bytes): columns that start on such a boundary
    // only, full workgroups of 1024 lanes, auto-reset, policy and env on one Philox key
    const bool quad_ok = ((reinterpret_cast<uintptr_t>(state) | reinterpret_cast<uintptr_t>(action) | reinterpret_cast<uintptr_t>(ob) |
                           reinterpret_cast<uintptr_t>(reward)) & 15u) == 0 && (reinterpret_cast<uintptr_t>(done) & 3u) == 0 &&
                         rec % 4 == 0 && action_seed == seed && (flags & POMDP_AUTO_RESET) && n % (4 * BLOCK) == 0;
    const dim3 qgrid((unsigned)(n / (4 * BLOCK)));
    bool launched = false;
    if constexpr (std::is_same<Env, TagEnv>::value) {
        if (quad_ok && n >= QUAD_MIN_TAG && p.num_opponents == 1) {
            if (k >= 16 && TagEnv::tab_ok(p)) {
                note_fused("tag_steps_quad_kernel", "true", "");
                hipLaunchKernelGGL(tag_steps_quad_kernel<true>, qgrid, dim3(BLOCK), 0, (hipStream_t)stream, state, action, ob, reward,
                                   done, n, make_key(seed, t), lane0, make_key(action_seed, t + 1), k, rec, gf, p);
            } else {
                note_fused("tag_steps_quad_kernel", "false", "");
                hipLaunchKernelGGL(tag_steps_quad_kernel<false>, qgrid, dim3(BLOCK), 0, (hipStream_t)stream, state, action, ob, reward,
                                   done, n, make_key(seed, t), lane0, make_key(action_seed, t + 1), k, rec, gf, p);
            }
            launched = true;
        }
    }
    if constexpr (has_next<Env>::value) {
        if (quad_ok && n >= QUAD_MIN_BATTLESHIP && k <= 255) {
            note_fused("battleship_steps_quad_kernel", Env::NAME, "");
            hipLaunchKernelGGL(battleship_steps_quad_kernel<Env::WORDS / 3>, qgrid, dim3(BLOCK), 0, (hipStream_t)stream, state, action,
                               ob, reward, done, n, make_key(seed, t), lane0, make_key(action_seed, t + 1), k, rec, gf, p);
            launched = true;
        }
    }
    if constexpr (std::is_same<Env, NetworkEnv>::value) {
        if (quad_ok && n >= QUAD_MIN_NETWORK) {
            note_fused("network_steps_quad_kernel", "", "");
#define POMDP_LAUNCH_NET(NB_)                                                                                            \
    hipLaunchKernelGGL(network_steps_quad_kernel<NB_>, qgrid, dim3(BLOCK), 0, (hipStream_t)stream, state, action, ob, reward, \
                       done, n, make_key(seed, t), lane0, make_key(action_seed, t + 1), k, rec, gf, p)
            switch ((p.n_machines + 7) / 8) {
            case 1: POMDP_LAUNCH_NET(1); break;
            case 2: POMDP_LAUNCH_NET(2); break;
            case 3: POMDP_LAUNCH_NET(3); break;
            default: POMDP_LAUNCH_NET(4); break;
            }
#undef POMDP_LAUNCH_NET
            launched = true;
        }
    }
    if constexpr (quad_fused<Env>::value) {
        if (quad_ok && n >= QUAD_MIN_GENERIC) {
            note_fused("steps_quad_generic_kernel", Env::NAME, "");
            hipLaunchKernelGGL((steps_quad_generic_kernel<Env>), qgrid, dim3(BLOCK), 0, (hipStream_t)stream, state, action, ob,
                               reward, done, n, make_key(seed, t), lane0, make_key(action_seed, t + 1), k, rec, gf, p);
            launched = true;
        }
    }
    if constexpr (quad_tab<Env>::value) {
        // from 16 steps per launch on the lane step reads the (position, action) table the workgroup builds first and a
        // thread owns a quad of consecutive lanes (steps_quad_kernel: RockSample and StochasticRock)
        if (quad_ok && n >= QUAD_MIN_ROCK && k >= 16 && p.num_rocks + 5 <= Env::TAB_ACTIONS) {
            note_fused("steps_quad_kernel", Env::NAME, "");
            hipLaunchKernelGGL((steps_quad_kernel<Env>), qgrid, dim3(BLOCK), 0, (hipStream_t)stream, state, action, ob, reward,
                               done, n, make_key(seed, t), lane0, make_key(action_seed, t + 1), k, rec, gf, p);
            launched = true;
        }
    }
    if constexpr (Env::POOLED_ANY_LPT) {
        if (!launched && lpt2 && (flags & POMDP_AUTO_RESET) && n % (4 * BLOCK) == 0 && n >= (1 << 20) && Env::WORDS == 1) {
            POMDP_LAUNCH_STEPS(4, true, qgrid);
            launched = true;
        }
    }
    if constexpr (Env::POOLED_LPT2) {
        if (!launched && lpt2) {
            if (simple) POMDP_LAUNCH_STEPS(2, true, grid); else POMDP_LAUNCH_STEPS(2, false, grid);
            launched = true;
        }
    }
    if constexpr (quad_tab<Env>::value && Env::QUAD_SENSOR) {
        // RockSample's small shards, one lane per thread: the table-driven lane step from 16 steps per launch on
        if (!launched && simple && k >= 16 && p.num_rocks + 5 <= Env::TAB_ACTIONS) {
            note_fused("steps_kernel", Env::NAME, ", 1, true, true");
            hipLaunchKernelGGL((steps_kernel<Env, 1, true, true>), grid, dim3(BLOCK), 0, (hipStream_t)stream, state, action, ob, reward,
                               done, err, n, make_key(seed, t), lane0, kflags, make_key(action_seed, t + 1), k, rec, p);
            launched = true;
        }
    }
    if (!launched) { if (simple) POMDP_LAUNCH_STEPS(1, true, grid); else POMDP_LAUNCH_STEPS(1, false, grid); }
#undef POMDP_LAUNCH_STEPS
    return (int)hipGetLastError();
}

using StochRock1 = RockEnv<1, true>;   // StochasticRockEnv, one / two state words
using StochRock2 = RockEnv<2, true>;

static bool rock_ok(const pomdp_rock_params *p)
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
static int bs_mask_words(const pomdp_battleship_params *p)
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

template <class Env>
static int launch_legal(const typename Env::Params &p, const uint32_t *state, int32_t *list, int32_t *len, int64_t n,
                        int stride, void *stream)
{
    if (!state || !list || !len || n < 0 || stride < 1) return POMDP_E_BADARG;
    if (n == 0) return 0;
    hipLaunchKernelGGL(legal_kernel<Env>, dim3(blocks_for(n)), dim3(BLOCK), 0, (hipStream_t)stream, p, state, list, len,
                       n, stride);
    return (int)hipGetLastError();
}
template <class Env>
static int launch_prob(const typename Env::Params &p, const uint32_t *state, const int32_t *action, const int32_t *ob,
                       double *out, int64_t n, void *stream)
{
    if (!state || !action || !ob || !out || n < 0) return POMDP_E_BADARG;
    if (n == 0) return 0;
    hipLaunchKernelGGL(prob_kernel<Env>, dim3(blocks_for(n)), dim3(BLOCK), 0, (hipStream_t)stream, p, state, action, ob,
                       out, n);
    return (int)hipGetLastError();
}
template <class Env>
static int launch_rollout(const typename Env::Params &p, const uint32_t *state, int64_t n_roots, int64_t sims,
                          int depth, double discount, int flags, uint64_t seed, uint32_t lane0, uint64_t t0, double *ret,
                          int32_t *n_steps, int32_t *first_action, int32_t *last_ob, uint8_t *terminated, void *stream)
{
    if (!state || !ret || !n_steps || !first_action || !last_ob || !terminated || n_roots < 0 || sims < 1 || depth < 0 ||
        bad_range(n_roots * sims, lane0) || (lane0 & 3u))                  // quad-shared blocks travel within the hardware quad
        return POMDP_E_BADARG;
    const int64_t n = n_roots * sims;
    if (n == 0) return 0;
    hipLaunchKernelGGL(rollout_kernel<Env>, dim3(blocks_for(n)), dim3(BLOCK), 0, (hipStream_t)stream, p, state, n_roots,
                       sims, depth, discount, (flags & POMDP_ROLLOUT_ALL_ACTIONS) ? 1 : 0, make_key(seed, t0), lane0, ret,
                       n_steps, first_action, last_ob, terminated);
    return (int)hipGetLastError();
}
static bool belief_ok(const pomdp_rock_belief *b)
{
    return b && b->count && b->measured && b->lkv && b->lkw && b->prob_valuable && b->check_ok;
}
static bool history_ok(const pomdp_history *h, bool rock)
{
    if (!(h && h->size && h->last_action && h->last_ob && (!rock || (h->total_sample && h->total_move && h->move_ok)))) return false;
    if (h->max_size < -1 || h->max_size > 62) return false;                    // window of at most 63 transitions
    return h->max_size < 0 || !rock || (h->ring && h->head);                   // a bounded RockSample history keeps its window
}
static const pomdp_rock_belief NO_BELIEF = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};

template <class Env>
static int launch_belief_update(const typename Env::Params &p, const uint32_t *state, const int32_t *action, const int32_t *ob,
                                const uint8_t *done, const pomdp_rock_belief *b, int64_t n, int flags, void *stream)
{
    if (n == 0) return 0;
    hipLaunchKernelGGL(belief_update_kernel<Env>, dim3(blocks_for(n)), dim3(BLOCK), 0, (hipStream_t)stream, p, state, action,
                       ob, done, *b, n, (flags & POMDP_AUTO_RESET) ? 1 : 0);
    return (int)hipGetLastError();
}
template <class Env>
static int launch_select_target(const typename Env::Params &p, const uint32_t *state, const pomdp_rock_belief *b,
                                int32_t *target, int64_t n, void *stream)
{
    if (n == 0) return 0;
    hipLaunchKernelGGL(select_target_kernel<Env>, dim3(blocks_for(n)), dim3(BLOCK), 0, (hipStream_t)stream, p, state, *b,
                       target, n);
    return (int)hipGetLastError();
}
template <class Env>
static int launch_preferred(const typename Env::Params &p, const uint32_t *state, const pomdp_rock_belief *b,
                            const pomdp_history *h, int32_t *list, int32_t *len, int64_t n, int stride, void *stream)
{
    if (n == 0) return 0;
    hipLaunchKernelGGL(preferred_kernel<Env>, dim3(blocks_for(n)), dim3(BLOCK), 0, (hipStream_t)stream, p, state,
                       b ? *b : NO_BELIEF, *h, list, len, n, stride);
    return (int)hipGetLastError();
}
template <class Env>
static int launch_heuristic_steps(const typename Env::Params &p, uint32_t *state, const pomdp_rock_belief *b,
                                  const pomdp_history *h, int K, int32_t *prev_ob, int32_t *action, int32_t *ob, void *reward,
                                  uint8_t *done, const pomdp_returns *returns, int64_t n, uint64_t seed, uint32_t lane0,
                                  uint64_t t0, int64_t k_steps, int flags, void *stream)
{
    if (n == 0) return 0;
    static const pomdp_returns NO_RETURNS = {0.0, nullptr, nullptr, nullptr};
    constexpr int64_t FUSE_MAX = 64;                      // steps per launch
    for (int64_t s = 0; s < k_steps; s += FUSE_MAX) {
        const int c = (int)(k_steps - s < FUSE_MAX ? k_steps - s : FUSE_MAX);
        bool ring = false;
        if constexpr (Env::HAS_ROCKS) ring = h->max_size >= 0 && K > 0 && h->ring && h->head;
        if constexpr (Env::HAS_ROCKS) {
            if (ring)
                hipLaunchKernelGGL((heuristic_steps_kernel<Env, true>), dim3(blocks_for(n)), dim3(BLOCK), 0, (hipStream_t)stream, p,
                                   state, b ? *b : NO_BELIEF, *h, K, returns ? *returns : NO_RETURNS, prev_ob, action, ob,
                                   (typename Env::Reward *)reward, done, n, make_key(seed, t0 + (uint64_t)s), lane0, flags, c);
        }
        if (!ring)
            hipLaunchKernelGGL((heuristic_steps_kernel<Env, false>), dim3(blocks_for(n)), dim3(BLOCK), 0, (hipStream_t)stream, p,
                               state, b ? *b : NO_BELIEF, *h, K, returns ? *returns : NO_RETURNS, prev_ob, action, ob,
                               (typename Env::Reward *)reward, done, n, make_key(seed, t0 + (uint64_t)s), lane0, flags, c);
        const int rc = (int)hipGetLastError();
        if (rc) return rc;
    }
    return 0;
}
} // namespace pomdp

using namespace pomdp;

// Resolve (env kind, params) to the env type the kernels are instantiated for, validate the params against what the
// packed layouts support, and call f(EnvTag<Env>{}, typed params).
template <class E> struct EnvTag { using Env = E; };
template <class F>
static int dispatch_env(int env, const void *params, F &&f)
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

extern "C" {
#ifdef POMDP_DEV_TIMELINE
int pomdp_dev_timeline(uint64_t *buf) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_timeline), &buf, sizeof(buf)); }
#endif

int pomdp_abi_version(void) { return POMDP_ABI_VERSION; }

const char *pomdp_last_fused_kernel(void) { return g_last_fused; }

const char *pomdp_error_string(int code)
{
    if (code == 0) return "ok";
    if (code == POMDP_E_BADARG) return "bad argument (NULL pointer, negative n, or lane range past 2^32)";
    if (code == POMDP_E_BADPARAMS) return "params outside the supported packed layout";
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "unknown error";
}

int pomdp_rock_reset(const pomdp_rock_params *p, uint32_t *state, int32_t *ob, int64_t n, uint64_t seed,
                     uint32_t lane0, uint64_t t, void *stream)
{
    if (!rock_ok(p)) return POMDP_E_BADPARAMS;
    // reset is identical for StochasticRockEnv (it inherits RockEnv.reset)
    return p->num_rocks <= 12 ? launch_reset<RockEnv<1>>(*p, state, ob, n, seed, lane0, t, stream)
                              : launch_reset<RockEnv<2>>(*p, state, ob, n, seed, lane0, t, stream);
}

int pomdp_rock_step(const pomdp_rock_params *p, uint32_t *state, const int32_t *action, int32_t *ob, int32_t *reward,
                    uint8_t *done, uint32_t *err, int64_t n, uint64_t seed, uint32_t lane0, uint64_t t, int flags,
                    void *stream)
{
    if (!rock_ok(p)) return POMDP_E_BADPARAMS;
    if (p->stochastic)
        return p->num_rocks <= 12
                   ? launch_step<StochRock1>(*p, state, action, ob, reward, done, err, n, seed, lane0, t, flags, stream)
                   : launch_step<StochRock2>(*p, state, action, ob, reward, done, err, n, seed, lane0, t, flags, stream);
    return p->num_rocks <= 12
               ? launch_step<RockEnv<1>>(*p, state, action, ob, reward, done, err, n, seed, lane0, t, flags, stream)
               : launch_step<RockEnv<2>>(*p, state, action, ob, reward, done, err, n, seed, lane0, t, flags, stream);
}

int pomdp_tag_reset(const pomdp_tag_params *p, uint32_t *state, int32_t *ob, int64_t n, uint64_t seed, uint32_t lane0,
                    uint64_t t, void *stream)
{
    if (!p || p->num_opponents < 1 || p->num_opponents > 4) return POMDP_E_BADPARAMS;
    return launch_reset<TagEnv>(*p, state, ob, n, seed, lane0, t, stream);
}

int pomdp_tag_step(const pomdp_tag_params *p, uint32_t *state, const int32_t *action, int32_t *ob, float *reward,
                   uint8_t *done, uint32_t *err, int64_t n, uint64_t seed, uint32_t lane0, uint64_t t, int flags,
                   void *stream)
{
    if (!p || p->num_opponents < 1 || p->num_opponents > 4) return POMDP_E_BADPARAMS;
    return launch_step<TagEnv>(*p, state, action, ob, reward, done, err, n, seed, lane0, t, flags, stream);
}

int pomdp_battleship_reset(const pomdp_battleship_params *p, uint32_t *state, int32_t *ob, int64_t n, uint64_t seed,
                           uint32_t lane0, uint64_t t, void *stream)
{
    switch (bs_mask_words(p)) {
    case 1: return launch_reset<BattleShipEnv<1>>(*p, state, ob, n, seed, lane0, t, stream);
    case 2: return launch_reset<BattleShipEnv<2>>(*p, state, ob, n, seed, lane0, t, stream);
    case 3: return launch_reset<BattleShipEnv<3>>(*p, state, ob, n, seed, lane0, t, stream);
    case 4: return launch_reset<BattleShipEnv<4>>(*p, state, ob, n, seed, lane0, t, stream);
    default: return POMDP_E_BADPARAMS;
    }
}

int pomdp_battleship_step(const pomdp_battleship_params *p, uint32_t *state, const int32_t *action, int32_t *ob,
                          int32_t *reward, uint8_t *done, uint32_t *err, int64_t n, uint64_t seed, uint32_t lane0,
                          uint64_t t, int flags, void *stream)
{
    switch (bs_mask_words(p)) {
    case 1: return launch_step<BattleShipEnv<1>>(*p, state, action, ob, reward, done, err, n, seed, lane0, t, flags, stream);
    case 2: return launch_step<BattleShipEnv<2>>(*p, state, action, ob, reward, done, err, n, seed, lane0, t, flags, stream);
    case 3: return launch_step<BattleShipEnv<3>>(*p, state, action, ob, reward, done, err, n, seed, lane0, t, flags, stream);
    case 4: return launch_step<BattleShipEnv<4>>(*p, state, action, ob, reward, done, err, n, seed, lane0, t, flags, stream);
    default: return POMDP_E_BADPARAMS;
    }
}

int pomdp_tiger_reset(const pomdp_tiger_params *p, uint32_t *state, int32_t *ob, int64_t n, uint64_t seed,
                      uint32_t lane0, uint64_t t, void *stream)
{
    if (!p) return POMDP_E_BADPARAMS;
    return launch_reset<TigerEnv>(*p, state, ob, n, seed, lane0, t, stream);
}

int pomdp_tiger_step(const pomdp_tiger_params *p, uint32_t *state, const int32_t *action, int32_t *ob, int32_t *reward,
                     uint8_t *done, uint32_t *err, int64_t n, uint64_t seed, uint32_t lane0, uint64_t t, int flags,
                     void *stream)
{
    if (!p) return POMDP_E_BADPARAMS;
    return launch_step<TigerEnv>(*p, state, action, ob, reward, done, err, n, seed, lane0, t, flags, stream);
}

int pomdp_network_reset(const pomdp_network_params *p, uint32_t *state, int32_t *ob, int64_t n, uint64_t seed,
                        uint32_t lane0, uint64_t t, void *stream)
{
    if (!p || p->n_machines < 1 || p->n_machines > 32) return POMDP_E_BADPARAMS;
    return launch_reset<NetworkEnv>(*p, state, ob, n, seed, lane0, t, stream);
}

int pomdp_network_step(const pomdp_network_params *p, uint32_t *state, const int32_t *action, int32_t *ob,
                       float *reward, uint8_t *done, uint32_t *err, int64_t n, uint64_t seed, uint32_t lane0,
                       uint64_t t, int flags, void *stream)
{
    if (!p || p->n_machines < 1 || p->n_machines > 32) return POMDP_E_BADPARAMS;
    return launch_step<NetworkEnv>(*p, state, action, ob, reward, done, err, n, seed, lane0, t, flags, stream);
}

int pomdp_step(const pomdp_step_args *a, const int32_t *action, uint64_t t, void *stream)
{
    if (!a || !a->params) return POMDP_E_BADARG;
    switch (a->env) {
    case POMDP_ENV_ROCK:
        return pomdp_rock_step((const pomdp_rock_params *)a->params, a->state, action, a->ob, (int32_t *)a->reward, a->done, a->err,
                               a->n, a->seed, a->lane0, t, a->flags, stream);
    case POMDP_ENV_TAG:
        return pomdp_tag_step((const pomdp_tag_params *)a->params, a->state, action, a->ob, (float *)a->reward, a->done, a->err,
                              a->n, a->seed, a->lane0, t, a->flags, stream);
    case POMDP_ENV_BATTLESHIP:
        return pomdp_battleship_step((const pomdp_battleship_params *)a->params, a->state, action, a->ob, (int32_t *)a->reward,
                                     a->done, a->err, a->n, a->seed, a->lane0, t, a->flags, stream);
    case POMDP_ENV_TIGER:
        return pomdp_tiger_step((const pomdp_tiger_params *)a->params, a->state, action, a->ob, (int32_t *)a->reward, a->done,
                                a->err, a->n, a->seed, a->lane0, t, a->flags, stream);
    case POMDP_ENV_NETWORK:
        return pomdp_network_step((const pomdp_network_params *)a->params, a->state, action, a->ob, (float *)a->reward, a->done,
                                  a->err, a->n, a->seed, a->lane0, t, a->flags, stream);
    default: return POMDP_E_BADARG;
    }
}

// Scalar mode (one lane, outputs in pinned host memory): the kernel publishes its outputs through a flag in pinned host
// memory with a system-scope release and the host polls the flag — the wake-up of a blocking synchronisation is most of
// a scalar step otherwise.  The stream stays ordered, so the next launch needs no wait.  Anything else, or a flag that
// does not show up within a millisecond (a failed launch), takes hipStreamSynchronize.
struct ScalarWait {
    uint32_t *flag = nullptr;
    uint32_t seq = 0;
    // the one allocation the library makes (documented in include/pomdp_hip.h): 64 bytes of pinned host memory per calling
    // thread, visible to every device (portable), freed when the thread ends
    ~ScalarWait() { if (flag) (void)hipHostFree(flag); }
    bool arm(int64_t n)
    {
        if (n != 1) return false;
        if (!flag) {
            if (hipHostMalloc((void **)&flag, 64, hipHostMallocPortable) != hipSuccess) { flag = nullptr; return false; }
            *flag = 0;
        }
        tl_host_flag = flag; tl_flag_value = ++seq;
        return true;
    }
    int wait(bool armed, int rc, void *stream)
    {
        const bool taken = armed && tl_host_flag == nullptr;   // the launcher passed the flag to its kernel
        tl_host_flag = nullptr;
        if (rc) return rc;
        if (taken) {
            // a one-lane launch on an idle stream publishes its flag ~10 us after the call; with earlier work queued on the
            // stream the flag cannot appear before that work is done, so the poll is bounded at 100 us (ten launches' worth)
            // and the blocking wait takes over — a query of the stream before every step would cost the common case more
            const auto t0 = std::chrono::steady_clock::now();
            for (uint32_t spins = 0;; ++spins) {
                if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) return 0;
                if ((spins & 255u) == 255u && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(100)) break;
            }
        }
        return (int)hipStreamSynchronize((hipStream_t)stream);
    }
};
static thread_local ScalarWait tl_scalar_wait;

int pomdp_step_sync(const pomdp_step_args *a, const int32_t *action, uint64_t t, void *stream)
{
    const bool armed = a && tl_scalar_wait.arm(a->n);
    return tl_scalar_wait.wait(armed, pomdp_step(a, action, t, stream), stream);
}

int pomdp_reset_sync(int env, const void *params, uint32_t *state, int32_t *ob, int64_t n, uint64_t seed, uint32_t lane0,
                     uint64_t t, void *stream)
{
    if (!params) return POMDP_E_BADARG;
    const bool armed = tl_scalar_wait.arm(n);
    int rc;
    switch (env) {
    case POMDP_ENV_ROCK: rc = pomdp_rock_reset((const pomdp_rock_params *)params, state, ob, n, seed, lane0, t, stream); break;
    case POMDP_ENV_TAG: rc = pomdp_tag_reset((const pomdp_tag_params *)params, state, ob, n, seed, lane0, t, stream); break;
    case POMDP_ENV_BATTLESHIP:
        rc = pomdp_battleship_reset((const pomdp_battleship_params *)params, state, ob, n, seed, lane0, t, stream); break;
    case POMDP_ENV_TIGER: rc = pomdp_tiger_reset((const pomdp_tiger_params *)params, state, ob, n, seed, lane0, t, stream); break;
    case POMDP_ENV_NETWORK: rc = pomdp_network_reset((const pomdp_network_params *)params, state, ob, n, seed, lane0, t, stream); break;
    default: rc = POMDP_E_BADARG;
    }
    return tl_scalar_wait.wait(armed, rc, stream);
}

int pomdp_stream_sync(void *stream) { return (int)hipStreamSynchronize((hipStream_t)stream); }

int pomdp_synthetic_actions(int32_t *action, int64_t n, uint64_t seed, uint32_t lane0, uint64_t t, uint32_t n_actions,
                            void *stream)
{
    if (!action || bad_range(n, lane0) || (lane0 & 3u) || (reinterpret_cast<uintptr_t>(action) & 15u) || n_actions == 0)
        return POMDP_E_BADARG;
    if (n == 0) return 0;
    hipLaunchKernelGGL(synthetic_actions_kernel, dim3(grid_for((n + 3) / 4)), dim3(BLOCK), 0, (hipStream_t)stream, action, n,
                       make_key(seed, t), lane0 >> 2, n_actions);
    return (int)hipGetLastError();
}

// the env's action count from its params; 0 = unknown env
static uint32_t env_action_count(int env, const void *params)
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

// params and buffers of the C-side episode loops, checked before anything is enqueued
static int check_driver_args(int env, const void *params, const void *state, const void *action, const void *ob,
                             const void *reward, const void *done, int64_t n, uint32_t lane0, int64_t k_steps)
{
    if (!params || !state || !action || !ob || !reward || !done || k_steps < 0 || bad_range(n, lane0) || (lane0 & 3u))
        return POMDP_E_BADARG;
    return dispatch_env(env, params, [](auto, const auto &) { return 0; });      // POMDP_E_BADPARAMS / unknown env
}

int pomdp_rollout_synthetic(int env, const void *params, uint32_t *state, int32_t *action, int32_t *ob, void *reward,
                            uint8_t *done, uint32_t *err, int64_t n, uint64_t seed, uint64_t action_seed,
                            uint32_t lane0, uint64_t t0, int64_t k_steps, int flags, void *stream)
{
    int rc = check_driver_args(env, params, state, action, ob, reward, done, n, lane0, k_steps);
    if (rc) return rc;
    if (k_steps == 0 || n == 0) return 0;
    const uint32_t n_actions = env_action_count(env, params);
    if (action_seed == seed && (flags & POMDP_FUSE_STEPS)) {
        // chained and fused: up to FUSE_MAX consecutive steps share one launch (steps_kernel and its quad-per-thread
        // forms); every step also leaves the actions of the following call counter in `action`; the first launch
        // derives the actions of t0 itself
        constexpr int64_t FUSE_MAX = 64;
        for (int64_t s = 0; s < k_steps; s += FUSE_MAX) {
            const int c = (int)(k_steps - s < FUSE_MAX ? k_steps - s : FUSE_MAX);
            const uint64_t t = t0 + (uint64_t)s;
            rc = dispatch_env(env, params, [&](auto tag, const auto &p) {
                using E = typename decltype(tag)::Env;
                return launch_steps_fused<E>(p, state, action, ob, (typename E::Reward *)reward, done, err, n, seed,
                                             action_seed, lane0, t, c, flags, 0, s == 0, stream);
            });
            if (rc) return rc;
        }
        return 0;
    }
    // actions of the first step from the stand-alone policy kernel
    rc = pomdp_synthetic_actions(action, n, action_seed, lane0, t0, n_actions, stream);
    if (rc) return rc;
    if (action_seed == seed) {
        // chained: one launch per step, which also leaves the actions of the following call counter in `action`
        for (int64_t s = 0; s < k_steps; ++s) {
            const uint64_t t = t0 + (uint64_t)s;
            rc = dispatch_env(env, params, [&](auto tag, const auto &p) {
                using E = typename decltype(tag)::Env;
                return launch_step_chain<E>(p, state, action, ob, (typename E::Reward *)reward, done, err, n, seed, action_seed,
                                            lane0, t, flags, stream);
            });
            if (rc) return rc;
        }
        return 0;
    }
    // distinct policy key: policy launch + step launch per step
    for (int64_t s = 0; s < k_steps; ++s) {
        const uint64_t t = t0 + (uint64_t)s;
        if (s > 0 && (rc = pomdp_synthetic_actions(action, n, action_seed, lane0, t, n_actions, stream))) return rc;
        rc = dispatch_env(env, params, [&](auto tag, const auto &p) {
            using E = typename decltype(tag)::Env;
            return launch_step<E>(p, state, action, ob, (typename E::Reward *)reward, done, err, n, seed, lane0, t, flags, stream);
        });
        if (rc) return rc;
    }
    return pomdp_synthetic_actions(action, n, action_seed, lane0, t0 + (uint64_t)k_steps, n_actions, stream);
}

int pomdp_collect_synthetic(int env, const void *params, uint32_t *state, int32_t *action, int32_t *ob, void *reward,
                            uint8_t *done, uint32_t *err, int64_t n, uint64_t seed, uint32_t lane0, uint64_t t0,
                            int64_t k_steps, int64_t pitch, int flags, void *stream)
{
    int rc = check_driver_args(env, params, state, action, ob, reward, done, n, lane0, k_steps);
    if (rc) return rc;
    if (pitch < n || !(flags & POMDP_AUTO_RESET)) return POMDP_E_BADARG;
    if (k_steps == 0 || n == 0) return 0;
    constexpr int64_t FUSE_MAX = 64;
    for (int64_t s = 0; s < k_steps; s += FUSE_MAX) {      // the first launch writes row 0 (the actions of t0) itself
        const int c = (int)(k_steps - s < FUSE_MAX ? k_steps - s : FUSE_MAX);
        rc = dispatch_env(env, params, [&](auto tag, const auto &p) {
            using E = typename decltype(tag)::Env;
            using R = typename E::Reward;
            return launch_steps_fused<E>(p, state, action + s * pitch, ob + s * pitch, (R *)reward + s * pitch,
                                         done + s * pitch, err, n, seed, seed, lane0, t0 + (uint64_t)s, c, flags, pitch,
                                         s == 0, stream);
        });
        if (rc) return rc;
    }
    return 0;
}

int pomdp_collect(const pomdp_collect_args *a, uint64_t t0, int64_t k_steps, void *stream)
{
    if (!a) return POMDP_E_BADARG;
    return pomdp_collect_synthetic(a->env, a->params, a->state, a->action, a->ob, a->reward, a->done, a->err, a->n, a->seed,
                                   a->lane0, t0, k_steps, a->pitch, a->flags, stream);
}

int pomdp_legal_actions(int env, const void *params, const uint32_t *state, int32_t *list, int32_t *len, int64_t n,
                        int stride, void *stream)
{
    if (!params) return POMDP_E_BADARG;
    return dispatch_env(env, params, [&](auto tag, const auto &p) {
        using E = typename decltype(tag)::Env;
        return launch_legal<E>(p, state, list, len, n, stride, stream);
    });
}

int pomdp_compute_prob(int env, const void *params, const uint32_t *state, const int32_t *action, const int32_t *ob,
                       double *out, int64_t n, void *stream)
{
    if (!params) return POMDP_E_BADARG;
    return dispatch_env(env, params, [&](auto tag, const auto &p) {
        using E = typename decltype(tag)::Env;
        return launch_prob<E>(p, state, action, ob, out, n, stream);
    });
}

int pomdp_rollout(int env, const void *params, const uint32_t *root_state, int64_t n_roots, int64_t sims_per_root,
                  int depth, double discount, int flags, uint64_t seed, uint32_t lane0, uint64_t t0, double *ret,
                  int32_t *n_steps, int32_t *first_action, int32_t *last_ob, uint8_t *terminated, void *stream)
{
    if (!params) return POMDP_E_BADARG;
    return dispatch_env(env, params, [&](auto tag, const auto &p) {
        using E = typename decltype(tag)::Env;
        return launch_rollout<E>(p, root_state, n_roots, sims_per_root, depth, discount, flags, seed, lane0, t0, ret, n_steps,
                                 first_action, last_ob, terminated, stream);
    });
}

int pomdp_rock_belief_reset(const pomdp_rock_params *p, const pomdp_rock_belief *b, const uint8_t *where, int64_t n,
                            void *stream)
{
    if (!p || !belief_ok(b) || n < 0) return POMDP_E_BADARG;
    if (!rock_ok(p)) return POMDP_E_BADPARAMS;
    if (n == 0) return 0;
    hipLaunchKernelGGL(belief_reset_kernel, dim3(blocks_for(n)), dim3(BLOCK), 0, (hipStream_t)stream, *b, p->num_rocks, where, n);
    return (int)hipGetLastError();
}

int pomdp_rock_belief_refresh(const pomdp_rock_params *p, const pomdp_rock_belief *b, int64_t n, void *stream)
{
    if (!p || !belief_ok(b) || n < 0) return POMDP_E_BADARG;
    if (!rock_ok(p)) return POMDP_E_BADPARAMS;
    if (n == 0) return 0;
    hipLaunchKernelGGL(belief_refresh_kernel, dim3(blocks_for(n)), dim3(BLOCK), 0, (hipStream_t)stream, *b, p->num_rocks, n);
    return (int)hipGetLastError();
}

int pomdp_rock_belief_update(const pomdp_rock_params *p, const uint32_t *state, const int32_t *action, const int32_t *ob,
                             const uint8_t *done, const pomdp_rock_belief *b, int64_t n, int flags, void *stream)
{
    if (!p || !state || !action || !ob || !done || !belief_ok(b) || n < 0) return POMDP_E_BADARG;
    if (!rock_ok(p)) return POMDP_E_BADPARAMS;
    if (p->num_rocks <= 12) return launch_belief_update<RockEnv<1>>(*p, state, action, ob, done, b, n, flags, stream);
    return launch_belief_update<RockEnv<2>>(*p, state, action, ob, done, b, n, flags, stream);
}

int pomdp_rock_select_target(const pomdp_rock_params *p, const uint32_t *state, const pomdp_rock_belief *b, int32_t *target,
                             int64_t n, void *stream)
{
    if (!p || !state || !target || !belief_ok(b) || n < 0) return POMDP_E_BADARG;
    if (!rock_ok(p)) return POMDP_E_BADPARAMS;
    if (p->num_rocks <= 12) return launch_select_target<RockEnv<1>>(*p, state, b, target, n, stream);
    return launch_select_target<RockEnv<2>>(*p, state, b, target, n, stream);
}

static int history_rocks(int env, const void *params)
{
    if (env != POMDP_ENV_ROCK) return (env >= POMDP_ENV_TAG && env <= POMDP_ENV_NETWORK) ? 0 : -1;
    const pomdp_rock_params *p = (const pomdp_rock_params *)params;
    return (p && rock_ok(p)) ? p->num_rocks : -1;
}

int pomdp_history_clear(int env, const void *params, const pomdp_history *h, const uint8_t *where, int64_t n, void *stream)
{
    const int K = history_rocks(env, params);
    if (K < 0 || !history_ok(h, K > 0) || n < 0) return POMDP_E_BADARG;
    if (n == 0) return 0;
    hipLaunchKernelGGL(history_clear_kernel, dim3(blocks_for(n)), dim3(BLOCK), 0, (hipStream_t)stream, *h, K, where, n);
    return (int)hipGetLastError();
}

int pomdp_history_append(int env, const void *params, const pomdp_history *h, const int32_t *observation,
                         const int32_t *action, const int32_t *next_observation, const uint8_t *done, int64_t n, int flags,
                         void *stream)
{
    const int K = history_rocks(env, params);
    if (K < 0 || !history_ok(h, K > 0) || !observation || !action || !next_observation || !done || n < 0)
        return POMDP_E_BADARG;
    if (n == 0) return 0;
    hipLaunchKernelGGL(history_append_kernel, dim3(blocks_for(n)), dim3(BLOCK), 0, (hipStream_t)stream, *h, K, observation,
                       action, next_observation, done, n, (flags & POMDP_AUTO_RESET) ? 1 : 0);
    return (int)hipGetLastError();
}

int pomdp_preferred_actions(int env, const void *params, const uint32_t *state, const pomdp_rock_belief *b,
                            const pomdp_history *h, int32_t *list, int32_t *len, int64_t n, int stride, void *stream)
{
    if (!params || !state || !list || !len || n < 0 || stride < 1) return POMDP_E_BADARG;
    if (!history_ok(h, env == POMDP_ENV_ROCK) || (env == POMDP_ENV_ROCK && !belief_ok(b))) return POMDP_E_BADARG;
    return dispatch_env(env, params, [&](auto tag, const auto &p) {
        using E = typename decltype(tag)::Env;
        return launch_preferred<E>(p, state, b, h, list, len, n, stride, stream);
    });
}

int pomdp_pick_actions(const int32_t *list, const int32_t *len, int stride, int32_t *action, int64_t n, uint64_t seed,
                       uint32_t lane0, uint64_t t, void *stream)
{
    if (!list || !len || !action || stride < 1 || bad_range(n, lane0)) return POMDP_E_BADARG;
    if (n == 0) return 0;
    hipLaunchKernelGGL(pick_actions_kernel, dim3(blocks_for(n)), dim3(BLOCK), 0, (hipStream_t)stream, list, len, stride,
                       action, n, make_key(seed, t), lane0);
    return (int)hipGetLastError();
}

int pomdp_heuristic_steps(int env, const void *params, uint32_t *state, const pomdp_rock_belief *b, const pomdp_history *h,
                          int32_t *prev_ob, int32_t *action, int32_t *ob, void *reward, uint8_t *done,
                          const pomdp_returns *returns, int64_t n, uint64_t seed, uint32_t lane0, uint64_t t0,
                          int64_t k_steps, int flags, void *stream)
{
    const int K = history_rocks(env, params);
    // (the policy's block — and RockSample's STEP / RESET blocks — are shared by global lanes 4 q .. 4 q + 3 and travel
    // within the hardware quad: a shard has to start on such a boundary)
    if (K < 0 || !params || !state || !prev_ob || !action || !ob || !reward || !done || k_steps < 0 || bad_range(n, lane0) || (lane0 & 3u))
        return POMDP_E_BADARG;
    if (!history_ok(h, K > 0) || (K > 0 && !belief_ok(b))) return POMDP_E_BADARG;
    if (returns && !(returns->ret && returns->disc && returns->ret_done)) return POMDP_E_BADARG;
    return dispatch_env(env, params, [&](auto tag, const auto &p) {
        using E = typename decltype(tag)::Env;
        return launch_heuristic_steps<E>(p, state, b, h, K, prev_ob, action, ob, reward, done, returns, n, seed, lane0, t0,
                                         k_steps, flags, stream);
    });
}

int pomdp_philox_blocks(const uint32_t *ctr_key, uint32_t *out, int64_t n_blocks, void *stream)
{
    if (!ctr_key || !out || n_blocks < 0) return POMDP_E_BADARG;
    if (n_blocks == 0) return 0;
    hipLaunchKernelGGL(philox_blocks_kernel, dim3((unsigned)((n_blocks + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       ctr_key, out, n_blocks);
    return (int)hipGetLastError();
}

} // extern "C"
