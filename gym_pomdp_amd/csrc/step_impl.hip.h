// step_impl.hip.h — the single-step kernels (reset, step, the quad-per-thread step of the RockSample family) and their
// launchers.  Included by the translation units that instantiate them (step_rock.hip, step_other.hip).
#pragma once
#include "kernels_common.hip.h"

namespace pomdp {

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
    bool fresh[LPT];
    uint32_t glane[LPT];
    int a_next[LPT];
#pragma unroll
    for (int j = 0; j < LPT; ++j) {
        fresh[j] = live[j] && d[j] && auto_reset; glane[j] = lane0 + idx[j]; a_next[j] = 0;
        if constexpr (has_next<Env>::value) { if (fresh[j]) Env::load_next(st[j], state_w, n, rel[j]); }   // the cached board moves in
    }
    Fin::run(sh, p, st, fresh, key, glane, akey, (uint32_t)n_act, a_next, aux, o);
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
}

// ONE step of RockSample with the caller's actions (env.step()) and quads of consecutive lanes per thread: a quad's
// STEP block is the thread's own (one Philox block for four lane-steps, computed under the latency of the loads, no
// exchange through LDS), state / action / ob / reward move as 16-byte accesses and the four done bytes as one word —
// 105 VALU instructions per lane-step where step_kernel<Env, 2> issues 187.
// TILES (the product runs 1; -DPOMDP_STEP_TILES=2 builds the B arm): a one-step launch is 9 MB of loads, the lane steps and
// 13 MB of stores; with one quad per thread every workgroup of the launch is resident at once and in the same phase, so the
// three run one after the other (round 5's timeline: 1.3 + 2 + 1.2 us).  With TILES > 1 a thread walks TILES
// tiles of 1024 lanes, software-pipelined: the loads of ALL its tiles are issued first, in tile order — HBM serves them
// roughly in that order, and gfx9 returns them in order, so `s_waitcnt vmcnt(loads still to come)` releases tile 0 while the
// later tiles are still streaming in —, then tile by tile the lane steps and the stores: tile i's stores drain under tile
// i + 1's lane steps (its loads were issued BEFORE those stores, so waiting for them does not wait for the stores).
// Workgroup w owns tiles w, w + G, ... of a G-workgroup launch.  Measured at 2^20 lanes (profiles/r06_step_tiles.txt): two
// tiles 7.6 against 6.5 us per call (Network 10.6 against 9.2) — a 2^20-lane batch is 4096 quad-waves, four per SIMD; two
// tiles per thread leave two, and two waves fill a SIMD's issue slots so much worse during the lane steps (the ~500
// dependent-heavy instructions of a tile) that the overlap is lost twice over.
// Same contract as step_kernel: a lane whose action is out of range is left untouched and counted in *err, without
// auto-reset a done lane stays frozen.  n a multiple of TILES x 1024 lanes on 16-byte column boundaries only (launch_step);
// anything else takes fewer tiles, or step_kernel.
template <class Env, int TILES>
__global__ __launch_bounds__(BLOCK) void step_quad_kernel(uint32_t *__restrict__ state, const int32_t *__restrict__ action,
                                                          int32_t *__restrict__ ob, int32_t *__restrict__ reward,
                                                          uint8_t *__restrict__ done, uint32_t *__restrict__ err, int64_t n,
                                                          RngKey key, uint32_t lane0, int flags, const typename Env::Params p)
{
    constexpr int W = Env::WORDS;
    using S = typename Env::S;
    __shared__ typename Env::Shared sh;
    const bool auto_reset = flags & POMDP_AUTO_RESET;
    uint32_t l0[TILES];
    u32x4 s_lo[TILES], s_hi[TILES], a4[TILES];
    uint32_t dn[TILES];
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
        l0[t] = (blockIdx.x + (uint32_t)t * gridDim.x) * (uint32_t)(4 * BLOCK) + 4u * threadIdx.x;
        s_lo[t] = ld_stream4(state + l0[t]);
        s_hi[t] = u32x4{0, 0, 0, 0};
        if (W == 2) s_hi[t] = ld_stream4(state + n + l0[t]);
        a4[t] = ld_stream4(reinterpret_cast<const uint32_t *>(action) + l0[t]);
        dn[t] = auto_reset ? 0u : ld_stream(reinterpret_cast<const uint32_t *>(done + l0[t]));   // frozen lanes (the reference would assert)
    }
    const auto staged = Env::stage_load(p, (int)threadIdx.x);
    Env::stage_store(sh, staged, (int)threadIdx.x);
    __syncthreads();
    const uint32_t n_act = (uint32_t)Env::n_actions(p);
    uint32_t n_bad = 0;
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
        const uint32_t glane0 = lane0 + l0[t];
        uint32_t *const done_w = reinterpret_cast<uint32_t *>(done + l0[t]);
        // the quad's words depend on lane ids only: tile 0's Philox runs under the load latency, tile t's under tile t - 1's stores
        constexpr uint32_t SENSOR_BLOCK = Env::SENSOR_BLOCK;
        const uint4 sw = Env::quad_block(key, glane0, SENSOR_BLOCK);
        // a lane's step draws EITHER its sensor reading OR (done) its next episode: both from this one block
        const uint32_t H[4] = {sw.x, sw.y, sw.z, sw.w}, R[4] = {sw.x, sw.y, sw.z, sw.w};
        uint32_t G[4] = {0, 0, 0, 0};
        if constexpr (Env::STOCHASTIC) { const uint4 gw = Env::quad_block(key, glane0, 0u); G[0] = gw.x; G[1] = gw.y; G[2] = gw.z; G[3] = gw.w; }
        typename Env::State st[4];
        typename Env::Aux aux[4];
        int r[4], d[4];
        bool live[4], fresh[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool valid = a4[t][j] < n_act, was_done = ((dn[t] >> (8 * j)) & 0xFFu) != 0u;
            live[j] = valid && !was_done;
            n_bad += (uint32_t)(!valid && !was_done);
            st[j].s = (S)((uint64_t)s_lo[t][j] | ((uint64_t)s_hi[t][j] << 32));
            typename Env::State nx = st[j];
            Env::step_pre(sh, p, nx, valid ? (int)a4[t][j] : 0, r[j], d[j], aux[j]);
            bool acts = live[j];
            if constexpr (Env::STOCHASTIC)                                             // applied iff binomial(1, p_move) says so (rock.py:443)
                acts = acts && (Env::k53_le(G[j], (uint32_t)(p.act_thr >> 26), (uint32_t)p.act_thr & Env::LO_MASK,
                                            [&]() { return Env::elem(Env::quad_block(key, glane0, 1u), (uint32_t)j); }) != (p.act_gt != 0));
            if (acts) st[j] = nx; else { r[j] = 0; d[j] = live[j] ? 0 : (int)was_done; aux[j].want = false; }
            fresh[j] = acts && d[j] != 0 && auto_reset;                                // done lanes start a new episode
        }
        // (A CHECK neither moves the agent nor ends the episode: the sensor reads the state the step left.)
        uint32_t o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            o[j] = (uint32_t)Env::sensor_ob(sh, st[j], aux[j], H[j], [&]() { return Env::elem(Env::quad_block(key, glane0, SENSOR_BLOCK + 1u), (uint32_t)j); });
        st_stream4(reinterpret_cast<uint32_t *>(ob) + l0[t], o[0], o[1], o[2], o[3]);
        st_stream4(reinterpret_cast<uint32_t *>(reward) + l0[t], (uint32_t)r[0], (uint32_t)r[1], (uint32_t)r[2], (uint32_t)r[3]);
        st_stream(done_w, (uint32_t)d[0] | ((uint32_t)d[1] << 8) | ((uint32_t)d[2] << 16) | ((uint32_t)d[3] << 24));
#pragma unroll
        for (int j = 0; j < 4; ++j)
            st[j].s = fresh[j] ? Env::fresh_state(p, R[j], key, glane0 + (uint32_t)j) : st[j].s;
        st_stream4(state + l0[t], (uint32_t)st[0].s, (uint32_t)st[1].s, (uint32_t)st[2].s, (uint32_t)st[3].s);
        if (W == 2)
            st_stream4(state + n + l0[t], (uint32_t)((uint64_t)st[0].s >> 32), (uint32_t)((uint64_t)st[1].s >> 32),
                       (uint32_t)((uint64_t)st[2].s >> 32), (uint32_t)((uint64_t)st[3].s >> 32));
    }
    if (n_bad && err) atomicAdd(err, n_bad);
}

// ONE step of Network with the caller's actions and a quad of consecutive lanes per thread (env.step() from 2^19 lanes).
// The top 16 bits of a step's doubles come from blocks shared by the quad, two draws per word (network.hip.h): the thread's
// own three blocks — computed under the latency of its loads — hold draws 0 .. 5 of each of its four lanes, which serve all
// but one lane-step in 10^4 under a random policy; a thread with a lane that has more to draw runs on block by block.  State,
// action, ob and reward move as 16-byte accesses.  Same contract as step_kernel: an out-of-range action leaves the lane
// untouched and is counted in *err; without auto-reset a lane whose done flag is set stays frozen (Network itself never
// sets it).
// TILES: as step_quad_kernel — every tile's loads first, in tile order, then lane steps and stores tile by tile.
template <class Env, int TILES>   // NetworkEnv (a template so that the header may be included by several translation units)
__global__ __launch_bounds__(BLOCK) void network_step_quad_kernel(uint32_t *__restrict__ state, const int32_t *__restrict__ action,
                                                                  int32_t *__restrict__ ob, float *__restrict__ reward,
                                                                  uint8_t *__restrict__ done, uint32_t *__restrict__ err, int64_t n,
                                                                  RngKey key, uint32_t lane0, int flags, const typename Env::Params p)
{
    __shared__ typename Env::Shared sh;
    const bool auto_reset = flags & POMDP_AUTO_RESET;
    uint32_t l0[TILES], dn_t[TILES];
    u32x4 s4_t[TILES], a4_t[TILES];
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
        l0[t] = (blockIdx.x + (uint32_t)t * gridDim.x) * (uint32_t)(4 * BLOCK) + 4u * threadIdx.x;
        s4_t[t] = ld_stream4(state + l0[t]);
        a4_t[t] = ld_stream4(reinterpret_cast<const uint32_t *>(action) + l0[t]);
        dn_t[t] = auto_reset ? 0u : ld_stream(reinterpret_cast<const uint32_t *>(done + l0[t]));
    }
    Env::stage(sh, p, (int)threadIdx.x);
    __syncthreads();
    const typename Env::Thr T = Env::thresholds(p);
    const int M2 = 2 * p.n_machines;
    const uint32_t n_act = (uint32_t)Env::n_actions(p);
    const uint32_t all_up = p.n_machines >= 32 ? 0xFFFFFFFFu : ((1u << p.n_machines) - 1u);
    uint32_t n_bad = 0;
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
    const uint32_t glane0 = lane0 + l0[t];
    uint32_t *const done_w = reinterpret_cast<uint32_t *>(done + l0[t]);
    const u32x4 s4 = s4_t[t], a4 = a4_t[t];
    const uint32_t dn = dn_t[t];
    // the quad's blocks depend on lane ids only: Philox under the load latency (tile 0) / the previous tile's stores
    const uint4 q0 = Env::quad_block(key, glane0, 0u), q1 = Env::quad_block(key, glane0, 1u), q2 = Env::quad_block(key, glane0, 2u);
    const uint32_t W0[4] = {q0.x, q0.y, q0.z, q0.w}, W1[4] = {q1.x, q1.y, q1.z, q1.w}, W2[4] = {q2.x, q2.y, q2.z, q2.w};
    uint32_t st[4], kill[4], todo[4], nbf[4], near[4];
    int base[4], a_eff[4];
    bool truthful[4], pend[4], need[4], live[4];
    bool any_need = false;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t s0 = s4[j] & all_up;                                    // bits at or above n_machines are not machines: every launch shape masks them
        st[j] = s0;
        nbf[j] = Env::nb_failed_of(sh, p, s0);
        todo[j] = s0;
        near[j] = 0xFFFFFFFFu;
        kill[j] = Env::draw2(W0[j], todo[j], nbf[j], T, near[j]);
        kill[j] |= Env::draw2(W1[j], todo[j], nbf[j], T, near[j]);
    }
    // draws 4 and 5 belong to a fifth and sixth up machine: 0.2 % of the lanes have one, four wave-steps in ten some lane
    if (__any((todo[0] | todo[1] | todo[2] | todo[3]) != 0u)) {                // wave-uniform
#pragma unroll
        for (int j = 0; j < 4; ++j) kill[j] |= Env::draw2(W2[j], todo[j], nbf[j], T, near[j]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const bool valid = a4[j] < n_act, was_done = ((dn >> (8 * j)) & 0xFFu) != 0u;
        live[j] = valid && !was_done;
        n_bad += (uint32_t)(!valid && !was_done);
        a_eff[j] = valid ? (int)a4[j] : M2;                                    // an invalid action draws like "no action"; the lane is discarded
        const uint32_t s0 = st[j];
        const int n_up = __popc(s0);
        base[j] = n_up + __popc(s0 & p.deg_gt2_mask);                          // network.py:87-92
        const bool has_action = a_eff[j] < M2;
        uint32_t near_a = 0xFFFFFFFFu;
        const bool tr = Env::truthful_of(n_up < 2 ? W0[j] : (n_up < 4 ? W1[j] : W2[j]), n_up & 1, T, near_a);   // half-word n_up of the twelve (n_up < 6)
        const bool here = has_action && n_up < 6;
        truthful[j] = here && tr;
        near[j] = min(near[j], here ? near_a : 0xFFFFFFFFu);
        pend[j] = has_action && !here;
        need[j] = live[j] && (todo[j] != 0u || pend[j]);
        any_need |= need[j];
    }
    for (uint32_t b = 3; any_need; ++b) {                                      // this thread's quad has more to draw: block by block
        const uint4 qb = Env::quad_block(key, glane0, b);
        const uint32_t Wb[4] = {qb.x, qb.y, qb.z, qb.w};
        any_need = false;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (need[j]) {
                const int left = __popc(todo[j]);
                kill[j] |= Env::draw2(Wb[j], todo[j], nbf[j], T, near[j]);
                if (pend[j] && left < 2) { truthful[j] = Env::truthful_of(Wb[j], left, T, near[j]); pend[j] = false; }
                need[j] = todo[j] != 0u || pend[j];
                any_need |= need[j];
            }
        }
    }
    uint32_t o4[4], r4[4], dpack = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int o = 0;
        float r = 0.f;
        uint32_t sn = s4[j];                                                   // a lane that does not step keeps its word as it came
        if (live[j]) {
            if (near[j] < Env::TIE) {                                          // a draw decided below its top 16 bits: the exact per-lane form
                typename Env::State e{st[j]};
                int d;
                Env::step_exact(sh, p, e, a_eff[j], key, glane0 + (uint32_t)j, o, r, d);
                sn = e.w;
            } else {                                                           // network.py:101-112
                typename Env::State e{st[j] & ~kill[j]};
                Env::finish(p, e.w, a_eff[j], base[j], truthful[j], o, r);
                sn = e.w;
            }
        }
        st[j] = sn;
        o4[j] = (uint32_t)o;
        r4[j] = __float_as_uint(r);
        dpack |= (((dn >> (8 * j)) & 0xFFu) != 0u ? 1u : 0u) << (8 * j);      // network.py:113: never done; a frozen lane keeps its flag
    }
    st_stream4(state + l0[t], st[0], st[1], st[2], st[3]);
    st_stream4(reinterpret_cast<uint32_t *>(ob) + l0[t], o4[0], o4[1], o4[2], o4[3]);
    st_stream4(reinterpret_cast<uint32_t *>(reward) + l0[t], r4[0], r4[1], r4[2], r4[3]);
    st_stream(done_w, dpack);
    }
    if (n_bad && err) atomicAdd(err, n_bad);
}

template <class Env>
int launch_reset(const typename Env::Params &p, uint32_t *state, int32_t *ob, int64_t n, uint64_t seed,
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

template <class Env>
int launch_step(const typename Env::Params &p, uint32_t *state, const int32_t *action, int32_t *ob,
                       typename Env::Reward *reward, uint8_t *done, uint32_t *err, int64_t n, uint64_t seed,
                       uint32_t lane0, uint64_t t, int flags, void *stream)
{
    if (!state || !action || !ob || !reward || !done || bad_range(n, lane0)) return POMDP_E_BADARG;
    if (n == 0) return 0;
    // Two lanes per thread where the env pools work across a wave's two 64-lane sub-batches (RockSample: the
    // Finisher specialisation above) and the batch still gives every CU several workgroups; one lane per thread
    // otherwise (measured equal within 2 % for the generic envs, tools/microbench.hip).
    if constexpr (Env::QUAD_STEP) {
        // quads of lanes per thread (step_quad_kernel) once the batch fills the chip with such workgroups
        const bool cols16 = ((reinterpret_cast<uintptr_t>(state) | reinterpret_cast<uintptr_t>(action) | reinterpret_cast<uintptr_t>(ob) |
                              reinterpret_cast<uintptr_t>(reward)) & 15u) == 0 && (reinterpret_cast<uintptr_t>(done) & 3u) == 0;
        if (n >= STEP_QUAD_MIN_LANES && n % (4 * BLOCK) == 0 && (lane0 & 3u) == 0 && cols16) {
            // STEP_TILES tiles per thread, software-pipelined, while that leaves every CU two workgroups (see the kernel)
            const int64_t tiles = n / (4 * BLOCK);
            if (STEP_TILES >= 2 && tiles % STEP_TILES == 0 && tiles / STEP_TILES >= 2 * 256)
                hipLaunchKernelGGL((step_quad_kernel<Env, STEP_TILES>), dim3((unsigned)(tiles / STEP_TILES)), dim3(BLOCK), 0, (hipStream_t)stream,
                                   state, action, ob, reward, done, err, n, make_key(seed, t), lane0, flags, p);
            else
                hipLaunchKernelGGL((step_quad_kernel<Env, 1>), dim3((unsigned)tiles), dim3(BLOCK), 0, (hipStream_t)stream, state,
                                   action, ob, reward, done, err, n, make_key(seed, t), lane0, flags, p);
            return (int)hipGetLastError();
        }
    }
    if constexpr (std::is_same<Env, NetworkEnv>::value) {
        const bool cols16 = ((reinterpret_cast<uintptr_t>(state) | reinterpret_cast<uintptr_t>(action) | reinterpret_cast<uintptr_t>(ob) |
                              reinterpret_cast<uintptr_t>(reward)) & 15u) == 0 && (reinterpret_cast<uintptr_t>(done) & 3u) == 0;
        if (n >= STEP_QUAD_MIN_LANES && n % (4 * BLOCK) == 0 && (lane0 & 3u) == 0 && cols16) {    // a thread's quad = a quad of the word contract
            const int64_t tiles = n / (4 * BLOCK);
            if (STEP_TILES >= 2 && tiles % STEP_TILES == 0 && tiles / STEP_TILES >= 2 * 256)
                hipLaunchKernelGGL((network_step_quad_kernel<Env, STEP_TILES>), dim3((unsigned)(tiles / STEP_TILES)), dim3(BLOCK), 0,
                                   (hipStream_t)stream, state, action, ob, reward, done, err, n, make_key(seed, t), lane0, flags, p);
            else
                hipLaunchKernelGGL((network_step_quad_kernel<Env, 1>), dim3((unsigned)tiles), dim3(BLOCK), 0, (hipStream_t)stream, state,
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
int launch_step_chain(const typename Env::Params &p, uint32_t *state, int32_t *action, int32_t *ob,
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

} // namespace pomdp
