// fused_impl.hip.h — the fused multi-step kernels (steps_kernel and the quad-per-thread loops of every env family) and
// launch_steps_fused, which picks among them.  Included by the translation units that instantiate it for their env types
// (fused_rock.hip, fused_stochrock.hip, fused_tag.hip, fused_battleship.hip, fused_misc.hip).
#pragma once
#include "kernels_common.hip.h"

namespace pomdp {

// ---- what every quad-per-thread loop shares ------------------------------------------------------------------------------
// The thread's place in the batch, its sink and its policy; the prologue around the first actions; the per-step key; the
// epilogue of a step (the policy's next actions become the current ones) and of the launch.  LPT lanes per thread: 4, or 2
// for BattleShip's small shards.  A contract change touches this and the env's own lane step, not six loops.
template <class L, class Pol, int LPT = 4>
struct FusedCtx {
    uint32_t l0, glane0;                                     // the thread's first lane within the shard / its global id
    uint64_t t0;                                             // call counter of the launch's first step
    typename lanes_out<L, LPT>::type out;
    typename lanes_policy<Pol, LPT>::type pol;
    uint32_t n_bad;                                          // tape only: out-of-range actions met
    __device__ __forceinline__ FusedCtx(int32_t *action, int32_t *ob, void *reward, uint8_t *done, int64_t rec, uint32_t lane0,
                                        const RngKey &key0, const RngKey &akey0, uint32_t n_act, int k_steps, const TapeRef &tape)
        : l0(blockIdx.x * (uint32_t)(LPT * BLOCK) + (uint32_t)LPT * threadIdx.x), glane0(lane0 + l0),
          t0(((uint64_t)key0.t_hi << 32) | key0.t_lo), out(action, ob, reward, done, rec, l0),
          pol(tape, l0, glane0, key0, akey0, n_act, k_steps), n_bad(0) {}
    // the actions of the launch's first step: the policy's own (the column sink keeps them in row 0, or reads them from there)
    __device__ __forceinline__ void first(int gen_first, int (&a_cur)[LPT])
    {
        if constexpr (LPT == 4) {
            const u32x4 a4 = out.first(pol, gen_first);
#pragma unroll
            for (int j = 0; j < 4; ++j) a_cur[j] = (int)a4[j];
        } else {
            uint32_t a2[LPT];
            pol.first(a2);
#pragma unroll
            for (int j = 0; j < LPT; ++j) a_cur[j] = (int)a2[j];
        }
    }
    // the env's Philox key at step s of the launch
    __device__ __forceinline__ RngKey key(const RngKey &key0, int s) const
    {
        RngKey k = key0;
        k.t_lo = (uint32_t)(t0 + (uint64_t)s); k.t_hi = (uint32_t)((t0 + (uint64_t)s) >> 32);
        return k;
    }
    // end of step s, after its stores: the actions of step s + 1 (a tape's row arrives here) become the current ones
    __device__ __forceinline__ void advance(int s, uint32_t (&a_next)[LPT], int (&a_cur)[LPT])
    {
        pol.end(s, a_next);
#pragma unroll
        for (int j = 0; j < LPT; ++j) a_cur[j] = (int)a_next[j];
    }
    __device__ __forceinline__ void finish(int k_steps)
    {
        pol.count_bad(n_bad);
        out.finish(k_steps);
    }
};
// The step loop of a fused launch: every load issued so far has landed (wait_loads: a storing loop may not wait on vmcnt), then
// four consecutive segments of falling issue priority (LoopPrio).  `s` runs over the launch's steps.
#define POMDP_FUSED_STEP_LOOP(prio_, s_)                                                                              \
    wait_loads();                                                                                                      \
    _Pragma("unroll 1") for (int seg_ = 0, s_ = 0; seg_ < 4; ++seg_)                                                   \
        for (const int seg_end_ = (prio_).segment(seg_); s_ < seg_end_; ++s_)

// k consecutive chained steps in ONE launch: exactly the memory state k launches of step_kernel<Env, LPT, true> leave —
// every step's ob / reward / done / state / next action is computed and written — but a lane's state and action stay in
// registers from one step to the next (nothing is re-read) and there is one launch ramp per k steps instead of per
// step.  Possible because a lane's step t+1 depends only on its own step t and all cooperation (pooled Philox passes,
// cooperative resets) is wave- or workgroup-local: no grid-wide synchronisation is involved.
// SIMPLE: every thread's lanes exist (n is a multiple of the workgroup's BLOCK * LPT lanes) and done lanes auto-reset, so
// no lane is ever out of range or frozen, and the actions are the driver's own (always valid): the bookkeeping for those
// cases is compiled out.

// L: the trajectory layout the steps are written in (traj_out.hip.h).  Blocked / Packed: `action` is the trajectory's base,
// ob / reward / done are not used, and the launch derives its first actions itself (FLAG_GEN_FIRST is always set).
// TAPE: the actions are the caller's (TapeRef: uint8 [k_steps][stride]) instead of the synthetic policy's — the general
// one-lane-per-thread form only (any n, out-of-range actions counted); each lane keeps a four-step window of its column.
template <class Env, int LPT, bool SIMPLE, bool TAB = false, class L = Columns, bool TAPE = false>
__global__ __launch_bounds__(BLOCK) void steps_kernel(uint32_t *__restrict__ state, int32_t *__restrict__ action,
                                                      int32_t *__restrict__ ob, typename Env::Reward *__restrict__ reward,
                                                      uint8_t *__restrict__ done, uint32_t *__restrict__ err, int64_t n,
                                                      RngKey key0, uint32_t lane0, int flags, RngKey akey0, int k_steps,
                                                      int64_t rec, const typename Env::Params p, TapeRef tape)
{
    // a tape drives the one-lane-per-thread instantiations: the general one, and (round 6) the SIMPLE ones of the envs whose blocks
    // are time-shared — full workgroups, auto-reset, but the actions are the caller's and may be out of range (counted, the lane
    // untouched), so `valid` / `live` stay run-time there
    static_assert(!TAPE || LPT == 1, "a tape drives the one-lane-per-thread instantiations");
    __shared__ typename Env::Shared sh;
    // TAB: the lane step reads the (position, action) table built below (RockEnv::step_rec: the lane's packed record and its
    // new state, fresh episode included, in one go); one lane per thread, the quad's blocks time-shared
    constexpr bool REC = TAB;
    __shared__ typename step_tab_of<Env, TAB>::type tab;
    static_assert(!TAB || (SIMPLE && LPT == 1), "the table-driven step serves the SIMPLE one-lane-per-thread instantiation");
    const bool auto_reset = SIMPLE || (flags & POMDP_AUTO_RESET);
    const uint32_t wg0 = blockIdx.x * (uint32_t)(BLOCK * LPT);
    const uint32_t last = SIMPLE ? (uint32_t)(BLOCK * LPT - 1) : (uint32_t)((uint64_t)(n - 1) - wg0);
    // rec = 0: every step overwrites the same n-element outputs (what the per-step launches do); rec = row pitch in
    // elements: step s writes row s of ob / reward / done and row s + 1 of action (row s being the actions it took)
    LaneOut<L, typename Env::Reward, LPT> out(action, ob, reward, done, rec, wg0);
    uint32_t *const state_w = state + wg0;
    constexpr bool COLS = L::ID == POMDP_LAYOUT_COLUMNS;     // only the column layout has a row of first actions / done flags to read
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
        a_cur[j] = (!COLS || (flags & FLAG_GEN_FIRST)) ? 0 : out.load_first(rc);
        Env::load(st[j], state_w, n, rc);
        if constexpr (has_next<Env>::value) Env::load_next(st[j], state_w, n, rc);   // once per launch, with the other words
        if constexpr (COLS) was_done[j] = auto_reset ? false : (ld_stream(done + wg0 + rc) != 0);
        else was_done[j] = false;
        out.begin(j, rc);                                    // Returns: the lane's running statistics
    }
    using Fin = Finisher<Env, LPT, true>;
    constexpr bool quad_policy = quad_policy_of<Fin>::value;
    uint4 aq = make_uint4(0, 0, 0, 0), sq = make_uint4(0, 0, 0, 0), rq = make_uint4(0, 0, 0, 0);
    uint4 nq0 = make_uint4(0, 0, 0, 0), nq1 = nq0, nq2 = nq0;        // Network: this lane's words of the quad's blocks 0 .. 2 of steps s .. s + 3
    const int n_act = Env::n_actions(p);
    const uint64_t t0 = ((uint64_t)key0.t_hi << 32) | key0.t_lo, ta0 = ((uint64_t)akey0.t_hi << 32) | akey0.t_lo;
    // TAPE: this lane's column of the tape (threads past n read lane n - 1's, like every other per-lane word)
    typename std::conditional<TAPE, TapeColumn<uint8_t>, NoColumn>::type col(tape, wg0 + (in_range[0] ? rel[0] : last), k_steps);
    if constexpr (TAPE) a_cur[0] = (int)col.first;
    else
    if (!COLS || (flags & FLAG_GEN_FIRST)) {             // wave-uniform: the policy's actions of the first call counter
        RngKey fkey = akey0;
        fkey.t_lo = (uint32_t)(ta0 - 1ull); fkey.t_hi = (uint32_t)((ta0 - 1ull) >> 32);
#pragma unroll
        for (int j = 0; j < LPT; ++j) {
            a_cur[j] = synthetic_action(fkey, glane[j], (uint32_t)n_act);
            if (in_range[j]) out.store_first(rel[j], a_cur[j]);
        }
    }
    out.first_done();
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
        Env::build_rec_tab(tab, sh, p, (int)threadIdx.x);
        __syncthreads();
    }
    wait_loads();
    const LoopPrio prio(k_steps);
    constexpr bool POLICY_WITH_STEP = quad_policy && !TAPE && POLICY_WITH_STEP_BLOCKS &&
                                      (Env::QUAD_SENSOR || quad_word_env<Env>::value || quad_words_of<Env>::value == 3);
    // The one-lane-per-thread loops with time-shared blocks serve the small shards (2^14 .. 2^18 lanes: one to four waves per
    // SIMD, a step is one wave's dependent chain).  Unrolled by four, a step's picks are register names instead of three
    // selects each and the every-fourth-step branch is straight-line code (UNROLL4).  A lone wave issues an instruction every
    // five to nine cycles whatever it depends on: what counts there is the NUMBER of instructions per step (the next group's
    // blocks drawn in instalments beside the step's own chain, the table entry asked for a step ahead: no gain, docs/HISTORY.md).
    constexpr bool UNROLL4 = quad_policy && LPT == 1 && (!TAPE || SIMPLE) && STEP_LOOP_UNROLL4;
    constexpr bool UNROLL4_TAIL = REC && L::ID != LAYOUT_RETURNS && !TAPE && STEP_LOOP_UNROLL4_TAIL;
    // A tape in the unrolled loop is read FOUR rows ahead: row r lives in slot r & 3 (compile-time names: a pending load is never
    // moved or selected); the top of step s asks for row s + 4 into the slot row s left when step s - 1 ended, the end of step s
    // reads row s + 1 — asked for three steps ago, which at a small shard's 0.3 us per step is about the latency of the load.
    // (One row ahead, as in the general loop, every step of a 2^17-lane shard waited for its load: 0.98 against 0.31 us.)
    constexpr bool TAPE4 = TAPE && UNROLL4;
    uint32_t tq[4] = {0, 0, 0, 0};
    if constexpr (TAPE4) { tq[1] = col.row(1); tq[2] = col.row(2); tq[3] = col.row(3); }
    constexpr int N_STEP_BLOCKS = quad_words_of<Env>::value == 3 ? 3 : 1;
    const uint32_t qe = glane[0] & 3u;                       // this lane's place in its quad: it draws the blocks of step s + qe
    // this lane's block of the policy for the group of four steps starting at s: that of step s + e, transposed within the quad
    // — component J is then THIS lane's word of step s + J.  Drawn in the same branch as the step's own time-shared blocks
    // (POLICY_WITH_STEP): the chains are independent, one block after the other costs a lone wave the latency of both.
    auto policy_quarter = [&](const int s) __attribute__((always_inline)) {
        const uint64_t te = ta0 + (uint64_t)s + (uint64_t)qe;
        aq = quad_transpose4(philox4x32_10(glane[0] >> 2, (uint32_t)te, (uint32_t)(te >> 32),
                                           (uint32_t)POMDP_STREAM_ACTION << 24, akey0.k0, akey0.k1), qe);
    };
    // ... and its blocks of the step stream (RockSample: the sensor block, which the fresh episodes of the steps' done lanes
    // start from as well; Tiger, Tag: the one STEP block; Network: three) — every Env::quad_block is stream STEP of lane >> 2
    auto step_quarter = [&](const int s) __attribute__((always_inline)) {
        const uint64_t te = t0 + (uint64_t)s + (uint64_t)qe;
        RngKey ke = key0;
        ke.t_lo = (uint32_t)te; ke.t_hi = (uint32_t)(te >> 32);
        if constexpr (N_STEP_BLOCKS == 3) {
            nq0 = quad_transpose4(stream_block(ke, glane[0] >> 2, POMDP_STREAM_STEP, 0u), qe);
            nq1 = quad_transpose4(stream_block(ke, glane[0] >> 2, POMDP_STREAM_STEP, 1u), qe);
            nq2 = quad_transpose4(stream_block(ke, glane[0] >> 2, POMDP_STREAM_STEP, 2u), qe);
        } else {
            sq = quad_transpose4(stream_block(ke, glane[0] >> 2, POMDP_STREAM_STEP, 0u), qe);
            rq = sq;
        }
    };
    // One step.  `sj_`: the step's place in its group of four (s & 3) — the time-shared blocks are drawn at place 0 and every
    // step picks its words by place: a run-time, wave-uniform select in the plain loop, a compile-time constant in the loop
    // unrolled by four (UNROLL4 below).
    auto step_body = [&](const int s, const auto sj_) __attribute__((always_inline)) {
        const int sj = sj_;
        // this step's word of a time-shared block.  In the plain loop (sj a run-time value) the words pass through an opaque
        // register definition first: as plain reads of the captured uint4 the compiler folds the selects into ONE read at a
        // run-time index — of an array it then keeps in scratch memory, with a `s_waitcnt vmcnt(0)` after the read, in every step
        // (the general tape loop after the step became a lambda: 32 B of scratch, 0.98 us per step at 2^17 lanes).
        auto pick4 = [&](const uint4 &q) __attribute__((always_inline)) -> uint32_t {
            uint32_t x = q.x, y = q.y, z = q.z, w = q.w;
            if constexpr (std::is_same<std::decay_t<decltype(sj_)>, int>::value) asm volatile("" : "+v"(x), "+v"(y), "+v"(z), "+v"(w));
            return sj == 0 ? x : sj == 1 ? y : sj == 2 ? z : w;
        };
        if constexpr (TAPE4) tq[std::decay_t<decltype(sj_)>::value] = col.row(s + 4);
        else if constexpr (TAPE) col.request(s);             // the row of step s + 1: first touched after this step's stores
        RngKey key = key0, akey = akey0;
        key.t_lo = (uint32_t)(t0 + (uint64_t)s); key.t_hi = (uint32_t)((t0 + (uint64_t)s) >> 32);
        akey.t_lo = (uint32_t)(ta0 + (uint64_t)s); akey.t_hi = (uint32_t)((ta0 + (uint64_t)s) >> 32);
        if constexpr (Fin::HAS_PREPASS) { if (s > 0) Fin::prepass(key, glane, akey); }
        int o[LPT], d[LPT];
        uint32_t recv[LPT];                              // REC: the lanes' packed records
        typename Env::Reward r[LPT];
        typename Fin::Aux aux[LPT];
        bool live[LPT], valid[LPT], fresh[LPT];
        int a_next[LPT];
        typename Env::State before[LPT];
#pragma unroll
        for (int j = 0; j < LPT; ++j) {
            before[j] = st[j];
            valid[j] = (SIMPLE && !TAPE) || (unsigned)a_cur[j] < (unsigned)n_act;
            live[j] = SIMPLE ? valid[j] : (in_range[j] && valid[j] && !was_done[j]);
            if constexpr (quad_policy && Env::QUAD_SENSOR) {
                // one lane per thread, the sensor block shared by the quad (RockSample shards below the pooled kernels' gates):
                // lane e computes the block of step s + e once per four steps and the words reach their lanes by the same
                // transpose as the policy's — one Philox block per lane per four steps instead of one per step
                if (sj == 0) { step_quarter(s); if constexpr (POLICY_WITH_STEP) policy_quarter(s); }
                const uint32_t H = pick4(sq);
                if constexpr (REC) {
                    // the step and, where it ends the episode, the fresh one (which starts from the same word H: auto-reset contract)
                    typename Env::S sj = st[j].s;
                    const uint32_t lane_ = glane[j];
                    Env::step_rec(sh, tab, sj, valid[j] ? (uint32_t)a_cur[j] : 0u, H, Env::fresh_state(p, H, key, lane_), recv[j],
                                  [&]() { return Env::elem(Env::quad_block(rare_key(key), lane_, 1u), lane_ & 3u); });
                    if constexpr (TAPE) recv[j] = valid[j] ? recv[j] : ((uint32_t)a_cur[j] & 0xFFu);   // the record keeps the byte: (ob, reward, done) = (0, 0, 0)
                    st[j].s = sj;
                    o[j] = (int)__builtin_amdgcn_ubfe(recv[j], 8u, 8u);
                    r[j] = (typename Env::Reward)__builtin_amdgcn_sbfe(recv[j], 16u, 8u);
                    d[j] = (int)(recv[j] >> 24);
                }
                else Env::step_with_H(sh, p, st[j], valid[j] ? a_cur[j] : 0, key, glane[j], H, o[j], r[j], d[j]);
            }
            else if constexpr (quad_policy && quad_word_env<Env>::value) {
                // Tiger, Tag: ONE word per lane-step from the quad's STEP block, time-shared the same way; the auto-reset of a
                // done lane reads the same word (fresh_w below)
                if (sj == 0) { step_quarter(s); if constexpr (POLICY_WITH_STEP) policy_quarter(s); }
                const uint32_t W = pick4(sq);
                Env::step_w(sh, p, st[j], valid[j] ? a_cur[j] : 0, key, glane[j], W, o[j], r[j], d[j]);
            }
            else if constexpr (quad_policy && quad_words_of<Env>::value == 3) {
                // the step's quad-shared blocks, time-shared: lane e of a quad computes the three blocks of step s + e once per
                // four steps, three 4 x 4 transposes hand every lane its own word of each block of each step
                if (sj == 0) { step_quarter(s); if constexpr (POLICY_WITH_STEP) policy_quarter(s); }
                auto pick = [&](const uint4 &q) { return pick4(q); };
                Env::step_words(sh, p, st[j], valid[j] ? a_cur[j] : 0, key, glane[j], pick(nq0), pick(nq1), pick(nq2), o[j], r[j], d[j]);
            }
            else Fin::lane_step(sh, p, st[j], valid[j] ? a_cur[j] : 0, key, glane[j], o[j], r[j], d[j], aux[j]);
            if (!live[j]) { r[j] = 0; d[j] = was_done[j]; }
            fresh[j] = live[j] && d[j] && auto_reset;
            a_next[j] = 0;
        }
        if constexpr (quad_policy) {
            if constexpr (!TAPE && !POLICY_WITH_STEP) { if (sj == 0) policy_quarter(s); }
            if constexpr (REC) {
                // step_rec already moved the fresh episode in
            } else if constexpr (Env::QUAD_SENSOR) {                         // RockSample: this lane's RESET word of step s
                const uint32_t rword = pick4(rq);
                st[0].s = fresh[0] ? Env::fresh_state(p, rword, key, glane[0]) : st[0].s;
            } else if constexpr (quad_word_env<Env>::value) {
                Env::fresh_w(sh, p, st[0], fresh[0], key, glane[0], pick4(sq));
            } else {
                Fin::resets_only(sh, p, st, fresh, key, glane);
            }
            const uint32_t word = pick4(aq);
            if constexpr (!TAPE) a_next[0] = (int)__umulhi(word, (uint32_t)n_act);
        } else {
            Fin::run(sh, p, st, fresh, key, glane, akey, (uint32_t)n_act, a_next, aux, o);
        }
#pragma unroll
        for (int j = 0; j < LPT; ++j) {
            if (!live[j]) { o[j] = 0; st[j] = before[j]; }                  // a lane that did not step keeps its state
            if (in_range[j]) out.put_next_action(rel[j], a_next[j]);
            ever_fresh[j] |= fresh[j];
            if (in_range[j]) {
                uint32_t rcode = 0;
                if constexpr (L::CODES && !REC) rcode = Env::reward_code(r[j]);
                if constexpr (REC && L::CODES) out.put_record(j, rel[j], recv[j]);
                else out.put(j, rel[j], a_cur[j], o[j], r[j], rcode, d[j]);
                if (!valid[j] && !was_done[j] && err) atomicAdd(err, 1u);
            }
            if constexpr (TAPE4) a_next[j] = (int)tq[(std::decay_t<decltype(sj_)>::value + 1) & 3];   // the tape's row of step s + 1, asked for three steps ago
            else if constexpr (TAPE) a_next[j] = (int)col.nxt;              // the tape's row of step s + 1, requested when this step began
            a_cur[j] = a_next[j];
            was_done[j] = auto_reset ? false : (d[j] != 0);
        }
        out.next_row();
        if constexpr (Fin::LOOP_BARRIER && !quad_policy) __syncthreads();
    };
    if constexpr (UNROLL4 && UNROLL4_TAIL) {
        // whole groups of four, then the launch's last one to three steps in the plain form (a fifth copy of the step: the
        // table-driven RockSample step affords it, the others lose registers and time to it)
        #pragma unroll 1
        for (int seg = 0, s = 0; seg < 4; ++seg) {                         // four priority segments (LoopPrio), ending at multiples of four
            const int seg_end = prio.template segment<4>(seg);
            #pragma unroll 1
            for (; s + 4 <= seg_end; s += 4) {
                step_body(s, std::integral_constant<int, 0>{});
                step_body(s + 1, std::integral_constant<int, 1>{});
                step_body(s + 2, std::integral_constant<int, 2>{});
                step_body(s + 3, std::integral_constant<int, 3>{});
            }
            #pragma unroll 1
            for (; s < seg_end; ++s) step_body(s, s & 3);
        }
    } else if constexpr (UNROLL4) {
        #pragma unroll 1
        for (int seg = 0, s = 0; seg < 4; ++seg)                           // four priority segments (LoopPrio), ending at multiples of four
        for (const int seg_end = prio.template segment<4>(seg); s < seg_end; s += 4) {
            step_body(s, std::integral_constant<int, 0>{});
            if (s + 1 < seg_end) {                                         // (only the launch's last group can be short)
                step_body(s + 1, std::integral_constant<int, 1>{});
                if (s + 2 < seg_end) {
                    step_body(s + 2, std::integral_constant<int, 2>{});
                    if (s + 3 < seg_end) step_body(s + 3, std::integral_constant<int, 3>{});
                }
            }
        }
    } else {
        #pragma unroll 1
        for (int seg = 0, s = 0; seg < 4; ++seg)                           // four priority segments (LoopPrio)
        for (const int seg_end = prio.segment(seg); s < seg_end; ++s) step_body(s, s & 3);
    }
    // the state is the loop's carry: it lived in registers and reaches memory once (a lane that never stepped writes back
    // what it read; BattleShip's ship words only if some step of the launch dealt a new board)
#pragma unroll
    for (int j = 0; j < LPT; ++j)
        if (in_range[j]) { Env::store(st[j], state_w, n, rel[j], ever_fresh[j]); out.finish(j, rel[j], k_steps); }
}

// The fused RockSample loop with a thread owning four CONSECUTIVE lanes — a quad.  RockSample's word contract shares the
// STEP block — which a done step's auto-reset reads as well (rock.hip.h) — and the policy's ACTION block among the four
// lanes of a quad, so with this mapping both are the thread's own: two Philox blocks per thread-step straight into
// registers, lane j taking element j — no exchange through LDS, no ballots, no task lists, and no dependence on the lane
// step: the compiler interleaves the chains with the table lookups.  A thread's outputs are four consecutive elements of each column: one 16-byte store per
// int32 column and one 4-byte store of the packed done bytes per step instead of twenty scalar stores; state and first
// actions come in the same way.  The lane step is the table-driven one.  Full workgroups of 1024 lanes and auto-reset
// only (the launcher's SIMPLE conditions).  Same results as steps_kernel: the mapping of lanes to threads is invisible
// to a lane's random words.
// LPT = 2 (round 6): HALF a quad per thread, for the shards that leave the quad loop two waves per SIMD or fewer (2^19 lanes
// — half of a 2^20-lane batch — ran the pooled two-lanes-per-thread steps_kernel at 0.46 of its issue floor).  The quad's
// STEP block (and StochasticRock's gate block) is time-shared by the quad's two threads exactly like the policy's block
// (pair_shared: thread e computes the block of step s + e at every even s, two DPP moves swap the halves): one block of each
// stream per thread per two steps — as many per lane-step as with a quad per thread — and twice the waves.  4-byte sinks and
// the returns sink (PairOut).
template <class Env, class L = Columns, class Pol = SyntheticQuad, int LPT = 4>
__global__ __launch_bounds__(BLOCK) void steps_quad_kernel(uint32_t *__restrict__ state, int32_t *__restrict__ action,
                                                           int32_t *__restrict__ ob, int32_t *__restrict__ reward,
                                                           uint8_t *__restrict__ done, int64_t n, RngKey key0, uint32_t lane0,
                                                           RngKey akey0, int k_steps, int64_t rec, int gen_first,
                                                           const typename Env::Params p, TapeRef tape)
{
    static_assert(LPT == 4 || LPT == 2, "a quad or half a quad per thread");
    constexpr int W = Env::WORDS;
    using S = typename Env::S;
    __shared__ typename Env::Shared sh;
    // the lane step yields the lane's packed record straight from RecTab (RockEnv::step_rec, ~31 vector instructions per
    // lane-step with one state word)
    __shared__ typename Env::RecTab tab;
    const uint32_t n_act = (uint32_t)Env::n_actions(p);
    // a quad per thread on a tape: the loop unrolled by two, the tape read two steps ahead (TapeQuadAhead)
    constexpr bool AHEAD2 = Pol::TAPE && LPT == 4 && TAPE_TWO_STEPS_AHEAD;
    using PolT = typename std::conditional<AHEAD2, TapeQuadAhead, Pol>::type;
    FusedCtx<L, PolT, LPT> cx(action, ob, reward, done, rec, lane0, key0, akey0, n_act, k_steps, tape);
    const uint32_t l0 = cx.l0, glane0 = cx.glane0;           // the thread's first lane within the shard; its global id (a multiple of LPT)
    const uint32_t e0 = LPT == 4 ? 0u : (glane0 & 2u);       // ... and that lane's element of its quad's blocks (LPT = 2: 0 or 2)
    typename Env::State st[LPT];
    int a_cur[LPT];
    uint32_t sp0 = 0, sp1 = 0, gp0 = 0, gp1 = 0;             // LPT = 2: the odd step's words of the quad's time-shared STEP / gate blocks
    {
        uint32_t s_lo[LPT], s_hi[LPT];
        if constexpr (LPT == 4) {
            const u32x4 lo = ld_stream4(state + l0);
            u32x4 hi = {0, 0, 0, 0};
            if (W == 2) hi = ld_stream4(state + n + l0);
#pragma unroll
            for (int j = 0; j < 4; ++j) { s_lo[j] = lo[j]; s_hi[j] = hi[j]; }
        } else {
            const u32x2 lo = ld_stream2(state + l0);
            u32x2 hi = {0, 0};
            if (W == 2) hi = ld_stream2(state + n + l0);
#pragma unroll
            for (int j = 0; j < 2; ++j) { s_lo[j] = lo[j]; s_hi[j] = hi[j]; }
        }
        cx.first(gen_first, a_cur);
#pragma unroll
        for (int j = 0; j < LPT; ++j) st[j].s = (S)((uint64_t)s_lo[j] | ((uint64_t)s_hi[j] << 32));
    }
    Env::stage(sh, p, (int)threadIdx.x);
    __syncthreads();
    Env::build_rec_tab(tab, sh, p, (int)threadIdx.x);
    __syncthreads();
    const int K = p.num_rocks;
    const uint32_t start = (uint32_t)p.start_x | ((uint32_t)p.start_y << 4);
    const LoopPrio prio(k_steps);
    auto step_body = [&](const int s, const auto par_) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par_)::value;           // AHEAD2: s & 1
        const RngKey key = cx.key(key0, s);
        // the quad's sensor words of this step (StochasticRock: block 2 of the stream — block 0 gates the actions, rock.py:443)
        // — the words its fresh episodes start from as well — and the actions of the next call counter
        constexpr uint32_t SENSOR_BLOCK = Env::SENSOR_BLOCK;
        uint32_t H[LPT], G[LPT];
        if constexpr (LPT == 4) {
            const uint4 sw = philox4x32_10(glane0 >> 2, key.t_lo, key.t_hi, ((uint32_t)POMDP_STREAM_STEP << 24) | SENSOR_BLOCK, key.k0, key.k1);
            H[0] = sw.x; H[1] = sw.y; H[2] = sw.z; H[3] = sw.w;
        } else {
            pair_shared(s, e0 != 0u, [&](int sb) { return Env::quad_block(cx.key(key0, sb), glane0, SENSOR_BLOCK); }, H, sp0, sp1);
        }
        uint32_t a_next[LPT];
        if constexpr (AHEAD2) cx.pol.template begin_par<PAR>(s, a_next);
        else cx.pol.begin(s, a_next);
        // (a lane's step draws EITHER its sensor reading OR its next episode: the fresh episodes start from the same words, R = H)
        bool acts[LPT];
#pragma unroll
        for (int j = 0; j < LPT; ++j) acts[j] = true;
        if constexpr (Env::STOCHASTIC) {                                       // the action is applied iff binomial(1, p_move) says so
            if constexpr (LPT == 4) {
                const uint4 gw = Env::quad_block(key, glane0, 0u);
                G[0] = gw.x; G[1] = gw.y; G[2] = gw.z; G[3] = gw.w;
            } else {
                pair_shared(s, e0 != 0u, [&](int sb) { return Env::quad_block(cx.key(key0, sb), glane0, 0u); }, G, gp0, gp1);
            }
#pragma unroll
            for (int j = 0; j < LPT; ++j)
                acts[j] = Env::k53_le(G[j], (uint32_t)(p.act_thr >> 26), (uint32_t)p.act_thr & Env::LO_MASK,
                                      [&]() { return Env::elem(Env::quad_block(key, glane0, 1u), e0 + (uint32_t)j); }) != (p.act_gt != 0);
        }
        uint32_t codes[LPT], rec[LPT], a_taken[LPT];
#pragma unroll
        for (int j = 0; j < LPT; ++j) a_taken[j] = (uint32_t)a_cur[j];
        if constexpr (LPT == 4) Env::reset_codes4(H, key, glane0, K, codes);
        else {
#pragma unroll
            for (int j = 0; j < LPT; ++j) codes[j] = Env::template reset_codes<true>(H[j], key, glane0 + (uint32_t)j, K);
        }
#pragma unroll
        for (int j = 0; j < LPT; ++j) {
            const uint32_t lane = glane0 + (uint32_t)j;
            S sj = st[j].s;
            // a tape may hold anything: an out-of-range action leaves the lane untouched, (ob, reward, done) = (0, 0, 0), and is counted
            const bool valid = !Pol::TAPE || a_taken[j] < n_act;
            Env::step_rec(sh, tab, sj, valid ? a_taken[j] : 0u, H[j], (S)((uint64_t)start | ((uint64_t)codes[j] << 8)), rec[j],
                          [&]() { return Env::elem(Env::quad_block(key, lane, SENSOR_BLOCK + 1u), e0 + (uint32_t)j); });
            if constexpr (Env::STOCHASTIC) {                                // the gate said no (rock.py:443): nothing happens
                sj = acts[j] ? sj : st[j].s;
                rec[j] = acts[j] ? rec[j] : a_taken[j];
            }
            if constexpr (Pol::TAPE) {
                sj = valid ? sj : st[j].s;
                rec[j] = valid ? rec[j] : a_taken[j];
                cx.n_bad += (uint32_t)!valid;
            }
            st[j].s = sj;
        }
        cx.out.put_records(rec, a_next);
        if constexpr (AHEAD2) {
            cx.pol.template end_par<PAR>(s, a_next);
#pragma unroll
            for (int j = 0; j < LPT; ++j) a_cur[j] = (int)a_next[j];
        } else cx.advance(s, a_next, a_cur);
    };
    if constexpr (AHEAD2) {
        wait_loads();
        _Pragma("unroll 1") for (int seg = 0, s = 0; seg < 4; ++seg)           // (LoopPrio's segments, ending at even steps)
            for (const int seg_end = prio.template segment<2>(seg); s < seg_end; s += 2) {
                step_body(s, std::integral_constant<int, 0>{});
                if (s + 1 < seg_end) step_body(s + 1, std::integral_constant<int, 1>{});
            }
    } else {
        POMDP_FUSED_STEP_LOOP(prio, s) step_body(s, std::integral_constant<int, 0>{});
    }
    // the state is the loop's carry: it reaches memory once
    cx.finish(k_steps);
    if constexpr (LPT == 4) {
        st_stream4(state + l0, (uint32_t)st[0].s, (uint32_t)st[1].s, (uint32_t)st[2].s, (uint32_t)st[3].s);
        if (W == 2)
            st_stream4(state + n + l0, (uint32_t)((uint64_t)st[0].s >> 32), (uint32_t)((uint64_t)st[1].s >> 32),
                       (uint32_t)((uint64_t)st[2].s >> 32), (uint32_t)((uint64_t)st[3].s >> 32));
    } else {
        st_stream2(state + l0, (uint32_t)st[0].s, (uint32_t)st[1].s);
        if (W == 2) st_stream2(state + n + l0, (uint32_t)((uint64_t)st[0].s >> 32), (uint32_t)((uint64_t)st[1].s >> 32));
    }
}

// Tag (one opponent) with a quad per thread: the policy's ACTION block is the thread's own, and so is the quad's STEP block —
// its four words are the four lanes' (ABI 13, tag.hip.h): a failed TAG's flight and the auto-reset after a successful one read
// them where they stand.  (Until round 5 both had per-lane blocks, pooled per wave of 256 lanes through LDS: two ballots, a rank
// and an LDS round trip per lane for one Philox pass — 70 of the loop's 273 vector instructions per thread-step.)  The outputs
// leave as 16-byte stores.
// LPT = 2 (round 6): half a quad per thread for the shards the quad loop leaves two waves per SIMD (2^19 lanes), the quad's STEP
// block time-shared by its two threads like the policy's (pair_shared; see steps_quad_kernel).
template <bool TAB, class L = Columns, class Pol = SyntheticQuad, int LPT = 4>   // TAB: the lane step reads the (cells, action) table built when the launch starts (from 16 steps per launch)
__global__ __launch_bounds__(BLOCK) void tag_steps_quad_kernel(uint32_t *__restrict__ state, int32_t *__restrict__ action,
                                                               int32_t *__restrict__ ob, float *__restrict__ reward,
                                                               uint8_t *__restrict__ done, int64_t n, RngKey key0,
                                                               uint32_t lane0, RngKey akey0, int k_steps, int64_t rec,
                                                               int gen_first, const TagEnv::Params p, TapeRef tape)
{
    using Env = TagEnv;
    static_assert(LPT == 4 || LPT == 2, "a quad or half a quad per thread");
    __shared__ Env::Shared sh;
    __shared__ typename std::conditional<TAB, Env::StepTab, NoTab>::type tab;
    const uint32_t n_act = (uint32_t)Env::n_actions(p);
    FusedCtx<L, Pol, LPT> cx(action, ob, reward, done, rec, lane0, key0, akey0, n_act, k_steps, tape);
    const uint32_t l0 = cx.l0, glane0 = cx.glane0;
    const uint32_t e0 = LPT == 4 ? 0u : (glane0 & 2u);       // the first lane's element of its quad's blocks
    Env::State st[LPT];
    int a_cur[LPT];
    uint32_t sp0 = 0, sp1 = 0;                               // LPT = 2: the odd step's words of the time-shared STEP block
    {
        if constexpr (LPT == 4) {
            const u32x4 s4 = ld_stream4(state + l0);
            cx.first(gen_first, a_cur);
#pragma unroll
            for (int j = 0; j < 4; ++j) st[j].w = s4[j];
        } else {
            const u32x2 s2 = ld_stream2(state + l0);
            cx.first(gen_first, a_cur);
#pragma unroll
            for (int j = 0; j < 2; ++j) st[j].w = s2[j];
        }
    }
    Env::stage(sh, p, (int)threadIdx.x);
    __syncthreads();
    if constexpr (TAB) {
        Env::build_tab(tab, sh, p, (int)threadIdx.x);
        __syncthreads();
    }
    const LoopPrio prio(k_steps);
    POMDP_FUSED_STEP_LOOP(prio, s) {
        const RngKey key = cx.key(key0, s);
        uint32_t a_next[LPT];
        cx.pol.begin(s, a_next);
        int o[LPT], d[LPT];
        float r[LPT];
        uint32_t a_taken[LPT], W[LPT];
#pragma unroll
        for (int j = 0; j < LPT; ++j) a_taken[j] = (uint32_t)a_cur[j];
        if constexpr (LPT == 4) {
            const uint4 qw = Env::quad_block(key, glane0, 0u);
            W[0] = qw.x; W[1] = qw.y; W[2] = qw.z; W[3] = qw.w;
        } else {
            pair_shared(s, e0 != 0u, [&](int sb) { return Env::quad_block(cx.key(key0, sb), glane0, 0u); }, W, sp0, sp1);
        }
#pragma unroll
        for (int j = 0; j < LPT; ++j) {
            Env::Flight f;
            const bool valid = !Pol::TAPE || a_taken[j] < n_act;      // a tape's out-of-range action: the lane is left untouched and counted
            const Env::State before = st[j];
            const int a = valid ? a_cur[j] : 0;
            if constexpr (TAB) Env::step_one_opponent_tab(tab, st[j], a, o[j], r[j], d[j], f);
            else Env::step_one_opponent_pre(sh, p, st[j], a, o[j], r[j], d[j], f);
            const uint32_t lane = glane0 + (uint32_t)j;
            if constexpr (Pol::TAPE) { if (!valid) { f.need = false; o[j] = 0; r[j] = 0.f; d[j] = 0; cx.n_bad++; } }
            Env::flee_word(sh, p, st[j], f, W[j], [&]() { return Env::elem(Env::quad_block(key, lane, 1u), e0 + (uint32_t)j); });
            if (d[j]) Env::auto_reset_word(p, st[j], W[j], key, lane);
            if constexpr (Pol::TAPE) st[j] = valid ? st[j] : before;
        }
        uint32_t o4[LPT], r4[LPT], d4[LPT], rc[LPT];
#pragma unroll
        for (int j = 0; j < LPT; ++j) {
            o4[j] = (uint32_t)o[j]; r4[j] = __float_as_uint(r[j]); d4[j] = (uint32_t)d[j];
            rc[j] = 0;
            if constexpr (L::CODES) rc[j] = Env::reward_code(r[j]);
        }
        cx.out.put(a_taken, a_next, o4, r4, rc, d4);
        cx.advance(s, a_next, a_cur);
    }
    cx.finish(k_steps);
    if constexpr (LPT == 4) st_stream4(state + l0, st[0].w, st[1].w, st[2].w, st[3].w);
    else st_stream2(state + l0, st[0].w, st[1].w);
}

// Network with a quad per thread.  The reference draws one double per UP machine and one for the action (network.py:94-109);
// the top 16 bits of those doubles come from blocks shared by the quad, two draws per word (network.hip.h), so the thread's
// own three blocks — twelve 32-bit words — hold draws 0 .. 5 of each of its four lanes.  Under a random policy a lane has 1.4
// machines up (2 % of the lanes four or more, 0.01 % six or more), so draws 0 .. 5 serve all but one lane-step in 10^4,
// straight-line: the machines they belong to, which of the two thresholds each is compared with and the reward base come
// from a table indexed by the machine set (SMALL: n_machines <= 10, the reference's default — three 32-bit entries per set,
// built when the launch starts), the six comparisons are three packed 16-bit saturating subtractions, the action's draw is
// the half-word after the last machine's.  A thread with a lane that has more to draw runs on, block by block (the whole
// wave does, a few lanes wide: 3 % of the wave-steps; with two inline blocks it was every wave-step, and 40 % of the loop).  A tie (a draw's 16 bits equal its threshold's, 2^-16 per draw) sends the lane
// through NetworkEnv::step_exact, the exact per-lane form.  The policy's ACTION block is the thread's own, the outputs
// leave as 16-byte stores, the reward comes from a table of the float32(float64) values the reference's arithmetic gives.
// Network never terminates, so there is no reset.
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ uint32_t pk_sub_sat_u16(uint32_t a, uint32_t b)       // per half: max(a - b, 0)  (v_pk_sub_u16 clamp)
{
    u16x2 x, y;
    __builtin_memcpy(&x, &a, 4); __builtin_memcpy(&y, &b, 4);
    const u16x2 d = __builtin_elementwise_sub_sat(x, y);
    uint32_t r;
    __builtin_memcpy(&r, &d, 4);
    return r;
}
template <int NB, class L = Columns, bool SMALL = false, class Pol = SyntheticQuad>   // NB: bytes of the machine set, ceil(n_machines / 8)
__global__ __launch_bounds__(BLOCK) void network_steps_quad_kernel(uint32_t *__restrict__ state, int32_t *__restrict__ action,
                                                                   int32_t *__restrict__ ob, float *__restrict__ reward,
                                                                   uint8_t *__restrict__ done, int64_t n, RngKey key0,
                                                                   uint32_t lane0, RngKey akey0, int k_steps, int64_t rec,
                                                                   int gen_first, const NetworkEnv::Params p, TapeRef tape)
{
    using Env = NetworkEnv;
    __shared__ Env::Shared sh;                               // the nibble tables of the exact per-lane form (ties only)
    __shared__ uint32_t nbf8[SMALL ? 1 : NB][SMALL ? 1 : 256];   // nbf8[k][v]: machines that see a failed neighbour when the down
                                                             // machines among 8 k .. 8 k + 7 are the set v (network.py:82-85)
    // SMALL, by machine set v (lbK: the K-th lowest up machine as a bit, 0 if none; selX: byte offset into tp of the pair's
    // packed thresholds — bit 2: the first of the pair sees a failed neighbour, bit 3: the second):
    //   sta = lb0 | lb1 << 10 | selA << 20 | base << 24      stb = lb2 | lb3 << 10 | selB << 20      stc = lb4 | lb5 << 10 | selC << 20
    //   nbft = the failed-neighbour set itself (the continuation's)
    //   tp[s] = (T16 of the pair's first draw - 1) << 16 | (T16 of its second - 1): d = sat(W - tp) per half is 0 where the
    //   draw leaves the machine up, 1 on a tie, >= 1 where it fails (fails iff k53 > thr, network.py:94-99)
    __shared__ uint32_t sta[SMALL ? 1024 : 1], stb[SMALL ? 1024 : 1], stc[SMALL ? 1024 : 1], nbft[SMALL ? 1024 : 1], tp[4];
    __shared__ float rtab[3][68];                            // reward by (no action / ping / reboot, 2 per up machine with > 2
                                                             // neighbours + 1 per other up machine): network.py:87-92, 103, 110
    const uint32_t n_act = (uint32_t)Env::n_actions(p);
    FusedCtx<L, Pol> cx(action, ob, reward, done, rec, lane0, key0, akey0, n_act, k_steps, tape);
    const uint32_t l0 = cx.l0, glane0 = cx.glane0;
    uint32_t st[4];
    int a_cur[4];
    {
        const u32x4 s4 = ld_stream4(state + l0);
        cx.first(gen_first, a_cur);
#pragma unroll
        for (int j = 0; j < 4; ++j) st[j] = s4[j];
    }
    Env::stage(sh, p, (int)threadIdx.x);
    const Env::Thr T = Env::thresholds(p);
    if constexpr (SMALL) {
        const uint32_t up_all = (1u << p.n_machines) - 1u;
        for (uint32_t v = threadIdx.x; v <= up_all; v += BLOCK) {
            const uint32_t down = ~v & up_all;
            uint32_t m = 0;
            for (int i = 0; i < p.n_machines; ++i) m |= ((p.nb_mask[i] & down) != 0u ? 1u : 0u) << i;
            uint32_t r = v, lb[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) { lb[k] = r & (0u - r); r ^= lb[k]; }
            const uint32_t base = (uint32_t)(__popc(v) + __popc(v & p.deg_gt2_mask));          // network.py:87-92
            sta[v] = lb[0] | (lb[1] << 10) | ((m & lb[0]) ? 4u << 20 : 0u) | ((m & lb[1]) ? 8u << 20 : 0u) | (base << 24);
            stb[v] = lb[2] | (lb[3] << 10) | ((m & lb[2]) ? 4u << 20 : 0u) | ((m & lb[3]) ? 8u << 20 : 0u);
            stc[v] = lb[4] | (lb[5] << 10) | ((m & lb[4]) ? 4u << 20 : 0u) | ((m & lb[5]) ? 8u << 20 : 0u);
            nbft[v] = m;
        }
        if (threadIdx.x < 4) {
            const uint32_t f = (T.fail >> 16) - 1u, nb = (T.nb >> 16) - 1u;     // the launcher checked both thresholds' top 16 bits >= 1
            tp[threadIdx.x] = (((threadIdx.x & 1u) ? nb : f) << 16) | ((threadIdx.x & 2u) ? nb : f);
        }
    } else {
#pragma unroll
    for (int kb = 0; kb < NB; ++kb) {                        // BLOCK threads = the 256 values of a byte
        uint32_t m = 0;
        for (int i = 0; i < p.n_machines; ++i) m |= (((p.nb_mask[i] >> (8 * kb)) & threadIdx.x) != 0u ? 1u : 0u) << i;
        nbf8[kb][threadIdx.x] = m;
    }
    }
    if (threadIdx.x < 3 * 68) {                              // r = float32(float64(base) - cost), as the reference computes it
        const int kind = (int)threadIdx.x / 68, b = (int)threadIdx.x % 68;
        double r = (double)b;
        if (kind == 1) r -= .1;
        if (kind == 2) r -= 2.5;
        rtab[kind][b] = (float)r;
    }
    __syncthreads();
    const uint32_t all_up = p.n_machines >= 32 ? 0xFFFFFFFFu : ((1u << p.n_machines) - 1u);
    const int M2 = 2 * p.n_machines;
    const LoopPrio prio(k_steps);
    POMDP_FUSED_STEP_LOOP(prio, s) {
        const RngKey key = cx.key(key0, s);
        uint32_t a_next[4];
        cx.pol.begin(s, a_next);
        const uint4 q0 = Env::quad_block(key, glane0, 0u), q1 = Env::quad_block(key, glane0, 1u), q2 = Env::quad_block(key, glane0, 2u);
        const uint32_t W0[4] = {q0.x, q0.y, q0.z, q0.w}, W1[4] = {q1.x, q1.y, q1.z, q1.w};
        const uint32_t W2[4] = {q2.x, q2.y, q2.z, q2.w};
        // a tape's out-of-range action draws like "no action" (the same random words are consumed); the lane is then left untouched
        bool valid[4] = {true, true, true, true};
        if constexpr (Pol::TAPE) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { valid[j] = (uint32_t)a_cur[j] < n_act; cx.n_bad += (uint32_t)!valid[j]; }
        }
        const uint32_t a_taken[4] = {(uint32_t)a_cur[0], (uint32_t)a_cur[1], (uint32_t)a_cur[2], (uint32_t)a_cur[3]};
        if constexpr (Pol::TAPE) {
#pragma unroll
            for (int j = 0; j < 4; ++j) a_cur[j] = valid[j] ? a_cur[j] : M2;
        }
        uint32_t kill[4], todo[4], near[4], tie16[4];
        int base[4];
        bool truthful[4], pend[4], need[4];
        bool any_need = false;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t s0 = st[j] & all_up;                                // set_state() may have handed in stray upper bits
            const int n_up = __popc(s0);
            near[j] = 0xFFFFFFFFu;
            if constexpr (SMALL) {
                const uint32_t ea = sta[s0], eb = stb[s0], ec = stc[s0];
                const uint32_t lb0 = ea & 1023u, lb1 = __builtin_amdgcn_ubfe(ea, 10u, 10u), lb2 = eb & 1023u, lb3 = __builtin_amdgcn_ubfe(eb, 10u, 10u);
                const uint32_t lb4 = ec & 1023u, lb5 = __builtin_amdgcn_ubfe(ec, 10u, 10u);
                const uint32_t ta_ = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(tp) + __builtin_amdgcn_ubfe(ea, 20u, 4u));
                const uint32_t tb_ = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(tp) + __builtin_amdgcn_ubfe(eb, 20u, 4u));
                const uint32_t tc_ = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(tp) + __builtin_amdgcn_ubfe(ec, 20u, 4u));
                const uint32_t d0 = pk_sub_sat_u16(W0[j], ta_), d1 = pk_sub_sat_u16(W1[j], tb_), d2 = pk_sub_sat_u16(W2[j], tc_);
                base[j] = (int)(ea >> 24);
                kill[j] = (d0 > 0xFFFFu ? lb0 : 0u) | ((d0 & 0xFFFFu) ? lb1 : 0u) | (d1 > 0xFFFFu ? lb2 : 0u) | ((d1 & 0xFFFFu) ? lb3 : 0u) |
                          (d2 > 0xFFFFu ? lb4 : 0u) | ((d2 & 0xFFFFu) ? lb5 : 0u);
                // a tie: some half of d is exactly 1 (a slot without a machine may raise a false alarm: the exact path, 2^-16 of the time)
                tie16[j] = ((d0 >> 16) == 1u) | ((d0 & 0xFFFFu) == 1u) | ((d1 >> 16) == 1u) | ((d1 & 0xFFFFu) == 1u) |
                           ((d2 >> 16) == 1u) | ((d2 & 0xFFFFu) == 1u);
                todo[j] = s0 & ~(lb0 | lb1 | lb2 | lb3 | lb4 | lb5);
            } else {
                base[j] = n_up + __popc(s0 & p.deg_gt2_mask);                  // network.py:87-92
                const uint32_t down = ~s0 & all_up;
                uint32_t f = nbf8[0][down & 255u];
#pragma unroll
                for (int kb = 1; kb < NB; ++kb) f |= nbf8[kb][(down >> (8 * kb)) & 255u];
                todo[j] = s0;
                kill[j] = Env::draw2(W0[j], todo[j], f, T, near[j]);
                kill[j] |= Env::draw2(W1[j], todo[j], f, T, near[j]);
                kill[j] |= Env::draw2(W2[j], todo[j], f, T, near[j]);
                tie16[j] = 0u;
            }
            const bool has_action = a_cur[j] < M2;
            const uint32_t aw = n_up < 2 ? W0[j] : (n_up < 4 ? W1[j] : W2[j]);   // the action's draw: half-word n_up of the twelve (n_up < 6)
            uint32_t near_a = 0xFFFFFFFFu;
            const bool tr = Env::truthful_of(aw, n_up & 1, T, near_a);
            const bool here = has_action && n_up < 6;
            truthful[j] = here && tr;
            near[j] = min(near[j], here ? near_a : 0xFFFFFFFFu);
            pend[j] = has_action && !here;
            need[j] = todo[j] != 0u || pend[j];
            any_need |= need[j];
        }
        if (any_need) {                                                        // this thread's quad has more to draw: block by block
            uint32_t nbf[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint32_t s0 = st[j] & all_up;
                if constexpr (SMALL) nbf[j] = nbft[s0];
                else {
                    const uint32_t down = ~s0 & all_up;
                    uint32_t f = nbf8[0][down & 255u];
#pragma unroll
                    for (int kb = 1; kb < NB; ++kb) f |= nbf8[kb][(down >> (8 * kb)) & 255u];
                    nbf[j] = f;
                }
            }
            for (uint32_t b = 3; any_need; ++b) {
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
        }
        uint32_t o4[4], r4[4], rc[4] = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int o;
            float r;
            if (Pol::TAPE && !valid[j]) {                                      // untouched: (ob, reward, done) = (0, 0, 0)
                o = 0; r = 0.f;
            } else if (tie16[j] != 0u || near[j] < Env::TIE) {                 // a draw decided below its top 16 bits: the exact per-lane form
                Env::State e{st[j] & all_up};
                int d;
                Env::step_exact(sh, p, e, a_cur[j], key, glane0 + (uint32_t)j, o, r, d);
                st[j] = e.w;
                if constexpr (L::CODES) rc[j] = Env::reward_code(r);
            } else {                                                           // network.py:101-112
                const int a = a_cur[j], machine = (a >> 1) & 31;
                const bool has_action = a < M2, reboot = has_action && (a & 1);
                uint32_t sn = st[j] & all_up & ~kill[j];
                sn |= reboot ? 1u << machine : 0u;
                const int up = (int)((sn >> machine) & 1u);                    // a rebooted machine is up: ob = truthful either way
                o = has_action ? (truthful[j] ? up : 1 - up) : 2;
                r = rtab[has_action ? 1 + (a & 1) : 0][base[j]];
                if constexpr (L::CODES) rc[j] = Env::reward_code(has_action ? 1 + (a & 1) : 0, base[j]);
                st[j] = sn;
            }
            o4[j] = (uint32_t)o;
            r4[j] = __float_as_uint(r);
        }
        const uint32_t d4[4] = {0u, 0u, 0u, 0u};                               // network.py:113: never done
        cx.out.put(a_taken, a_next, o4, r4, rc, d4);
        cx.advance(s, a_next, a_cur);
    }
    cx.finish(k_steps);
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
// The Returns sink keeps ten more registers per lane across the loop (traj_out.hip.h); with three or four mask words that is
// 137-143 registers by the compiler's own choice — three waves per SIMD, where a 2^20-lane launch has four workgroups per CU
// to place.  waves_per_eu(4) holds it to 128 (1 = no constraint, for the trajectory sinks).
template <class L, int MW, bool VIS_LDS = false> struct quad_waves { static constexpr int value = (L::ID == LAYOUT_RETURNS && MW >= 3 && !VIS_LDS) ? 4 : 1; };
// LPT = 2 (round 6): the same loop with HALF a quad per thread, for the shards that leave the quad loop two waves per SIMD —
// configs[3]'s 2^19 lanes per GPU ran at 0.64-0.79 of the VALU's issue slots.  Two lanes per thread are four waves per SIMD
// there; the policy's block is time-shared by the two threads of a quad (SyntheticPair), the ship masks lie [word][lane of
// the pair][thread], the sinks are the 4-byte ones and the returns sink (PairOut).
template <int MW, class L = Columns, class Pol = SyntheticQuad, int LPT = 4, bool VIS_LDS = false>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(quad_waves<L, MW, VIS_LDS>::value))) void battleship_steps_quad_kernel(uint32_t *__restrict__ state, int32_t *__restrict__ action,
                                                                      int32_t *__restrict__ ob, int32_t *__restrict__ reward,
                                                                      uint8_t *__restrict__ done, int64_t n, RngKey key0,
                                                                      uint32_t lane0, RngKey akey0, int k_steps, int64_t rec,
                                                                      int gen_first, const pomdp_battleship_params p, TapeRef tape)
{
    using Env = BattleShipEnv<MW>;
    using Mask = typename Env::Mask;
    // The two masks a step only READS — the ships of this episode and of the next — live in LDS, [word][lane of the quad]
    // [thread] (conflict-free whatever word a thread asks for: a row is 256 words, so thread t always hits bank t % 64); a
    // shot tests ONE word of the ship mask (an LDS read issued when the step begins).
    // VIS_LDS (round 6; shards up to 2^19 lanes): the cells already shot at live there too — one more read, one bit test, one
    // LDS write of the visited word, about twenty vector instructions per lane-step where picking the word out of a
    // register-resident 128-bit mask takes fifty; `remaining` (the top six bits of the last visited word in memory) rides
    // in a register of its own.  48 bytes of LDS per lane: a 2^20-lane batch (4096 lanes per CU) does not fit a CU's 160 KB
    // at once, so there the visited mask stays in registers (10x10, packed records, 2^20 lanes: 1.59 us per step in
    // registers, 1.72 in LDS with three workgroups resident instead of four).
    static_assert(LPT == 4 || LPT == 2, "a quad or half a quad per thread");
    constexpr int LOG = LPT == 4 ? 2 : 1;
    __shared__ uint32_t occ_lds[MW][LPT][BLOCK], next_lds[MW][LPT][BLOCK], vis_lds[VIS_LDS ? MW : 1][VIS_LDS ? LPT : 1][VIS_LDS ? BLOCK : 1];
    __shared__ uint8_t task_lds[BLOCK / 64][64 * LPT];       // task rank -> lane within the wave's 64 * LPT
    __shared__ uint8_t ts_lds[BLOCK / 64][64 * LPT];         // ... and the step at which that lane's current board was dealt
    __shared__ typename Env::SeqTables seq;                  // the column patterns of the board builder
    Env::stage_seq(seq, p, (int)threadIdx.x);
    const int wv = (int)(threadIdx.x >> 6), me = (int)(threadIdx.x & 63u), tid = (int)threadIdx.x;
    const uint32_t n_act = (uint32_t)Env::n_actions(p);
    FusedCtx<L, Pol, LPT> cx(action, ob, reward, done, rec, lane0, key0, akey0, n_act, k_steps, tape);
    const uint32_t l0 = cx.l0, glane0 = cx.glane0, wave0 = glane0 - (uint32_t)LPT * (uint32_t)me;
    int rem[LPT];                                            // VIS_LDS: total_remaining (battleship.py:100-104)
    Mask vis[VIS_LDS ? 1 : LPT];                             // !VIS_LDS: the visited mask, remaining in the top bits of its last word
    int a_cur[LPT];
    {
        typename std::conditional<LPT == 4, u32x4, u32x2>::type w[3 * MW];
#pragma unroll
        for (int q = 0; q < 3 * MW; ++q) {
            if constexpr (LPT == 4) w[q] = ld_stream4(state + (int64_t)q * n + l0);
            else w[q] = ld_stream2(state + (int64_t)q * n + l0);
        }
        cx.first(gen_first, a_cur);
#pragma unroll
        for (int j = 0; j < LPT; ++j) {
            rem[j] = (int)(w[2 * MW - 1][j] >> 26);
            if constexpr (!VIS_LDS) vis[j].lo = vis[j].hi = 0;
#pragma unroll
            for (int q = 0; q < MW; ++q) {
                occ_lds[q][j][tid] = w[q][j];
                if constexpr (VIS_LDS) vis_lds[q][j][tid] = q == MW - 1 ? (w[MW + q][j] & 0x03FFFFFFu) : w[MW + q][j];
                else vis[j].set_word(q, w[MW + q][j]);
                next_lds[q][j][tid] = w[2 * MW + q][j];
            }
        }
    }
    __syncthreads();
    const int cells = p.x_size * p.y_size;
    const uint64_t t0 = cx.t0;
    int pend[LPT];                                           // >= 0: the step at which the lane's board was dealt; its `next` is yet to be built
#pragma unroll
    for (int j = 0; j < LPT; ++j) pend[j] = -1;
    // the boards of every waiting lane of the wave (wave-uniform control flow; the scratch and the mask slots are the wave's own)
    auto build_boards = [&]() {
        int ntask = 0;
#pragma unroll
        for (int j = 0; j < LPT; ++j) {
            const uint64_t m = __ballot(pend[j] >= 0);
            const int rank = ntask + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            ntask += __popcll(m);
            if (pend[j] >= 0) { task_lds[wv][rank & (64 * LPT - 1)] = (uint8_t)(LPT * me + j); ts_lds[wv][rank & (64 * LPT - 1)] = (uint8_t)pend[j]; }
            pend[j] = -1;
        }
        if (ntask == 0) return;                                                // wave-uniform
        // the task list is read by OTHER lanes of this wave (take): pin the LDS order the hand-off relies on — no
        // instruction on hardware (a wave's LDS operations execute in order), but the compiler may not move the reads up
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // A pool of 64 builders works the task list off: every lane feeds its board one word of its stream per iteration
        // (four iterations per Philox block, the blocks computed by all lanes at once, each with its own counter), and a
        // lane whose board is complete takes the next unclaimed task at the following block boundary instead of idling
        // until the slowest board of its batch is done (a 5x5 board takes 42 words on average and over a hundred at worst).
        const typename Env::BuildConsts bc = Env::build_consts(p);
        typename Env::Builder bld;
        int my = me < ntask ? me : -1, next_task = ntask < 64 ? ntask : 64;    // this lane's task; the first unclaimed one
        int bidx = 0;
        uint32_t bt_lo = 0, bt_hi = 0, blk = 0;
        auto take = [&](int q) {                                               // lanes with q >= 0 start on task q
            const int idx = q >= 0 ? (int)task_lds[wv][q & (64 * LPT - 1)] : 0, s0 = q >= 0 ? (int)ts_lds[wv][q & (64 * LPT - 1)] : 0;
            const uint64_t td = t0 + (uint64_t)s0;                             // battleship.py:131-137 on stream NEXT of that step's call counter
            if (q >= 0) { bidx = idx; bt_lo = (uint32_t)td; bt_hi = (uint32_t)(td >> 32); blk = 0; bld.start(p.max_len); }
        };
        bld.idle();
        take(my);
        while (__any(my >= 0)) {
            const uint4 b4 = philox4x32_10(wave0 + (uint32_t)bidx, bt_lo, bt_hi, ((uint32_t)POMDP_STREAM_NEXT << 24) | (blk & 0xFFFFFFu), key0.k0, key0.k1);
            ++blk;
            Env::feed2(bld, seq, bc, b4.x, b4.y);                             // == feed(x), feed(y): one placement test per pair
            Env::feed2(bld, seq, bc, b4.z, b4.w);
            const bool fin = my >= 0 && !bld.busy();
            const uint64_t fm = __ballot(fin);
            if (fm != 0ull) {                                                  // wave-uniform
                if (fin) {                                                     // straight into the owner's slot: lane bidx % LPT of thread bidx / LPT
#pragma unroll
                    for (int w = 0; w < MW; ++w) next_lds[w][bidx & (LPT - 1)][64 * wv + (bidx >> LOG)] = (uint32_t)(bld.occ >> (32 * w));
                }
                const int r = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(fm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)fm, 0u));
                if (fin) { my = next_task + r < ntask ? next_task + r : -1; bld.idle(); take(my); }
                next_task += __popcll(fm);
            }
        }
        // the builders wrote next_lds slots that their OWNERS read (swap-in, final store): same hand-off, same pin
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    // The ladder from 32 steps per launch only: the board builders that follow the loop are instruction-bound where the loop
    // is store-bound, and waves that leave the loop at different times build their boards under the others' stores (10x10,
    // 20 steps per launch: 5.8 us per step without the ladder, 6.3 with it; 5x5 at 64 steps: 5.0 without, 4.3 with).
    const LoopPrio prio(k_steps, k_steps >= 32);
    POMDP_FUSED_STEP_LOOP(prio, s) {
        uint32_t a_taken[LPT];
        bool valid[LPT];
#pragma unroll
        for (int j = 0; j < LPT; ++j) { a_taken[j] = (uint32_t)a_cur[j]; valid[j] = true; }
        if constexpr (Pol::TAPE) {               // a tape's out-of-range shot: the lane is left untouched, (0, 0, 0), and counted
#pragma unroll
            for (int j = 0; j < LPT; ++j) { valid[j] = a_taken[j] < n_act; cx.n_bad += (uint32_t)!valid[j]; a_cur[j] = valid[j] ? a_cur[j] : 0; }
        }
        uint32_t ow[LPT], vw[LPT];
#pragma unroll
        for (int j = 0; j < LPT; ++j) {                                         // the word of the ship mask and of the visited mask this shot tests
            ow[j] = occ_lds[a_cur[j] >> 5][j][tid];
            vw[j] = VIS_LDS ? vis_lds[a_cur[j] >> 5][j][tid] : 0u;
        }
        uint32_t a_next[LPT];
        cx.pol.begin(s, a_next);
        uint32_t o4[LPT], r4[LPT], d4[LPT];
        bool d[LPT], again = false, any_d = false;
#pragma unroll
        for (int j = 0; j < LPT; ++j) {                                        // battleship.py:91-122: draws nothing
            const int a = a_cur[j];
            int r;
            uint32_t ob_j;
            if constexpr (VIS_LDS) {
                const uint32_t m = 1u << (a & 31);
                const bool visited = (vw[j] & m) != 0u, hit = (ow[j] & m) != 0u;
                const bool fresh_hit = !visited && hit && valid[j];
                r = visited ? -10 : -1;
                rem[j] -= (int)fresh_hit;
                d[j] = rem[j] == 0 && valid[j];
                if (d[j]) r += cells;
                if (valid[j]) vis_lds[a >> 5][j][tid] = vw[j] | m;             // (a cell shot at before: the same word again)
                ob_j = (uint32_t)fresh_hit;
                if constexpr (Pol::TAPE) { if (!valid[j]) r = 0; }
            } else {
                const Mask vis_before = vis[j];
                const uint32_t last = vis[j].word(MW - 1);
                int remaining = (int)(last >> 26);
                const bool visited = Env::bit(vis[j], a), hit = (ow[j] >> (a & 31)) & 1u;
                if (visited) r = -10;
                else { r = -1; remaining -= (int)hit; Env::set_bit(vis[j], a); }
                d[j] = remaining == 0;
                if (d[j]) r += cells;
                vis[j].set_word(MW - 1, (vis[j].word(MW - 1) & 0x03FFFFFFu) | ((uint32_t)remaining << 26));
                ob_j = (uint32_t)(!visited && hit);
                if constexpr (Pol::TAPE) { if (!valid[j]) { vis[j] = vis_before; d[j] = false; r = 0; ob_j = 0; } }
            }
            again |= d[j] && pend[j] >= 0;
            any_d |= d[j];
            o4[j] = ob_j; r4[j] = (uint32_t)r;
            d4[j] = (uint32_t)d[j];
        }
        cx.out.put(a_taken, a_next, o4, r4, r4, d4);                           // the int8 reward IS its code
        cx.advance(s, a_next, a_cur);
        if (__any(again)) build_boards();                                      // a second episode ended before the lane's next board exists
        if (__any(any_d)) {                                                    // wave-uniform: the cached boards move in
#pragma unroll
            for (int j = 0; j < LPT; ++j) {
                if (d[j]) {
                    int ships = 0;
#pragma unroll
                    for (int q = 0; q < MW; ++q) {
                        const uint32_t x = next_lds[q][j][tid];
                        occ_lds[q][j][tid] = x;
                        if constexpr (VIS_LDS) vis_lds[q][j][tid] = 0u;
                        ships += __popc(x);
                    }
                    rem[j] = ships;
                    if constexpr (!VIS_LDS) { vis[j].lo = vis[j].hi = 0; vis[j].set_word(MW - 1, (uint32_t)ships << 26); }
                    pend[j] = s;
                }
            }
        }
    }
    build_boards();
    cx.finish(k_steps);
#pragma unroll
    for (int q = 0; q < MW; ++q) {
        uint32_t vq[LPT];
#pragma unroll
        for (int j = 0; j < LPT; ++j) {
            if constexpr (VIS_LDS) vq[j] = vis_lds[q][j][tid] | (q == MW - 1 ? (uint32_t)rem[j] << 26 : 0u);
            else vq[j] = vis[j].word(q);
        }
        if constexpr (LPT == 4) {
            st_stream4(state + (int64_t)q * n + l0, occ_lds[q][0][tid], occ_lds[q][1][tid], occ_lds[q][2][tid], occ_lds[q][3][tid]);
            st_stream4(state + (int64_t)(MW + q) * n + l0, vq[0], vq[1], vq[2], vq[3]);
            st_stream4(state + (int64_t)(2 * MW + q) * n + l0, next_lds[q][0][tid], next_lds[q][1][tid], next_lds[q][2][tid], next_lds[q][3][tid]);
        } else {
            st_stream2(state + (int64_t)q * n + l0, occ_lds[q][0][tid], occ_lds[q][1][tid]);
            st_stream2(state + (int64_t)(MW + q) * n + l0, vq[0], vq[1]);
            st_stream2(state + (int64_t)(2 * MW + q) * n + l0, next_lds[q][0][tid], next_lds[q][1][tid]);
        }
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

template <class Env, class L = Columns, class Pol = SyntheticQuad>
__global__ __launch_bounds__(BLOCK) void steps_quad_generic_kernel(uint32_t *__restrict__ state, int32_t *__restrict__ action,
                                                                   int32_t *__restrict__ ob,
                                                                   typename Env::Reward *__restrict__ reward,
                                                                   uint8_t *__restrict__ done, int64_t n, RngKey key0,
                                                                   uint32_t lane0, RngKey akey0, int k_steps, int64_t rec,
                                                                   int gen_first, const typename Env::Params p, TapeRef tape)
{
    static_assert(Env::WORDS == 1 && sizeof(typename Env::Reward) == 4, "one state word, 4-byte rewards");
    __shared__ typename Env::Shared sh;
    const uint32_t n_act = (uint32_t)Env::n_actions(p);
    FusedCtx<L, Pol> cx(action, ob, reward, done, rec, lane0, key0, akey0, n_act, k_steps, tape);
    const uint32_t l0 = cx.l0, glane0 = cx.glane0;
    typename Env::State st[4];
    int a_cur[4];
    {
#pragma unroll
        for (int j = 0; j < 4; ++j) Env::load(st[j], state, n, l0 + (uint32_t)j);
        cx.first(gen_first, a_cur);
    }
    Env::stage(sh, p, (int)threadIdx.x);
    __syncthreads();
    const LoopPrio prio(k_steps);
    POMDP_FUSED_STEP_LOOP(prio, s) {
        const RngKey key = cx.key(key0, s);
        uint32_t a_next[4];
        cx.pol.begin(s, a_next);
        uint32_t o4[4], r4[4], d4[4], rc[4] = {0, 0, 0, 0};
        const uint32_t a_taken[4] = {(uint32_t)a_cur[0], (uint32_t)a_cur[1], (uint32_t)a_cur[2], (uint32_t)a_cur[3]};
        uint32_t W[4] = {0, 0, 0, 0};                                          // Tiger: the quad's STEP block IS the thread's four words
        if constexpr (quad_word_env<Env>::value) {
            const uint4 qw = Env::quad_block(key, glane0, 0u);
            W[0] = qw.x; W[1] = qw.y; W[2] = qw.z; W[3] = qw.w;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int o, d;
            typename Env::Reward r;
            const uint32_t lane = glane0 + (uint32_t)j;
            const bool valid = !Pol::TAPE || a_taken[j] < n_act;               // a tape's out-of-range action: untouched, (0, 0, 0), counted
            const typename Env::State before = st[j];
            const int a = valid ? a_cur[j] : 0;
            if constexpr (quad_word_env<Env>::value)
                Env::step_word(p, st[j], a, W[j], [&]() { return Env::elem(Env::quad_block(key, lane, 1u), (uint32_t)j); }, o, r, d);
            else
                Env::step(sh, p, st[j], a, key, lane, o, r, d);
            if constexpr (Pol::TAPE) { if (!valid) { st[j] = before; o = 0; r = 0; d = 0; cx.n_bad++; } }
            Env::reset_where(sh, p, st[j], d != 0, key, lane);                 // wave-convergent: every lane calls it
            o4[j] = (uint32_t)o;
            __builtin_memcpy(&r4[j], &r, 4);
            if constexpr (L::CODES) rc[j] = Env::reward_code(r);
            d4[j] = (uint32_t)(d != 0);
        }
        cx.out.put(a_taken, a_next, o4, r4, rc, d4);
        cx.advance(s, a_next, a_cur);
    }
    cx.finish(k_steps);
#pragma unroll
    for (int j = 0; j < 4; ++j) Env::store(st[j], state, n, l0 + (uint32_t)j, true);
}

// the same as k launch_step_chain calls at t, t + 1, ..., in one launch.  gen_first: the launch derives the actions of
// call counter t itself (and writes them to `action`) instead of reading them — the caller skips the policy launch.
// L: the trajectory layout (traj_out.hip.h).  Blocked / Packed / Narrow (trajectory collection only): `action` is the
// trajectory's base — row-major, `rec` lanes per row —, ob / reward / done are ignored, auto-reset and the shared policy key
// are required and every launch derives its first actions itself.  Returns<Env> (pomdp_collect_returns): `action` = the
// statistics' double rows, `ob` = their int32 rows, `reward` = the discount's bit pattern, `rec` = their pitch.
// tape.base != nullptr (pomdp_collect_tape*): the k steps take the caller's actions, row s of the tape at step s; the
// column layout then has no `action` column (ColumnsNoAct).  The quad-per-thread loops when the batch qualifies for them
// (and the tape's rows start on 4-byte boundaries), the general one-lane-per-thread loop otherwise.
template <class Env, class L>
static int launch_steps_fused_l(const typename Env::Params &p, uint32_t *state, int32_t *action, int32_t *ob,
                                typename Env::Reward *reward, uint8_t *done, uint32_t *err, int64_t n, uint64_t seed,
                                uint64_t action_seed, uint32_t lane0, uint64_t t, int k, int flags, int64_t rec, bool gen_first,
                                TapeRef tape, void *stream)
{
    constexpr bool COLS = L::ID == POMDP_LAYOUT_COLUMNS;
    const bool taped = tape.base != nullptr;
    if (!state || (!action && !(COLS && taped)) || bad_range(n, lane0) || (lane0 & 3u) || k < 1) return POMDP_E_BADARG;
    if (COLS && (!ob || !reward || !done)) return POMDP_E_BADARG;
    if (taped && (std::is_same<L, Columns>::value || tape.stride < n)) return POMDP_E_BADARG;
    if (taped) action_seed = seed;                          // no policy key is involved
    constexpr bool RETS = L::ID == LAYOUT_RETURNS;
    if (!COLS) {
        if (!(flags & POMDP_AUTO_RESET) || action_seed != seed || rec < n) return POMDP_E_BADARG;
        if (L::ID == POMDP_LAYOUT_BLOCKED && rec % TRAJ_BLOCK_LANES != 0) return POMDP_E_BADARG;
        if (L::ID == POMDP_LAYOUT_NARROW && rec % 4 != 0) return POMDP_E_BADARG;
        if (RETS && !ob) return POMDP_E_BADARG;
        gen_first = true;
        if (!RETS) { ob = nullptr; reward = nullptr; }
        done = nullptr;
    }
    if (n == 0) return 0;
    char lname[32] = "";
    snprintf(lname, sizeof lname, "%s%s%s", COLS ? "" : ", ", COLS ? "" : L::NAME, taped ? ", Tape" : "");
    // two lanes per thread from 2^19 lanes, where the batch does not qualify for a quad-per-thread loop: at 2^18 lanes the
    // one-lane-per-thread loops take 0.90 (RockSample; 1.09 with two) and 1.08 us per step (Tag)
    const bool lpt2 = Env::POOLED_LPT2 && n >= 2 * LPT2_MIN_LANES && !taped;
    const bool simple = (flags & POMDP_AUTO_RESET) && n % (lpt2 ? 2 * BLOCK : BLOCK) == 0 && !taped;
    const dim3 grid(lpt2 ? (unsigned)((n + 2 * BLOCK - 1) / (2 * BLOCK)) : blocks_for(n));
    const int kflags = (flags & POMDP_AUTO_RESET) | (gen_first ? FLAG_GEN_FIRST : 0);
    const int gf = gen_first ? 1 : 0;
    char variant[64];
#define POMDP_LAUNCH_STEPS(LPT_, SIMPLE_, GRID_)                                                                       \
    do {                                                                                                                 \
        snprintf(variant, sizeof variant, ", " #LPT_ ", " #SIMPLE_ "%s%s", COLS ? "" : ", false", lname);                 \
        note_fused("steps_kernel", Env::NAME, variant);                                                                  \
        hipLaunchKernelGGL((steps_kernel<Env, LPT_, SIMPLE_, false, L>), GRID_, dim3(BLOCK), 0, (hipStream_t)stream, state, \
                           action, ob, reward, done, err, n, make_key(seed, t), lane0, kflags, make_key(action_seed, t + 1), \
                           k, rec, p, tape);                                                                             \
    } while (0)
    // a quad-per-thread loop with the policy the launch asked for: KERNEL<..., L, Pol>
#define POMDP_QUAD_ARGS state, action, ob, reward, done, n, make_key(seed, t), lane0, make_key(action_seed, t + 1), k, rec, gf, p, tape
#define POMDP_LAUNCH_QUAD(...)                                                                                           \
    do {      /* (only the pairs that can occur are instantiated: Columns never rides a tape, ColumnsNoAct always does) */ \
        if constexpr (!std::is_same<L, Columns>::value) {                                                                \
            if (taped) hipLaunchKernelGGL((__VA_ARGS__, TapeQuad>), qgrid, dim3(BLOCK), 0, (hipStream_t)stream, POMDP_QUAD_ARGS); \
        }                                                                                                                \
        if constexpr (!std::is_same<L, ColumnsNoAct>::value) {                                                           \
            if (!taped) hipLaunchKernelGGL((__VA_ARGS__, SyntheticQuad>), qgrid, dim3(BLOCK), 0, (hipStream_t)stream, POMDP_QUAD_ARGS); \
        }                                                                                                                \
    } while (0)
    // RockSample's pooled passes exist for any number of lanes per thread; in the fused loop (no load latency to hide)
    // four per thread, with fuller passes, beat two by 5 % from 2^20 lanes up (3.97 vs 4.16 us per step) when the state
    // is one word; with two state words (K > 12) the extra registers cost more (5.04 vs 4.74 us).  Only the geometries
    // an env can take are instantiated.
    // the quad-per-thread loops move 16 bytes at a time (4 for the done bytes): columns that start on such a boundary
    // only, full workgroups of 1024 lanes, auto-reset, policy and env on one Philox key
    bool quad_ok = (reinterpret_cast<uintptr_t>(state) & 15u) == 0 && rec % 4 == 0 && action_seed == seed &&
                   (flags & POMDP_AUTO_RESET) && n % (4 * BLOCK) == 0;
    if (COLS)
        quad_ok = quad_ok && ((reinterpret_cast<uintptr_t>(taped ? nullptr : action) | reinterpret_cast<uintptr_t>(ob) | reinterpret_cast<uintptr_t>(reward)) & 15u) == 0 &&
                  (reinterpret_cast<uintptr_t>(done) & 3u) == 0;
    else
        quad_ok = quad_ok && (reinterpret_cast<uintptr_t>(action) & 15u) == 0 && (!RETS || (reinterpret_cast<uintptr_t>(ob) & 15u) == 0);
    if (taped) quad_ok = quad_ok && (reinterpret_cast<uintptr_t>(tape.base) & 3u) == 0 && tape.stride % 4 == 0;   // a quad's row = one dword
    const dim3 qgrid((unsigned)(n / (4 * BLOCK)));
    bool launched = false;
    // half a quad per thread (4-byte and returns sinks): KERNEL<..., L, Pol, 2> over n / 2 threads
#define POMDP_LAUNCH_PAIR(...)                                                                                           \
    do {                                                                                                                 \
        const dim3 pgrid((unsigned)(n / (2 * BLOCK)));                                                                   \
        if (taped) hipLaunchKernelGGL((__VA_ARGS__, TapeQuad, 2>), pgrid, dim3(BLOCK), 0, (hipStream_t)stream, POMDP_QUAD_ARGS); \
        else hipLaunchKernelGGL((__VA_ARGS__, SyntheticQuad, 2>), pgrid, dim3(BLOCK), 0, (hipStream_t)stream, POMDP_QUAD_ARGS); \
    } while (0)
    if constexpr (std::is_same<Env, TagEnv>::value) {
        if constexpr (pair_sink<L>::value) {
            if (quad_ok && n >= TAG_PAIR_MIN_LANES && n <= TAG_PAIR_MAX_LANES && p.num_opponents == 1 && k >= 16 && TagEnv::tab_ok(p)) {
                char pname[40];
                snprintf(pname, sizeof pname, "%s, 2", lname);
                note_fused("tag_steps_quad_kernel", "true", pname);
                POMDP_LAUNCH_PAIR(tag_steps_quad_kernel<true, L);
                launched = true;
            }
        }
        if (!launched && quad_ok && n >= QUAD_MIN_TAG && p.num_opponents == 1) {
            if (k >= 16 && TagEnv::tab_ok(p)) {
                note_fused("tag_steps_quad_kernel", "true", lname);
                POMDP_LAUNCH_QUAD(tag_steps_quad_kernel<true, L);
            } else {
                note_fused("tag_steps_quad_kernel", "false", lname);
                POMDP_LAUNCH_QUAD(tag_steps_quad_kernel<false, L);
            }
            launched = true;
        }
    }
    if constexpr (has_next<Env>::value) {
        if constexpr (pair_sink<L>::value) {
            // half a quad per thread where the quad loop would leave a SIMD two waves or fewer (configs[3]'s 2^19-lane shard)
            if (quad_ok && n >= BS_PAIR_MIN_LANES && n <= BS_PAIR_MAX_LANES && k <= 256) {
                char pname[40];
                snprintf(pname, sizeof pname, "%s, 2", lname);
                note_fused("battleship_steps_quad_kernel", Env::NAME, pname);
                const dim3 pgrid((unsigned)(n / (2 * BLOCK)));
                if (taped) hipLaunchKernelGGL((battleship_steps_quad_kernel<Env::WORDS / 3, L, TapeQuad, 2, true>), pgrid, dim3(BLOCK), 0, (hipStream_t)stream, POMDP_QUAD_ARGS);
                else hipLaunchKernelGGL((battleship_steps_quad_kernel<Env::WORDS / 3, L, SyntheticQuad, 2, true>), pgrid, dim3(BLOCK), 0, (hipStream_t)stream, POMDP_QUAD_ARGS);
                launched = true;
            }
        }
        if (!launched && quad_ok && n >= QUAD_MIN_BATTLESHIP && k <= 256) {   // the pool keeps a lane's deal step in a byte
            note_fused("battleship_steps_quad_kernel", Env::NAME, lname);
            // the visited mask in LDS while a CU's share of the batch fits there (see the kernel)
            if (n * (Env::WORDS / 3) <= BS_VIS_LDS_MAX_LANES * 4) {          // 12 bytes of LDS per mask word and lane: 2^19 lanes of four words,
                                                                              // 2^20 of two (the reference's 5 x 5 board: 2.39 against 2.73 us per step), ...
                if constexpr (!std::is_same<L, Columns>::value) {
                    if (taped) hipLaunchKernelGGL((battleship_steps_quad_kernel<Env::WORDS / 3, L, TapeQuad, 4, true>), qgrid, dim3(BLOCK), 0, (hipStream_t)stream, POMDP_QUAD_ARGS);
                }
                if constexpr (!std::is_same<L, ColumnsNoAct>::value) {
                    if (!taped) hipLaunchKernelGGL((battleship_steps_quad_kernel<Env::WORDS / 3, L, SyntheticQuad, 4, true>), qgrid, dim3(BLOCK), 0, (hipStream_t)stream, POMDP_QUAD_ARGS);
                }
            } else POMDP_LAUNCH_QUAD(battleship_steps_quad_kernel<Env::WORDS / 3, L);
            launched = true;
        }
    }
    if constexpr (std::is_same<Env, NetworkEnv>::value) {
        if (quad_ok && n >= QUAD_MIN_NETWORK) {
            // named as a profiler shows the instantiation: <bytes of the machine set, layout, state-table form>
            // the state-table form: the reference's default has 10 machines; its packed thresholds are T16 - 1
            const bool small = p.n_machines <= 10 && (p.fail_thr >> 37) != 0 && (p.fail_nb_thr >> 37) != 0;
            snprintf(variant, sizeof variant, "%d, %s, %s%s", small ? 2 : (p.n_machines + 7) / 8, L::NAME, small ? "true" : "false", taped ? ", Tape" : "");
            note_fused("network_steps_quad_kernel", variant, "");
            if (small) {                     // the state-table form (the reference's default has 10 machines)
                POMDP_LAUNCH_QUAD(network_steps_quad_kernel<2, L, true);
            } else
            switch ((p.n_machines + 7) / 8) {
            case 1: POMDP_LAUNCH_QUAD(network_steps_quad_kernel<1, L, false); break;
            case 2: POMDP_LAUNCH_QUAD(network_steps_quad_kernel<2, L, false); break;
            case 3: POMDP_LAUNCH_QUAD(network_steps_quad_kernel<3, L, false); break;
            default: POMDP_LAUNCH_QUAD(network_steps_quad_kernel<4, L, false); break;
            }
            launched = true;
        }
    }
    if constexpr (quad_fused<Env>::value) {
        if (quad_ok && n >= QUAD_MIN_GENERIC) {
            note_fused("steps_quad_generic_kernel", Env::NAME, lname);
            POMDP_LAUNCH_QUAD(steps_quad_generic_kernel<Env, L);
            launched = true;
        }
    }
    if constexpr (quad_tab<Env>::value) {
        // from 16 steps per launch on the lane step reads the (position, action) table the workgroup builds first and a
        // thread owns a quad of consecutive lanes (steps_quad_kernel: RockSample and StochasticRock)
        if constexpr (pair_sink<L>::value && !RETS) {
            // half a quad per thread where a quad per thread would leave a SIMD two waves or fewer (the shards of a 2^20-lane batch)
            if (quad_ok && n >= (Env::STOCHASTIC ? STOCHROCK_PAIR_MIN_LANES : ROCK_PAIR_MIN_LANES) &&
                n <= (Env::STOCHASTIC ? STOCHROCK_PAIR_MAX_LANES : ROCK_PAIR_MAX_LANES) && k >= 16 && p.num_rocks + 5 <= Env::TAB_ACTIONS) {
                char pname[40];
                snprintf(pname, sizeof pname, "%s, 2", lname);
                note_fused("steps_quad_kernel", Env::NAME, pname);
                const dim3 pgrid((unsigned)(n / (2 * BLOCK)));
                if (taped) hipLaunchKernelGGL((steps_quad_kernel<Env, L, TapeQuad, 2>), pgrid, dim3(BLOCK), 0, (hipStream_t)stream, POMDP_QUAD_ARGS);
                else hipLaunchKernelGGL((steps_quad_kernel<Env, L, SyntheticQuad, 2>), pgrid, dim3(BLOCK), 0, (hipStream_t)stream, POMDP_QUAD_ARGS);
                launched = true;
            }
        }
        if (!launched && quad_ok && n >= (Env::STOCHASTIC ? QUAD_MIN_STOCHROCK : QUAD_MIN_ROCK) && k >= 16 && p.num_rocks + 5 <= Env::TAB_ACTIONS) {
            note_fused("steps_quad_kernel", Env::NAME, lname);
            POMDP_LAUNCH_QUAD(steps_quad_kernel<Env, L);
            launched = true;
        }
    }
    if (taped) {
        if constexpr (!std::is_same<L, Columns>::value) {
            if constexpr (quad_policy_of<Finisher<Env, 1, true>>::value) {
                // full workgroups of an auto-reset batch below the quad gates: the small shards' loops (time-shared blocks, unrolled
                // by four, RockSample's table-driven step) with the tape read four rows ahead
                if (!launched && TAPE_SMALL_SHARD_LOOPS && (flags & POMDP_AUTO_RESET) && n % BLOCK == 0) {
                    bool tab = false;
                    if constexpr (quad_tab<Env>::value && Env::QUAD_SENSOR) tab = k >= 16 && p.num_rocks + 5 <= Env::TAB_ACTIONS;
                    snprintf(variant, sizeof variant, ", 1, true, %s%s", tab ? "true" : "false", lname);
                    note_fused("steps_kernel", Env::NAME, variant);
                    if constexpr (quad_tab<Env>::value && Env::QUAD_SENSOR) {
                        if (tab) hipLaunchKernelGGL((steps_kernel<Env, 1, true, true, L, true>), grid, dim3(BLOCK), 0, (hipStream_t)stream, state, action, ob,
                                                    reward, done, err, n, make_key(seed, t), lane0, kflags, make_key(action_seed, t + 1), k, rec, p, tape);
                    }
                    if (!tab) hipLaunchKernelGGL((steps_kernel<Env, 1, true, false, L, true>), grid, dim3(BLOCK), 0, (hipStream_t)stream, state, action, ob,
                                                 reward, done, err, n, make_key(seed, t), lane0, kflags, make_key(action_seed, t + 1), k, rec, p, tape);
                    launched = true;
                }
            }
            if (!launched) {                                 // any batch, any alignment: one lane per thread, the general form
                snprintf(variant, sizeof variant, ", 1, false%s%s", COLS ? "" : ", false", lname);
                note_fused("steps_kernel", Env::NAME, variant);
                hipLaunchKernelGGL((steps_kernel<Env, 1, false, false, L, true>), grid, dim3(BLOCK), 0, (hipStream_t)stream, state, action, ob,
                                   reward, done, err, n, make_key(seed, t), lane0, kflags, make_key(action_seed, t + 1), k, rec, p, tape);
            }
        }
        return (int)hipGetLastError();
    }
    if constexpr (!std::is_same<L, ColumnsNoAct>::value) {  // the synthetic policy's own forms
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
            snprintf(variant, sizeof variant, ", 1, true, true%s", lname);
            note_fused("steps_kernel", Env::NAME, variant);
            hipLaunchKernelGGL((steps_kernel<Env, 1, true, true, L>), grid, dim3(BLOCK), 0, (hipStream_t)stream, state, action, ob, reward,
                               done, err, n, make_key(seed, t), lane0, kflags, make_key(action_seed, t + 1), k, rec, p, tape);
            launched = true;
        }
    }
    if (!launched) { if (simple) POMDP_LAUNCH_STEPS(1, true, grid); else POMDP_LAUNCH_STEPS(1, false, grid); }
    }
#undef POMDP_LAUNCH_STEPS
#undef POMDP_LAUNCH_QUAD
#undef POMDP_QUAD_ARGS
    return (int)hipGetLastError();
}

template <class Env>
int launch_steps_fused(const typename Env::Params &p, uint32_t *state, int32_t *action, int32_t *ob,
                       typename Env::Reward *reward, uint8_t *done, uint32_t *err, int64_t n, uint64_t seed,
                       uint64_t action_seed, uint32_t lane0, uint64_t t, int k, int flags, int64_t rec, bool gen_first,
                       int layout, TapeRef tape, void *stream)
{
#define POMDP_FUSED_L(L_) launch_steps_fused_l<Env, L_>(p, state, action, ob, reward, done, err, n, seed, action_seed, lane0, t, k, flags, rec, gen_first, tape, stream)
    switch (layout) {
    case POMDP_LAYOUT_COLUMNS: return tape.base ? POMDP_FUSED_L(ColumnsNoAct) : POMDP_FUSED_L(Columns);
    case POMDP_LAYOUT_BLOCKED: return POMDP_FUSED_L(Blocked);
    case POMDP_LAYOUT_PACKED: return POMDP_FUSED_L(Packed);
    case POMDP_LAYOUT_NARROW: return POMDP_FUSED_L(Narrow);
    case LAYOUT_RETURNS: return POMDP_FUSED_L(Returns<Env>);
    default: return POMDP_E_BADARG;
    }
#undef POMDP_FUSED_L
}

} // namespace pomdp
