// planner.hip — planner hooks: _generate_legal, _compute_prob, fused rollouts, side statistics, History, _generate_preferred, the heuristic-policy loop.
// Part of libpomdp_hip.so; built by gym_pomdp_amd/_native.py (hipcc --offload-arch=gfx950 -O3 -std=c++17 -c, one object per file).
#include "kernels_common.hip.h"

namespace pomdp {

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
// Six waves per SIMD (at most 80 registers) where the compiler's own choice was five (81-86): the loop's CHECK / fresh-episode
// branches stall on memory, and the sixth wave fills those slots — RockSample(7,8) 4.64 -> 4.52, (15,15) 5.98 -> 5.85, Tag
// 4.35 -> 4.15 us per step at 2^20 lanes; four waves: 4.95 / 6.42 / 4.69, eight (spilling): 5.17 / 6.58 / 4.22.  BattleShip
// (three state words and more) keeps the compiler's own choice — waves_per_eu(1) constrains nothing: 81-105 registers, five
// or four waves; six would spill 100+ bytes per lane.  Every launch runs at least one step (the launcher's k >= 1): the
// outputs written after the loop are those of the launch's last step.
template <class Env> struct heur_waves { static constexpr int value = Env::WORDS <= 2 ? 6 : 1; };
template <class Env, bool RING>   // RING: a bounded RockSample history (history_push keeps its window)
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(heur_waves<Env>::value)))
void heuristic_steps_kernel(const typename Env::Params p, uint32_t *__restrict__ state,
                                                                pomdp_rock_belief b, pomdp_history h, int K, pomdp_returns R,
                                                                int32_t *__restrict__ prev_ob, int32_t *__restrict__ action,
                                                                int32_t *__restrict__ ob, typename Env::Reward *__restrict__ reward,
                                                                uint8_t *__restrict__ done, int64_t n, RngKey key0, uint32_t lane0,
                                                                int flags, int k_steps)
{
#pragma clang fp contract(off)
    // the number of rocks is 0 at compile time for the envs without rocks: their loop carries none of the per-rock code
    // (Tag 4.03 -> 3.69 us per step, BattleShip 10 x 10 81 -> 72 registers; RockSample's own code is untouched — its
    // allocation sits on the 80-register edge and any rewording of these lines has cost it 8 %)
    K = Env::HAS_ROCKS ? K : 0;
    __shared__ typename Env::Shared sh;
    const bool auto_reset = flags & POMDP_AUTO_RESET;
    const uint32_t idx = blockIdx.x * (uint32_t)BLOCK + threadIdx.x;
    const bool in_range = (uint64_t)idx < (uint64_t)n;
    const uint32_t i = in_range ? idx : (uint32_t)(n - 1);     // memory index only: threads past n read lane n - 1's words
    // the lane id — and with it the quad element e that picks which step's quad-shared block this thread computes — comes
    // from the UNCLAMPED index, as in rollout_kernel / steps_kernel: the padding threads of a ragged last quad (n % 4 != 0)
    // still supply the blocks of steps base + 1 .. 3 to the quad's in-range lanes through quad_transpose4
    const uint32_t lane = lane0 + idx;
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
    // what the launch's LAST step returns: every step overwrites the same n-element outputs, so only the last one's values
    // are ever visible — they are written once, after the loop (round 4).  Inside the loop they were four stores per step
    // that every CHECK's read-modify-write of the side statistics then waited behind (loads and stores share one counter).
    int out_a = -1, out_o = 0, out_d = 0;
    typename Env::Reward out_r = 0;
    const int hcap = h.max_size >= 0 ? h.max_size + 1 : 0x7FFFFFFF;            // len(history) stops there (rock.py:541-544)
    const uint64_t t0 = ((uint64_t)key0.t_hi << 32) | key0.t_lo;
    const uint32_t e = lane & 3u;
    // Random words four steps at a time, as in the rollout kernel: the policy's ACTION block is shared by the four lanes
    // of a quad (and so is RockSample's STEP block), so lane e of a quad computes the block(s) of step base + e and the
    // words travel by DPP quad-broadcast — one block per lane per four steps instead of four.
    // (no priority ladder here — LoopPrio, kernels_common.hip.h: this loop's launches run several rounds of workgroups and
    // its CHECK steps load; measured 2.17 -> 2.02e11 steps/s on RockSample(7,8) with it, no change on Tag)
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
            if constexpr (Env::QUAD_SENSOR) {
                // RockSample: the fresh episode starts from the word the step's sensor draw would have read (auto-reset
                // contract, rock.hip.h) — this lane's word of the quad's block is already here; reset_where would compute
                // that block again, per lane, in every wave-step in which some lane's episode ends
                st.s = fresh ? Env::fresh_state(p, comp<J>(sq), key, lane) : st.s;
            } else {
                Env::reset_where(sh, p, st, fresh, key, lane);                 // wave-cooperative: every lane calls it
            }
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
            out_a = live ? a : -1; out_o = o; out_r = r; out_d = d;
            if (live) was_done = auto_reset ? false : (d != 0);
        };
        one_step(std::integral_constant<int, 0>{});
        one_step(std::integral_constant<int, 1>{});
        one_step(std::integral_constant<int, 2>{});
        one_step(std::integral_constant<int, 3>{});
    }
    if (!in_range) return;
    st_stream(action + i, (int32_t)out_a);
    st_stream(ob + i, (int32_t)out_o);
    st_stream(reward + i, out_r);
    st_stream(done + i, (uint8_t)out_d);
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
    constexpr bool TAB = ROLLOUT_TAB<Env>::value;            // RockSample: the (position, action) table of the fused loops,
    constexpr bool REC = TAB;                                // whose lane step yields the packed record (step_rec)
    __shared__ typename step_tab_of<Env, TAB>::type tab;
    Env::stage(sh, p, (int)threadIdx.x);
    __syncthreads();
    if constexpr (TAB) {
        Env::build_rec_tab(tab, sh, p, (int)threadIdx.x);
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
        if constexpr (Env::QUAD_SENSOR || quad_word_env<Env>::value) {   // this lane's share: the quad's STEP block of step base + (lane & 3)
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
                if constexpr (REC) {
                    // a simulation stops at its terminal step, so the state after one is never read: no fresh episode to pass
                    uint32_t rec;
                    Env::step_rec(sh, tab, nx.s, (uint32_t)a, comp<J>(sq), nx.s, rec,
                                  [&]() { return Env::elem(Env::quad_block(key, lane, 1u), lane & 3u); });
                    o2 = (int)__builtin_amdgcn_ubfe(rec, 8u, 8u);
                    r = (double)(int32_t)__builtin_amdgcn_sbfe(rec, 16u, 8u);
                    d2 = (int)(rec >> 24);
                }
                else Env::step_with_H(sh, p, nx, a, key, lane, comp<J>(sq), o2, r, d2);
            } else if constexpr (quad_word_env<Env>::value) {          // Tiger, Tag: the lane's word of the quad's block
                Env::step_w(sh, p, nx, a, key, lane, comp<J>(sq), o2, r, d2);
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
        first_action[i] = first;
        if (n_steps) n_steps[i] = k;                          // kernel arguments: wave-uniform
        if (last_ob) last_ob[i] = o;
        if (terminated) terminated[i] = (uint8_t)d;
    }
}

// The planning step's reduction (include/pomdp_hip.h: pomdp_plan): one workgroup per root turns the root's simulations into
// visits / mean return per first action and picks the best one.  The order of the float64 additions is part of the
// contract — chunks of 64 simulations by index, simulation-index order within a chunk, chunk order across chunks, each sum
// starting from +0.0 — so the layout follows it: the root's returns and first actions are staged through LDS a tile of
// PLAN_TILE = 16 chunks at a time; the tile's (chunk, action) pairs are dealt out one per thread, each walking its chunk in
// index order (an addition of nothing leaves the sum alone, which is what skipping a simulation means, and a sum that
// started at +0.0 is never -0.0); the per-chunk sums go through LDS and thread a adds them in chunk order into its running
// total across tiles.  No atomics, no cross-workgroup traffic.  LDS is sized by the action count (dynamic: 13 KB for
// RockSample(15,15)'s 20 actions), so that a CU holds eight roots at once — configs[4]'s 2048 roots are one round of
// workgroups; a chunk's returns sit 65 doubles apart so that threads on different chunks read different banks.
constexpr int PLAN_CHUNKS = 16, PLAN_TILE = PLAN_CHUNKS * POMDP_PLAN_CHUNK, PLAN_ROW = POMDP_PLAN_CHUNK + 1;
static inline size_t plan_lds_bytes(int n_act)
{
    return sizeof(double) * (PLAN_CHUNKS * PLAN_ROW + (size_t)PLAN_CHUNKS * n_act + n_act) + sizeof(int32_t) * n_act +
           PLAN_TILE + (size_t)PLAN_CHUNKS * n_act;
}
__global__ __launch_bounds__(BLOCK) void plan_reduce_kernel(const double *__restrict__ ret, const int32_t *__restrict__ first_action,
                                                            int64_t sims, int n_act, pomdp_plan_out out)
{
#pragma clang fp contract(off)
    extern __shared__ double plan_lds[];
    double *const r_lds = plan_lds;                                        // [PLAN_CHUNKS][PLAN_ROW]
    double *const part = r_lds + PLAN_CHUNKS * PLAN_ROW;                   // [PLAN_CHUNKS][n_act]
    double *const q_lds = part + PLAN_CHUNKS * n_act;                      // [n_act]
    int32_t *const n_lds = reinterpret_cast<int32_t *>(q_lds + n_act);     // [n_act]
    uint8_t *const a_lds = reinterpret_cast<uint8_t *>(n_lds + n_act);     // [PLAN_TILE]; 255: no step taken / out of range
    uint8_t *const cnt = a_lds + PLAN_TILE;                                // [PLAN_CHUNKS][n_act], <= 64 each
    const int64_t root = blockIdx.x;
    const int tid = (int)threadIdx.x;
    const double *const rr = ret + root * sims;
    const int32_t *const fa = first_action + root * sims;
    double total = 0.0;                                      // thread a < n_act: action a
    int32_t visits = 0;
    for (int64_t base = 0; base < sims; base += PLAN_TILE) {
        const int tile = (int)(sims - base < PLAN_TILE ? sims - base : PLAN_TILE);
        for (int j = tid; j < tile; j += BLOCK) {
            r_lds[(j >> 6) * PLAN_ROW + (j & 63)] = rr[base + j];
            const int32_t f = fa[base + j];
            a_lds[j] = (uint8_t)((uint32_t)f < (uint32_t)n_act ? f : 255);
        }
        __syncthreads();
        const int n_chunks = (tile + POMDP_PLAN_CHUNK - 1) / POMDP_PLAN_CHUNK;
        for (int item = tid; item < n_chunks * n_act; item += BLOCK) {
            const int c = item / n_act, a = item - c * n_act;
            const int len = tile - c * POMDP_PLAN_CHUNK < POMDP_PLAN_CHUNK ? tile - c * POMDP_PLAN_CHUNK : POMDP_PLAN_CHUNK;
            const double *const rc = r_lds + c * PLAN_ROW;
            const uint8_t *const ac = a_lds + c * POMDP_PLAN_CHUNK;
            double p = 0.0;
            int k = 0;
            for (int j = 0; j < len; ++j) {
                const bool hit = (int)ac[j] == a;
                const double with = p + rc[j];
                p = hit ? with : p;
                k += (int)hit;
            }
            part[c * n_act + a] = p;
            cnt[c * n_act + a] = (uint8_t)k;
        }
        __syncthreads();
        if (tid < n_act)
            for (int c = 0; c < n_chunks; ++c) { total = total + part[c * n_act + tid]; visits += (int32_t)cnt[c * n_act + tid]; }
        __syncthreads();                                     // the next tile overwrites r_lds / part
    }
    if (tid < n_act) {
        const double q = visits > 0 ? total / (double)visits : 0.0;
        out.q[root * out.stride + tid] = q;
        out.visits[root * out.stride + tid] = visits;
        q_lds[tid] = q;
        n_lds[tid] = visits;
    }
    __syncthreads();
    if (tid == 0) {                                          // n_act <= 255 values: the first strict maximum among the visited actions
        int b = -1;
        double bq = 0.0;
        for (int a = 0; a < n_act; ++a)
            if (n_lds[a] > 0 && (b < 0 || q_lds[a] > bq)) { b = a; bq = q_lds[a]; }
        out.best[root] = b;
        if (out.value) out.value[root] = b >= 0 ? bq : 0.0;
    }
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
    if (!state || !ret || !first_action || n_roots < 0 || sims < 1 || depth < 0 ||
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
    if (h->max_size < -1 || h->max_size > 0x7FFFFFFE) return false;           // any window the caller has a (max_size + 1) x n byte ring for
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
    const int64_t FUSE_MAX = fuse_max();                  // steps per launch
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

extern "C" {

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

static bool plan_out_ok(const pomdp_plan_out *o, int n_act)
{
    return o && o->q && o->visits && o->best && n_act >= 1 && n_act <= 255 && o->stride >= n_act;
}

int pomdp_plan_reduce(const double *ret, const int32_t *first_action, int64_t n_roots, int64_t sims_per_root, int n_actions,
                      const pomdp_plan_out *out, void *stream)
{
    if (!ret || !first_action || n_roots < 0 || n_roots > 0x7FFFFFFF || sims_per_root < 1 || !plan_out_ok(out, n_actions))
        return POMDP_E_BADARG;
    if (n_roots == 0) return 0;
    hipLaunchKernelGGL(plan_reduce_kernel, dim3((unsigned)n_roots), dim3(BLOCK), plan_lds_bytes(n_actions), (hipStream_t)stream, ret,
                       first_action, sims_per_root, n_actions, *out);
    return (int)hipGetLastError();
}

int pomdp_plan(int env, const void *params, const uint32_t *root_state, int64_t n_roots, int64_t sims_per_root, int depth,
               double discount, int flags, uint64_t seed, uint32_t lane0, uint64_t t0, double *sim_ret,
               int32_t *sim_first_action, const pomdp_plan_out *out, void *stream)
{
    if (!params || !out) return POMDP_E_BADARG;
    // everything is checked before anything is enqueued
    int rc = dispatch_env(env, params, [](auto, const auto &) { return 0; });   // POMDP_E_BADPARAMS / unknown env
    if (rc) return rc;
    const int n_act = (int)env_action_count(env, params);
    if (n_roots > 0x7FFFFFFF || !plan_out_ok(out, n_act)) return POMDP_E_BADARG;
    rc = pomdp_rollout(env, params, root_state, n_roots, sims_per_root, depth, discount, flags, seed, lane0, t0, sim_ret, nullptr,
                       sim_first_action, nullptr, nullptr, stream);
    if (rc) return rc;
    return pomdp_plan_reduce(sim_ret, sim_first_action, n_roots, sims_per_root, n_act, out, stream);
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

} // extern "C"
