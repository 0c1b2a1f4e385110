// traj_out.hip.h — where a fused launch puts a step's (action, ob, reward, done): the three trajectory layouts of
// include/pomdp_hip.h (POMDP_LAYOUT_*), as the "sinks" the fused step loops of fused_impl.hip.h are instantiated with.
//
//   Columns  the default ABI: four separate [step][pitch] columns (action int32 — row s + 1 receives the NEXT actions —
//            ob int32, reward int32 | float, done uint8).  A wave-step writes four 1 KB / 256 B pieces that lie a whole
//            column (4 MB x steps) apart: four concurrent write streams.  How fast they drain depends on where the
//            allocation's pages lie (129-162 us for the same 64-step launch).
//   Blocked  the same 13 bytes per lane-step, same int32 / float values, ONE write stream: row s of the trajectory is
//            n / 256 blocks of 3328 bytes, block q = lanes 256 q .. 256 q + 255 = action int32[256] | ob int32[256] |
//            reward[256] | done uint8[256].  A wave of a quad-per-thread loop owns exactly one block per step: its four
//            stores land in one contiguous, 256-byte-aligned 3328-byte piece, a workgroup's in 13 KB, a step's in one
//            contiguous n x 13 bytes.
//   Packed   one 32-bit record per lane-step: action | ob << 8 | reward code << 16 | done << 24 (every env's actions and
//            observations fit a byte; the reward code is the int8 value itself for the integer-valued rewards of
//            RockSample / Tag / BattleShip / Tiger and an index into Network's 3 x 68 reward table: Env::reward_code).
//            4 bytes per lane-step instead of 13: one 16-byte store per thread-step of a quad-per-thread loop, and the
//            fused loops become bound by instruction issue instead of by the write stream.
//
//   Narrow   the Packed record's four bytes as four typed planes: row s = action uint8[pitch] | ob uint8[pitch] | reward code
//            (u)int8[pitch] | done uint8[pitch].  The same 4 bytes per lane-step, and every plane of every step is an array a
//            consumer reads in place — nothing to decode.  A quad-per-thread loop transposes its four records (eight
//            v_perm_b32) and stores four bytes per plane: four 256-byte pieces per wave-step instead of one 1 KB piece.
//   Returns  no trajectory at all: the reduction the reference's callers apply to the stream — r += discount * rw; discount
//            *= .95 per step, one return per episode (network.py:175-191, rock.py:553-575) — kept per lane in registers
//            across the launch and written once when it ends (include/pomdp_hip.h: pomdp_collect_returns).
//
// Blocked, Packed and Narrow rows hold the action TAKEN at step s (row s of every field belongs to step s); they have no row
// of "next actions" — a launch derives its first actions from the synthetic policy itself (they are a function of (seed,
// lane, t) only), which is what the Columns layout's gen_first launches do as well.
#pragma once
#include "envs.hip.h"

namespace pomdp {

// CODES: the sink reads the step's reward code (Env::reward_code) — the loops compute it only then
constexpr int LAYOUT_RETURNS = 100;                         // launcher-internal: not a trajectory layout of the ABI
// ACT: the columns include `action` (row s + 1 = the policy's next actions).  A tape-driven launch leaves it out — the caller
// holds the actions already — and writes ob / reward / done only: 9 bytes per lane-step.
template <bool ACT> struct ColumnsT { static constexpr int ID = POMDP_LAYOUT_COLUMNS; static constexpr const char *NAME = "Columns"; static constexpr bool CODES = false; };
using Columns = ColumnsT<true>;
using ColumnsNoAct = ColumnsT<false>;
struct Blocked { static constexpr int ID = POMDP_LAYOUT_BLOCKED; static constexpr const char *NAME = "Blocked"; static constexpr bool CODES = false; };
struct Packed  { static constexpr int ID = POMDP_LAYOUT_PACKED;  static constexpr const char *NAME = "Packed";  static constexpr bool CODES = true; };
struct Narrow  { static constexpr int ID = POMDP_LAYOUT_NARROW;  static constexpr const char *NAME = "Narrow";  static constexpr bool CODES = true; };
// Env: whose reward codes the sink turns back into the reference's float64 rewards (Env::code_reward)
template <class Env> struct Returns { static constexpr int ID = LAYOUT_RETURNS; static constexpr const char *NAME = "Returns"; static constexpr bool CODES = true; using E = Env; };

constexpr int TRAJ_BLOCK_LANES = 256;                       // lanes per block of the Blocked layout = one wave's four per thread
constexpr int TRAJ_BLOCK_BYTES = 13 * TRAJ_BLOCK_LANES;     // 3 x 1024 bytes of int32 / float + 256 done bytes

static __device__ __forceinline__ uint32_t pack_record(uint32_t a, uint32_t o, uint32_t rcode, uint32_t d)
{
    // a, o < 256 and d in {0, 1} by construction; the reward code may carry an int8's sign bits
    return a | (o << 8) | ((rcode & 0xFFu) << 16) | (d << 24);
}

// ---- where a fused loop's actions come from ------------------------------------------------------------------------------
// Synthetic: the bench's uniform random policy — stream ACTION of (seed, quad, t), generated inside the loop (the launches of
// pomdp_collect_synthetic / _layout / _returns).  Tape: the CALLER's actions (pomdp_collect_tape*), `uint8 [k_steps][stride]`
// in HBM, row s = the actions of the launch's step s.  gfx9 counts loads and stores on one counter and returns them in
// order, so waiting for a load also waits for every store issued BEFORE it: the row of step s + 1 is therefore requested at
// the TOP of step s (begin) — the only older stores are step s - 1's, a whole step old by the time anybody waits — and first
// touched at the END of step s (end), after the step's own stores have been issued, with the step's arithmetic in between to
// cover the latency.  Request and use sit in the same loop iteration, so the compiler's own `s_waitcnt vmcnt(n)` counts
// exactly the stores issued after the load; nothing pending is carried round the loop.
struct TapeRef {
    const uint8_t *base;     // row 0 of this launch; nullptr: the synthetic policy
    int64_t stride;          // bytes from one step's row to the next
    uint32_t *err;           // device counter of out-of-range actions (may be nullptr)
};

template <class T>           // T = uint32_t: the four lanes of a quad (one dword per row); uint8_t: one lane
struct TapeColumn {
    const uint8_t *p;        // this thread's element of row 0
    int64_t stride;
    int last;                // k_steps - 1: rows past it are never requested
    T first, nxt;
    __device__ __forceinline__ T row(int r) const
    {
        r = r < last ? r : last;
        return ld_stream(reinterpret_cast<const T *>(p + (int64_t)r * stride));
    }
    __device__ __forceinline__ TapeColumn(const TapeRef &t, uint32_t col, int k_steps)
        : p(t.base + col), stride(t.stride), last(k_steps - 1), nxt(0) { first = row(0); }
    __device__ __forceinline__ void request(int s) { nxt = row(s + 1); }   // top of step s: the actions of step s + 1
};
struct NoColumn {            // the same interface for a loop that is not tape-driven: nothing is loaded
    uint8_t first, nxt;
    __device__ __forceinline__ NoColumn(const TapeRef &, uint32_t, int) : first(0), nxt(0) {}
    __device__ __forceinline__ void request(int) {}
};

// quad-per-thread loops: l0 = the thread's first lane within the shard, glane0 its global id.  begin(s, a_next) at the top of
// step s, end(s, a_next) after the step's stores: between them a_next holds the actions of step s + 1 (Synthetic) or nothing yet.
struct SyntheticQuad {
    static constexpr bool TAPE = false;
    uint32_t glane0, n_act, k0, k1;
    uint64_t ta0;
    // (the policy shares the env's Philox key in every launch that takes these loops: the key words are the ENV key's, so that
    // the compiler keeps one copy of them in scalar registers — a second copy made Network's loop spill scalars)
    __device__ __forceinline__ SyntheticQuad(const TapeRef &, uint32_t, uint32_t glane0_, const RngKey &key0, const RngKey &akey0, uint32_t n_act_, int)
        : glane0(glane0_), n_act(n_act_), k0(key0.k0), k1(key0.k1), ta0(((uint64_t)akey0.t_hi << 32) | akey0.t_lo) {}
    __device__ __forceinline__ uint4 block(uint64_t ta) const
    {
        return philox4x32_10(glane0 >> 2, (uint32_t)ta, (uint32_t)(ta >> 32), (uint32_t)POMDP_STREAM_ACTION << 24, k0, k1);
    }
    // the actions of the call counter BEFORE akey0's: what pomdp_synthetic_actions would have written for the launch's first step
    __device__ __forceinline__ u32x4 first() const
    {
        const uint4 w = block(ta0 - 1ull);
        return u32x4{__umulhi(w.x, n_act), __umulhi(w.y, n_act), __umulhi(w.z, n_act), __umulhi(w.w, n_act)};
    }
    __device__ __forceinline__ void begin(int s, uint32_t (&a)[4]) const
    {
        const uint4 pw = block(ta0 + (uint64_t)s);
        a[0] = __umulhi(pw.x, n_act); a[1] = __umulhi(pw.y, n_act); a[2] = __umulhi(pw.z, n_act); a[3] = __umulhi(pw.w, n_act);
    }
    __device__ __forceinline__ void end(int, uint32_t (&)[4]) const {}
    __device__ __forceinline__ void count_bad(uint32_t) const {}
};
struct TapeQuad {
    static constexpr bool TAPE = true;
    TapeColumn<uint32_t> col;
    uint32_t *err;
    static __device__ __forceinline__ void unpack(uint32_t w, uint32_t (&a)[4])
    {
        a[0] = w & 0xFFu; a[1] = __builtin_amdgcn_ubfe(w, 8u, 8u); a[2] = __builtin_amdgcn_ubfe(w, 16u, 8u); a[3] = w >> 24;
    }
    __device__ __forceinline__ TapeQuad(const TapeRef &t, uint32_t l0, uint32_t, const RngKey &, const RngKey &, uint32_t, int k_steps)
        : col(t, l0, k_steps), err(t.err) {}
    __device__ __forceinline__ u32x4 first() const
    {
        uint32_t a[4];
        unpack(col.first, a);
        return u32x4{a[0], a[1], a[2], a[3]};
    }
    __device__ __forceinline__ void begin(int s, uint32_t (&a)[4]) { col.request(s); a[0] = a[1] = a[2] = a[3] = 0; }
    __device__ __forceinline__ void end(int, uint32_t (&a)[4]) const { unpack(col.nxt, a); }
    __device__ __forceinline__ void count_bad(uint32_t n_bad) const { if (n_bad && err) atomicAdd(err, n_bad); }
};

// The same tape read TWO steps ahead, for a loop unrolled by two (steps_quad_kernel): row r lives in register r & 1; the top
// of step s asks for row s + 2 into the register row s left (its actions were taken over when step s - 1 ended), the end of
// step s reads row s + 1 — asked for at the top of step s - 1, two steps' arithmetic and one step's stores ago.  The register
// names are compile-time (PAR = s & 1): a pending load is never moved, selected or indexed, only waited for where it is read,
// and the wait counts the younger loads and stores exactly (tests/test_host_logic.py checks the compiled loop).
struct TapeQuadAhead {
    static constexpr bool TAPE = true;
    TapeColumn<uint32_t> col;
    uint32_t *err;
    uint32_t even, odd;      // rows 2 i, 2 i + 1 of the tape as they come round
    __device__ __forceinline__ TapeQuadAhead(const TapeRef &t, uint32_t l0, uint32_t, const RngKey &, const RngKey &, uint32_t, int k_steps)
        : col(t, l0, k_steps), err(t.err), even(0), odd(col.row(1)) {}
    __device__ __forceinline__ u32x4 first() const
    {
        uint32_t a[4];
        TapeQuad::unpack(col.first, a);
        return u32x4{a[0], a[1], a[2], a[3]};
    }
    template <int PAR> __device__ __forceinline__ void begin_par(int s, uint32_t (&a)[4])
    {
        if constexpr (PAR == 0) even = col.row(s + 2); else odd = col.row(s + 2);
        a[0] = a[1] = a[2] = a[3] = 0;
    }
    template <int PAR> __device__ __forceinline__ void end_par(int, uint32_t (&a)[4]) const { TapeQuad::unpack(PAR == 0 ? odd : even, a); }
    __device__ __forceinline__ void count_bad(uint32_t n_bad) const { if (n_bad && err) atomicAdd(err, n_bad); }
};

// ---- two lanes per thread: half a quad (BattleShip's small shards, fused_impl.hip.h) ----------------------------------------
// The policy's block belongs to a quad, i.e. to a PAIR of neighbouring threads: thread e (0 / 1: lanes 0-1 / 2-3 of the quad)
// computes the block of step s + e at every even s, and the pair swaps the halves the partner needs (two DPP moves) — one
// Philox block per thread per two steps, as many per lane-step as in the quad-per-thread loops.
static __device__ __forceinline__ uint32_t pair_swap(uint32_t v)      // v of the neighbouring thread (lanes 2 i <-> 2 i + 1 of the wave)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]
}
// the two words of a quad-shared block stream that belong to this thread's two lanes, at step s: at every even s thread `hi`
// (lanes 2, 3 of the quad) computes the block of step s + 1, its neighbour the block of step s, and they swap the halves
template <class BlockOfStep>
static __device__ __forceinline__ void pair_shared(int s, bool hi, BlockOfStep block_of_step, uint32_t (&w)[2], uint32_t &odd0, uint32_t &odd1)
{
    if ((s & 1) == 0) {                                       // wave-uniform
        const uint4 b = block_of_step(s + (hi ? 1 : 0));
        const uint32_t r0 = pair_swap(hi ? b.x : b.z), r1 = pair_swap(hi ? b.y : b.w);
        w[0] = hi ? r0 : b.x; w[1] = hi ? r1 : b.y;
        odd0 = hi ? b.z : r0; odd1 = hi ? b.w : r1;           // kept by the caller for step s + 1
    } else { w[0] = odd0; w[1] = odd1; }
}
struct SyntheticPair {
    static constexpr bool TAPE = false;
    uint32_t glane0, n_act, k0, k1, odd[2];
    uint64_t ta0;
    bool hi;                 // this thread holds lanes 2, 3 of its quad
    __device__ __forceinline__ SyntheticPair(const TapeRef &, uint32_t, uint32_t glane0_, const RngKey &key0, const RngKey &akey0, uint32_t n_act_, int)
        : glane0(glane0_), n_act(n_act_), k0(key0.k0), k1(key0.k1), odd{0, 0}, ta0(((uint64_t)akey0.t_hi << 32) | akey0.t_lo), hi((glane0_ & 2u) != 0u) {}
    __device__ __forceinline__ uint4 block(uint64_t ta) const
    {
        return philox4x32_10(glane0 >> 2, (uint32_t)ta, (uint32_t)(ta >> 32), (uint32_t)POMDP_STREAM_ACTION << 24, k0, k1);
    }
    __device__ __forceinline__ void first(uint32_t (&a)[2]) const
    {
        const uint4 w = block(ta0 - 1ull);
        a[0] = __umulhi(hi ? w.z : w.x, n_act); a[1] = __umulhi(hi ? w.w : w.y, n_act);
    }
    __device__ __forceinline__ void begin(int s, uint32_t (&a)[2])
    {
        if ((s & 1) == 0) {                                   // wave-uniform: this thread's block is step s + hi's
            const uint4 w = block(ta0 + (uint64_t)s + (hi ? 1ull : 0ull));
            const uint32_t r0 = pair_swap(hi ? w.x : w.z), r1 = pair_swap(hi ? w.y : w.w);   // what the partner lacks <-> what it sends
            a[0] = __umulhi(hi ? r0 : w.x, n_act); a[1] = __umulhi(hi ? r1 : w.y, n_act);     // step s: the low thread's block
            odd[0] = __umulhi(hi ? w.z : r0, n_act); odd[1] = __umulhi(hi ? w.w : r1, n_act); // step s + 1: the high thread's
        } else { a[0] = odd[0]; a[1] = odd[1]; }
    }
    __device__ __forceinline__ void end(int, uint32_t (&)[2]) const {}
    __device__ __forceinline__ void count_bad(uint32_t) const {}
};
struct TapePair {
    static constexpr bool TAPE = true;
    TapeColumn<uint16_t> col;
    uint32_t *err;
    __device__ __forceinline__ TapePair(const TapeRef &t, uint32_t l0, uint32_t, const RngKey &, const RngKey &, uint32_t, int k_steps)
        : col(t, l0, k_steps), err(t.err) {}
    __device__ __forceinline__ void first(uint32_t (&a)[2]) const { a[0] = col.first & 0xFFu; a[1] = (uint32_t)col.first >> 8; }
    __device__ __forceinline__ void begin(int s, uint32_t (&a)[2]) { col.request(s); a[0] = a[1] = 0; }
    __device__ __forceinline__ void end(int, uint32_t (&a)[2]) const { a[0] = col.nxt & 0xFFu; a[1] = (uint32_t)col.nxt >> 8; }
    __device__ __forceinline__ void count_bad(uint32_t n_bad) const { if (n_bad && err) atomicAdd(err, n_bad); }
};
template <class Pol, int LPT> struct lanes_policy { using type = Pol; };
template <> struct lanes_policy<SyntheticQuad, 2> { using type = SyntheticPair; };
template <> struct lanes_policy<TapeQuad, 2> { using type = TapePair; };

// ---- a thread that owns a quad of consecutive lanes (the quad-per-thread loops) ------------------------------------
// l0: the thread's first lane within the shard (a multiple of 4).  first(): the actions of the launch's first step.
// put(): one step's results of the four lanes — a_cur the actions taken, a_next the policy's actions of the next call
// counter, o / r (raw 32-bit patterns) / rc (reward codes, read by Packed only) / d (0 or 1) — then on to the next row.
template <class L> struct QuadOut;

template <bool ACT> struct QuadOut<ColumnsT<ACT>> {
    uint32_t *action_w, *ob_w, *reward_w, *done_w;
    int64_t rec;
    __device__ __forceinline__ QuadOut(void *action, void *ob, void *reward, void *done, int64_t rec_, uint32_t l0)
        : action_w(ACT ? reinterpret_cast<uint32_t *>(action) + l0 : nullptr), ob_w(reinterpret_cast<uint32_t *>(ob) + l0),
          reward_w(reinterpret_cast<uint32_t *>(reward) + l0), done_w(reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(done) + l0)),
          rec(rec_) {}
    // read from row 0 of `action`, or (gen_first, wave-uniform) the policy's own first actions, written to that row
    template <class Pol>
    __device__ __forceinline__ u32x4 first(const Pol &pol, int gen_first)
    {
        if constexpr (!ACT) return pol.first();
        u32x4 a;
        if (!gen_first) a = ld_stream4(action_w);
        else {
            a = pol.first();
            st_stream4(action_w, a[0], a[1], a[2], a[3]);
        }
        action_w += rec;
        return a;
    }
    __device__ __forceinline__ void put(const uint32_t (&)[4], const uint32_t (&a_next)[4], const uint32_t (&o)[4],
                                        const uint32_t (&r)[4], const uint32_t (&)[4], const uint32_t (&d)[4])
    {
        if constexpr (ACT) st_stream4(action_w, a_next[0], a_next[1], a_next[2], a_next[3]);
        st_stream4(ob_w, o[0], o[1], o[2], o[3]);
        st_stream4(reward_w, r[0], r[1], r[2], r[3]);
        st_stream(done_w, d[0] | (d[1] << 8) | (d[2] << 16) | (d[3] << 24));
        action_w += rec; ob_w += rec; reward_w += rec; done_w += rec / 4;
    }
    // the same from the lanes' packed records (a lane step that produces the record directly: RockEnv::step_rec) — unpacked
    // here; valid for the envs whose reward code is the int8 reward itself
    __device__ __forceinline__ void put_records(const uint32_t (&rec)[4], const uint32_t (&a_next)[4])
    {
        uint32_t a[4], o[4], r[4], d[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            a[j] = rec[j] & 0xFFu;
            o[j] = __builtin_amdgcn_ubfe(rec[j], 8u, 8u);
            r[j] = (uint32_t)__builtin_amdgcn_sbfe(rec[j], 16u, 8u);
            d[j] = rec[j] >> 24;
        }
        put(a, a_next, o, r, r, d);
    }
    __device__ __forceinline__ void finish(int) {}
};

template <> struct QuadOut<Blocked> {
    uint8_t *w, *wd;                                        // the thread's 16 bytes of the block's action section; its 4 done bytes
    int64_t row_bytes;
    __device__ __forceinline__ QuadOut(void *base, void *, void *, void *, int64_t rec_, uint32_t l0)
        : w(reinterpret_cast<uint8_t *>(base) + (int64_t)(l0 >> 8) * TRAJ_BLOCK_BYTES + (l0 & 255u) * 4u),
          wd(reinterpret_cast<uint8_t *>(base) + (int64_t)(l0 >> 8) * TRAJ_BLOCK_BYTES + 3 * 1024 + (l0 & 255u)),
          row_bytes(rec_ * 13) {}
    template <class Pol> __device__ __forceinline__ u32x4 first(const Pol &pol, int) { return pol.first(); }
    __device__ __forceinline__ void put(const uint32_t (&a_cur)[4], const uint32_t (&)[4], const uint32_t (&o)[4],
                                        const uint32_t (&r)[4], const uint32_t (&)[4], const uint32_t (&d)[4])
    {
        st_stream4(reinterpret_cast<uint32_t *>(w), a_cur[0], a_cur[1], a_cur[2], a_cur[3]);
        st_stream4(reinterpret_cast<uint32_t *>(w + 1024), o[0], o[1], o[2], o[3]);
        st_stream4(reinterpret_cast<uint32_t *>(w + 2048), r[0], r[1], r[2], r[3]);
        st_stream(reinterpret_cast<uint32_t *>(wd), d[0] | (d[1] << 8) | (d[2] << 16) | (d[3] << 24));
        w += row_bytes; wd += row_bytes;
    }
    // the same from the lanes' packed records (a lane step that produces the record directly: RockEnv::step_rec) — unpacked
    // here; valid for the envs whose reward code is the int8 reward itself
    __device__ __forceinline__ void put_records(const uint32_t (&rec)[4], const uint32_t (&a_next)[4])
    {
        uint32_t a[4], o[4], r[4], d[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            a[j] = rec[j] & 0xFFu;
            o[j] = __builtin_amdgcn_ubfe(rec[j], 8u, 8u);
            r[j] = (uint32_t)__builtin_amdgcn_sbfe(rec[j], 16u, 8u);
            d[j] = rec[j] >> 24;
        }
        put(a, a_next, o, r, r, d);
    }
    __device__ __forceinline__ void finish(int) {}
};

template <> struct QuadOut<Packed> {
    uint32_t *w;
    int64_t rec;
    __device__ __forceinline__ QuadOut(void *base, void *, void *, void *, int64_t rec_, uint32_t l0)
        : w(reinterpret_cast<uint32_t *>(base) + l0), rec(rec_) {}
    template <class Pol> __device__ __forceinline__ u32x4 first(const Pol &pol, int) { return pol.first(); }
    __device__ __forceinline__ void put(const uint32_t (&a_cur)[4], const uint32_t (&)[4], const uint32_t (&o)[4],
                                        const uint32_t (&)[4], const uint32_t (&rc)[4], const uint32_t (&d)[4])
    {
        st_stream4(w, pack_record(a_cur[0], o[0], rc[0], d[0]), pack_record(a_cur[1], o[1], rc[1], d[1]),
                   pack_record(a_cur[2], o[2], rc[2], d[2]), pack_record(a_cur[3], o[3], rc[3], d[3]));
        w += rec;
    }
    __device__ __forceinline__ void put_records(const uint32_t (&r)[4], const uint32_t (&)[4])
    {
        st_stream4(w, r[0], r[1], r[2], r[3]);
        w += rec;
    }
    __device__ __forceinline__ void finish(int) {}
};

template <> struct QuadOut<Narrow> {
    uint32_t *w;                                            // the thread's four bytes of the row's action plane
    int64_t plane, row;                                     // in 32-bit words: a plane is `pitch` bytes, a row four planes
    __device__ __forceinline__ QuadOut(void *base, void *, void *, void *, int64_t rec_, uint32_t l0)
        : w(reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(base) + l0)), plane(rec_ / 4), row(rec_) {}
    template <class Pol> __device__ __forceinline__ u32x4 first(const Pol &pol, int) { return pol.first(); }
    // 4 x 4 byte transpose: byte b of record j -> byte j of plane b (v_perm_b32: selector 0-3 = bytes of the second
    // operand, 4-7 = bytes of the first)
    __device__ __forceinline__ void put_records(const uint32_t (&r)[4], const uint32_t (&)[4])
    {
        const uint32_t t0 = __builtin_amdgcn_perm(r[1], r[0], 0x05010400u), t1 = __builtin_amdgcn_perm(r[1], r[0], 0x07030602u);
        const uint32_t t2 = __builtin_amdgcn_perm(r[3], r[2], 0x05010400u), t3 = __builtin_amdgcn_perm(r[3], r[2], 0x07030602u);
        st_stream(w, __builtin_amdgcn_perm(t2, t0, 0x05040100u));
        st_stream(w + plane, __builtin_amdgcn_perm(t2, t0, 0x07060302u));
        st_stream(w + 2 * plane, __builtin_amdgcn_perm(t3, t1, 0x05040100u));
        st_stream(w + 3 * plane, __builtin_amdgcn_perm(t3, t1, 0x07060302u));
        w += row;
    }
    __device__ __forceinline__ void put(const uint32_t (&a_cur)[4], const uint32_t (&)[4], const uint32_t (&o)[4],
                                        const uint32_t (&)[4], const uint32_t (&rc)[4], const uint32_t (&d)[4])
    {
        st_stream(w, a_cur[0] | (a_cur[1] << 8) | (a_cur[2] << 16) | (a_cur[3] << 24));
        st_stream(w + plane, o[0] | (o[1] << 8) | (o[2] << 16) | (o[3] << 24));
        st_stream(w + 2 * plane, (rc[0] & 0xFFu) | ((rc[1] & 0xFFu) << 8) | ((rc[2] & 0xFFu) << 16) | (rc[3] << 24));
        st_stream(w + 3 * plane, d[0] | (d[1] << 8) | (d[2] << 16) | (d[3] << 24));
        w += row;
    }
    __device__ __forceinline__ void finish(int) {}
};

// The workgroup's table of the rewards its env's codes stand for, as the float64 values the reference's callers add up
// (Env::code_reward).  Filled by the sinks' constructors — every fused loop constructs its sink before the barrier that
// follows its table staging, and reads it only inside the step loop.
template <class Env>
static __device__ __forceinline__ double (&reward_f64_lds())[256]
{
    __shared__ double t[256];
    return t;
}

typedef double f64x2 __attribute__((ext_vector_type(2)));
template <class Env, class = void> struct never_done : std::false_type {};
template <class Env> struct never_done<Env, std::enable_if_t<Env::NEVER_DONE>> : std::true_type {};

// The per-lane reduction of the Returns sinks, r += discount * rw; discount *= _discount (rock.py:569-570, network.py:186-187)
// — separate multiply and add — with the episode's return banked when a step ends it.  The loops this rides in are bound
// by VALU issue, so the banking avoids 64-bit selects (v_cndmask_b32 pairs at 4 cycles each): `m` is all ones on a done
// step and zero otherwise — one v_bfe_i32, opaque to the compiler, which would turn the masking back into selects — and
//     ret_sum += total & m        (+0.0 on a step that ends nothing: no change)
//     ret      = total & ~m       (+0.0 = 0x0...0: the fresh episode's return)
//     disc     = disc * discount & ~m | 1.0 & m
//     episodes -= m
//     ret_done = total & m | ret_done & ~m
// are plain 32-bit ANDs at 2 cycles (two v_bfi_b32 for the last).  (An exec-masked store of ret_done where an episode ends,
// instead of two more registers per lane, measured slower: 8-byte pieces of partial lines — Tiger, a third of whose steps
// end an episode, 2.24 -> 3.73 us per step.)
static __device__ __forceinline__ uint32_t mask_of_bit(uint32_t word, int bit)
{
    uint32_t m;
    if (bit == 24) asm("v_bfe_i32 %0, %1, 24, 1" : "=v"(m) : "v"(word));
    else asm("v_bfe_i32 %0, %1, 0, 1" : "=v"(m) : "v"(word));
    return m;
}
static __device__ __forceinline__ double f64_and(double v, uint32_t m)
{
    uint64_t b;
    __builtin_memcpy(&b, &v, 8);
    b = ((uint64_t)((uint32_t)(b >> 32) & m) << 32) | ((uint32_t)b & m);
    __builtin_memcpy(&v, &b, 8);
    return v;
}
template <class Env>
static __device__ __forceinline__ void returns_step(double &ret, double &disc, double &ret_sum, uint32_t &episodes, double &ret_done,
                                                    double discount, uint32_t rcode, uint32_t m)
{
#pragma clang fp contract(off)
    const double term = disc * reward_f64_lds<Env>()[rcode & 0xFFu];
    const double total = ret + term;
    const double next = disc * discount;
    if constexpr (never_done<Env>::value) { ret = total; disc = next; }          // network.py:113: nothing is ever banked
    else {
        uint64_t tb, db;
        __builtin_memcpy(&tb, &total, 8);
        __builtin_memcpy(&db, &ret_done, 8);
        db = ((uint64_t)(((uint32_t)(tb >> 32) & m) | ((uint32_t)(db >> 32) & ~m)) << 32) | (((uint32_t)tb & m) | ((uint32_t)db & ~m));
        __builtin_memcpy(&ret_done, &db, 8);
        ret_sum = ret_sum + f64_and(total, m);
        ret = f64_and(total, ~m);
        uint64_t nb;
        __builtin_memcpy(&nb, &next, 8);
        nb = ((uint64_t)(((uint32_t)(nb >> 32) & ~m) | (0x3FF00000u & m)) << 32) | ((uint32_t)nb & ~m);   // 1.0 = 0x3FF00000:00000000
        __builtin_memcpy(&disc, &nb, 8);
        episodes -= m;
    }
}

template <class Env> struct QuadOut<Returns<Env>> {
    static constexpr bool BANK = !never_done<Env>::value;
    double *acc_w;
    uint32_t *cnt_w;
    int64_t pitch;
    double discount;
    double ret[4], disc[4], ret_done[4], ret_sum[4];
    uint32_t episodes[4];
    // acc: double [4][pitch], cnt: int32 [2][pitch] (include/pomdp_hip.h: pomdp_return_stats); the discount rides in the slot
    // of the reward pointer as its bit pattern (the kernels' signatures are the trajectory sinks')
    __device__ __forceinline__ QuadOut(void *acc, void *cnt, void *discount_bits, void *, int64_t pitch_, uint32_t l0)
        : acc_w(reinterpret_cast<double *>(acc) + l0), cnt_w(reinterpret_cast<uint32_t *>(cnt) + l0), pitch(pitch_)
    {
        const uint64_t bits = reinterpret_cast<uint64_t>(discount_bits);
        __builtin_memcpy(&discount, &bits, 8);
        reward_f64_lds<Env>()[threadIdx.x & 255u] = Env::code_reward(threadIdx.x & 255u);
        auto row = [&](int q, double (&v)[4]) {
            const f64x2 lo = __builtin_nontemporal_load(reinterpret_cast<const f64x2 *>(acc_w + q * pitch));
            const f64x2 hi = __builtin_nontemporal_load(reinterpret_cast<const f64x2 *>(acc_w + q * pitch) + 1);
            v[0] = lo[0]; v[1] = lo[1]; v[2] = hi[0]; v[3] = hi[1];
        };
        row(0, ret); row(1, disc);
        if constexpr (BANK) {
            row(2, ret_done); row(3, ret_sum);
            const u32x4 e = ld_stream4(cnt_w);
            episodes[0] = e[0]; episodes[1] = e[1]; episodes[2] = e[2]; episodes[3] = e[3];
        }
    }
    template <class Pol> __device__ __forceinline__ u32x4 first(const Pol &pol, int) { return pol.first(); }
    __device__ __forceinline__ void put_records(const uint32_t (&r)[4], const uint32_t (&)[4])
    {
#pragma unroll
        for (int j = 0; j < 4; ++j)                           // the done byte of a record is 0 or 1
            returns_step<Env>(ret[j], disc[j], ret_sum[j], episodes[j], ret_done[j], discount, r[j] >> 16, mask_of_bit(r[j], 24));
    }
    __device__ __forceinline__ void put(const uint32_t (&)[4], const uint32_t (&)[4], const uint32_t (&)[4],
                                        const uint32_t (&)[4], const uint32_t (&rc)[4], const uint32_t (&d)[4])
    {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            returns_step<Env>(ret[j], disc[j], ret_sum[j], episodes[j], ret_done[j], discount, rc[j], BANK ? mask_of_bit(d[j], 0) : 0u);
    }
    __device__ __forceinline__ void finish(int k_steps)
    {
        auto row = [&](int q, const double (&v)[4]) {
            __builtin_nontemporal_store(f64x2{v[0], v[1]}, reinterpret_cast<f64x2 *>(acc_w + q * pitch));
            __builtin_nontemporal_store(f64x2{v[2], v[3]}, reinterpret_cast<f64x2 *>(acc_w + q * pitch) + 1);
        };
        row(0, ret); row(1, disc);
        if constexpr (BANK) {
            row(2, ret_done); row(3, ret_sum);
            st_stream4(cnt_w, episodes[0], episodes[1], episodes[2], episodes[3]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)                          // steps += k: no value comes back, nothing to wait for
            (void)__hip_atomic_fetch_add(cnt_w + pitch + j, (uint32_t)k_steps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
};

// ---- a thread that owns TWO consecutive lanes: the 4-byte sinks and the returns sink (the 13-byte layouts are bound by their
// store stream whatever the geometry: they keep the quad-per-thread loops).  put(): as QuadOut's, two lanes wide.
template <class L> struct PairOut;
template <> struct PairOut<Packed> {
    uint32_t *w;
    int64_t rec;
    __device__ __forceinline__ PairOut(void *base, void *, void *, void *, int64_t rec_, uint32_t l0) : w(reinterpret_cast<uint32_t *>(base) + l0), rec(rec_) {}
    __device__ __forceinline__ void put(const uint32_t (&a_cur)[2], const uint32_t (&)[2], const uint32_t (&o)[2], const uint32_t (&)[2],
                                        const uint32_t (&rc)[2], const uint32_t (&d)[2])
    {
        st_stream2(w, pack_record(a_cur[0], o[0], rc[0], d[0]), pack_record(a_cur[1], o[1], rc[1], d[1]));
        w += rec;
    }
    __device__ __forceinline__ void put_records(const uint32_t (&r)[2], const uint32_t (&)[2])
    {
        st_stream2(w, r[0], r[1]);
        w += rec;
    }
    __device__ __forceinline__ void finish(int) {}
};
template <> struct PairOut<Narrow> {
    uint16_t *w;                                            // the thread's two bytes of the row's action plane
    int64_t plane, row;                                     // in 16-bit units
    __device__ __forceinline__ PairOut(void *base, void *, void *, void *, int64_t rec_, uint32_t l0)
        : w(reinterpret_cast<uint16_t *>(reinterpret_cast<uint8_t *>(base) + l0)), plane(rec_ / 2), row(2 * rec_) {}
    __device__ __forceinline__ void put(const uint32_t (&a_cur)[2], const uint32_t (&)[2], const uint32_t (&o)[2], const uint32_t (&)[2],
                                        const uint32_t (&rc)[2], const uint32_t (&d)[2])
    {
        st_stream(w, (uint16_t)(a_cur[0] | (a_cur[1] << 8)));
        st_stream(w + plane, (uint16_t)(o[0] | (o[1] << 8)));
        st_stream(w + 2 * plane, (uint16_t)((rc[0] & 0xFFu) | ((rc[1] & 0xFFu) << 8)));
        st_stream(w + 3 * plane, (uint16_t)(d[0] | (d[1] << 8)));
        w += row;
    }
    __device__ __forceinline__ void put_records(const uint32_t (&r)[2], const uint32_t (&)[2])
    {
        // byte b of record j -> byte j of plane b
        st_stream(w, (uint16_t)__builtin_amdgcn_perm(r[1], r[0], 0x0C0C0400u));
        st_stream(w + plane, (uint16_t)__builtin_amdgcn_perm(r[1], r[0], 0x0C0C0501u));
        st_stream(w + 2 * plane, (uint16_t)__builtin_amdgcn_perm(r[1], r[0], 0x0C0C0602u));
        st_stream(w + 3 * plane, (uint16_t)__builtin_amdgcn_perm(r[1], r[0], 0x0C0C0703u));
        w += row;
    }
    __device__ __forceinline__ void finish(int) {}
};
template <class Env> struct PairOut<Returns<Env>> {
    static constexpr bool BANK = !never_done<Env>::value;
    double *acc_w;
    uint32_t *cnt_w;
    int64_t pitch;
    double discount;
    double ret[2], disc[2], ret_done[2], ret_sum[2];
    uint32_t episodes[2];
    __device__ __forceinline__ PairOut(void *acc, void *cnt, void *discount_bits, void *, int64_t pitch_, uint32_t l0)
        : acc_w(reinterpret_cast<double *>(acc) + l0), cnt_w(reinterpret_cast<uint32_t *>(cnt) + l0), pitch(pitch_)
    {
        const uint64_t bits = reinterpret_cast<uint64_t>(discount_bits);
        __builtin_memcpy(&discount, &bits, 8);
        reward_f64_lds<Env>()[threadIdx.x & 255u] = Env::code_reward(threadIdx.x & 255u);
        auto row = [&](int q, double (&v)[2]) {
            const f64x2 x = __builtin_nontemporal_load(reinterpret_cast<const f64x2 *>(acc_w + q * pitch));
            v[0] = x[0]; v[1] = x[1];
        };
        row(0, ret); row(1, disc);
        if constexpr (BANK) {
            row(2, ret_done); row(3, ret_sum);
            const u32x2 e = ld_stream2(cnt_w);
            episodes[0] = e[0]; episodes[1] = e[1];
        }
    }
    __device__ __forceinline__ void put(const uint32_t (&)[2], const uint32_t (&)[2], const uint32_t (&)[2], const uint32_t (&)[2],
                                        const uint32_t (&rc)[2], const uint32_t (&d)[2])
    {
#pragma unroll
        for (int j = 0; j < 2; ++j)
            returns_step<Env>(ret[j], disc[j], ret_sum[j], episodes[j], ret_done[j], discount, rc[j], BANK ? mask_of_bit(d[j], 0) : 0u);
    }
    __device__ __forceinline__ void put_records(const uint32_t (&r)[2], const uint32_t (&)[2])
    {
#pragma unroll
        for (int j = 0; j < 2; ++j)                           // the done byte of a record is 0 or 1
            returns_step<Env>(ret[j], disc[j], ret_sum[j], episodes[j], ret_done[j], discount, r[j] >> 16, mask_of_bit(r[j], 24));
    }
    __device__ __forceinline__ void finish(int k_steps)
    {
        auto row = [&](int q, const double (&v)[2]) { __builtin_nontemporal_store(f64x2{v[0], v[1]}, reinterpret_cast<f64x2 *>(acc_w + q * pitch)); };
        row(0, ret); row(1, disc);
        if constexpr (BANK) {
            row(2, ret_done); row(3, ret_sum);
            st_stream2(cnt_w, episodes[0], episodes[1]);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
            (void)__hip_atomic_fetch_add(cnt_w + pitch + j, (uint32_t)k_steps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
};
template <class L> struct pair_sink : std::false_type {};               // the sinks PairOut exists for
template <> struct pair_sink<Packed> : std::true_type {};
template <> struct pair_sink<Narrow> : std::true_type {};
template <class Env> struct pair_sink<Returns<Env>> : std::true_type {};
template <class L, int LPT> struct lanes_out { using type = QuadOut<L>; };
template <class L> struct lanes_out<L, 2> { using type = PairOut<L>; };

// ---- a thread whose lanes are 256 apart (steps_kernel: lane j of a thread is base + tid + 256 j, j < LPT) ----------------
// wg0: the workgroup's first lane within the shard (a multiple of 256); rel = tid + 256 j.  begin(j, rel): before the loop,
// with the lane's (clamped) index; put(j, rel, ...): one step's results of an in-range lane; finish(j, rel, k): after the loop.
template <class L, class RT, int LPT> struct LaneOut;

template <bool ACT, class RT, int LPT> struct LaneOut<ColumnsT<ACT>, RT, LPT> {
    int32_t *action_w, *ob_w;
    RT *reward_w;
    uint8_t *done_w;
    int64_t rec;
    __device__ __forceinline__ LaneOut(void *action, void *ob, void *reward, void *done, int64_t rec_, uint32_t wg0)
        : action_w(reinterpret_cast<int32_t *>(action) + wg0), ob_w(reinterpret_cast<int32_t *>(ob) + wg0),
          reward_w(reinterpret_cast<RT *>(reward) + wg0), done_w(reinterpret_cast<uint8_t *>(done) + wg0), rec(rec_) {}
    __device__ __forceinline__ void begin(int, uint32_t) {}
    __device__ __forceinline__ int load_first(uint32_t rel) const { return ACT ? ld_stream(action_w + rel) : 0; }
    __device__ __forceinline__ void store_first(uint32_t rel, int a) const { if constexpr (ACT) st_stream(action_w + rel, (int32_t)a); }
    __device__ __forceinline__ void first_done() { action_w += rec; }                  // row 0 of `action` is behind us
    __device__ __forceinline__ void put_next_action(uint32_t rel, int a_next) const { if constexpr (ACT) st_stream(action_w + rel, (int32_t)a_next); }
    __device__ __forceinline__ void put(int, uint32_t rel, int, int o, RT r, uint32_t, int d)
    {
        st_stream(ob_w + rel, (int32_t)o);
        st_stream(reward_w + rel, r);
        st_stream(done_w + rel, (uint8_t)d);
    }
    __device__ __forceinline__ void next_row() { action_w += rec; ob_w += rec; reward_w += rec; done_w += rec; }
    __device__ __forceinline__ void finish(int, uint32_t, int) {}
};

template <class RT, int LPT> struct LaneOut<Blocked, RT, LPT> {
    uint8_t *w;                                             // block of the workgroup's sub-batch 0, action section
    int64_t row_bytes;
    __device__ __forceinline__ LaneOut(void *base, void *, void *, void *, int64_t rec_, uint32_t wg0)
        : w(reinterpret_cast<uint8_t *>(base) + (int64_t)(wg0 >> 8) * TRAJ_BLOCK_BYTES), row_bytes(rec_ * 13) {}
    __device__ __forceinline__ void begin(int, uint32_t) {}
    __device__ __forceinline__ int load_first(uint32_t) const { return 0; }
    __device__ __forceinline__ void store_first(uint32_t, int) const {}
    __device__ __forceinline__ void first_done() {}
    __device__ __forceinline__ void put_next_action(uint32_t, int) const {}
    __device__ __forceinline__ void put(int, uint32_t rel, int a_cur, int o, RT r, uint32_t, int d)
    {
        uint8_t *b = w + (rel >> 8) * (uint32_t)TRAJ_BLOCK_BYTES;                      // sub-batch j's block
        const uint32_t i = rel & 255u;
        st_stream(reinterpret_cast<int32_t *>(b) + i, (int32_t)a_cur);
        st_stream(reinterpret_cast<int32_t *>(b + 1024) + i, (int32_t)o);
        st_stream(reinterpret_cast<RT *>(b + 2048) + i, r);
        st_stream(b + 3072 + i, (uint8_t)d);
    }
    __device__ __forceinline__ void next_row() { w += row_bytes; }
    __device__ __forceinline__ void finish(int, uint32_t, int) {}
};

template <class RT, int LPT> struct LaneOut<Packed, RT, LPT> {
    uint32_t *w;
    int64_t rec;
    __device__ __forceinline__ LaneOut(void *base, void *, void *, void *, int64_t rec_, uint32_t wg0)
        : w(reinterpret_cast<uint32_t *>(base) + wg0), rec(rec_) {}
    __device__ __forceinline__ void begin(int, uint32_t) {}
    __device__ __forceinline__ int load_first(uint32_t) const { return 0; }
    __device__ __forceinline__ void store_first(uint32_t, int) const {}
    __device__ __forceinline__ void first_done() {}
    __device__ __forceinline__ void put_next_action(uint32_t, int) const {}
    __device__ __forceinline__ void put(int, uint32_t rel, int a_cur, int o, RT, uint32_t rcode, int d)
    {
        st_stream(w + rel, pack_record((uint32_t)a_cur & 0xFFu, (uint32_t)o & 0xFFu, rcode, (uint32_t)(d != 0)));
    }
    __device__ __forceinline__ void put_record(int, uint32_t rel, uint32_t record) { st_stream(w + rel, record); }
    __device__ __forceinline__ void next_row() { w += rec; }
    __device__ __forceinline__ void finish(int, uint32_t, int) {}
};

template <class RT, int LPT> struct LaneOut<Narrow, RT, LPT> {
    uint8_t *w;                                             // the workgroup's first byte of the row's action plane
    int64_t pitch;
    __device__ __forceinline__ LaneOut(void *base, void *, void *, void *, int64_t rec_, uint32_t wg0)
        : w(reinterpret_cast<uint8_t *>(base) + wg0), pitch(rec_) {}
    __device__ __forceinline__ void begin(int, uint32_t) {}
    __device__ __forceinline__ int load_first(uint32_t) const { return 0; }
    __device__ __forceinline__ void store_first(uint32_t, int) const {}
    __device__ __forceinline__ void first_done() {}
    __device__ __forceinline__ void put_next_action(uint32_t, int) const {}
    __device__ __forceinline__ void put(int, uint32_t rel, int a_cur, int o, RT, uint32_t rcode, int d)
    {
        st_stream(w + rel, (uint8_t)a_cur);
        st_stream(w + pitch + rel, (uint8_t)o);
        st_stream(w + 2 * pitch + rel, (uint8_t)rcode);
        st_stream(w + 3 * pitch + rel, (uint8_t)(d != 0));
    }
    __device__ __forceinline__ void put_record(int j, uint32_t rel, uint32_t record)
    {
        put(j, rel, (int)(record & 0xFFu), (int)((record >> 8) & 0xFFu), RT(0), record >> 16, (int)(record >> 24));
    }
    __device__ __forceinline__ void next_row() { w += 4 * pitch; }
    __device__ __forceinline__ void finish(int, uint32_t, int) {}
};

template <class Env, class RT, int LPT> struct LaneOut<Returns<Env>, RT, LPT> {
    static constexpr bool BANK = !never_done<Env>::value;
    double *acc_w;
    uint32_t *cnt_w;
    int64_t pitch;
    double discount;
    double ret[LPT], disc[LPT], ret_done[LPT], ret_sum[LPT];
    uint32_t episodes[LPT];
    __device__ __forceinline__ LaneOut(void *acc, void *cnt, void *discount_bits, void *, int64_t pitch_, uint32_t wg0)
        : acc_w(reinterpret_cast<double *>(acc) + wg0), cnt_w(reinterpret_cast<uint32_t *>(cnt) + wg0), pitch(pitch_)
    {
        const uint64_t bits = reinterpret_cast<uint64_t>(discount_bits);
        __builtin_memcpy(&discount, &bits, 8);
        reward_f64_lds<Env>()[threadIdx.x & 255u] = Env::code_reward(threadIdx.x & 255u);
    }
    __device__ __forceinline__ void begin(int j, uint32_t rel)
    {
        ret[j] = ld_stream(acc_w + rel); disc[j] = ld_stream(acc_w + pitch + rel);
        if constexpr (BANK) {
            ret_done[j] = ld_stream(acc_w + 2 * pitch + rel); ret_sum[j] = ld_stream(acc_w + 3 * pitch + rel);
            episodes[j] = ld_stream(cnt_w + rel);
        }
    }
    __device__ __forceinline__ int load_first(uint32_t) const { return 0; }
    __device__ __forceinline__ void store_first(uint32_t, int) const {}
    __device__ __forceinline__ void first_done() {}
    __device__ __forceinline__ void put_next_action(uint32_t, int) const {}
    __device__ __forceinline__ void put(int j, uint32_t rel, int, int, RT, uint32_t rcode, int d)
    {
        returns_step<Env>(ret[j], disc[j], ret_sum[j], episodes[j], ret_done[j], discount, rcode, BANK ? mask_of_bit((uint32_t)(d != 0), 0) : 0u);
    }
    __device__ __forceinline__ void put_record(int j, uint32_t rel, uint32_t record)
    {
        returns_step<Env>(ret[j], disc[j], ret_sum[j], episodes[j], ret_done[j], discount, record >> 16, mask_of_bit(record, 24));
    }
    __device__ __forceinline__ void next_row() {}
    __device__ __forceinline__ void finish(int j, uint32_t rel, int k_steps)
    {
        st_stream(acc_w + rel, ret[j]); st_stream(acc_w + pitch + rel, disc[j]);
        if constexpr (BANK) {
            st_stream(acc_w + 2 * pitch + rel, ret_done[j]); st_stream(acc_w + 3 * pitch + rel, ret_sum[j]);
            st_stream(cnt_w + rel, episodes[j]);
        }
        (void)__hip_atomic_fetch_add(cnt_w + pitch + rel, (uint32_t)k_steps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
};

} // namespace pomdp
