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
// Blocked and Packed rows hold the action TAKEN at step s (row s of every field belongs to step s); they have no row of
// "next actions" — a launch derives its first actions from the synthetic policy itself (they are a function of (seed,
// lane, t) only), which is what the Columns layout's gen_first launches do as well.
#pragma once
#include "envs.hip.h"

namespace pomdp {

struct Columns { static constexpr int ID = POMDP_LAYOUT_COLUMNS; static constexpr const char *NAME = "Columns"; };
struct Blocked { static constexpr int ID = POMDP_LAYOUT_BLOCKED; static constexpr const char *NAME = "Blocked"; };
struct Packed  { static constexpr int ID = POMDP_LAYOUT_PACKED;  static constexpr const char *NAME = "Packed"; };

constexpr int TRAJ_BLOCK_LANES = 256;                       // lanes per block of the Blocked layout = one wave's four per thread
constexpr int TRAJ_BLOCK_BYTES = 13 * TRAJ_BLOCK_LANES;     // 3 x 1024 bytes of int32 / float + 256 done bytes

static __device__ __forceinline__ uint32_t pack_record(uint32_t a, uint32_t o, uint32_t rcode, uint32_t d)
{
    // a, o < 256 and d in {0, 1} by construction; the reward code may carry an int8's sign bits
    return a | (o << 8) | ((rcode & 0xFFu) << 16) | (d << 24);
}

// the four policy words of a quad at the call counter BEFORE akey0's, as actions (what pomdp_synthetic_actions would write)
static __device__ __forceinline__ u32x4 gen_actions4(uint32_t glane0, const RngKey &akey0, uint32_t n_act)
{
    const uint64_t tf = (((uint64_t)akey0.t_hi << 32) | akey0.t_lo) - 1ull;
    const uint4 w = philox4x32_10(glane0 >> 2, (uint32_t)tf, (uint32_t)(tf >> 32), (uint32_t)POMDP_STREAM_ACTION << 24, akey0.k0, akey0.k1);
    return u32x4{__umulhi(w.x, n_act), __umulhi(w.y, n_act), __umulhi(w.z, n_act), __umulhi(w.w, n_act)};
}

// ---- a thread that owns a quad of consecutive lanes (the quad-per-thread loops) ------------------------------------
// l0: the thread's first lane within the shard (a multiple of 4).  first(): the actions of the launch's first step.
// put(): one step's results of the four lanes — a_cur the actions taken, a_next the policy's actions of the next call
// counter, o / r (raw 32-bit patterns) / rc (reward codes, read by Packed only) / d (0 or 1) — then on to the next row.
template <class L> struct QuadOut;

template <> struct QuadOut<Columns> {
    uint32_t *action_w, *ob_w, *reward_w, *done_w;
    int64_t rec;
    __device__ __forceinline__ QuadOut(void *action, void *ob, void *reward, void *done, int64_t rec_, uint32_t l0)
        : action_w(reinterpret_cast<uint32_t *>(action) + l0), ob_w(reinterpret_cast<uint32_t *>(ob) + l0),
          reward_w(reinterpret_cast<uint32_t *>(reward) + l0), done_w(reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(done) + l0)),
          rec(rec_) {}
    // read from row 0 of `action`, or (gen_first, wave-uniform) the quad's block of the synthetic policy, written to that row
    __device__ __forceinline__ u32x4 first(int gen_first, uint32_t glane0, const RngKey &akey0, uint32_t n_act)
    {
        u32x4 a;
        if (!gen_first) a = ld_stream4(action_w);
        else {
            a = gen_actions4(glane0, akey0, n_act);
            st_stream4(action_w, a[0], a[1], a[2], a[3]);
        }
        action_w += rec;
        return a;
    }
    __device__ __forceinline__ void put(const uint32_t (&)[4], const uint32_t (&a_next)[4], const uint32_t (&o)[4],
                                        const uint32_t (&r)[4], const uint32_t (&)[4], const uint32_t (&d)[4])
    {
        st_stream4(action_w, a_next[0], a_next[1], a_next[2], a_next[3]);
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
};

template <> struct QuadOut<Blocked> {
    uint8_t *w, *wd;                                        // the thread's 16 bytes of the block's action section; its 4 done bytes
    int64_t row_bytes;
    __device__ __forceinline__ QuadOut(void *base, void *, void *, void *, int64_t rec_, uint32_t l0)
        : w(reinterpret_cast<uint8_t *>(base) + (int64_t)(l0 >> 8) * TRAJ_BLOCK_BYTES + (l0 & 255u) * 4u),
          wd(reinterpret_cast<uint8_t *>(base) + (int64_t)(l0 >> 8) * TRAJ_BLOCK_BYTES + 3 * 1024 + (l0 & 255u)),
          row_bytes(rec_ * 13) {}
    __device__ __forceinline__ u32x4 first(int, uint32_t glane0, const RngKey &akey0, uint32_t n_act) { return gen_actions4(glane0, akey0, n_act); }
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
};

template <> struct QuadOut<Packed> {
    uint32_t *w;
    int64_t rec;
    __device__ __forceinline__ QuadOut(void *base, void *, void *, void *, int64_t rec_, uint32_t l0)
        : w(reinterpret_cast<uint32_t *>(base) + l0), rec(rec_) {}
    __device__ __forceinline__ u32x4 first(int, uint32_t glane0, const RngKey &akey0, uint32_t n_act) { return gen_actions4(glane0, akey0, n_act); }
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
};

// ---- a thread whose lanes are 256 apart (steps_kernel: lane j of a thread is base + tid + 256 j) ------------------------
// wg0: the workgroup's first lane within the shard (a multiple of 256); rel = tid + 256 j.
template <class L, class RT> struct LaneOut;

template <class RT> struct LaneOut<Columns, RT> {
    int32_t *action_w, *ob_w;
    RT *reward_w;
    uint8_t *done_w;
    int64_t rec;
    static constexpr bool HAS_ACTION_ROWS = true;
    __device__ __forceinline__ LaneOut(void *action, void *ob, void *reward, void *done, int64_t rec_, uint32_t wg0)
        : action_w(reinterpret_cast<int32_t *>(action) + wg0), ob_w(reinterpret_cast<int32_t *>(ob) + wg0),
          reward_w(reinterpret_cast<RT *>(reward) + wg0), done_w(reinterpret_cast<uint8_t *>(done) + wg0), rec(rec_) {}
    __device__ __forceinline__ int load_first(uint32_t rel) const { return ld_stream(action_w + rel); }
    __device__ __forceinline__ void store_first(uint32_t rel, int a) const { st_stream(action_w + rel, (int32_t)a); }
    __device__ __forceinline__ void first_done() { action_w += rec; }                  // row 0 of `action` is behind us
    __device__ __forceinline__ void put_next_action(uint32_t rel, int a_next) const { st_stream(action_w + rel, (int32_t)a_next); }
    __device__ __forceinline__ void put(uint32_t rel, int, int o, RT r, uint32_t, int d) const
    {
        st_stream(ob_w + rel, (int32_t)o);
        st_stream(reward_w + rel, r);
        st_stream(done_w + rel, (uint8_t)d);
    }
    __device__ __forceinline__ void next_row() { action_w += rec; ob_w += rec; reward_w += rec; done_w += rec; }
};

template <class RT> struct LaneOut<Blocked, RT> {
    uint8_t *w;                                             // block of the workgroup's sub-batch 0, action section
    int64_t row_bytes;
    static constexpr bool HAS_ACTION_ROWS = false;
    __device__ __forceinline__ LaneOut(void *base, void *, void *, void *, int64_t rec_, uint32_t wg0)
        : w(reinterpret_cast<uint8_t *>(base) + (int64_t)(wg0 >> 8) * TRAJ_BLOCK_BYTES), row_bytes(rec_ * 13) {}
    __device__ __forceinline__ int load_first(uint32_t) const { return 0; }
    __device__ __forceinline__ void store_first(uint32_t, int) const {}
    __device__ __forceinline__ void first_done() {}
    __device__ __forceinline__ void put_next_action(uint32_t, int) const {}
    __device__ __forceinline__ void put(uint32_t rel, int a_cur, int o, RT r, uint32_t, int d) const
    {
        uint8_t *b = w + (rel >> 8) * (uint32_t)TRAJ_BLOCK_BYTES;                      // sub-batch j's block
        const uint32_t i = rel & 255u;
        st_stream(reinterpret_cast<int32_t *>(b) + i, (int32_t)a_cur);
        st_stream(reinterpret_cast<int32_t *>(b + 1024) + i, (int32_t)o);
        st_stream(reinterpret_cast<RT *>(b + 2048) + i, r);
        st_stream(b + 3072 + i, (uint8_t)d);
    }
    __device__ __forceinline__ void next_row() { w += row_bytes; }
};

template <class RT> struct LaneOut<Packed, RT> {
    uint32_t *w;
    int64_t rec;
    static constexpr bool HAS_ACTION_ROWS = false;
    __device__ __forceinline__ LaneOut(void *base, void *, void *, void *, int64_t rec_, uint32_t wg0)
        : w(reinterpret_cast<uint32_t *>(base) + wg0), rec(rec_) {}
    __device__ __forceinline__ int load_first(uint32_t) const { return 0; }
    __device__ __forceinline__ void store_first(uint32_t, int) const {}
    __device__ __forceinline__ void first_done() {}
    __device__ __forceinline__ void put_next_action(uint32_t, int) const {}
    __device__ __forceinline__ void put(uint32_t rel, int a_cur, int o, RT, uint32_t rcode, int d) const
    {
        st_stream(w + rel, pack_record((uint32_t)a_cur & 0xFFu, (uint32_t)o & 0xFFu, rcode, (uint32_t)(d != 0)));
    }
    __device__ __forceinline__ void put_record(uint32_t rel, uint32_t record) const { st_stream(w + rel, record); }
    __device__ __forceinline__ void next_row() { w += rec; }
};

} // namespace pomdp
