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

#include "../../include/pomdp_hip.h"
#include "envs.hip.h"
#include "philox.hip.h"

namespace pomdp {

constexpr int BLOCK = 256;        // 4 waves: one per SIMD
constexpr int MAX_BLOCKS = 256 * 8; // 256 CUs x 8 resident workgroups, grid-stride beyond that

static inline int grid_for(int64_t n)
{
    const int64_t b = (n + BLOCK - 1) / BLOCK;
    return (int)(b < 1 ? 1 : (b > MAX_BLOCKS ? MAX_BLOCKS : b));
}

// ---------------------------------------------------------------------------
// reset: every lane starts a fresh episode from stream RESET of (seed, lane, t)
// ---------------------------------------------------------------------------
template <class Env>
__global__ __launch_bounds__(BLOCK) void reset_kernel(const typename Env::Params p, uint32_t *__restrict__ state,
                                                      int32_t *__restrict__ ob, int64_t n, RngKey key, uint32_t lane0)
{
    __shared__ typename Env::Shared sh;
    Env::stage(sh, p);
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
        typename Env::State st;
        const int o = Env::reset(sh, p, st, key, lane0 + (uint32_t)i);
        Env::store(st, state, n, i, true);
        if (ob) ob[i] = o;
    }
}

// ---------------------------------------------------------------------------
// step: transition + observation + reward (+ same-call auto-reset of done lanes)
// ---------------------------------------------------------------------------
template <class Env>
__global__ __launch_bounds__(BLOCK) void step_kernel(const typename Env::Params p, uint32_t *__restrict__ state,
                                                     const int32_t *__restrict__ action, int32_t *__restrict__ ob,
                                                     typename Env::Reward *__restrict__ reward,
                                                     uint8_t *__restrict__ done, uint32_t *__restrict__ err,
                                                     int64_t n, RngKey key, uint32_t lane0, int flags)
{
    __shared__ typename Env::Shared sh;
    Env::stage(sh, p);
    __syncthreads();
    const bool auto_reset = flags & POMDP_AUTO_RESET;
    const int n_act = Env::n_actions(p);
    const int64_t stride = (int64_t)gridDim.x * BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
        const int a = action[i];
        int o = 0, d = 0;
        typename Env::Reward r = 0;
        if (!auto_reset && done[i]) {
            d = 1;                                   // frozen lane (the reference would assert)
        } else if ((unsigned)a >= (unsigned)n_act) {
            if (err) atomicAdd(err, 1u);             // the reference asserts; the lane is left untouched
        } else {
            const uint32_t lane = lane0 + (uint32_t)i;
            typename Env::State st;
            Env::load(st, state, n, i);
            Env::step(sh, p, st, a, key, lane, o, r, d);
            const bool fresh = d && auto_reset;
            if (fresh) Env::reset(sh, p, st, key, lane);
            Env::store(st, state, n, i, fresh);
        }
        ob[i] = o;
        reward[i] = r;
        done[i] = (uint8_t)d;
    }
}

// ---------------------------------------------------------------------------
// helpers
// ---------------------------------------------------------------------------
// one thread = four consecutive lanes = one Philox block = one 16-byte store
__global__ __launch_bounds__(BLOCK) void synthetic_actions_kernel(int4 *__restrict__ action, int64_t n4, RngKey key,
                                                                 uint32_t q0, uint32_t n_actions)
{
    const int64_t stride = (int64_t)gridDim.x * BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n4; i += stride) {
        const uint4 w = philox4x32_10(q0 + (uint32_t)i, key.t_lo, key.t_hi, (uint32_t)POMDP_STREAM_ACTION << 24,
                                      key.k0, key.k1);
        action[i] = make_int4((int)__umulhi(w.x, n_actions), (int)__umulhi(w.y, n_actions),
                              (int)__umulhi(w.z, n_actions), (int)__umulhi(w.w, n_actions));
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

template <class Env>
static int launch_reset(const typename Env::Params &p, uint32_t *state, int32_t *ob, int64_t n, uint64_t seed,
                        uint32_t lane0, uint64_t t, void *stream)
{
    if (!state || bad_range(n, lane0)) return POMDP_E_BADARG;
    if (n == 0) return 0;
    hipLaunchKernelGGL(reset_kernel<Env>, dim3(grid_for(n)), dim3(BLOCK), 0, (hipStream_t)stream, p, state, ob, n,
                       make_key(seed, t), lane0);
    return (int)hipGetLastError();
}

template <class Env>
static int launch_step(const typename Env::Params &p, uint32_t *state, const int32_t *action, int32_t *ob,
                       typename Env::Reward *reward, uint8_t *done, uint32_t *err, int64_t n, uint64_t seed,
                       uint32_t lane0, uint64_t t, int flags, void *stream)
{
    if (!state || !action || !ob || !reward || !done || bad_range(n, lane0)) return POMDP_E_BADARG;
    if (n == 0) return 0;
    hipLaunchKernelGGL(step_kernel<Env>, dim3(grid_for(n)), dim3(BLOCK), 0, (hipStream_t)stream, p, state, action, ob,
                       reward, done, err, n, make_key(seed, t), lane0, flags);
    return (int)hipGetLastError();
}

static bool rock_ok(const pomdp_rock_params *p)
{
    return p && p->size >= 1 && p->size <= 15 && p->num_rocks >= 1 && p->num_rocks <= 16 &&
           (unsigned)p->start_x < (unsigned)p->size && (unsigned)p->start_y < (unsigned)p->size;
}
static int bs_mask_words(const pomdp_battleship_params *p)
{
    if (!p || p->x_size < 1 || p->y_size < 1 || p->x_size > 16 || p->y_size > 16) return 0;
    const int cells = p->x_size * p->y_size;
    if (cells > 122 || p->max_len < 2 || p->max_len > 10) return 0;
    return (cells + 6 + 31) / 32;
}

} // namespace pomdp

using namespace pomdp;

extern "C" {

int pomdp_abi_version(void) { return POMDP_ABI_VERSION; }

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
    return p->num_rocks <= 12 ? launch_reset<RockEnv<1>>(*p, state, ob, n, seed, lane0, t, stream)
                              : launch_reset<RockEnv<2>>(*p, state, ob, n, seed, lane0, t, stream);
}

int pomdp_rock_step(const pomdp_rock_params *p, uint32_t *state, const int32_t *action, int32_t *ob, int32_t *reward,
                    uint8_t *done, uint32_t *err, int64_t n, uint64_t seed, uint32_t lane0, uint64_t t, int flags,
                    void *stream)
{
    if (!rock_ok(p)) return POMDP_E_BADPARAMS;
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

int pomdp_synthetic_actions(int32_t *action, int64_t n, uint64_t seed, uint32_t lane0, uint64_t t, uint32_t n_actions,
                            void *stream)
{
    if (!action || bad_range(n, lane0) || (n & 3) || (lane0 & 3u) || n_actions == 0) return POMDP_E_BADARG;
    if (n == 0) return 0;
    hipLaunchKernelGGL(synthetic_actions_kernel, dim3(grid_for(n / 4)), dim3(BLOCK), 0, (hipStream_t)stream,
                       (int4 *)action, n / 4, make_key(seed, t), lane0 >> 2, n_actions);
    return (int)hipGetLastError();
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
