// api.hip — the C ABI of include/pomdp_hip.h: per-env reset / step, bound-argument and scalar-mode entry points, the synthetic policy, the C-side episode loops.
// Part of libpomdp_hip.so; built by gym_pomdp_amd/_native.py (hipcc --offload-arch=gfx950 -O3 -std=c++17 -c, one object per file).
#include "kernels_common.hip.h"
#include <atomic>
#include <cstring>

namespace pomdp {

thread_local uint32_t *tl_host_flag = nullptr;
thread_local uint32_t tl_flag_value = 0;
thread_local char g_last_fused[96] = "";
std::atomic<int> g_fuse_max{POMDP_FUSE_MAX_DEFAULT};     // process-wide, set and read by any host thread (relaxed: a knob, not a fence)

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

// PACKED records -> the default ABI's columns (pomdp_decode_packed).  One thread = four consecutive records of one row:
// a 16-byte load, three 16-byte stores and the four done bytes.  The reward column's bit patterns come from a 256-entry
// table in LDS (int32 of the int8 code; float for Tag; Network's float32(base - cost)) built once per workgroup.
static __device__ __forceinline__ uint32_t decoded_reward_bits(int env, uint32_t code)
{
    if (env == POMDP_ENV_NETWORK) return __float_as_uint((float)NetworkEnv::code_reward(code));
    const int r = (int)(int8_t)code;
    return env == POMDP_ENV_TAG ? __float_as_uint((float)r) : (uint32_t)r;
}
template <bool VEC>
__global__ __launch_bounds__(BLOCK) void decode_packed_kernel(const uint32_t *__restrict__ records, int64_t n, int64_t k_steps,
                                                             int64_t pitch_in, uint32_t *__restrict__ action,
                                                             uint32_t *__restrict__ ob, uint32_t *__restrict__ reward,
                                                             uint8_t *__restrict__ done, int64_t pitch_out, int env)
{
    __shared__ uint32_t rtab[256];
    rtab[threadIdx.x] = decoded_reward_bits(env, threadIdx.x);
    __syncthreads();
    const int64_t per_row = VEC ? (n + 3) >> 2 : n, total = per_row * k_steps, stride = (int64_t)gridDim.x * BLOCK;
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < total; i += stride) {
        const int64_t s = i / per_row, q = i - s * per_row;
        if (VEC && 4 * q + 4 <= n) {
            const int64_t in = s * pitch_in + 4 * q, o = s * pitch_out + 4 * q;
            const u32x4 r = ld_stream4(records + in);
            st_stream4(action + o, r[0] & 0xFFu, r[1] & 0xFFu, r[2] & 0xFFu, r[3] & 0xFFu);
            st_stream4(ob + o, __builtin_amdgcn_ubfe(r[0], 8u, 8u), __builtin_amdgcn_ubfe(r[1], 8u, 8u), __builtin_amdgcn_ubfe(r[2], 8u, 8u),
                       __builtin_amdgcn_ubfe(r[3], 8u, 8u));
            st_stream4(reward + o, rtab[__builtin_amdgcn_ubfe(r[0], 16u, 8u)], rtab[__builtin_amdgcn_ubfe(r[1], 16u, 8u)],
                       rtab[__builtin_amdgcn_ubfe(r[2], 16u, 8u)], rtab[__builtin_amdgcn_ubfe(r[3], 16u, 8u)]);
            // done bytes 0 / 1 of the four records -> four consecutive bytes
            st_stream(reinterpret_cast<uint32_t *>(done + o), (r[0] >> 24) | ((r[1] >> 24) << 8) | ((r[2] >> 24) << 16) | (r[3] & 0xFF000000u));
        } else {
            const int64_t l0 = VEC ? 4 * q : q, l1 = VEC ? n : q + 1;                    // the ragged tail of a row, or one lane
            for (int64_t l = l0; l < l1; ++l) {
                const uint32_t r = records[s * pitch_in + l];
                const int64_t o = s * pitch_out + l;
                action[o] = r & 0xFFu; ob[o] = (r >> 8) & 0xFFu; reward[o] = rtab[(r >> 16) & 0xFFu]; done[o] = (uint8_t)(r >> 24);
            }
        }
    }
}

} // namespace pomdp

extern "C" {

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
        const int64_t FUSE_MAX = fuse_max();
        for (int64_t s = 0; s < k_steps; s += FUSE_MAX) {
            const int c = (int)(k_steps - s < FUSE_MAX ? k_steps - s : FUSE_MAX);
            const uint64_t t = t0 + (uint64_t)s;
            rc = dispatch_env(env, params, [&](auto tag, const auto &p) {
                using E = typename decltype(tag)::Env;
                return launch_steps_fused<E>(p, state, action, ob, (typename E::Reward *)reward, done, err, n, seed,
                                             action_seed, lane0, t, c, flags, 0, s == 0, POMDP_LAYOUT_COLUMNS, NO_TAPE, stream);
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

// rows s .. s + c - 1 of a caller's tape as one launch's TapeRef (nullptr: the synthetic policy)
static TapeRef tape_rows(const pomdp_tape *tape, int64_t s, uint32_t *err)
{
    if (!tape) return NO_TAPE;
    return TapeRef{tape->actions + s * tape->stride, tape->stride, err};
}
static bool tape_ok(const pomdp_tape *tape, int64_t n) { return tape && tape->actions && tape->stride >= n; }

// pomdp_collect_synthetic (tape == nullptr; `action` = its [k + 1][pitch] rows) / pomdp_collect_tape (no action column)
static int collect_columns(int env, const void *params, uint32_t *state, const pomdp_tape *tape, int32_t *action, int32_t *ob,
                           void *reward, uint8_t *done, uint32_t *err, int64_t n, uint64_t seed, uint32_t lane0, uint64_t t0,
                           int64_t k_steps, int64_t pitch, int flags, void *stream)
{
    int rc = check_driver_args(env, params, state, tape ? (const void *)ob : (const void *)action, ob, reward, done, n, lane0, k_steps);
    if (rc) return rc;
    if (pitch < n || !(flags & POMDP_AUTO_RESET) || (tape && !tape_ok(tape, n))) return POMDP_E_BADARG;
    if (k_steps == 0 || n == 0) return 0;
    const int64_t FUSE_MAX = pomdp_fuse_steps(env, POMDP_LAYOUT_COLUMNS);
    for (int64_t s = 0; s < k_steps; s += FUSE_MAX) {      // the first launch writes row 0 (the actions of t0) itself
        const int c = (int)(k_steps - s < FUSE_MAX ? k_steps - s : FUSE_MAX);
        rc = dispatch_env(env, params, [&](auto tag, const auto &p) {
            using E = typename decltype(tag)::Env;
            using R = typename E::Reward;
            return launch_steps_fused<E>(p, state, tape ? nullptr : action + s * pitch, ob + s * pitch, (R *)reward + s * pitch,
                                         done + s * pitch, err, n, seed, seed, lane0, t0 + (uint64_t)s, c, flags, pitch,
                                         s == 0, POMDP_LAYOUT_COLUMNS, tape_rows(tape, s, err), stream);
        });
        if (rc) return rc;
    }
    return 0;
}

int pomdp_collect_synthetic(int env, const void *params, uint32_t *state, int32_t *action, int32_t *ob, void *reward,
                            uint8_t *done, uint32_t *err, int64_t n, uint64_t seed, uint32_t lane0, uint64_t t0,
                            int64_t k_steps, int64_t pitch, int flags, void *stream)
{
    return collect_columns(env, params, state, nullptr, action, ob, reward, done, err, n, seed, lane0, t0, k_steps, pitch, flags, stream);
}

int pomdp_collect_tape(int env, const void *params, uint32_t *state, const pomdp_tape *tape, int32_t *ob, void *reward,
                       uint8_t *done, uint32_t *err, int64_t n, uint64_t seed, uint32_t lane0, uint64_t t0, int64_t k_steps,
                       int64_t pitch, int flags, void *stream)
{
    if (!tape) return POMDP_E_BADARG;
    return collect_columns(env, params, state, tape, nullptr, ob, reward, done, err, n, seed, lane0, t0, k_steps, pitch, flags, stream);
}

int pomdp_collect(const pomdp_collect_args *a, uint64_t t0, int64_t k_steps, void *stream)
{
    if (!a) return POMDP_E_BADARG;
    return pomdp_collect_synthetic(a->env, a->params, a->state, a->action, a->ob, a->reward, a->done, a->err, a->n, a->seed,
                                   a->lane0, t0, k_steps, a->pitch, a->flags, stream);
}

// Trajectory collection into one of the single-stream layouts (include/pomdp_hip.h: POMDP_LAYOUT_BLOCKED / _PACKED): the
// launches of pomdp_collect_synthetic with another sink (traj_out.hip.h).
static int collect_layout(int env, const void *params, uint32_t *state, const pomdp_tape *tape, void *traj, uint32_t *err, int64_t n,
                          uint64_t seed, uint32_t lane0, uint64_t t0, int64_t k_steps, int64_t pitch, int layout, int flags, void *stream)
{
    if (layout != POMDP_LAYOUT_BLOCKED && layout != POMDP_LAYOUT_PACKED && layout != POMDP_LAYOUT_NARROW) return POMDP_E_BADARG;
    int rc = check_driver_args(env, params, state, traj, traj, traj, traj, n, lane0, k_steps);
    if (rc) return rc;
    if (pitch < n || !(flags & POMDP_AUTO_RESET) || (tape && !tape_ok(tape, n))) return POMDP_E_BADARG;
    if (layout == POMDP_LAYOUT_BLOCKED && pitch % 256 != 0) return POMDP_E_BADARG;
    if (layout == POMDP_LAYOUT_NARROW && pitch % 4 != 0) return POMDP_E_BADARG;
    // a Packed record (and a Narrow plane) keeps action and observation in a byte each: every env's fit by construction
    // except Tag's "opponent seen" value, which is a constructor argument (tag.py:94)
    if (layout != POMDP_LAYOUT_BLOCKED && env == POMDP_ENV_TAG && (uint32_t)((const pomdp_tag_params *)params)->obs_cells > 255u)
        return POMDP_E_BADPARAMS;
    if (k_steps == 0 || n == 0) return 0;
    const int64_t row_bytes = layout == POMDP_LAYOUT_BLOCKED ? pitch * 13 : pitch * 4;
    const int64_t FUSE_MAX = pomdp_fuse_steps(env, layout);
    for (int64_t s = 0; s < k_steps; s += FUSE_MAX) {
        const int c = (int)(k_steps - s < FUSE_MAX ? k_steps - s : FUSE_MAX);
        int32_t *base = reinterpret_cast<int32_t *>(reinterpret_cast<uint8_t *>(traj) + s * row_bytes);
        rc = dispatch_env(env, params, [&](auto tag, const auto &p) {
            using E = typename decltype(tag)::Env;
            return launch_steps_fused<E>(p, state, base, nullptr, nullptr, nullptr, err, n, seed, seed, lane0, t0 + (uint64_t)s, c,
                                         flags, pitch, true, layout, tape_rows(tape, s, err), stream);
        });
        if (rc) return rc;
    }
    return 0;
}

int pomdp_collect_layout(int env, const void *params, uint32_t *state, void *traj, uint32_t *err, int64_t n, uint64_t seed,
                         uint32_t lane0, uint64_t t0, int64_t k_steps, int64_t pitch, int layout, int flags, void *stream)
{
    return collect_layout(env, params, state, nullptr, traj, err, n, seed, lane0, t0, k_steps, pitch, layout, flags, stream);
}

int pomdp_collect_tape_layout(int env, const void *params, uint32_t *state, const pomdp_tape *tape, void *traj, uint32_t *err,
                              int64_t n, uint64_t seed, uint32_t lane0, uint64_t t0, int64_t k_steps, int64_t pitch, int layout,
                              int flags, void *stream)
{
    if (!tape) return POMDP_E_BADARG;
    return collect_layout(env, params, state, tape, traj, err, n, seed, lane0, t0, k_steps, pitch, layout, flags, stream);
}

int pomdp_collect_traj(const pomdp_traj_args *a, uint64_t t0, int64_t k_steps, void *stream)
{
    if (!a) return POMDP_E_BADARG;
    return pomdp_collect_layout(a->env, a->params, a->state, a->traj, a->err, a->n, a->seed, a->lane0, t0, k_steps, a->pitch,
                                a->layout, a->flags, stream);
}

static int collect_returns(int env, const void *params, uint32_t *state, const pomdp_tape *tape, const pomdp_return_stats *stats,
                           uint32_t *err, int64_t n, uint64_t seed, uint32_t lane0, uint64_t t0, int64_t k_steps, int flags, void *stream)
{
    if (!stats || !stats->acc || !stats->cnt) return POMDP_E_BADARG;
    int rc = check_driver_args(env, params, state, stats->acc, stats->cnt, stats->acc, stats->acc, n, lane0, k_steps);
    if (rc) return rc;
    if (stats->pitch < n || !(flags & POMDP_AUTO_RESET) || !(stats->discount == stats->discount) || (tape && !tape_ok(tape, n)))
        return POMDP_E_BADARG;
    if (k_steps == 0 || n == 0) return 0;
    // the launches of pomdp_collect_synthetic with the Returns sink (traj_out.hip.h); the discount travels as its bit pattern
    uint64_t bits;
    memcpy(&bits, &stats->discount, 8);
    const int64_t FUSE_MAX = fuse_max();
    for (int64_t s = 0; s < k_steps; s += FUSE_MAX) {
        const int c = (int)(k_steps - s < FUSE_MAX ? k_steps - s : FUSE_MAX);
        rc = dispatch_env(env, params, [&](auto tag, const auto &p) {
            using E = typename decltype(tag)::Env;
            return launch_steps_fused<E>(p, state, reinterpret_cast<int32_t *>(stats->acc), stats->cnt,
                                         reinterpret_cast<typename E::Reward *>(bits), nullptr, err, n, seed, seed, lane0,
                                         t0 + (uint64_t)s, c, flags, stats->pitch, true, LAYOUT_RETURNS, tape_rows(tape, s, err), stream);
        });
        if (rc) return rc;
    }
    return 0;
}

int pomdp_collect_returns(int env, const void *params, uint32_t *state, const pomdp_return_stats *stats, uint32_t *err,
                          int64_t n, uint64_t seed, uint32_t lane0, uint64_t t0, int64_t k_steps, int flags, void *stream)
{
    return collect_returns(env, params, state, nullptr, stats, err, n, seed, lane0, t0, k_steps, flags, stream);
}

int pomdp_collect_tape_returns(int env, const void *params, uint32_t *state, const pomdp_tape *tape,
                               const pomdp_return_stats *stats, uint32_t *err, int64_t n, uint64_t seed, uint32_t lane0,
                               uint64_t t0, int64_t k_steps, int flags, void *stream)
{
    if (!tape) return POMDP_E_BADARG;
    return collect_returns(env, params, state, tape, stats, err, n, seed, lane0, t0, k_steps, flags, stream);
}

int pomdp_decode_packed(int env, const uint32_t *records, int64_t n, int64_t k_steps, int64_t pitch_in, int32_t *action,
                        int32_t *ob, void *reward, uint8_t *done, int64_t pitch_out, void *stream)
{
    if (!records || !action || !ob || !reward || !done || n < 0 || k_steps < 0 || pitch_in < n || pitch_out < n ||
        env < POMDP_ENV_ROCK || env > POMDP_ENV_NETWORK)
        return POMDP_E_BADARG;
    if (n == 0 || k_steps == 0) return 0;
    const bool vec = ((reinterpret_cast<uintptr_t>(records) | reinterpret_cast<uintptr_t>(action) | reinterpret_cast<uintptr_t>(ob) |
                       reinterpret_cast<uintptr_t>(reward)) & 15u) == 0 && (reinterpret_cast<uintptr_t>(done) & 3u) == 0 &&
                     pitch_in % 4 == 0 && pitch_out % 4 == 0;
    const int64_t items = (vec ? (n + 3) / 4 : n) * k_steps;
    const int64_t b = (items + BLOCK - 1) / BLOCK;
    const dim3 grid((unsigned)(b > 256 * 16 ? 256 * 16 : b));                   // 16 workgroups per CU, grid-stride beyond
    if (vec)
        hipLaunchKernelGGL(decode_packed_kernel<true>, grid, dim3(BLOCK), 0, (hipStream_t)stream, records, n, k_steps, pitch_in,
                           (uint32_t *)action, (uint32_t *)ob, (uint32_t *)reward, done, pitch_out, env);
    else
        hipLaunchKernelGGL(decode_packed_kernel<false>, grid, dim3(BLOCK), 0, (hipStream_t)stream, records, n, k_steps, pitch_in,
                           (uint32_t *)action, (uint32_t *)ob, (uint32_t *)reward, done, pitch_out, env);
    return (int)hipGetLastError();
}

int pomdp_fuse_steps(int env, int layout)
{
    const bool wide = layout == POMDP_LAYOUT_COLUMNS || layout == POMDP_LAYOUT_BLOCKED;
    const bool store_bound = env == POMDP_ENV_ROCK || env == POMDP_ENV_TAG || env == POMDP_ENV_TIGER;
    const int f = g_fuse_max.load(std::memory_order_relaxed);
    return (wide && store_bound && f > 64) ? 64 : f;
}

int pomdp_fuse_max(int v)
{
    const int old = g_fuse_max.load(std::memory_order_relaxed);
    if (v >= 1) g_fuse_max.store(v > FUSE_MAX_LIMIT ? FUSE_MAX_LIMIT : v, std::memory_order_relaxed);
    return old;
}

double pomdp_packed_reward(int env, uint32_t code)
{
    code &= 0xFFu;
    if (env != POMDP_ENV_NETWORK) return (double)(int8_t)code;       // the reward itself
    const int kind = (int)code / NetworkEnv::REWARD_BASES, base = (int)code % NetworkEnv::REWARD_BASES;
    double r = (double)base;                                           // network.py:87-92, 103, 110 — as the kernels' rtab
    if (kind == 1) r -= .1;
    if (kind == 2) r -= 2.5;
    return (double)(float)r;
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

