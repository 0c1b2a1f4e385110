// microbench.hip — standalone A/B harness for the RockSample step kernel (not part of the product).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o tools/microbench tools/microbench.hip \
//        -Lgym_pomdp_amd/_lib -lpomdp_hip -Wl,-rpath,'$ORIGIN/../gym_pomdp_amd/_lib'
// (the product's kernel templates come from its headers, its launchers and C ABI from the library it links against)
#include "../gym_pomdp_amd/csrc/step_impl.hip.h"
#include "../gym_pomdp_amd/csrc/fused_impl.hip.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <chrono>

using namespace pomdp;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

// V0: pure traffic (same 21 B/lane, trivial ALU)
__global__ __launch_bounds__(256) void traffic_kernel(uint32_t *__restrict__ state, const int32_t *__restrict__ action,
                                                      int32_t *__restrict__ ob, int32_t *__restrict__ reward,
                                                      uint8_t *__restrict__ done, int64_t n)
{
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const uint32_t s = state[i];
        const int a = action[i];
        state[i] = s + (uint32_t)a;
        ob[i] = a & 3;
        reward[i] = a - 5;
        done[i] = (uint8_t)(s & 1u);
    }
}

template <int BS>
__global__ __launch_bounds__(BS) void traffic_kernel_bs(uint32_t *__restrict__ state, const int32_t *__restrict__ action,
                                                      int32_t *__restrict__ ob, int32_t *__restrict__ reward,
                                                      uint8_t *__restrict__ done, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * BS + threadIdx.x;
    if (i < n) {
        const uint32_t s = state[i];
        const int a = action[i];
        state[i] = s + (uint32_t)a;
        ob[i] = a & 3;
        reward[i] = a - 5;
        done[i] = (uint8_t)(s & 1u);
    }
}

// the same 21 B/lane with the product's streaming accessors (nt loads, write-through stores), one or two lanes per thread,
// plus NB Philox blocks per thread of register-only work: the floor for a kernel that moves the step kernel's bytes
template <int LPT, int NB>
__global__ __launch_bounds__(256) void traffic_stream(uint32_t *__restrict__ state, const int32_t *__restrict__ action,
                                                      int32_t *__restrict__ ob, int32_t *__restrict__ reward,
                                                      uint8_t *__restrict__ done, int64_t n, RngKey key)
{
    const uint32_t wg0 = blockIdx.x * (uint32_t)(256 * LPT);
    uint32_t s[LPT]; int a[LPT];
#pragma unroll
    for (int j = 0; j < LPT; ++j) { s[j] = ld_stream(state + wg0 + threadIdx.x + j * 256); a[j] = ld_stream(action + wg0 + threadIdx.x + j * 256); }
    uint32_t acc = 0;
#pragma unroll
    for (int b = 0; b < NB; ++b) { const uint4 w = stream_block(key, wg0 + threadIdx.x, 0u, (uint32_t)b); acc ^= w.x ^ w.y ^ w.z ^ w.w; }
#pragma unroll
    for (int j = 0; j < LPT; ++j) {
        const uint32_t i = wg0 + threadIdx.x + j * 256;
        st_stream(state + i, s[j] + (uint32_t)a[j] + (acc & 1u));
        st_stream(ob + i, (int32_t)(a[j] & 3));
        st_stream(reward + i, (int32_t)(a[j] - 5));
        st_stream(done + i, (uint8_t)(s[j] & 1u));
    }
}

// BattleShip-shaped traffic: 8 state words in, 4 out, + action/ob/reward/done
__global__ __launch_bounds__(256) void bs_traffic_soa(uint32_t *__restrict__ state, const int32_t *__restrict__ action,
                                                      int32_t *__restrict__ ob, int32_t *__restrict__ reward,
                                                      uint8_t *__restrict__ done, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint32_t w[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) w[j] = state[j * n + i];
    const int a = action[i];
    const uint32_t x = w[0] ^ w[1] ^ w[2] ^ w[3];
#pragma unroll
    for (int j = 4; j < 8; ++j) state[j * n + i] = w[j] + x + (uint32_t)a;
    ob[i] = a & 1; reward[i] = a - 5; done[i] = (uint8_t)(x & 1u);
}
__global__ __launch_bounds__(256) void bs_traffic_v4(uint4 *__restrict__ state, const int32_t *__restrict__ action,
                                                     int32_t *__restrict__ ob, int32_t *__restrict__ reward,
                                                     uint8_t *__restrict__ done, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint4 o = state[i], v = state[n + i];
    const int a = action[i];
    const uint32_t x = o.x ^ o.y ^ o.z ^ o.w;
    state[n + i] = make_uint4(v.x + x + a, v.y + x, v.z + x, v.w + x);
    ob[i] = a & 1; reward[i] = a - 5; done[i] = (uint8_t)(x & 1u);
}

// V0v: same traffic, 4 lanes per thread, 16-byte accesses
__global__ __launch_bounds__(256) void traffic_kernel_v4(uint4 *__restrict__ state, const int4 *__restrict__ action,
                                                         int4 *__restrict__ ob, int4 *__restrict__ reward,
                                                         uint32_t *__restrict__ done, int64_t n4)
{
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const uint4 s = state[i];
        const int4 a = action[i];
        state[i] = make_uint4(s.x + a.x, s.y + a.y, s.z + a.z, s.w + a.w);
        ob[i] = make_int4(a.x & 3, a.y & 3, a.z & 3, a.w & 3);
        reward[i] = make_int4(a.x - 5, a.y - 5, a.z - 5, a.w - 5);
        done[i] = (s.x & 1u) | ((s.y & 1u) << 8) | ((s.z & 1u) << 16) | ((s.w & 1u) << 24);
    }
}

// Philox-only: k blocks per lane, one store
template <int NBLK>
__global__ __launch_bounds__(256) void philox_kernel(uint32_t *__restrict__ out, int64_t n, RngKey key)
{
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        uint32_t acc = 0;
#pragma unroll
        for (int b = 0; b < NBLK; ++b) { const uint4 w = stream_block(key, (uint32_t)i, 0, b); acc ^= w.x ^ w.y ^ w.z ^ w.w; }
        out[i] = acc;
    }
}

__global__ void empty_kernel() {}

// the same floor with a thread owning four CONSECUTIVE lanes: one 16-byte store per int32 column and one 4-byte store for
// the done bytes per thread-step instead of twenty scalar stores
template <int NB, int POLICY = 0>   // POLICY 0: nontemporal, 1: plain cached, 2: write-through (sc1)
__global__ __launch_bounds__(256) void fused_store_floor_v4(int32_t *__restrict__ action, int32_t *__restrict__ ob,
                                                            int32_t *__restrict__ reward, uint8_t *__restrict__ done, int k,
                                                            int64_t rec, RngKey key)
{
    const uint32_t l0 = blockIdx.x * 1024u + 4u * threadIdx.x;
    typedef int v4i __attribute__((ext_vector_type(4)));
    v4i *a_w = reinterpret_cast<v4i *>(action + l0), *o_w = reinterpret_cast<v4i *>(ob + l0), *r_w = reinterpret_cast<v4i *>(reward + l0);
    uint32_t *d_w = reinterpret_cast<uint32_t *>(done + l0);
    uint32_t v = threadIdx.x;
    for (int s = 0; s < k; ++s) {
#pragma unroll
        for (int b = 0; b < NB; ++b) { const uint4 w = stream_block(key, l0 >> 2, (uint32_t)s, (uint32_t)b); v ^= w.x ^ w.y ^ w.z ^ w.w; }
        auto put = [&](v4i *q, v4i x) {
            if (POLICY == 0) __builtin_nontemporal_store(x, q);
            else if (POLICY == 1) *q = x;
            else asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(q), "v"(x) : "memory");
        };
        put(a_w, v4i{(int)v, (int)v + 1, (int)v + 2, (int)v + 3});
        put(o_w, v4i{(int)(v & 3u), 0, 1, 2});
        put(r_w, v4i{(int)(v >> 7), 0, 10, -10});
        if (POLICY == 1) *d_w = v & 0x01010101u; else st_stream(d_w, v & 0x01010101u);
        v = v * 5u + 1u;
        a_w += rec / 4; o_w += rec / 4; r_w += rec / 4; d_w += rec / 4;
    }
}

// the fused loop's store pattern with NB Philox blocks per THREAD-step of register work and nothing else: k steps, four lanes per
// thread, each step writes action / ob / reward (4 B) and done (1 B) per lane with the product's write-through stores; rows
// advance by rec elements per step.  The floor for a launch that has to emit 13 B per lane-step.
template <int NB>
__global__ __launch_bounds__(256) void fused_store_floor(int32_t *__restrict__ action, int32_t *__restrict__ ob,
                                                         int32_t *__restrict__ reward, uint8_t *__restrict__ done, int k,
                                                         int64_t rec, RngKey key)
{
    const uint32_t wg0 = blockIdx.x * 1024u;
    int32_t *a_w = action + wg0, *o_w = ob + wg0, *r_w = reward + wg0;
    uint8_t *d_w = done + wg0;
    uint32_t v = threadIdx.x;
    for (int s = 0; s < k; ++s) {
#pragma unroll
        for (int b = 0; b < NB; ++b) { const uint4 w = stream_block(key, wg0 + threadIdx.x, (uint32_t)s, (uint32_t)b); v ^= w.x ^ w.y ^ w.z ^ w.w; }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t rel = threadIdx.x + 256u * j;
            st_stream(a_w + rel, (int32_t)(v + j));
            st_stream(o_w + rel, (int32_t)(v & 3u));
            st_stream(r_w + rel, (int32_t)(v >> 7));
            st_stream(d_w + rel, (uint8_t)(v & 1u));
        }
        v = v * 5u + 1u;
        a_w += rec; o_w += rec; r_w += rec; d_w += rec;
    }
}

template <class F> static float time_it(F f, int iters)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 10; ++i) f(i);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) f(100 + i);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / iters;
}

int main(int argc, char **argv)
{
    const int64_t n = argc > 1 ? atoll(argv[1]) : (1 << 20);
    const int iters = getenv("MB_ITERS") ? atoi(getenv("MB_ITERS")) : 500;
    uint32_t *state; int32_t *action, *ob, *reward; uint8_t *done; uint32_t *err;
    CK(hipMalloc(&state, n * 8)); CK(hipMalloc(&action, n * 4)); CK(hipMalloc(&ob, n * 4)); CK(hipMalloc(&reward, n * 4));
    CK(hipMalloc(&done, n)); CK(hipMalloc(&err, 4)); CK(hipMemset(err, 0, 4));
    uint32_t *bsstate; CK(hipMalloc(&bsstate, n * 32)); CK(hipMemset(bsstate, 0, n * 32));
    // RockSample(7,8) params
    pomdp_rock_params p = {};
    p.size = 7; p.num_rocks = 8; p.start_x = 0; p.start_y = 3;
    const int rp[8][2] = {{2,0},{0,1},{3,1},{6,3},{2,4},{3,4},{5,5},{1,6}};
    for (int i = 0; i < 256; ++i) p.grid[i] = -1;
    for (int i = 0; i < 8; ++i) { p.rock_x[i] = rp[i][0]; p.rock_y[i] = rp[i][1]; p.grid[rp[i][0] * 16 + rp[i][1]] = i; }
    for (int d = 0; d < 32; ++d) p.thr[d] = 8000000000000000ull;
    pomdp_rock_reset(&p, state, ob, n, 1, 0, 0, nullptr);
    pomdp_synthetic_actions(action, n, 2, 0, 1, 13, nullptr);
    CK(hipDeviceSynchronize());

    printf("n = %lld lanes, %d iters, times in us per launch\n", (long long)n, iters);
    printf("empty kernel                : %8.2f\n", time_it([&](int) { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, 0); }, iters));
    for (int blocks : {1024, 2048, 4096, 8192}) {
        printf("traffic 21B  grid=%5d      : %8.2f\n", blocks,
               time_it([&](int) { hipLaunchKernelGGL(traffic_kernel, dim3(blocks), dim3(256), 0, 0, state, action, ob, reward, done, n); }, iters));
    }
    printf("traffic 21B  bs=64  1 lane/thr : %8.2f\n", time_it([&](int) { hipLaunchKernelGGL(traffic_kernel_bs<64>, dim3((unsigned)(n / 64)), dim3(64), 0, 0, state, action, ob, reward, done, n); }, iters));
    printf("traffic 21B  bs=256 1 lane/thr : %8.2f\n", time_it([&](int) { hipLaunchKernelGGL(traffic_kernel_bs<256>, dim3((unsigned)(n / 256)), dim3(256), 0, 0, state, action, ob, reward, done, n); }, iters));
    printf("traffic 21B  bs=512 1 lane/thr : %8.2f\n", time_it([&](int) { hipLaunchKernelGGL(traffic_kernel_bs<512>, dim3((unsigned)(n / 512)), dim3(512), 0, 0, state, action, ob, reward, done, n); }, iters));
    printf("traffic 21B  bs=1024 1 lane/thr: %8.2f\n", time_it([&](int) { hipLaunchKernelGGL(traffic_kernel_bs<1024>, dim3((unsigned)(n / 1024)), dim3(1024), 0, 0, state, action, ob, reward, done, n); }, iters));
    printf("streamed 21B  1 lane/thr, +0/1/2/4/6 philox blocks : %6.2f %6.2f %6.2f %6.2f %6.2f\n",
           time_it([&](int t) { hipLaunchKernelGGL((traffic_stream<1, 0>), dim3((unsigned)(n / 256)), dim3(256), 0, 0, state, action, ob, reward, done, n, make_key(1, t)); }, iters),
           time_it([&](int t) { hipLaunchKernelGGL((traffic_stream<1, 1>), dim3((unsigned)(n / 256)), dim3(256), 0, 0, state, action, ob, reward, done, n, make_key(1, t)); }, iters),
           time_it([&](int t) { hipLaunchKernelGGL((traffic_stream<1, 2>), dim3((unsigned)(n / 256)), dim3(256), 0, 0, state, action, ob, reward, done, n, make_key(1, t)); }, iters),
           time_it([&](int t) { hipLaunchKernelGGL((traffic_stream<1, 4>), dim3((unsigned)(n / 256)), dim3(256), 0, 0, state, action, ob, reward, done, n, make_key(1, t)); }, iters),
           time_it([&](int t) { hipLaunchKernelGGL((traffic_stream<1, 6>), dim3((unsigned)(n / 256)), dim3(256), 0, 0, state, action, ob, reward, done, n, make_key(1, t)); }, iters));
    printf("streamed 21B  2 lanes/thr, +0/2/4/8/12 philox blocks: %6.2f %6.2f %6.2f %6.2f %6.2f\n",
           time_it([&](int t) { hipLaunchKernelGGL((traffic_stream<2, 0>), dim3((unsigned)(n / 512)), dim3(256), 0, 0, state, action, ob, reward, done, n, make_key(1, t)); }, iters),
           time_it([&](int t) { hipLaunchKernelGGL((traffic_stream<2, 2>), dim3((unsigned)(n / 512)), dim3(256), 0, 0, state, action, ob, reward, done, n, make_key(1, t)); }, iters),
           time_it([&](int t) { hipLaunchKernelGGL((traffic_stream<2, 4>), dim3((unsigned)(n / 512)), dim3(256), 0, 0, state, action, ob, reward, done, n, make_key(1, t)); }, iters),
           time_it([&](int t) { hipLaunchKernelGGL((traffic_stream<2, 8>), dim3((unsigned)(n / 512)), dim3(256), 0, 0, state, action, ob, reward, done, n, make_key(1, t)); }, iters),
           time_it([&](int t) { hipLaunchKernelGGL((traffic_stream<2, 12>), dim3((unsigned)(n / 512)), dim3(256), 0, 0, state, action, ob, reward, done, n, make_key(1, t)); }, iters));
    printf("battleship traffic SoA dwords : %8.2f\n", time_it([&](int) { hipLaunchKernelGGL(bs_traffic_soa, dim3((unsigned)(n / 256)), dim3(256), 0, 0, bsstate, action, ob, reward, done, n); }, iters));
    printf("battleship traffic 16B/lane   : %8.2f\n", time_it([&](int) { hipLaunchKernelGGL(bs_traffic_v4, dim3((unsigned)(n / 256)), dim3(256), 0, 0, (uint4 *)bsstate, action, ob, reward, done, n); }, iters));
    for (int blocks : {256, 512, 1024, 2048}) {
        printf("traffic 21B v4 grid=%5d    : %8.2f\n", blocks,
               time_it([&](int) { hipLaunchKernelGGL(traffic_kernel_v4, dim3(blocks), dim3(256), 0, 0, (uint4 *)state, (const int4 *)action, (int4 *)ob, (int4 *)reward, (uint32_t *)done, n / 4); }, iters));
    }
    RngKey key = make_key(1, 5);
    printf("philox 1 blk/lane           : %8.2f\n", time_it([&](int) { hipLaunchKernelGGL(philox_kernel<1>, dim3(2048), dim3(256), 0, 0, (uint32_t *)ob, n, key); }, iters));
    printf("philox 2 blk/lane           : %8.2f\n", time_it([&](int) { hipLaunchKernelGGL(philox_kernel<2>, dim3(2048), dim3(256), 0, 0, (uint32_t *)ob, n, key); }, iters));
    printf("philox 4 blk/lane           : %8.2f\n", time_it([&](int) { hipLaunchKernelGGL(philox_kernel<4>, dim3(2048), dim3(256), 0, 0, (uint32_t *)ob, n, key); }, iters));
    printf("philox 8 blk/lane           : %8.2f\n", time_it([&](int) { hipLaunchKernelGGL(philox_kernel<8>, dim3(2048), dim3(256), 0, 0, (uint32_t *)ob, n, key); }, iters));
    printf("synthetic actions           : %8.2f\n", time_it([&](int t) { pomdp_synthetic_actions(action, n, 2, 0, t, 13, nullptr); }, iters));
    pomdp_synthetic_actions(action, n, 2, 0, 1, 13, nullptr);
    printf("rock step (auto-reset)      : %8.2f\n", time_it([&](int t) { pomdp_rock_step(&p, state, action, ob, reward, done, err, n, 1, 0, t, 1, nullptr); }, iters));
    {
        auto run = [&](auto env_tag, const char *name) {
            using E = decltype(env_tag);
            printf("%-28s: LPT1 %8.2f", name, time_it([&](int t) {
                hipLaunchKernelGGL((step_kernel<E, 1>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, state, action, ob, reward, done, err, n, make_key(1, t), 0u, 1, RngKey(), p);
            }, iters));
            printf("  LPT2 %8.2f", time_it([&](int t) {
                hipLaunchKernelGGL((step_kernel<E, 2>), dim3((unsigned)((n + 511) / 512)), dim3(256), 0, 0, state, action, ob, reward, done, err, n, make_key(1, t), 0u, 1, RngKey(), p);
            }, iters));
            printf("  LPT4 %8.2f\n", time_it([&](int t) {
                hipLaunchKernelGGL((step_kernel<E, 4>), dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, 0, state, action, ob, reward, done, err, n, make_key(1, t), 0u, 1, RngKey(), p);
            }, iters));
        };
        run(RockEnv<1>{}, "step full");
        {
            using E = RockEnv<1>;
            printf("%-28s: LPT1 %8.2f", "chain step", time_it([&](int t) {
                hipLaunchKernelGGL((step_kernel<E, 1, true>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, state, action, ob, reward, done, err, n, make_key(1, t), 0u, 1, make_key(1, t + 1), p);
            }, iters));
            printf("  LPT2 %8.2f", time_it([&](int t) {
                hipLaunchKernelGGL((step_kernel<E, 2, true>), dim3((unsigned)((n + 511) / 512)), dim3(256), 0, 0, state, action, ob, reward, done, err, n, make_key(1, t), 0u, 1, make_key(1, t + 1), p);
            }, iters));
            printf("  LPT4 %8.2f", time_it([&](int t) {
                hipLaunchKernelGGL((step_kernel<E, 4, true>), dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, 0, state, action, ob, reward, done, err, n, make_key(1, t), 0u, 1, make_key(1, t + 1), p);
            }, iters));
            for (int lds_kb : {24, 36, 48, 72}) {   // occupancy limited by a dummy dynamic-LDS request
                printf("  LPT2/lds%dk %6.2f", lds_kb, time_it([&](int t) {
                    hipLaunchKernelGGL((step_kernel<E, 2, true>), dim3((unsigned)((n + 511) / 512)), dim3(256), lds_kb * 1024, 0, state, action, ob, reward, done, err, n, make_key(1, t), 0u, 1, make_key(1, t + 1), p);
                }, iters));
            }
            printf("\n");
        }
    }
    {   // split the batch over S streams: policy + step per part, parts are independent
        for (int S : {1, 2, 4}) {
            std::vector<hipStream_t> st(S);
            for (auto &x : st) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
            const int64_t part = n / S;
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            auto body = [&](int t) {
                for (int k = 0; k < S; ++k) {
                    pomdp_synthetic_actions(action + k * part, part, 2, (uint32_t)(k * part), t, 13, st[k]);
                    pomdp_rock_step(&p, state + k * part, action + k * part, ob + k * part, reward + k * part, done + k * part, err, part, 1, (uint32_t)(k * part), t, 1, st[k]);
                }
            };
            for (int i = 0; i < 10; ++i) body(i);
            CK(hipDeviceSynchronize());
            auto t0 = std::chrono::high_resolution_clock::now();
            for (int i = 0; i < iters; ++i) body(100 + i);
            CK(hipDeviceSynchronize());
            auto t1 = std::chrono::high_resolution_clock::now();
            printf("policy+step over %d stream(s): %8.2f us/step\n", S, std::chrono::duration<double, std::micro>(t1 - t0).count() / iters);
            for (auto &x : st) CK(hipStreamDestroy(x));
        }
    }
    printf("pomdp_rollout_synthetic (C driver, chained), per step: %8.2f\n",
           time_it([&](int t) { pomdp_rollout_synthetic(POMDP_ENV_ROCK, &p, state, action, ob, reward, done, err, n, 1, 1, 0, (uint64_t)t * 100, 100, 1, nullptr); }, iters / 100 + 1) / 100);
    {   // fused multi-step launches: 64 chained steps per launch (SIMPLE = full workgroups, auto-reset), per step
        using E = RockEnv<1>;
        const int reps = iters / 64 + 1;
        printf("steps_kernel (64 steps per launch), per step: LPT2 simple %6.2f", time_it([&](int t) {
            hipLaunchKernelGGL((steps_kernel<E, 2, true>), dim3((unsigned)(n / 512)), dim3(256), 0, 0, state, action, ob, reward, done, err, n, make_key(1, 64 * t), 0u, 1, make_key(1, 64 * t + 1), 64, 0, p);
        }, reps) / 64);
        printf("  LPT2 general %6.2f", time_it([&](int t) {
            hipLaunchKernelGGL((steps_kernel<E, 2, false>), dim3((unsigned)((n + 511) / 512)), dim3(256), 0, 0, state, action, ob, reward, done, err, n, make_key(1, 64 * t), 0u, 1, make_key(1, 64 * t + 1), 64, 0, p);
        }, reps) / 64);
        printf("  LPT4 simple %6.2f\n", time_it([&](int t) {
            hipLaunchKernelGGL((steps_kernel<E, 4, true>), dim3((unsigned)(n / 1024)), dim3(256), 0, 0, state, action, ob, reward, done, err, n, make_key(1, 64 * t), 0u, 1, make_key(1, 64 * t + 1), 64, 0, p);
        }, reps) / 64);
    }
    {   // the LPT4 SIMPLE fused loop with the table-driven and the arithmetic lane step (the ablation variants of round 1 —
        // no sensor block / no auto-reset / no LDS lookups — were built from RockEnv<W, ABLATE>, which the product no longer
        // carries: `git show 2884871:tools/microbench.hip` has them)
        const int reps = iters / 64 + 1;
        auto fused4 = [&](auto tag, const char *what) {
            using E = decltype(tag);
            printf("  steps_kernel LPT4 simple, %-44s %6.2f us/step\n", what, time_it([&](int t) {
                hipLaunchKernelGGL((steps_kernel<E, 4, true>), dim3((unsigned)(n / 1024)), dim3(256), 0, 0, state, action, ob, reward, done, err, n, make_key(1, 64 * t), 0u, 1, make_key(1, 64 * t + 1), 64, 0, p);
            }, reps) / 64);
        };
        printf("  steps_kernel LPT4 simple, lane step from the (position, action) table     %6.2f us/step  <- as shipped from 16 steps per launch\n", time_it([&](int t) {
            hipLaunchKernelGGL((steps_kernel<RockEnv<1>, 4, true, true>), dim3((unsigned)(n / 1024)), dim3(256), 0, 0, state, action, ob, reward, done, err, n, make_key(1, 64 * t), 0u, 1, make_key(1, 64 * t + 1), 64, 0, p);
        }, reps) / 64);
        fused4(RockEnv<1>{}, "arithmetic lane step");
        int32_t *ta, *to, *tr; uint8_t *td;                         // 64-row trajectory buffers
        CK(hipMalloc(&ta, 65 * n * 4)); CK(hipMalloc(&to, 64 * n * 4)); CK(hipMalloc(&tr, 64 * n * 4)); CK(hipMalloc(&td, 64 * n));
        printf("  steps_kernel LPT4 simple, one row per step (rec = n)                   %6.2f us/step\n", time_it([&](int t) {
            hipLaunchKernelGGL((steps_kernel<RockEnv<1>, 4, true>), dim3((unsigned)(n / 1024)), dim3(256), 0, 0, state, ta, to, tr, td, err, n, make_key(1, 64 * t), 0u, 1, make_key(1, 64 * t + 1), 64, n, p);
        }, reps) / 64);
        printf("  store floor (13 B per lane-step, 64 steps per launch), Philox blocks per thread-step 0 / 1 / 2 / 3, same row | own row:\n   ");
        for (int64_t rec : {(int64_t)0, n}) {
            printf(" %6.2f", time_it([&](int t) { hipLaunchKernelGGL((fused_store_floor<0>), dim3((unsigned)(n / 1024)), dim3(256), 0, 0, ta, to, tr, td, 64, rec, make_key(1, t)); }, reps) / 64);
            printf(" %6.2f", time_it([&](int t) { hipLaunchKernelGGL((fused_store_floor<1>), dim3((unsigned)(n / 1024)), dim3(256), 0, 0, ta, to, tr, td, 64, rec, make_key(1, t)); }, reps) / 64);
            printf(" %6.2f", time_it([&](int t) { hipLaunchKernelGGL((fused_store_floor<2>), dim3((unsigned)(n / 1024)), dim3(256), 0, 0, ta, to, tr, td, 64, rec, make_key(1, t)); }, reps) / 64);
            printf(" %6.2f  |", time_it([&](int t) { hipLaunchKernelGGL((fused_store_floor<3>), dim3((unsigned)(n / 1024)), dim3(256), 0, 0, ta, to, tr, td, 64, rec, make_key(1, t)); }, reps) / 64);
        }
        printf("\n   ... a thread owning four consecutive lanes (16-byte stores), 0 / 2 blocks:");
        for (int64_t rec : {(int64_t)0, n}) {
            printf(" %6.2f", time_it([&](int t) { hipLaunchKernelGGL((fused_store_floor_v4<0>), dim3((unsigned)(n / 1024)), dim3(256), 0, 0, ta, to, tr, td, 64, rec, make_key(1, t)); }, reps) / 64);
            printf(" %6.2f  |", time_it([&](int t) { hipLaunchKernelGGL((fused_store_floor_v4<2>), dim3((unsigned)(n / 1024)), dim3(256), 0, 0, ta, to, tr, td, 64, rec, make_key(1, t)); }, reps) / 64);
        }
        printf("\n   ... the same with plain cached | write-through stores, 2 blocks, same row, own row:");
        for (int64_t rec : {(int64_t)0, n}) {
            printf(" %6.2f", time_it([&](int t) { hipLaunchKernelGGL((fused_store_floor_v4<2, 1>), dim3((unsigned)(n / 1024)), dim3(256), 0, 0, ta, to, tr, td, 64, rec, make_key(1, t)); }, reps) / 64);
            printf(" %6.2f  |", time_it([&](int t) { hipLaunchKernelGGL((fused_store_floor_v4<2, 2>), dim3((unsigned)(n / 1024)), dim3(256), 0, 0, ta, to, tr, td, 64, rec, make_key(1, t)); }, reps) / 64);
        }
        printf("\n");
        CK(hipFree(ta)); CK(hipFree(to)); CK(hipFree(tr)); CK(hipFree(td));
    }
    {   // hipGraph replay of 100 chained step launches vs the same launches issued one by one
        using E = RockEnv<1>;
        hipStream_t gs; CK(hipStreamCreateWithFlags(&gs, hipStreamNonBlocking));
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(gs, hipStreamCaptureModeGlobal));
        for (int t = 0; t < 100; ++t)
            hipLaunchKernelGGL((step_kernel<E, 2, true>), dim3((unsigned)((n + 511) / 512)), dim3(256), 0, gs, state, action, ob, reward, done, err, n, make_key(1, 1000 + t), 0u, 1, make_key(1, 1001 + t), p);
        CK(hipStreamEndCapture(gs, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, gs));
        CK(hipStreamSynchronize(gs));
        const int reps = iters / 100 + 1;
        auto t0 = std::chrono::high_resolution_clock::now();
        for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ge, gs));
        CK(hipStreamSynchronize(gs));
        auto t1 = std::chrono::high_resolution_clock::now();
        printf("hipGraph of 100 chained steps, per step: %8.2f\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / (reps * 100.0));
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g)); CK(hipStreamDestroy(gs));
    }
    {   // does the CPU's run-ahead matter?  chained step launches on the null stream / a created stream, paced by a busy-wait
        using E = RockEnv<1>;
        hipStream_t cs; CK(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
        for (hipStream_t strm : {(hipStream_t)0, cs}) {
            for (double pace_us : {0.0, 3.0, 5.0, 6.5}) {
                hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
                auto launch = [&](int t) {
                    hipLaunchKernelGGL((step_kernel<E, 2, true>), dim3((unsigned)((n + 511) / 512)), dim3(256), 0, strm, state, action, ob, reward, done, err, n, make_key(1, t), 0u, 1, make_key(1, t + 1), p);
                };
                for (int i = 0; i < 10; ++i) launch(i);
                CK(hipDeviceSynchronize());
                CK(hipEventRecord(e0, strm));
                auto w0 = std::chrono::high_resolution_clock::now();
                for (int i = 0; i < iters; ++i) {
                    auto l0 = std::chrono::high_resolution_clock::now();
                    launch(100 + i);
                    while (std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - l0).count() < pace_us) {}
                }
                auto w1 = std::chrono::high_resolution_clock::now();
                CK(hipEventRecord(e1, strm));
                CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                printf("chain step, %s stream, launch paced to %.1f us: %6.2f us/step by events, cpu issue %.2f us/launch\n",
                       strm ? "created" : "null", pace_us, ms * 1e3f / iters,
                       std::chrono::duration<double, std::micro>(w1 - w0).count() / iters);
            }
        }
        CK(hipStreamDestroy(cs));
    }
    {   // chained step launches (step + next policy in one kernel), batch split over S streams, launches interleaved
        using E = RockEnv<1>;
        for (int S : {1, 2, 4, 8}) {
            std::vector<hipStream_t> st(S);
            for (auto &x : st) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
            const int64_t part = n / S;
            auto body = [&](int t) {
                for (int k = 0; k < S; ++k) {
                    const int64_t o = k * part;
                    if (part >= (1 << 18))
                        hipLaunchKernelGGL((step_kernel<E, 2, true>), dim3((unsigned)((part + 511) / 512)), dim3(256), 0, st[k], state + o, action + o, ob + o, reward + o, done + o, err, part, make_key(1, t), (uint32_t)o, 1, make_key(1, t + 1), p);
                    else
                        hipLaunchKernelGGL((step_kernel<E, 1, true>), dim3((unsigned)((part + 255) / 256)), dim3(256), 0, st[k], state + o, action + o, ob + o, reward + o, done + o, err, part, make_key(1, t), (uint32_t)o, 1, make_key(1, t + 1), p);
                }
            };
            for (int i = 0; i < 10; ++i) body(i);
            CK(hipDeviceSynchronize());
            auto t0 = std::chrono::high_resolution_clock::now();
            for (int i = 0; i < iters; ++i) body(100 + i);
            CK(hipDeviceSynchronize());
            auto t1 = std::chrono::high_resolution_clock::now();
            printf("chained step over %d stream(s): %8.2f us/step\n", S, std::chrono::duration<double, std::micro>(t1 - t0).count() / iters);
            for (auto &x : st) CK(hipStreamDestroy(x));
        }
    }
    printf("rock reset                  : %8.2f\n", time_it([&](int t) { pomdp_rock_reset(&p, state, ob, n, 1, 0, t, nullptr); }, iters));
    return 0;
}
