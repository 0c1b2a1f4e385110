// philox.hip.h — per-lane counter-based random words for the gfx950 kernels.
//
// Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as
// 1, 2, 3", SC'11).  key = (seed lo, seed hi), ctr = (lane, t lo, t hi,
// stream << 24 | block).  The key schedule is wave-uniform, so the compiler keeps
// it in SGPRs; one block costs 20 32x32->64 multiplies per lane.
//
// On top of the words sit numpy's legacy RandomState constructions, which is what
// makes the kernels bit-exact against the reference's np.random draws
// (SURVEY.md §8c): res53 doubles (as their 53-bit integer numerator, so every
// Bernoulli is an integer compare against a captured threshold) and the
// masked-rejection randint.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pomdp {

// Streaming access to the per-lane columns.  Every byte of a column is read or written exactly once per launch and
// the per-XCD L2 keeps nothing across a kernel boundary, so
//   - stores are device-scope write-through (`global_store ... sc1`): the lines leave L2 while the kernel is still
//     computing instead of in one write-back burst when it ends;
//   - loads carry the `nt` bit (no allocation on the way in).
// Measured on the RockSample step kernel at 2^20 lanes (round-1 tools/microbench.hip, plain cached access as the A arm):
// 9.06 us cached -> 7.9 us with nt loads and stores -> 7.2 us with write-through stores.  Load policy alone: no effect.
template <bool STREAM = true, class T>
__device__ __forceinline__ T ld_stream(const T *p) { return STREAM ? __builtin_nontemporal_load(p) : *p; }
template <bool STREAM = true, class T>
__device__ __forceinline__ void st_stream(T *p, T v)
{
    if (STREAM) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}

// four consecutive 32-bit columns entries of one thread, streamed (16-byte accesses, no allocation in L2)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4 ld_stream4(const uint32_t *p) { return __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p)); }
__device__ __forceinline__ void st_stream4(uint32_t *p, uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
    __builtin_nontemporal_store(u32x4{a, b, c, d}, reinterpret_cast<u32x4 *>(p));
}

// ... and two (the two-lanes-per-thread loops)
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u32x2 ld_stream2(const uint32_t *p) { return __builtin_nontemporal_load(reinterpret_cast<const u32x2 *>(p)); }
__device__ __forceinline__ void st_stream2(uint32_t *p, uint32_t a, uint32_t b)
{
    __builtin_nontemporal_store(u32x2{a, b}, reinterpret_cast<u32x2 *>(p));
}

struct RngKey {          // wave-uniform part of the counter/key
    uint32_t k0, k1;     // seed lo, hi
    uint32_t t_lo, t_hi; // call counter of the batched env
};

__device__ __forceinline__ uint4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                               uint32_t k0, uint32_t k1)
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        // three-input xor in one instruction (gfx950 v_bitop3_b32, truth table 0x96); hipcc emits two
        // v_xor_b32 for the plain C expression, which is a third of a Philox block's instructions
        const uint32_t n0 = __builtin_amdgcn_bitop3_b32((uint32_t)(p1 >> 32), c1, k0, 0x96);
        const uint32_t n2 = __builtin_amdgcn_bitop3_b32((uint32_t)(p0 >> 32), c3, k1, 0x96);
        c1 = (uint32_t)p1;
        c3 = (uint32_t)p0;
        c0 = n0;
        c2 = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return make_uint4(c0, c1, c2, c3);
}

// The key as a rare branch sees it: a copy the compiler cannot see through.  A Philox block's first rounds on the wave-uniform
// counter words are cheap to speculate, and two rare branches of one step that draw the same block (a sensor tie, a reset tie)
// make them a common subexpression: hoisted in front of both — into EVERY step (RockSample, one lane per thread: two scalar
// and three vector multiplies per step for draws that happen once in 2^27).
__device__ __forceinline__ RngKey rare_key(const RngKey &k)
{
    RngKey r = k;
    asm volatile("" : "+s"(r.t_lo), "+s"(r.t_hi));
    return r;
}

__device__ __forceinline__ uint4 stream_block(const RngKey &k, uint32_t lane, uint32_t stream, uint32_t block)
{
    return philox4x32_10(lane, k.t_lo, k.t_hi, (stream << 24) | block, k.k0, k.k1);
}

// numerator of numpy's legacy double: (w0 >> 5) * 2^26 + (w1 >> 6)
__device__ __forceinline__ uint64_t k53(uint32_t w0, uint32_t w1)
{
    return ((uint64_t)(w0 >> 5) << 26) + (uint64_t)(w1 >> 6);
}

// Sequential consumer of one (lane, t, stream) word stream; blocks are generated on demand.
struct WordStream {
    const RngKey &key;
    uint32_t lane, stream, widx;
    uint4 blk;

    __device__ __forceinline__ WordStream(const RngKey &k, uint32_t lane_, uint32_t stream_)
        : key(k), lane(lane_), stream(stream_), widx(0), blk(make_uint4(0, 0, 0, 0)) {}

    __device__ __forceinline__ uint32_t next32()
    {
        const uint32_t sel = widx & 3u;
        if (sel == 0) blk = stream_block(key, lane, stream, (widx >> 2) & 0xFFFFFFu);
        ++widx;
        return sel == 0 ? blk.x : sel == 1 ? blk.y : sel == 2 ? blk.z : blk.w;
    }
    __device__ __forceinline__ uint64_t next_k53()
    {
        const uint32_t a = next32();
        const uint32_t b = next32();
        return k53(a, b);
    }
    // np.random.randint(n): mask = smear(n - 1); redraw while (word & mask) > n - 1
    __device__ __forceinline__ uint32_t randint(uint32_t n)
    {
        const uint32_t rng = n - 1u;
        if (rng == 0u) return 0u;
        const uint32_t mask = 0xFFFFFFFFu >> __clz(rng);
        uint32_t v;
        do { v = next32() & mask; } while (v > rng);
        return v;
    }
};

} // namespace pomdp
