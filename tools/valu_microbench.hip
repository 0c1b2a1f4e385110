// valu_microbench.hip — issue cost of the VALU instruction classes the POMDP kernels are made of, on gfx950 (dev aid;
// nothing here ships).  SURVEY.md §8d asks for an integer-op rate instead of a bytes figure for the compute-bound
// kernels; the rate's ceiling depends on the instruction mix (a wave64 v_add_u32 and a v_mad_u64_u32 do not cost the
// same number of issue cycles), so this measures each class: U independent accumulators, each instruction depending only
// on its own accumulator, R iterations, W waves per SIMD running the same loop — ONE workgroup of 256 W threads per
// sampled CU (its waves are dealt round-robin to the CU's four SIMDs, so W per SIMD is certain; 32 workgroups per launch so
// that no two share a CU) — cycles per wave-instruction per SIMD = the workgroup's span in shader cycles (s_memtime: first
// wave's start to last wave's end) / (W * R * U); the same span by the 100 MHz s_memrealtime counter gives the clock.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/valu_microbench tools/valu_microbench.hip && tools/valu_microbench > out.json
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <string>
#include <algorithm>
#include "../gym_pomdp_amd/csrc/philox.hip.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int U = 16;      // independent accumulators per lane
constexpr int R = 2048;    // loop iterations

// one instruction per accumulator per iteration; the template parameter picks the class
enum Op { ADD, AND, XOR, BITOP3, CNDMASK, LSHR, LSHL_OR, AND_OR, MUL_HI, MUL_LO, MAD_U64, SAD_U8, BFE, CMP, MOV, DPP_MOV, BCNT, ROTR, MIN_U32,
          ADD3, PERM, MAD_U32_U24, LSHL_B64, ADD_CO, ADD_F64, MUL_F64, CVT_F64_I32, OR3, LSHL_ADD, XAD, SUB, OR, FFBL, N_OPS };
static const char *OP_NAME[N_OPS] = {"v_add_u32", "v_and_b32", "v_xor_b32", "v_bitop3_b32", "v_cndmask_b32", "v_lshrrev_b32", "v_lshl_or_b32",
                                     "v_and_or_b32", "v_mul_hi_u32", "v_mul_lo_u32", "v_mad_u64_u32", "v_sad_u8", "v_bfe_u32", "v_cmp_lt_u32",
                                     "v_mov_b32", "v_mov_b32_dpp", "v_bcnt_u32_b32", "v_alignbit_b32", "v_min_u32", "v_add3_u32", "v_perm_b32",
                                     "v_mad_u32_u24", "v_lshlrev_b64", "v_add_co_u32", "v_add_f64", "v_mul_f64", "v_cvt_f64_i32", "v_or3_b32",
                                     "v_lshl_add_u32", "v_xad_u32", "v_sub_u32", "v_or_b32", "v_ffbl_b32"};

template <int OP>
__global__ __launch_bounds__(1024) void bench(uint32_t *out, uint64_t *cycles, uint32_t seed)
{
    uint32_t x[U];
    uint64_t y[U];
    const uint32_t c = seed * 2654435761u + threadIdx.x;
    const uint64_t mask64 = 0x5555555555555555ull * (uint64_t)(seed | 1u);
#pragma unroll
    for (int i = 0; i < U; ++i) { x[i] = c + (uint32_t)i * 0x9E3779B9u; y[i] = ((uint64_t)x[i] << 32) | (uint32_t)i; }
    const uint64_t r0 = wall_clock64(), t0 = __builtin_readcyclecounter();
    for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int i = 0; i < U; ++i) {
            if (OP == ADD) asm volatile("v_add_u32 %0, %1, %0" : "+v"(x[i]) : "v"(c));
            if (OP == AND) asm volatile("v_and_b32 %0, %1, %0" : "+v"(x[i]) : "v"(c));
            if (OP == XOR) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(x[i]) : "v"(c));
            if (OP == BITOP3) asm volatile("v_bitop3_b32 %0, %1, %2, %0 bitop3:0x96" : "+v"(x[i]) : "v"(c), "s"(seed));
            if (OP == CNDMASK) asm volatile("v_cndmask_b32_e64 %0, %1, %0, %2" : "+v"(x[i]) : "v"(c), "s"(mask64));
            if (OP == LSHR) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(x[i]));
            if (OP == LSHL_OR) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(x[i]) : "v"(c));
            if (OP == AND_OR) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(c), "s"(seed));
            if (OP == MUL_HI) asm volatile("v_mul_hi_u32 %0, %1, %0" : "+v"(x[i]) : "v"(c));
            if (OP == MUL_LO) asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(x[i]) : "v"(c));
            if (OP == MAD_U64) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(y[i]) : "v"(c), "s"(seed) : "vcc");
            if (OP == SAD_U8) asm volatile("v_sad_u8 %0, %0, %1, 0" : "+v"(x[i]) : "v"(c));
            if (OP == BFE) asm volatile("v_bfe_u32 %0, %0, 1, 31" : "+v"(x[i]));
            if (OP == CMP) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(x[i]), "v"(c) : "vcc");
            if (OP == MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(x[i]) : "v"(c));
            if (OP == DPP_MOV) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x[i]));
            if (OP == BCNT) asm volatile("v_bcnt_u32_b32 %0, %0, 0" : "+v"(x[i]));
            if (OP == ROTR) asm volatile("v_alignbit_b32 %0, %0, %0, 7" : "+v"(x[i]));
            if (OP == MIN_U32) asm volatile("v_min_u32 %0, %1, %0" : "+v"(x[i]) : "v"(c));
            if (OP == ADD3) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(c), "s"(seed));
            if (OP == PERM) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(c), "s"(seed));
            if (OP == MAD_U32_U24) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x[i]) : "v"(c), "s"(seed));
            if (OP == LSHL_B64) asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(y[i]));
            if (OP == ADD_CO) asm volatile("v_add_co_u32 %0, vcc, %1, %0" : "+v"(x[i]) : "v"(c) : "vcc");
            if (OP == ADD_F64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(y[i]) : "v"(y[(i + 1) % U]));
            if (OP == MUL_F64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(y[i]) : "v"(y[(i + 1) % U]));
            if (OP == CVT_F64_I32) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(y[i]) : "v"(x[i]));
            if (OP == OR3) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(c), "s"(seed));
            if (OP == LSHL_ADD) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(x[i]) : "v"(c));
            if (OP == XAD) asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(c), "s"(seed));
            if (OP == SUB) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(x[i]) : "v"(c));
            if (OP == OR) asm volatile("v_or_b32 %0, %1, %0" : "+v"(x[i]) : "v"(c));
            if (OP == FFBL) asm volatile("v_ffbl_b32 %0, %0" : "+v"(x[i]));
        }
    }
    const uint64_t t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < U; ++i) acc ^= x[i] ^ (uint32_t)y[i] ^ (uint32_t)(y[i] >> 32);
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) { uint64_t *c2 = cycles + 4 * (blockIdx.x * 16 + (threadIdx.x >> 6)); c2[0] = t0; c2[1] = t1; c2[2] = r0; c2[3] = r1; }
}

// the product's Philox4x32-10 block (20 v_mad_u64_u32 + 20 v_bitop3_b32 + key schedule on the scalar unit), four independent
// chains per lane: shader cycles per BLOCK per SIMD
constexpr int PR = 512;
__global__ __launch_bounds__(1024) void bench_philox(uint32_t *out, uint64_t *cycles, uint32_t seed)
{
    uint4 a = make_uint4(threadIdx.x, 1, 2, 3), b = make_uint4(threadIdx.x, 5, 6, 7), c = make_uint4(threadIdx.x, 9, 10, 11), d = make_uint4(threadIdx.x, 13, 14, 15);
    const uint64_t r0 = wall_clock64(), t0 = __builtin_readcyclecounter();
    for (int r = 0; r < PR; ++r) {
        a = pomdp::philox4x32_10(a.x, a.y, a.z, a.w, seed, (uint32_t)r);
        b = pomdp::philox4x32_10(b.x, b.y, b.z, b.w, seed, (uint32_t)r);
        c = pomdp::philox4x32_10(c.x, c.y, c.z, c.w, seed, (uint32_t)r);
        d = pomdp::philox4x32_10(d.x, d.y, d.z, d.w, seed, (uint32_t)r);
    }
    const uint64_t t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a.x ^ b.y ^ c.z ^ d.w;
    if ((threadIdx.x & 63) == 0) { uint64_t *c2 = cycles + 4 * (blockIdx.x * 16 + (threadIdx.x >> 6)); c2[0] = t0; c2[1] = t1; c2[2] = r0; c2[3] = r1; }
}

typedef void (*kern_t)(uint32_t *, uint64_t *, uint32_t);
template <int OP> struct Tab { static void fill(kern_t *t) { t[OP] = bench<OP>; Tab<OP + 1>::fill(t); } };
template <> struct Tab<N_OPS> { static void fill(kern_t *) {} };

constexpr int GRID = 32;   // workgroups per launch: far fewer than CUs, so each has a CU to itself

// median over the workgroups of (span in shader cycles, span in ns by the 100 MHz real-time counter)
static void spans(const std::vector<uint64_t> &h, int waves, double &cyc, double &ns)
{
    std::vector<double> c, r;
    for (int g = 0; g < GRID; ++g) {
        uint64_t t0 = ~0ull, t1 = 0, r0 = ~0ull, r1 = 0;
        for (int w = 0; w < waves; ++w) {
            const uint64_t *e = &h[4 * ((size_t)g * 16 + w)];
            t0 = std::min(t0, e[0]); t1 = std::max(t1, e[1]); r0 = std::min(r0, e[2]); r1 = std::max(r1, e[3]);
        }
        c.push_back((double)(t1 - t0)); r.push_back((double)(r1 - r0) * 10.0);
    }
    std::sort(c.begin(), c.end()); std::sort(r.begin(), r.end());
    cyc = c[c.size() / 2]; ns = r[r.size() / 2];
}

int main()
{
    kern_t k[N_OPS];
    Tab<0>::fill(k);
    uint32_t *out; uint64_t *cyc;
    CHECK(hipMalloc(&out, (size_t)GRID * 1024 * 4));
    CHECK(hipMalloc(&cyc, (size_t)GRID * 16 * 4 * 8));
    std::vector<uint64_t> h((size_t)GRID * 16 * 4);
    printf("{\"U\": %d, \"R\": %d, \"unit\": \"cycles = shader cycles (s_memtime) per wave64 instruction per SIMD with W waves per SIMD "
           "(one workgroup of 256 W threads alone on its CU, median of %d workgroups); ghz = the same span by s_memrealtime\", \"ops\": {\n", U, R, GRID);
    for (int op = 0; op <= N_OPS; ++op) {
        if (op == N_OPS) printf("},\n \"philox4x32_10_block\": {");
        else printf("  \"%s\": {", OP_NAME[op]);
        for (int wi = 0, W = 1; W <= 4; W *= 2, ++wi) {
            for (int rep = 0; rep < 3; ++rep) {
                if (op == N_OPS) hipLaunchKernelGGL(bench_philox, dim3(GRID), dim3(256 * W), 0, 0, out, cyc, 12345u + rep);
                else hipLaunchKernelGGL(k[op], dim3(GRID), dim3(256 * W), 0, 0, out, cyc, 12345u + rep);
            }
            CHECK(hipDeviceSynchronize());
            CHECK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
            double c, ns;
            spans(h, 4 * W, c, ns);
            const double n = op == N_OPS ? (double)W * PR * 4 : (double)W * R * U;
            printf("%s\"W%d\": {\"cycles\": %.3f, \"ns\": %.4f, \"ghz\": %.3f}", wi ? ", " : "", W, c / n, ns / n, c / ns);
        }
        printf("}%s\n", op + 1 < N_OPS ? "," : "");
    }
    printf("}\n");
    return 0;
}
