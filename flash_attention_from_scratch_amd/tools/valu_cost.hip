// valu_cost.hip -- issue cost (cycles per wave-instruction, one wave per SIMD, 64 independent instructions
// between two s_memtime stamps) of the vector instructions the softmax stream is made of, alone and beside
// MFMAs.  Build: hipcc --offload-arch=gfx950 -O3 valu_cost.hip -o valu_cost
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
// 64 independent instructions: 8 destination registers round robin, sources never written in the block
#define BLOCK(name, ins)                                                                                         \
    __global__ void __launch_bounds__(256, 1) k_##name(unsigned long long *out, float seed) {                    \
        float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, d0, d1, d2, d3, d4, d5, d6, d7;               \
        d0 = d1 = d2 = d3 = d4 = d5 = d6 = d7 = seed;                                                              \
        unsigned long long t0, t1;                                                                                \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));        \
        asm volatile(REP8(ins) : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7)   \
                     : "v"(a0), "v"(a1), "v"(a2), "v"(a3));                                                        \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));                                          \
        if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;                                                 \
        if (d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7 == 12345.0f) out[1] = 1;                                         \
    }
#define I8(op, tail) op " %0, " tail "\n\t" op " %1, " tail "\n\t" op " %2, " tail "\n\t" op " %3, " tail "\n\t" op " %4, " tail "\n\t" op " %5, " tail "\n\t" op " %6, " tail "\n\t" op " %7, " tail "\n\t"
BLOCK(fma, I8("v_fma_f32", "%8, %9, %10"))
BLOCK(add, I8("v_add_f32", "%8, %9"))
BLOCK(exp, I8("v_exp_f32", "%8"))
BLOCK(exp_legacy, I8("v_exp_legacy_f32", "%8"))
BLOCK(exp_f16, I8("v_exp_f16", "%8"))
BLOCK(log, I8("v_log_f32", "%8"))
BLOCK(rcp, I8("v_rcp_f32", "%8"))
BLOCK(cvt_pk, I8("v_cvt_pk_bf16_f32", "%8, %9"))
BLOCK(max3, I8("v_max3_f32", "%8, %9, %10"))
BLOCK(ldexp, I8("v_ldexp_f32", "%8, %9"))
BLOCK(dot2, I8("v_dot2_f32_bf16", "%8, %9, %10"))
BLOCK(dot2c, I8("v_dot2c_f32_bf16", "%8, %9"))
BLOCK(exp_then_fma, "v_exp_f32 %0, %8\n\tv_fma_f32 %1, %8, %9, %10\n\tv_exp_f32 %2, %9\n\tv_fma_f32 %3, %8, %9, %10\n\tv_exp_f32 %4, %10\n\tv_fma_f32 %5, %8, %9, %10\n\tv_exp_f32 %6, %11\n\tv_fma_f32 %7, %8, %9, %10\n\t")

// beside MFMAs: 8 x { v_mfma ; N fillers } -- cycles per MFMA gap
#define MBLOCK(name, fill)                                                                                       \
    __global__ void __launch_bounds__(256, 1) m_##name(unsigned long long *out, float seed, const bf16x8 *ab) {   \
        float a0 = seed, a1 = seed + 1, a2 = seed + 2, d0, d1, d2, d3, d4, d5, d6, d7;                              \
        d0 = d1 = d2 = d3 = d4 = d5 = d6 = d7 = seed;                                                              \
        bf16x8 a = ab[threadIdx.x & 63], b = ab[64 + (threadIdx.x & 63)];                                          \
        f32x16 c0 = {}, c1 = {};                                                                                  \
        unsigned long long t0, t1;                                                                                \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));        \
        asm volatile(REP8("v_mfma_f32_32x32x16_bf16 %8, %10, %11, %8\n\t" fill "v_mfma_f32_32x32x16_bf16 %9, %10, %11, %9\n\t" fill) \
                     : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7), "+v"(c0), "+v"(c1) \
                     : "v"(a), "v"(b), "v"(a0), "v"(a1), "v"(a2));                                                 \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));                                          \
        if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;                                                 \
        if (d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7 + c0[0] + c1[1] == 12345.0f) out[1] = 1;                         \
    }
MBLOCK(bare, "")
MBLOCK(fma4, "v_fma_f32 %0, %12, %13, %14\n\tv_fma_f32 %1, %12, %13, %14\n\tv_fma_f32 %2, %12, %13, %14\n\tv_fma_f32 %3, %12, %13, %14\n\t")
MBLOCK(fma6, "v_fma_f32 %0, %12, %13, %14\n\tv_fma_f32 %1, %12, %13, %14\n\tv_fma_f32 %2, %12, %13, %14\n\tv_fma_f32 %3, %12, %13, %14\n\tv_fma_f32 %4, %12, %13, %14\n\tv_fma_f32 %5, %12, %13, %14\n\t")
MBLOCK(exp1, "v_exp_f32 %0, %12\n\t")
MBLOCK(exp2, "v_exp_f32 %0, %12\n\tv_exp_f32 %1, %13\n\t")
MBLOCK(exp1_fma3, "v_exp_f32 %0, %12\n\tv_fma_f32 %1, %12, %13, %14\n\tv_fma_f32 %2, %12, %13, %14\n\tv_fma_f32 %3, %12, %13, %14\n\t")
MBLOCK(exp2_fma5, "v_exp_f32 %0, %12\n\tv_exp_f32 %1, %13\n\tv_fma_f32 %2, %12, %13, %14\n\tv_fma_f32 %3, %12, %13, %14\n\tv_fma_f32 %4, %12, %13, %14\n\tv_fma_f32 %5, %12, %13, %14\n\tv_fma_f32 %6, %12, %13, %14\n\t")
MBLOCK(expf16_2_fma5, "v_exp_f16 %0, %12\n\tv_exp_f16 %1, %13\n\tv_fma_f32 %2, %12, %13, %14\n\tv_fma_f32 %3, %12, %13, %14\n\tv_fma_f32 %4, %12, %13, %14\n\tv_fma_f32 %5, %12, %13, %14\n\tv_fma_f32 %6, %12, %13, %14\n\t")

#define MABLOCK(name, fill)                                                                                       \
    __global__ void __launch_bounds__(256, 1) ma_##name(unsigned long long *out, float seed, const bf16x8 *ab) {   \
        float a0 = seed, a1 = seed + 1, a2 = seed + 2, d0, d1, d2, d3, d4, d5, d6, d7;                              \
        d0 = d1 = d2 = d3 = d4 = d5 = d6 = d7 = seed;                                                              \
        bf16x8 a = ab[threadIdx.x & 63], b = ab[64 + (threadIdx.x & 63)];                                          \
        f32x16 c0 = {}, c1 = {};                                                                                  \
        unsigned long long t0, t1;                                                                                \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));        \
        asm volatile(REP8("v_mfma_f32_32x32x16_bf16 %8, %10, %11, %8\n\t" fill "v_mfma_f32_32x32x16_bf16 %9, %10, %11, %9\n\t" fill) \
                     : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7), "+a"(c0), "+a"(c1) \
                     : "v"(a), "v"(b), "v"(a0), "v"(a1), "v"(a2));                                                 \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));                                          \
        if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;                                                 \
        if (d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7 + c0[0] + c1[1] == 12345.0f) out[1] = 1;                         \
    }
MABLOCK(bare, "")
MABLOCK(fma4, "v_fma_f32 %0, %12, %13, %14\n\tv_fma_f32 %1, %12, %13, %14\n\tv_fma_f32 %2, %12, %13, %14\n\tv_fma_f32 %3, %12, %13, %14\n\t")
MABLOCK(fma6, "v_fma_f32 %0, %12, %13, %14\n\tv_fma_f32 %1, %12, %13, %14\n\tv_fma_f32 %2, %12, %13, %14\n\tv_fma_f32 %3, %12, %13, %14\n\tv_fma_f32 %4, %12, %13, %14\n\tv_fma_f32 %5, %12, %13, %14\n\t")
MABLOCK(exp2_fma5, "v_exp_f32 %0, %12\n\tv_exp_f32 %1, %13\n\tv_fma_f32 %2, %12, %13, %14\n\tv_fma_f32 %3, %12, %13, %14\n\tv_fma_f32 %4, %12, %13, %14\n\tv_fma_f32 %5, %12, %13, %14\n\tv_fma_f32 %6, %12, %13, %14\n\t")

// row sums on the matrix pipe: v_mfma_f32_4x4x4_16b_bf16 with A = ones adds a lane's four packed values into its accumulator
#define M4BLOCK(name, fill)                                                                                      \
    __global__ void __launch_bounds__(256, 1) m_##name(unsigned long long *out, float seed, const bf16x8 *ab) {   \
        typedef float f32x4_ __attribute__((ext_vector_type(4)));                                                  \
        typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));                                               \
        float a0 = seed, a1 = seed + 1, a2 = seed + 2, d0, d1, d2, d3, d4, d5, d6, d7;                              \
        d0 = d1 = d2 = d3 = d4 = d5 = d6 = d7 = seed;                                                              \
        bf16x8 a = ab[threadIdx.x & 63], b = ab[64 + (threadIdx.x & 63)];                                          \
        f32x16 c0 = {}, c1 = {};                                                                                  \
        f32x4_ s0 = {}, s1 = {};                                                                                  \
        u32x2_ ones = {0x3f803f80u, 0x3f803f80u}, pk = {0x3e003e00u, 0x3d003d00u};                                 \
        unsigned long long t0, t1;                                                                                \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));        \
        asm volatile(REP8("v_mfma_f32_32x32x16_bf16 %8, %12, %13, %8\n\t" fill "v_mfma_f32_32x32x16_bf16 %9, %12, %13, %9\n\t" fill) \
                     : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7), "+v"(c0), "+v"(c1), "+v"(s0), "+v"(s1) \
                     : "v"(a), "v"(b), "v"(a0), "v"(a1), "v"(a2), "v"(ones), "v"(pk));                             \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));                                          \
        if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;                                                 \
        if (d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7 + c0[0] + c1[1] + s0[0] + s1[1] == 12345.0f) out[1] = 1;        \
    }
// operands: %8 %9 big accumulators, %10 %11 = s0 s1, %12 %13 = a b, %14 %15 %16 = a0 a1 a2, %17 ones, %18 pk
#undef MFMA_BIG
M4BLOCK(sum4_only, "v_mfma_f32_4x4x4_16b_bf16 %10, %17, %18, %10\n\t")
M4BLOCK(sum4_x2, "v_mfma_f32_4x4x4_16b_bf16 %10, %17, %18, %10\n\tv_mfma_f32_4x4x4_16b_bf16 %11, %17, %18, %11\n\t")
M4BLOCK(unit_adds, "v_exp_f32 %0, %14\n\tv_exp_f32 %1, %15\n\tv_fma_f32 %2, %14, %15, %16\n\tv_fma_f32 %3, %14, %15, %16\n\tv_add_f32 %4, %14, %15\n\tv_add_f32 %5, %14, %16\n\tv_cvt_pk_bf16_f32 %6, %14, %15\n\t")
M4BLOCK(unit_noadds, "v_exp_f32 %0, %14\n\tv_exp_f32 %1, %15\n\tv_fma_f32 %2, %14, %15, %16\n\tv_fma_f32 %3, %14, %15, %16\n\tv_cvt_pk_bf16_f32 %6, %14, %15\n\t")
M4BLOCK(unit_sum4, "v_exp_f32 %0, %14\n\tv_exp_f32 %1, %15\n\tv_fma_f32 %2, %14, %15, %16\n\tv_fma_f32 %3, %14, %15, %16\n\tv_cvt_pk_bf16_f32 %6, %14, %15\n\tv_mfma_f32_4x4x4_16b_bf16 %10, %17, %18, %10\n\t")

// row sums of the ROUNDED P by one v_dot2_f32_bf16 with a packed-ones operand per pair (reads the previous unit's pack)
M4BLOCK(unit_dot2, "v_exp_f32 %0, %14\n\tv_exp_f32 %1, %15\n\tv_fma_f32 %2, %14, %15, %16\n\tv_fma_f32 %3, %14, %15, %16\n\tv_dot2_f32_bf16 %4, %6, %16, %4\n\tv_cvt_pk_bf16_f32 %6, %14, %15\n\t")
M4BLOCK(unit_dot2c, "v_exp_f32 %0, %14\n\tv_exp_f32 %1, %15\n\tv_fma_f32 %2, %14, %15, %16\n\tv_fma_f32 %3, %14, %15, %16\n\tv_dot2c_f32_bf16 %4, %6, %16\n\tv_cvt_pk_bf16_f32 %6, %14, %15\n\t")

// the unit with its two row-sum adds as ONE v_pk_add_f32 on register pairs (physical registers named in the asm)
__global__ void __launch_bounds__(256, 1) m_unit_pkadd(unsigned long long *out, float seed, const bf16x8 *ab) {
    float a0 = seed, a1 = seed + 1, a2 = seed + 2;
    bf16x8 a = ab[threadIdx.x & 63], b = ab[64 + (threadIdx.x & 63)];
    f32x16 c0 = {}, c1 = {};
    unsigned long long t0, t1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
#define UNIT_PK "v_exp_f32 v200, %4\n\tv_exp_f32 v201, %5\n\tv_fma_f32 v204, %4, %5, %6\n\tv_fma_f32 v205, %4, %5, %6\n\tv_pk_add_f32 v[202:203], v[202:203], v[200:201]\n\tv_cvt_pk_bf16_f32 v206, v200, v201\n\t"
    asm volatile(REP8("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\t" UNIT_PK "v_mfma_f32_32x32x16_bf16 %1, %2, %3, %1\n\t" UNIT_PK)
                 : "+v"(c0), "+v"(c1) : "v"(a), "v"(b), "v"(a0), "v"(a1), "v"(a2)
                 : "v200", "v201", "v202", "v203", "v204", "v205", "v206");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    if (c0[0] + c1[1] == 12345.0f) out[1] = 1;
}

int main() {
    unsigned long long *out; CHECK(hipMalloc(&out, 64));
    bf16x8 *ab; CHECK(hipMalloc(&ab, 128 * 16)); CHECK(hipMemset(ab, 0x3c, 128 * 16));
    unsigned long long h[2];
#define RUN(name, n) do { for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_##name, dim3(256), dim3(256), 0, 0, out, 1.5f); CHECK(hipDeviceSynchronize()); \
        CHECK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost)); printf("%-28s %6.2f cycles per instruction (%llu / %d)\n", #name, (double)h[0] / (n), h[0], n); } while (0)
    RUN(fma, 64); RUN(add, 64); RUN(exp, 64); RUN(exp_legacy, 64); RUN(exp_f16, 64); RUN(log, 64); RUN(rcp, 64); RUN(cvt_pk, 64); RUN(max3, 64); RUN(ldexp, 64); RUN(dot2, 64); RUN(dot2c, 64);
    RUN(exp_then_fma, 64);
#define MRUN(name) do { for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(m_##name, dim3(256), dim3(256), 0, 0, out, 1.5f, ab); CHECK(hipDeviceSynchronize()); \
        CHECK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost)); printf("MFMA + %-21s %6.2f cycles per MFMA gap (%llu / 16)\n", #name, (double)h[0] / 16, h[0]); } while (0)
    MRUN(sum4_only); MRUN(sum4_x2); MRUN(unit_adds); MRUN(unit_noadds); MRUN(unit_sum4); MRUN(unit_pkadd); MRUN(unit_dot2); MRUN(unit_dot2c);
    MRUN(bare); MRUN(fma4); MRUN(fma6); MRUN(exp1); MRUN(exp2); MRUN(exp1_fma3); MRUN(exp2_fma5); MRUN(expf16_2_fma5);
#define MARUN(name) do { for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(ma_##name, dim3(256), dim3(256), 0, 0, out, 1.5f, ab); CHECK(hipDeviceSynchronize()); \
        CHECK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost)); printf("MFMA(acc in AGPR) + %-10s %6.2f cycles per MFMA gap (%llu / 16)\n", #name, (double)h[0] / 16, h[0]); } while (0)
    MARUN(bare); MARUN(fma4); MARUN(fma6); MARUN(exp2_fma5);
    return 0;
}
