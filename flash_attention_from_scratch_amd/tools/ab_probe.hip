// ab_probe.hip -- can two SPECIALISED waves per SIMD keep the matrix pipe full?
// Planning probe for a producer/consumer split of the attention loop: wave A of a
// SIMD runs {1 MFMA + NA softmax-like VALU fillers} per step (the QK^T + softmax role), wave B
// {1 MFMA + NB LDS reads} (the P.V role).  Registers only, random operands (the chip is power
// limited: constant data clocks higher), no global traffic.  Reports TFLOP/s of:
//   A alone (one wave per SIMD), B alone, A and B together (two waves per SIMD).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

template <int NA>
__device__ __forceinline__ void role_a(int iters, float *out, unsigned seed) {
    f32x16 acc[4];
    bf16x8 a[2], b[2];
    float s[16], rs0 = 0.f, rs1 = 0.f, vm = 0.f;
    unsigned x = seed * 2654435761u + threadIdx.x * 40503u;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            x = x * 1664525u + 1013904223u; a[i][j] = (__bf16)(((int)(x >> 16) % 2001 - 1000) * 1e-3f);
            x = x * 1664525u + 1013904223u; b[i][j] = (__bf16)(((int)(x >> 16) % 2001 - 1000) * 1e-3f);
        }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { x = x * 1664525u + 1013904223u; s[r] = ((int)(x >> 16) % 2001 - 1000) * 4e-3f; }
    unsigned pk = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[u & 3]) : "v"(a[u & 1]), "v"(b[(u >> 1) & 1]));
            if (NA == 6) {  // variant: row sum by v_dot2 on the packed pair (1 instruction instead of 2 adds)
                float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[2 * u], 0.1275f, -1.0f));
                float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[2 * u + 1], 0.1275f, -1.0f));
                typedef __bf16 pair_t __attribute__((ext_vector_type(2)));
                pair_t pr; pr[0] = (__bf16)p0; pr[1] = (__bf16)p1;
                unsigned pu = __builtin_bit_cast(unsigned, pr);
                asm volatile("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(rs0) : "v"(pu), "v"(0x3f803f80u));
                pk ^= pu;
                asm volatile("" : "+v"(rs0), "+v"(pk));
            }
            if (NA >= 7) {  // one softmax unit: 2 fma, 2 exp2, 2 add, 1 pack
                float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[2 * u], 0.1275f, -1.0f));
                float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[2 * u + 1], 0.1275f, -1.0f));
                rs0 += p0; rs1 += p1;
                typedef __bf16 pair_t __attribute__((ext_vector_type(2)));
                pair_t pr; pr[0] = (__bf16)p0; pr[1] = (__bf16)p1;
                pk ^= __builtin_bit_cast(unsigned, pr);
                asm volatile("" : "+v"(rs0), "+v"(rs1), "+v"(pk));
            }
            if (NA >= 9) {  // + a row-max unit (2 x max3)
                float t;
                asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(t) : "v"(vm), "v"(s[(2 * u + 3) & 15]), "v"(s[(2 * u + 5) & 15]));
                asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(vm) : "v"(t), "v"(s[(2 * u + 7) & 15]), "v"(s[(2 * u + 9) & 15]));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float r = rs0 + rs1 + vm + __builtin_bit_cast(float, pk & 0x3f800000u);
#pragma unroll
    for (int i = 0; i < 4; ++i) r += acc[i][0] + acc[i][7];
    if (r == 12345.678f) out[threadIdx.x] = r;
}

// role K: the kernel's own mix per MFMA -- ~3.5 softmax VALU + 0.5 max3 + 0.75 LDS operand reads
// (+ one 1-KiB LDS-DMA piece per 8 MFMAs if DMA) -- to see which part is priced above a plain VALU
template <int LDSR, int DMA>
__device__ __forceinline__ void role_k(int iters, float *out, unsigned seed, char *lds, const char *gsrc) {
    f32x16 acc[4];
    bf16x8 a[2], b[2], ring[2];
    float s[16], rs0 = 0.f, rs1 = 0.f, vm = 0.f;
    unsigned x = seed * 2654435761u + threadIdx.x * 40503u, pk = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            x = x * 1664525u + 1013904223u; a[i][j] = (__bf16)(((int)(x >> 16) % 2001 - 1000) * 1e-3f);
            x = x * 1664525u + 1013904223u; b[i][j] = (__bf16)(((int)(x >> 16) % 2001 - 1000) * 1e-3f);
        }
    ring[0] = a[0]; ring[1] = a[1];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { x = x * 1664525u + 1013904223u; s[r] = ((int)(x >> 16) % 2001 - 1000) * 4e-3f; }
    const char *p = lds + (threadIdx.x & 63) * 16;
    const unsigned lds_base = (unsigned)(unsigned long long)(lds) + 8192 + (threadIdx.x >> 6) * 1024;
    const unsigned goff = (threadIdx.x & 63) * 16;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (LDSR && (u & 1) == 0) {  // operands in pairs, one counted wait per two MFMAs
                __builtin_amdgcn_s_waitcnt(0xC07F);
                asm volatile("" ::"v"(ring[0]), "v"(ring[1]));
                ring[0] = *(const bf16x8 *)(p + u * 512);
                if (LDSR >= 2) ring[1] = *(const bf16x8 *)(p + u * 512 + 4096);
            }
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[u & 3]) : "v"(a[u & 1]), "v"(b[(u >> 1) & 1]));
            if (DMA && u == 3)
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(goff), "s"(gsrc), "s"(__builtin_amdgcn_readfirstlane(lds_base)) : "memory");
            if (u & 1) {  // a softmax unit on every other MFMA, a max pair on the others
                float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[2 * u], 0.1275f, -1.0f));
                float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(s[2 * u + 1], 0.1275f, -1.0f));
                rs0 += p0; rs1 += p1;
                typedef __bf16 pair_t __attribute__((ext_vector_type(2)));
                pair_t pr; pr[0] = (__bf16)p0; pr[1] = (__bf16)p1;
                pk ^= __builtin_bit_cast(unsigned, pr);
                asm volatile("" : "+v"(rs0), "+v"(rs1), "+v"(pk));
            } else {
                float t;
                asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(t) : "v"(vm), "v"(s[(2 * u + 3) & 15]), "v"(s[(2 * u + 5) & 15]));
                vm = t;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (DMA && (it & 7) == 7) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float r = rs0 + rs1 + vm + __builtin_bit_cast(float, pk & 0x3f800000u) + (float)ring[0][0] + (float)ring[1][1];
#pragma unroll
    for (int i = 0; i < 4; ++i) r += acc[i][0] + acc[i][7];
    if (r == 12345.678f) out[threadIdx.x] = r;
}

template <int NB>
__device__ __forceinline__ void role_b(int iters, float *out, unsigned seed, char *lds) {
    f32x16 acc[8];
    bf16x8 a[2], b[2];
    unsigned x = seed * 2246822519u + threadIdx.x * 374761393u;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            x = x * 1664525u + 1013904223u; a[i][j] = (__bf16)(((int)(x >> 16) % 2001 - 1000) * 1e-3f);
            x = x * 1664525u + 1013904223u; b[i][j] = (__bf16)(((int)(x >> 16) % 2001 - 1000) * 1e-3f);
        }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const char *p = lds + (threadIdx.x & 63) * 16;
    bf16x8 ring[2] = {a[0], a[1]};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[u]) : "v"(a[u & 1]), "v"(b[(u >> 1) & 1]));
            if (NB >= 1) {  // operand read two steps ahead of its (pretended) use, as the kernel does
                asm volatile("" ::"v"(ring[u & 1]));
                ring[u & 1] = *(const bf16x8 *)(p + u * 1024);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][9];
    if (r == 12345.678f) out[threadIdx.x] = r;
}

template <int MODE, int NA, int NB>  // MODE 0: A only (256 threads), 1: B only, 2: A + B (512 threads), 3: kernel-like mix
__global__ void __launch_bounds__(MODE == 2 ? 512 : 256, 1) probe(int iters, float *out) {
    __shared__ char lds[16384];
    for (int i = threadIdx.x; i < 16384 / 4; i += blockDim.x) ((float *)lds)[i] = (float)(i * 37 % 101) * 0.01f;
    __syncthreads();
    const int wave = threadIdx.x >> 6;
    if (MODE == 3) role_k<NA, NB>(iters, out, blockIdx.x, lds, (const char *)out + 4096);
    else if (MODE == 0) role_a<NA>(iters, out, blockIdx.x);
    else if (MODE == 1) role_b<NB>(iters, out, blockIdx.x, lds);
    else { if (wave < 4) role_a<NA>(iters, out, blockIdx.x); else role_b<NB>(iters, out, blockIdx.x, lds); }
}

template <int MODE, int NA, int NB> void run(const char *name) {
    float *out; CHECK(hipMalloc(&out, 4096 + 65536)); CHECK(hipMemset(out, 0, 4096 + 65536));
    const int iters = 4000, blocks = 256 * 4;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((probe<MODE, NA, NB>), dim3(blocks), dim3(MODE == 2 ? 512 : 256), 0, 0, iters, out);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep >= 1 && ms < best) best = ms;
    }
    const double waves = (MODE == 2 ? 8.0 : 4.0) * blocks;
    const double mfma = waves * iters * 8.0;
    const double tf = mfma * 32768.0 * 2 / 2 / (best * 1e-3) / 1e12;   // 32*32*16*2 flops per MFMA
    // cycles per MFMA per SIMD at the nominal 2.4 GHz (the real clock is lower under load)
    const double simd_mfma = mfma / (256.0 * 4.0);
    printf("%-46s %8.3f ms  %7.1f TFLOP/s  %5.1f ns per MFMA per SIMD\n", name, best, tf, best * 1e6 / simd_mfma);
    CHECK(hipFree(out));
}

int main() {
    run<1, 0, 0>("B: bare MFMAs, one wave per SIMD");
    run<1, 0, 1>("B: MFMA + 1 ds_read_b128");
    run<0, 7, 0>("A: MFMA + softmax unit (7 VALU)");
    run<0, 6, 0>("A: MFMA + softmax unit with v_dot2 row sum (6 VALU)");
    run<0, 9, 0>("A: MFMA + softmax unit + 2 max3 (9 VALU)");
    run<2, 7, 1>("A (7 VALU) + B (1 LDS read), two waves per SIMD");
    run<2, 9, 1>("A (9 VALU) + B (1 LDS read), two waves per SIMD");
    run<2, 0, 0>("A bare + B bare, two waves per SIMD");
    run<3, 0, 0>("kernel-like VALU mix only (3.5 + 0.5 per MFMA)");
    run<3, 1, 0>("  + 0.5 LDS operand reads per MFMA, paired waits");
    run<3, 2, 0>("  + 1.0 LDS operand reads per MFMA, paired waits");
    run<3, 2, 1>("  + one LDS-DMA piece per 8 MFMAs");
    return 0;
}
