// occupancy_probe.hip -- what would two waves per SIMD buy the persistent kernel's instruction mix?  
// A synthetic visit loop with the per-MFMA filler mix of the hand-placed kernels -- per MFMA 3.5 vector instructions of softmax
// unit work {v_fma, v_exp, v_add, half a v_cvt_pk} and, per mix,
//     mix 2 (64 rows per wave: an LDS operand feeds two MFMAs)   0.75 LDS reads (ds_read_b128 / ds_read_b64_tr_b16) + 0.25 waits
//     mix 1 (32 rows per wave: every MFMA takes a fresh operand)  1.5  LDS reads                                     + 0.5  waits
// pinned gap by gap (sched_barrier(0) behind every MFMA + its fillers, as in fa_fwd_kernel64.hpp), accumulators and B operand in
// registers, operands from a conflict-free LDS image -- run with ONE wave per SIMD (256 threads, one workgroup per CU) and with
// TWO (two workgroups of 256 per CU, <= 256 registers per lane each).  Prints wave cycles per MFMA (s_memtime around the loop,
// mean over waves) and the SIMD's MFMA interval (= that / waves per SIMD): 32 is the matrix pipe's own rate.  No DMA, no
// barriers, no softmax dependencies on the MFMA results: an UPPER bound of what occupancy can hide, not a kernel.
//     hipcc -O3 --offload-arch=gfx950 -fno-slp-vectorize occupancy_probe.hip -o occupancy_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
struct s16x8 { s16x4 lo, hi; };
#define LDS_P(T) __attribute__((address_space(3))) T

template <int N, typename F> __device__ __forceinline__ void static_for(F &&f) {
    if constexpr (N > 0) { static_for<N - 1>(f); f(std::integral_constant<int, N - 1>{}); }
}

// MIX: 32-row Q tiles per wave whose MFMAs share an LDS operand (2: the 64-row kernel's stream, 1: the ring form's)
template <int MIX, int OCC>
__global__ void __launch_bounds__(256, OCC) probe(unsigned long long *out, const float *seed, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16384 / 4; i += 256) ((unsigned *)smem)[i] = 0x3c003c00u;
    __syncthreads();
    f32x16 acc[4] = {};
    bf16x8 q = __builtin_bit_cast(bf16x8, *(const float4 *)(seed + 4 * lane));
    float S[16], rs0 = 0.0f, rs1 = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) S[i] = seed[i];
    const float c = seed[20], m = seed[21];
    unsigned pk[4] = {};
    bf16x8 ring[2];
    unsigned a_k = lane * 16, a_v = (lane & 31) * 8 + (lane >> 5) * 4096;
    ring[0] = ring[1] = *(const bf16x8 *)(smem + a_k);
    unsigned long long t0, t1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
    for (int it = 0; it < iters; ++it) {
        asm volatile("" : "+v"(a_k), "+v"(a_v));   // (addresses "change": no hoisting of the reads)
        static_for<32>([&](auto g_) {
            constexpr int g = decltype(g_)::value;
            // operand for the MFMA(s) two steps ahead: a K fragment (ds_read_b128) in the first half of the visit, a V^T fragment
            // (two ds_read_b64_tr_b16) in the second -- every MFMA (MIX 1) or every second one (MIX 2)
            if constexpr (g % MIX == 0) {
                if constexpr ((g / MIX) % 2 == 0) __builtin_amdgcn_s_waitcnt(0xC07F | (1 << 8));   // counted: one read may fly
                if constexpr (g < 16) {
                    ring[(g / MIX) & 1] = *(const bf16x8 *)(smem + a_k + 1024 * (g & 7));   // (immediate offsets: no address arithmetic)
                } else {
                    s16x8 av;
                    av.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_P(s16x4) *)(smem + a_v + 512 * (g & 7)));
                    av.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_P(s16x4) *)(smem + a_v + 512 * (g & 7) + 256));
                    ring[(g / MIX) & 1] = __builtin_bit_cast(bf16x8, av);
                }
            }
            acc[g & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[((g / MIX) + 1) & 1], q, acc[g & 3], 0, 0, 0);
            // one softmax unit per two MFMAs: {2 fma, 2 exp2, 2 add, 1 pack} = 3.5 vector instructions per MFMA
            if constexpr (g % 2 == 0) {
                constexpr int e = (g / 2) % 8;
                asm volatile("" : "+v"(S[2 * e]), "+v"(S[2 * e + 1]));
                float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(S[2 * e], c, m));
                float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(S[2 * e + 1], c, m));
                rs0 += p0;
                rs1 += p1;
                asm volatile("" : "+v"(rs0), "+v"(rs1));
                typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
                bf16x2 h = {(__bf16)p0, (__bf16)p1};
                pk[e & 3] = __builtin_bit_cast(unsigned, h);
                asm volatile("" : "+v"(pk[e & 3]));
            } else {
                // (the other half of the unit's instructions ride here in the real stream; the probe keeps them in the even gaps and
                // leaves the odd ones to the MFMA and the operand reads: the per-MFMA totals are what is compared)
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
    float sink = rs0 + rs1 + acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + __builtin_bit_cast(float, pk[0] ^ pk[1] ^ pk[2] ^ pk[3]);
    if (lane == 0) out[(blockIdx.x * 4 + (threadIdx.x >> 6))] = t1 - t0;
    if (sink == 12345.678f) out[0] = 0;
}

template <int MIX, int OCC> static void run(const char *name, unsigned long long *out, const float *seed) {
    const int iters = 512, grid = 256 * OCC;
    auto kern = probe<MIX, OCC>;
    const int lds = 163840 / OCC;   // all of a CU's LDS between OCC workgroups: exactly OCC of them are resident per CU
    CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, out, seed, iters);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, out, seed, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long *h = (unsigned long long *)malloc(grid * 4 * 8);
    CHECK(hipMemcpy(h, out, grid * 4 * 8, hipMemcpyDeviceToHost));
    double sum = 0; for (int i = 0; i < grid * 4; ++i) sum += (double)h[i];
    const double per_wave = sum / (grid * 4) / (iters * 32.0);
    const double tflops = 2.0 * 32 * 32 * 16 * (double)grid * 4 * iters * 32 / (ms * 1e-3) / 1e12;
    int occ = 0; CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 256, lds));
    printf("%-58s wave cycles per MFMA %6.2f   SIMD's MFMA interval %6.2f   %7.1f TFLOP/s-equivalent   (resident workgroups per CU: %d)\n",
           name, per_wave, per_wave / OCC, tflops, occ);
    free(h);
}

int main() {
    unsigned long long *out; CHECK(hipMalloc(&out, 512 * 4 * 8));
    float hs[256 + 32]; for (int i = 0; i < 288; ++i) hs[i] = 0.01f * (i % 37) - 0.2f;
    hs[20] = 0.1275f; hs[21] = -0.3f;
    float *seed; CHECK(hipMalloc(&seed, sizeof(hs))); CHECK(hipMemcpy(seed, hs, sizeof(hs), hipMemcpyHostToDevice));
    for (int rep = 0; rep < 2; ++rep) {
        run<2, 1>("mix 2 (64 rows per wave), one wave per SIMD", out, seed);
        run<2, 2>("mix 2 (64 rows per wave), two waves per SIMD", out, seed);
        run<1, 1>("mix 1 (32 rows per wave: the ring form), one wave per SIMD", out, seed);
        run<1, 2>("mix 1 (32 rows per wave: the ring form), two waves per SIMD", out, seed);
    }
    return 0;
}
