// mfma_peak.hip -- what the matrix pipe sustains on THIS box: a register-only loop of
// v_mfma_f32_32x32x16_bf16 (4 independent accumulators per wave, no memory traffic), on zero
// operands and on random operands.  The gap between the two is the chip's power management
// (clock follows the power budget), and the random-operand number -- not the 2.5 PF datasheet
// figure -- is the ceiling any bf16 MFMA kernel with real data can approach here.
// Build: hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

__global__ void __launch_bounds__(512) mfma_loop(const bf16x8 *ab, float *out, int iters) {
    const int lane = threadIdx.x & 63;
    bf16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        a[i] = ab[(i * 64 + lane)];
        b[i] = ab[((4 + i) * 64 + lane)];
    }
    f32x16 c[4] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(i + u) & 3], b[i], c[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += c[i][r];
    if (s == 12345.678f) out[threadIdx.x] = s;  // keep the chain alive
}

static double run(const bf16x8 *d_ab, float *d_out, int blocks, int threads, int iters) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    double best = 1e30, sum = 0;
    for (int rep = 0; rep < 8; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(threads), 0, 0, d_ab, d_out, iters);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep >= 2) { sum += ms; if (ms < best) best = ms; }
    }
    const double flop = (double)blocks * (threads / 64) * iters * 16.0 * 2.0 * 32 * 32 * 16;
    return flop / (sum / 6 * 1e-3) / 1e12;
}

int main() {
    std::vector<uint16_t> h(8 * 64 * 8);
    bf16x8 *d_ab; float *d_out;
    CHECK(hipMalloc(&d_ab, h.size() * 2)); CHECK(hipMalloc(&d_out, 4096));
    srand(3);
    const int iters = 4000;
    for (int mode = 0; mode < 3; ++mode) {
        for (size_t i = 0; i < h.size(); ++i) {
            float x = mode == 0 ? 0.0f : mode == 1 ? ((rand() & 0xffff) / 65536.0f - 0.5f) * 3.4f : 1.0f;
            uint32_t u; memcpy(&u, &x, 4); h[i] = (uint16_t)(u >> 16);
        }
        CHECK(hipMemcpy(d_ab, h.data(), h.size() * 2, hipMemcpyHostToDevice));
        const char *name = mode == 0 ? "zeros " : mode == 1 ? "random" : "ones  ";
        printf("operands %s : 1 wave/SIMD %7.1f TFLOP/s | 2 waves/SIMD %7.1f TFLOP/s\n", name,
               run(d_ab, d_out, 256 * 4, 256, iters), run(d_ab, d_out, 256 * 4, 512, iters / 2));
    }
    return 0;
}
