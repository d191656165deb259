// ablate.hip -- cost attribution by ablation (guide 5.4 rule 17 style): the same kernel
// with one cost removed at a time, timed with hipEvents on the headline shape.
#include "../csrc/fa_fwd_kernel.hpp"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

static uint16_t *q, *k, *v, *o;
static const int B = 4, H = 16, D = 128, S = 4096;

template <int NW, int BC, bool PIPE, int ABL, bool DMA = true, int QT = 1>
void run(const char *name) {
    fa::KernelArgs a;
    a.q = q; a.k = k; a.v = v; a.o = o;
    a.batch_stride = (int64_t)S * H * D; a.seq_stride = H * D; a.head_stride = D;
    a.seq_len = S; a.n_heads = H; a.n_bh = B * H; a.n_q_blocks = S / (32 * QT * NW); a.n_kv_blocks = S / BC;
    auto kern = fa::fa_fwd_kernel<15, QT, NW, BC, true, true, true, PIPE, DMA, false, 128, ABL>;
    const int lds = (4 * BC * 256 > 32 * QT * NW * 256) ? 4 * BC * 256 : 32 * QT * NW * 256;
    CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e9, sum = 0;
    for (int rep = 0; rep < 12; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(a.n_bh * a.n_q_blocks), dim3(NW * 64), lds, 0, a);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep >= 2) { sum += ms; if (ms < best) best = ms; }
    }
    const double tf = 4.0 * B * H * (double)S * S * D / (sum / 10 * 1e-3) / 1e12;
    printf("%-44s NW=%d BC=%3d pipe=%d dma=%d abl=%2d : mean %.4f ms  min %.4f ms  %7.1f TF\n", name, NW, BC, PIPE, (int)DMA, ABL, sum / 10, best, tf);
}

int main() {
    const size_t n = (size_t)B * S * H * D;
    std::vector<uint16_t> h(n);
    CHECK(hipMalloc(&q, n * 2)); CHECK(hipMalloc(&k, n * 2)); CHECK(hipMalloc(&v, n * 2)); CHECK(hipMalloc(&o, n * 2));
    srand(1);
    for (int t = 0; t < 3; ++t) {
        for (size_t i = 0; i < n; ++i) {
            float x = ((rand() & 0xffff) / 65536.0f - 0.5f) * 3.4f;
            uint32_t u; memcpy(&u, &x, 4); h[i] = (uint16_t)(u >> 16);
        }
        CHECK(hipMemcpy(t == 0 ? q : t == 1 ? k : v, h.data(), n * 2, hipMemcpyHostToDevice));
    }
    run<4, 64, false, 0, true, 2>("QT2 plain (warm-up)");
    run<4, 64, false, 0, true, 2>("QT2 plain BC64");
    run<4, 64, false, 1, true, 2>("QT2 plain BC64 no-exp");
    run<4, 64, false, 2, true, 2>("QT2 plain BC64 no-softmax");
    run<4, 64, false, 4, true, 2>("QT2 plain BC64 no-LDS-reads");
    run<4, 64, false, 6, true, 2>("QT2 plain BC64 no-softmax no-LDS");
    run<4, 64, false, 8, true, 2>("QT2 plain BC64 no-barrier");
    run<4, 64, false, 30, true, 2>("QT2 plain BC64 MFMA only");
    run<8, 128, false, 0>("QT1 plain NW8 BC128");
    return 0;
}
