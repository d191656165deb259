// segment_timer.hip -- per-segment s_memtime stamps of one workgroup of the forward
// kernel (debug build with -DFA_TRACE; the shipped library carries no trace code).
// Prints, per wave, the mean cycles between consecutive stamps over the steady-state
// visits.  Usage: segment_timer <variant: plain|pingpong> [seq_len]
#define FA_TRACE 1
#include "../csrc/fa_fwd_kernel.hpp"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)

template <bool PIPE>
int run(int S) {
    const int B = 4, H = 16, D = 128;
    const size_t n = (size_t)B * S * H * D;
    std::vector<uint16_t> h(n);
    uint16_t *q, *k, *v, *o;
    unsigned long long *tr;
    CHECK(hipMalloc(&q, n * 2)); CHECK(hipMalloc(&k, n * 2)); CHECK(hipMalloc(&v, n * 2)); CHECK(hipMalloc(&o, n * 2));
    CHECK(hipMalloc(&tr, 8 * 64 * 8 * 8));
    CHECK(hipMemset(tr, 0, 8 * 64 * 8 * 8));
    srand(1);
    for (int t = 0; t < 3; ++t) {
        for (size_t i = 0; i < n; ++i) {
            float x = ((rand() & 0xffff) / 65536.0f - 0.5f) * 3.4f;  // ~unit variance
            uint32_t u; memcpy(&u, &x, 4); h[i] = (uint16_t)(u >> 16);
        }
        CHECK(hipMemcpy(t == 0 ? q : t == 1 ? k : v, h.data(), n * 2, hipMemcpyHostToDevice));
    }
    fa::KernelArgs a;
    a.q = q; a.k = k; a.v = v; a.o = o;
    a.batch_stride = (int64_t)S * H * D; a.seq_stride = H * D; a.head_stride = D;
    a.seq_len = S; a.n_heads = H; a.n_bh = B * H; a.n_q_blocks = S / 256; a.n_kv_blocks = S / 64;
    a.trace = tr; a.trace_block = 700;
    auto kern = fa::fa_fwd_kernel<15, 1, 8, 64, true, true, true, PIPE, true, false, 128>;
    CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(kern, dim3(a.n_bh * a.n_q_blocks), dim3(512), 65536, 0, a);
        CHECK(hipDeviceSynchronize());
    }
    std::vector<unsigned long long> t(8 * 64 * 8);
    CHECK(hipMemcpy(t.data(), tr, t.size() * 8, hipMemcpyDeviceToHost));
    const int nv = a.n_kv_blocks < 64 ? a.n_kv_blocks : 64;
    printf("variant %s S=%d visits=%d  (cycles; stamps: see fa_fwd_kernel.hpp FA_STAMP ids)\n", PIPE ? "pingpong" : "plain", S, nv);
    for (int w = 0; w < 8; ++w) {
        double seg[8] = {0}; double per = 0; int cnt = 0;
        for (int j = 8; j + 1 < nv - 2; ++j) {
            const unsigned long long *r = &t[(w * 64 + j) * 8], *rn = &t[(w * 64 + j + 1) * 8];
            // order of stamps in time for this wave
            int order_pp0[5] = {0, 1, 2, 3, 3}, order_pp1[5] = {0, 1, 2, 3, 3}, order_plain[5] = {0, 1, 2, 3, 4};
            const int *ord = PIPE ? (w < 4 ? order_pp0 : order_pp1) : order_plain;
            const int ns = 5;
            for (int s = 0; s + 1 < ns; ++s) seg[s] += (double)(r[ord[s + 1]] - r[ord[s]]);
            seg[ns - 1] += (double)(rn[0] - r[ord[ns - 1]]);
            per += (double)(rn[0] - r[0]);
            ++cnt;
        }
        printf(" wave %d: period %7.0f | segs", w, per / cnt);
        for (int s = 0; s < 5; ++s) printf(" %6.0f", seg[s] / cnt);
        const unsigned hw = (unsigned)t[(w * 64 + 63) * 8 + 7];
        printf("   | t0[8]=%lld hw_id=%08x wave_slot=%u simd=%u cu=%u\n", (long long)(t[(w * 64 + 8) * 8] - t[(0 * 64 + 8) * 8]), hw, hw & 15, (hw >> 4) & 3, (hw >> 8) & 15);
    }
    if (PIPE) printf(" segs: wait+barrier | dma issue + m/alpha/rescale | qk(it+1)+exp(it)+pv(it) | loop | -\n");
    else printf(" segs: wait+barrier | issue+qk | softmax+rescale | pv | loop\n");
    return 0;
}

int main(int argc, char **argv) {
    const int S = argc > 2 ? atoi(argv[2]) : 4096;
    if (argc > 1 && !strcmp(argv[1], "plain")) return run<false>(S);
    return run<true>(S);
}
