// check_qt1.hip -- the one-Q-tile-per-wave form (QTP = 1) against the 64-rows-per-wave lazy kernel on the same inputs:
// the two run the same arithmetic per 32-row tile, so their outputs must agree bit for bit.  Prints, per shape, how many
// rows differ and where the first ones are.  check_qt1 [S B H]...
#include "../csrc/fa_fwd_kernel64.hpp"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)
static float bf(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
int main(int argc, char **argv) {
    const int shapes[][3] = {{256, 1, 8}, {512, 1, 8}, {512, 2, 16}, {1024, 2, 16}, {4096, 1, 16}};
    auto k2 = fa::fa_fwd_kernel64<15, false, 0, false, false, false, 2>;
    auto k1 = fa::fa_fwd_kernel64<15, false, 0, false, false, false, 1>;
    CHECK(hipFuncSetAttribute((const void *)k2, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
    CHECK(hipFuncSetAttribute((const void *)k1, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
    int bad = 0;
    for (auto &sh : shapes) {
        const int S = sh[0], B = sh[1], H = sh[2], D = 128;
        const size_t n = (size_t)B * S * H * D;
        std::vector<uint16_t> h(n), o1(n), o2(n);
        uint16_t *q, *k, *v, *o;
        CHECK(hipMalloc(&q, n * 2)); CHECK(hipMalloc(&k, n * 2)); CHECK(hipMalloc(&v, n * 2)); CHECK(hipMalloc(&o, n * 2));
        srand(S + B);
        for (int t = 0; t < 3; ++t) {
            for (size_t i = 0; i < n; ++i) { float x = ((rand() & 0xffff) / 65536.0f - 0.5f) * 3.4f; uint32_t u; memcpy(&u, &x, 4); h[i] = (uint16_t)(u >> 16); }
            CHECK(hipMemcpy(t == 0 ? q : t == 1 ? k : v, h.data(), n * 2, hipMemcpyHostToDevice));
        }
        fa::KernelArgs a;
        a.q = q; a.k = k; a.v = v; a.o = o;
        a.batch_stride = (int64_t)S * H * D; a.seq_stride = H * D; a.head_stride = D;
        a.seq_len = S; a.n_heads = H; a.n_bh = B * H; a.n_kv_blocks = S / 64; a.causal = 0;
        for (int which = 0; which < 2; ++which) {
            CHECK(hipMemset(o, 0xff, n * 2));
            a.n_q_blocks = which ? S / 128 : S / 256;
            const int items = a.n_bh * a.n_q_blocks;
            hipLaunchKernelGGL(which ? k1 : k2, dim3(items < 256 ? items : 256), dim3(256), 163840, 0, a);
            CHECK(hipDeviceSynchronize());
            CHECK(hipMemcpy(which ? o1.data() : o2.data(), o, n * 2, hipMemcpyDeviceToHost));
        }
        size_t rows_bad = 0; int shown = 0;
        for (int b = 0; b < B; ++b) for (int s = 0; s < S; ++s) for (int hh = 0; hh < H; ++hh) {
            const size_t off = (((size_t)b * S + s) * H + hh) * D;
            int nd = 0; double worst = 0;
            for (int d = 0; d < D; ++d) if (o1[off + d] != o2[off + d]) { ++nd; const double e = fabs(bf(o1[off + d]) - bf(o2[off + d])); if (e > worst || e != e) worst = e != e ? 1e30 : e; }
            if (nd) { ++rows_bad; if (shown < 12) { printf("  S=%d b=%d h=%d row %4d (item %d, wave %d, row in tile %2d): %3d of 128 differ, worst %.3g  e.g. d0: %g vs %g\n", S, b, hh, s, s / 128, (s % 128) / 32, s % 32, nd, worst, bf(o1[off]), bf(o2[off])); ++shown; } }
        }
        printf("S=%d B=%d H=%d: %zu of %d rows differ\n", S, B, H, rows_bad, B * S * H);
        bad += rows_bad != 0;
        CHECK(hipFree(q)); CHECK(hipFree(k)); CHECK(hipFree(v)); CHECK(hipFree(o));
    }
    return bad ? 1 : 0;
}
