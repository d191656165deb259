// check_nw8.hip -- the EIGHT-wave one-Q-tile-per-wave form (QTP = 1, NW = 8: 256-row items, two waves per SIMD sharing the four-stage
// rings; round 6, experimental) against the 64-rows-per-wave lazy kernel on the same inputs, then timed against the shipped kernels.
// From check_qt1.hip:
// the two run the same arithmetic per 32-row tile, so their outputs must agree bit for bit.  Prints, per shape, how many
// rows differ and where the first ones are.  The speculative one-tile form walks K / V first-to-last since round 6 (its
// first pass; fa_fwd_kernel64.hpp FWD), so its fp32 sums add up in the other order: it is held to 2 ulp of the yardstick
// instead (rows that agree bit for bit are counted -- the items it gives up on are redone BY the lazy schedule).  Spiky
// data: a large key early in the sequence of every second head (the lazy forms visit it last and move their reference
// there; the speculative form takes it as its reference) and a larger one near the end (which fails its first pass).
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
    typedef void (*kfn)(const fa::KernelArgs);
    // 0: the 64-row lazy kernel (the yardstick), 1: one tile per wave, lazy, 2: one tile per wave, speculative
    kfn kern[3] = {(kfn)fa::fa_fwd_kernel64<15, false, 0, false, false, false, 2>,
                   (kfn)fa::fa_fwd_kernel64<15, false, 0, false, false, false, 1, false, 8>,
                   (kfn)fa::fa_fwd_kernel64<15, false, 0, false, true, false, 1, false, 8>};
    const char *names[3] = {"64-row lazy", "eight-wave one-tile lazy", "eight-wave one-tile speculative"};
    const int only_time = argc > 1 && !strcmp(argv[1], "time");
    for (auto f : kern) CHECK(hipFuncSetAttribute((const void *)f, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
    uint32_t *stats; CHECK(hipMalloc(&stats, 8));
    int bad = 0;
    for (int spiky = 0; spiky < (only_time ? 0 : 2); ++spiky)
    for (auto &sh : shapes) {
        const int S = sh[0], B = sh[1], H = sh[2], D = 128;
        const size_t n = (size_t)B * S * H * D;
        std::vector<uint16_t> h(n), out[3];
        uint16_t *q, *k, *v, *o;
        CHECK(hipMalloc(&q, n * 2)); CHECK(hipMalloc(&k, n * 2)); CHECK(hipMalloc(&v, n * 2)); CHECK(hipMalloc(&o, n * 2));
        srand(S + B);
        for (int t = 0; t < 3; ++t) {
            for (size_t i = 0; i < n; ++i) { float x = ((rand() & 0xffff) / 65536.0f - 0.5f) * 3.4f; uint32_t u; memcpy(&u, &x, 4); h[i] = (uint16_t)(u >> 16); }
            if (t == 1 && spiky)  // key 5 of every second head x 80 (logits ~100 binades above an N(0, 1) tile's), key S - 6 x 240
                for (int b = 0; b < B; ++b) for (int hh = 0; hh < H; hh += 2) for (int d = 0; d < D; ++d)
                    for (int far = 0; far < 2; ++far) {
                        uint16_t &w = h[(((size_t)b * S + (far ? S - 6 : 5)) * H + hh) * D + d];
                        float x = bf(w) * (far ? 240.0f : 80.0f); uint32_t u; memcpy(&u, &x, 4); w = (uint16_t)(u >> 16);
                    }
            CHECK(hipMemcpy(t == 0 ? q : t == 1 ? k : v, h.data(), n * 2, hipMemcpyHostToDevice));
        }
        fa::KernelArgs a;
        a.q = q; a.k = k; a.v = v; a.o = o;
        a.batch_stride = (int64_t)S * H * D; a.seq_stride = H * D; a.head_stride = D;
        a.seq_len = S; a.n_heads = H; a.n_bh = B * H; a.n_kv_blocks = S / 64; a.causal = 0;
        a.stats = stats;
        uint32_t st[3][2] = {};
        for (int which = 0; which < 3; ++which) {
            CHECK(hipMemset(o, 0xff, n * 2));
            CHECK(hipMemset(stats, 0, 8));
            a.n_q_blocks = S / 256;   // (eight waves x 32 rows = the 64-row kernel's 256-row items)
            const int items = a.n_bh * a.n_q_blocks;
            hipLaunchKernelGGL(kern[which], dim3(items < 256 ? items : 256), dim3(which ? 512 : 256), 163840, 0, a);
            CHECK(hipDeviceSynchronize());
            CHECK(hipMemcpy(st[which], stats, 8, hipMemcpyDeviceToHost));
            out[which].resize(n);
            CHECK(hipMemcpy(out[which].data(), o, n * 2, hipMemcpyDeviceToHost));
        }
        for (int which = 1; which < 3; ++which) {
            const auto &o1 = out[which], &o2 = out[0];
            size_t rows_bad = 0, rows_equal = 0; int shown = 0;
            const bool exact = which == 1;   // (the speculative form: 2 ulp, see the header)
            for (int b = 0; b < B; ++b) for (int s = 0; s < S; ++s) for (int hh = 0; hh < H; ++hh) {
                const size_t off = (((size_t)b * S + s) * H + hh) * D;
                int nd = 0; double worst = 0;
                for (int d = 0; d < D; ++d) if (o1[off + d] != o2[off + d]) {
                    const double e = fabs(bf(o1[off + d]) - bf(o2[off + d])), tol = exact ? 0.0 : 0.0078125 * (1.0 + fabs(bf(o2[off + d])));
                    if (!(e <= tol)) { ++nd; if (e > worst || e != e) worst = e != e ? 1e30 : e; }
                }
                rows_equal += !memcmp(&o1[off], &o2[off], D * 2);
                if (nd) { ++rows_bad; if (shown < 6) { printf("  S=%d b=%d h=%d row %4d (item %d, wave %d, row in tile %2d): %3d of 128 differ, worst %.3g  e.g. d0: %g vs %g\n", S, b, hh, s, s / 256, (s % 256) / 32, s % 32, nd, worst, bf(o1[off]), bf(o2[off])); ++shown; } }
            }
            printf("%s vs %s%s  S=%d B=%d H=%d: %zu of %d rows differ%s, %zu bit-identical   (items / redone: %u / %u)\n", names[which], names[0],
                   spiky ? ", spiky keys" : "", S, B, H, rows_bad, B * S * H, exact ? "" : " by more than 2 ulp", rows_equal, st[which][0], st[which][1]);
            bad += rows_bad != 0;
        }
        CHECK(hipFree(q)); CHECK(hipFree(k)); CHECK(hipFree(v)); CHECK(hipFree(o));
    }
    printf(bad ? "FAILED\n" : "the lazy forms agree bit for bit, the speculative one within 2 ulp\n");
    // ---- timing: B = 16, H = 16, S in {512, 1024, 4096}; interleaved, the shipped 64-row kernels beside the eight-wave forms
    {
        kfn tk[4] = {(kfn)fa::fa_fwd_kernel64<15, false, 0, false, true, false, 2>, kern[2], kern[0], kern[1]};
        const char *tn[4] = {"64-row speculative (shipped default)", "eight-wave speculative", "64-row lazy", "eight-wave lazy"};
        const int thr[4] = {256, 512, 256, 512};
        CHECK(hipFuncSetAttribute((const void *)tk[0], hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        for (int S : {512, 1024, 4096, 16384}) {
            const int B = S > 4096 ? 4 : 16, H = 16, D = 128;
            const size_t n = (size_t)B * S * H * D;
            std::vector<uint16_t> h(n);
            uint16_t *q, *k, *v, *o;
            CHECK(hipMalloc(&q, n * 2)); CHECK(hipMalloc(&k, n * 2)); CHECK(hipMalloc(&v, n * 2)); CHECK(hipMalloc(&o, n * 2));
            srand(7);
            for (int t = 0; t < 3; ++t) {
                for (size_t i = 0; i < n; ++i) { float x = ((rand() & 0xffff) / 65536.0f - 0.5f) * 3.4f; uint32_t u; memcpy(&u, &x, 4); h[i] = (uint16_t)(u >> 16); }
                CHECK(hipMemcpy(t == 0 ? q : t == 1 ? k : v, h.data(), n * 2, hipMemcpyHostToDevice));
            }
            fa::KernelArgs a;
            a.q = q; a.k = k; a.v = v; a.o = o;
            a.batch_stride = (int64_t)S * H * D; a.seq_stride = H * D; a.head_stride = D;
            a.seq_len = S; a.n_heads = H; a.n_bh = B * H; a.n_kv_blocks = S / 64; a.causal = 0; a.stats = nullptr;
            a.n_q_blocks = S / 256;
            double sum[4] = {}; int cnt[4] = {};
            for (int round = -1; round < 8; ++round)
                for (int w = 0; w < 4; ++w)
                    for (int i = 0; i < 8; ++i) {
                        CHECK(hipEventRecord(e0));
                        hipLaunchKernelGGL(tk[w], dim3(256), dim3(thr[w]), 163840, 0, a);
                        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                        if (round >= 0 && i >= 2) { sum[w] += ms; ++cnt[w]; }
                    }
            const double fl = 4.0 * B * H * (double)S * S * D;
            for (int w = 0; w < 4; ++w) printf("S=%5d B=%2d  %-40s %.4f ms  %7.1f TFLOP/s\n", S, B, tn[w], sum[w] / cnt[w], fl / (sum[w] / cnt[w] * 1e-3) / 1e12);
            CHECK(hipFree(q)); CHECK(hipFree(k)); CHECK(hipFree(v)); CHECK(hipFree(o));
        }
    }
    return bad ? 1 : 0;
}
