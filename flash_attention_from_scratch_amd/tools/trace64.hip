// trace64.hip -- s_memtime stamps inside ONE visit of the 64-rows-per-wave schedule (debug build
// with -DFA_TRACE; the shipped library carries no trace code).  Stamps: visit top, after the
// barrier, then every 4 MFMA gaps (ideal: 4 x 32 = 128 cycles apiece), visit end.
//
// Builds (csrc/Makefile, target `tools`): -DFA_TRACE=1 (default) stamps every 4 MFMAs of one visit -> lib/trace64;
// =2 the last 16 gaps of a visit; =3 one stamp per visit + the prologue pieces -> lib/trace64_tl; =4 ONE stamp per item,
// kept in a VGPR lane and stored at the exit (nothing added to the visits) -> lib/trace64_items; =5 the same plus stamps
// inside the first seam -> lib/trace64_seam.
//
// Caveats (check a trace build with tools/isa_lint64.py before believing it): the stamps change the
// code hipcc generates.  Modes 1 and 2 have been seen with the rescale path's 128 v_accvgpr_read
// hoisted behind the P.V MFMAs of two of the four visit variants (lint finding AGPR; those visits' last
// eight gaps then read 2-3x too long); mode 3's per-visit stamp (s_memtime + lgkmcnt(0) + a store)
// costs ~0.8 k cycles per visit, so its visit times are ~25 % above the product kernel's
// (PMC: SQ_VALU_MFMA_BUSY_CYCLES / SQ_WAVE_CYCLES = 71 %, ~2.75 k cycles per visit) -- use it for the
// prologue and the seams, which it does not disturb.
#ifndef FA_TRACE
#define FA_TRACE 1
#endif
#include "../csrc/fa_fwd_kernel64.hpp"
#ifndef TRACE_PSQ
#define TRACE_PSQ 0  // 1: the pre-scaled-Q build
#endif
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

#if FA_TRACE >= 4
// item timeline (-DFA_TRACE=4): trace64 [seq_len] [batch] [heads] -- the product kernel's walk with one stamp per item
// (no memory operation added to the visits): per workgroup  entry | S(0) formed | top of each item's first visit | exit
int main(int argc, char **argv) {
    const int S = argc > 1 ? atoi(argv[1]) : 4096, B = argc > 2 ? atoi(argv[2]) : 4, H = argc > 3 ? atoi(argv[3]) : 16, D = 128;
    const size_t n = (size_t)B * S * H * D;
    std::vector<uint16_t> h(n);
    uint16_t *q, *k, *v, *o; unsigned long long *tr;
    CHECK(hipMalloc(&q, n * 2)); CHECK(hipMalloc(&k, n * 2)); CHECK(hipMalloc(&v, n * 2)); CHECK(hipMalloc(&o, n * 2));
    const int grid = argc > 4 ? atoi(argv[4]) : 256;  // fewer workgroups (a multiple of 8): the same walk with a smaller chip-wide burst at the seams
    CHECK(hipMalloc(&tr, 4 * 256 * 64 * 4));
    srand(1);
    for (int t = 0; t < 3; ++t) {
        for (size_t i = 0; i < n; ++i) {
            float x = ((rand() & 0xffff) / 65536.0f - 0.5f) * 3.4f;
            uint32_t u; memcpy(&u, &x, 4); h[i] = (uint16_t)(u >> 16);
        }
        CHECK(hipMemcpy(t == 0 ? q : t == 1 ? k : v, h.data(), n * 2, hipMemcpyHostToDevice));
    }
    fa::KernelArgs a;
    a.q = q; a.k = k; a.v = v; a.o = o;
    a.batch_stride = (int64_t)S * H * D; a.seq_stride = H * D; a.head_stride = D;
    a.seq_len = S; a.n_heads = H; a.n_bh = B * H; a.n_q_blocks = S / 256; a.n_kv_blocks = S / 64; a.causal = 0;
    a.trace = tr; a.trace_block = -1; a.trace_visit = -1;
#ifndef TRACE_SPEC
#define TRACE_SPEC 1
#endif
    auto kern = fa::fa_fwd_kernel64<15, false, 0, false, TRACE_SPEC != 0, TRACE_PSQ != 0>;
    CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int n_items = B * H * (S / 256), per_wg = (n_items + grid - 1) / grid, nkv = S / 64;
    for (int rep = 0; rep < 3; ++rep) {
        for (int warm = 0; warm < (rep ? 2000 : 5); ++warm) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 163840, 0, a);  // rep > 0: clocks settled under load
        CHECK(hipMemset(tr, 0, 4 * 256 * 64 * 4));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 163840, 0, a);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned> t(4 * 256 * 64);
        CHECK(hipMemcpy(t.data(), tr, t.size() * 4, hipMemcpyDeviceToHost));
        printf("== S=%d B=%d H=%d grid %d: %d items per workgroup of %d visits; event %.1f us\n", S, B, H, grid, per_wg, nkv, ms * 1000);
        // wave 0 of every workgroup; times in s_memtime ticks (100 MHz on this chip? the tool prints the ratio to the event below)
        unsigned first = ~0u, last = 0;
        for (int w = 0; w < grid; ++w) { const unsigned *r = t.data() + w * 64; if (r[0] < first) first = r[0]; if (r[63] - first > last - first) last = r[63]; }
        printf("first entry -> last exit: %u ticks = %.4f ticks per ns of the event bracket\n", last - first, (last - first) / (ms * 1e6));
        double pro = 0, run = 0, fin = 0, ent = 0, ext = 0; std::vector<double> item(per_wg, 0.0);
        for (int w = 0; w < grid; ++w) {
            const unsigned *r = t.data() + w * 64;
            pro += r[1] - r[0]; run += r[63] - r[0]; ent += r[0] - first; ext += last - r[63];
            for (int i = 0; i + 1 < per_wg; ++i) item[i] += r[2 + i] - r[1 + i];
            fin += r[63] - r[per_wg];
        }
        {   // the chip-wide 100-MHz counter at every workgroup's entry (slot 46) and exit (slot 47): the clock of the walk, and where
            // the walks lie inside the launch (VERDICT r05 task 2: what a launch spends outside its workgroups)
            unsigned r_first = ~0u, r_last = 0; double clk = 0, in_us = 0, out_us = 0, walk_us = 0;
            for (int w = 0; w < grid; ++w) { const unsigned *r = t.data() + w * 64; if (r[46] < r_first) r_first = r[46]; }
            r_last = r_first;
            for (int w = 0; w < grid; ++w) { const unsigned *r = t.data() + w * 64; if (r[47] - r_first > r_last - r_first) r_last = r[47]; }
            std::vector<double> ent_us(grid), ext_us(grid);
            for (int w = 0; w < grid; ++w) {
                const unsigned *r = t.data() + w * 64;
                clk += (double)(r[63] - r[0]) / ((double)(r[47] - r[46]) * 0.01);   // cycles per us = MHz
                ent_us[w] = (r[46] - r_first) * 0.01; ext_us[w] = (r_last - r[47]) * 0.01;
                in_us += ent_us[w]; out_us += ext_us[w]; walk_us += (r[47] - r[46]) * 0.01;
            }
            std::sort(ent_us.begin(), ent_us.end()); std::sort(ext_us.begin(), ext_us.end());
            printf("realtime: first entry -> last exit %.2f us of the %.2f us event bracket | walk clock %.0f MHz | a workgroup's walk %.2f us (mean) |"
                   " entry behind the first: mean %.2f us, median %.2f, max %.2f | exit before the last: mean %.2f us, median %.2f, max %.2f\n",
                   (r_last - r_first) * 0.01, ms * 1000, clk / grid, walk_us / grid, in_us / grid, ent_us[grid / 2], ent_us[grid - 1],
                   out_us / grid, ext_us[grid / 2], ext_us[grid - 1]);
        }
        printf("mean over workgroups: entry +%.0f | prologue (entry -> first visit) %.0f |", ent / grid, pro / grid);
        for (int i = 0; i + 1 < per_wg; ++i) printf(" item %d (%d visits + seam) %.0f |", i, nkv, item[i] / grid);
        printf(" last item (%d visits + epilogue) %.0f | total %.0f | exit before the last %.0f\n", nkv, fin / grid, run / grid, ext / grid);
        const int show[4] = {0, 37 % grid, grid / 2, grid - 1};
        for (int si = 0; si < 4; ++si) {
            const unsigned *r = t.data() + show[si] * 64;
            printf("wg %3d wave 0: entry +%u | S(0) formed +%u | first visit +%u |", show[si], r[0] - first, r[62] - r[0], r[1] - r[0]);
            for (int i = 0; i + 1 < per_wg; ++i) printf(" %u", r[2 + i] - r[1 + i]);
            printf(" | last %u | exit +%u\n", r[63] - r[per_wg], r[63] - first);
        }
        if (FA_TRACE == 5 && per_wg > 1) {  // the first seam in detail (mean over workgroups, wave 0)
            const char *name[11] = {"visit n-2", "visit n-1 (forms the next S(0))", "epilogue (store_item)", "next coordinates", "Q tile 0 request",
                                    "row max + state", "O = 0", "to the top of visit 0", "visit 0", "visit 1", "visit 2"};
            const int a_[11] = {48, 49, 50, 51, 53, 54, 55, 52, 2, 56, 57}, b_[11] = {49, 50, 51, 53, 54, 55, 52, 2, 56, 57, 58};
            printf("first seam:");
            for (int i = 0; i < 11; ++i) { double sum = 0; for (int w = 0; w < grid; ++w) sum += (double)(t[w * 64 + b_[i]] - t[w * 64 + a_[i]]); printf(" %s %.0f |", name[i], sum / grid); }
            { double sum = 0; for (int w = 0; w < grid; ++w) sum += (double)(t[w * 64 + 59] - t[w * 64 + 58]); printf(" visit 3 %.0f\n", sum / grid); }
        }
        // the four waves of one workgroup: skew at the exit
        for (int wv = 0; wv < 4; ++wv) { const unsigned *r = t.data() + (wv * 256 + 5) * 64; printf("wg 5 wave %d: total %u%s", wv, r[63] - r[0], wv == 3 ? "\n" : " | "); }
    }
    return 0;
}
#elif FA_TRACE == 3
// timeline mode (-DFA_TRACE=3): trace64 [seq_len] [batch] -- one launch, every workgroup's wave 0 stamps
// entry | S(0) formed | visit tops ... | exit (a seam shows as a long visit)
int main(int argc, char **argv) {
    const int S = argc > 1 ? atoi(argv[1]) : 512, B = argc > 2 ? atoi(argv[2]) : 16, H = 16, D = 128;
    const size_t n = (size_t)B * S * H * D;
    std::vector<uint16_t> h(n);
    uint16_t *q, *k, *v, *o; unsigned long long *tr;
    CHECK(hipMalloc(&q, n * 2)); CHECK(hipMalloc(&k, n * 2)); CHECK(hipMalloc(&v, n * 2)); CHECK(hipMalloc(&o, n * 2));
    const int grid = 256;
    CHECK(hipMalloc(&tr, 4 * grid * 96 * 8));
    char *flush; const size_t flush_bytes = (size_t)1 << 30;
    CHECK(hipMalloc(&flush, flush_bytes));
    srand(1);
    for (int t = 0; t < 3; ++t) {
        for (size_t i = 0; i < n; ++i) {
            float x = ((rand() & 0xffff) / 65536.0f - 0.5f) * 3.4f;
            uint32_t u; memcpy(&u, &x, 4); h[i] = (uint16_t)(u >> 16);
        }
        CHECK(hipMemcpy(t == 0 ? q : t == 1 ? k : v, h.data(), n * 2, hipMemcpyHostToDevice));
    }
    fa::KernelArgs a;
    a.q = q; a.k = k; a.v = v; a.o = o;
    a.batch_stride = (int64_t)S * H * D; a.seq_stride = H * D; a.head_stride = D;
    a.seq_len = S; a.n_heads = H; a.n_bh = B * H; a.n_q_blocks = S / 256; a.n_kv_blocks = S / 64; a.causal = 0;
    a.trace = tr; a.trace_block = -1; a.trace_visit = -1;
#ifndef TRACE_ABL
#define TRACE_ABL 0
#endif
#ifndef TRACE_SPEC
#define TRACE_SPEC 1  // 1: the speculative-softmax build (the default kernel), 0: the lazy-rescale build
#endif
    auto kern = fa::fa_fwd_kernel64<15, false, TRACE_ABL, false, TRACE_SPEC != 0, TRACE_PSQ != 0>;
    CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int mode = 0; mode < 2; ++mode) {  // 0: warm caches, back to back; 1: cache flushed before the launch
        for (int warm = 0; warm < 5; ++warm) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 163840, 0, a);
        if (mode == 1) { CHECK(hipMemset(flush, 1, flush_bytes)); }
        CHECK(hipMemset(tr, 0, 4 * grid * 96 * 8));
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 163840, 0, a);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned> t(grid * 96);  // wave 0's stamps: low 32 bits of s_memtime, 96 slots per workgroup
        CHECK(hipMemcpy(t.data(), tr, grid * 96 * 4, hipMemcpyDeviceToHost));
        int cnt = 0;
        while (cnt < 96 && t[cnt]) ++cnt;
        printf("== %s: S=%d B=%d  event %.1f us (%d stamps per workgroup: entry, S(0) formed, visit tops, exit)\n",
               mode ? "flushed" : "warm", S, B, ms * 1000, cnt);
        const int show[6] = {0, 1, 37, 128, 200, 255};
        for (int si = 0; si < 6; ++si) {
            const unsigned *r = t.data() + show[si] * 96;
            // r[0] entry, r[1..6] prologue stamps, r[7] S(0) formed, r[8..] visit tops, r[cnt-1] exit
            printf("wg %3d: total %7u | prologue %5u = K0 req %u | Q,K1,V0 req %u | landed %u | Q read+barrier %u | S0 issue+rest req %u | max+K1 wait %u | barrier %u |",
                   show[si], r[cnt - 1] - r[0], r[7] - r[0], r[1] - r[0], r[2] - r[1], r[3] - r[2], r[4] - r[3], r[5] - r[4], r[6] - r[5], r[7] - r[6]);
            for (int i = 8; i < cnt; ++i) printf(" %u", r[i] - r[i - 1]);
            printf("\n");
        }
        {   // spread over the workgroups: how long each one ran, and when it entered / left relative to the first entry
            unsigned e0 = ~0u; for (int w = 0; w < grid; ++w) e0 = t[w * 96] < e0 ? t[w * 96] : e0;
            double tmin = 1e18, tmax = 0, emax = 0, xmin = 1e18, xmax = 0;
            for (int w = 0; w < grid; ++w) {
                const double tot = (double)(t[w * 96 + cnt - 1] - t[w * 96]), en = (double)(t[w * 96] - e0), ex = (double)(t[w * 96 + cnt - 1] - e0);
                tmin = tot < tmin ? tot : tmin; tmax = tot > tmax ? tot : tmax; emax = en > emax ? en : emax;
                xmin = ex < xmin ? ex : xmin; xmax = ex > xmax ? ex : xmax;
            }
            printf("spread: workgroup run time min %.0f max %.0f | last entry +%.0f | first exit +%.0f last exit +%.0f (cycles after the first entry)\n",
                   tmin, tmax, emax, xmin, xmax);
        }
        printf("mean  : total %7.0f |", [&] { double s_ = 0; for (int w = 0; w < grid; ++w) s_ += (double)(t[w * 96 + cnt - 1] - t[w * 96]); return s_ / grid; }());
        for (int i = 1; i < cnt; ++i) { double sum = 0; for (int w = 0; w < grid; ++w) sum += (double)(t[w * 96 + i] - t[w * 96 + i - 1]); printf(" %.0f", sum / grid); }
        printf("\n");
    }
    return 0;
}
#else
int main(int argc, char **argv) {
    const int B = 16, H = 16, D = 128, S = 4096;
    const bool zeros = argc > 1 && !strcmp(argv[1], "zeros");
    const size_t n = (size_t)B * S * H * D;
    std::vector<uint16_t> h(n);
    uint16_t *q, *k, *v, *o; unsigned long long *tr;
    CHECK(hipMalloc(&q, n * 2)); CHECK(hipMalloc(&k, n * 2)); CHECK(hipMalloc(&v, n * 2)); CHECK(hipMalloc(&o, n * 2));
    CHECK(hipMalloc(&tr, 4 * 24 * 8));
    srand(1);
    for (int t = 0; t < 3; ++t) {
        for (size_t i = 0; i < n; ++i) {
            float x = zeros ? 0.0f : ((rand() & 0xffff) / 65536.0f - 0.5f) * 3.4f;
            uint32_t u; memcpy(&u, &x, 4); h[i] = (uint16_t)(u >> 16);
        }
        CHECK(hipMemcpy(t == 0 ? q : t == 1 ? k : v, h.data(), n * 2, hipMemcpyHostToDevice));
    }
    fa::KernelArgs a;
    a.q = q; a.k = k; a.v = v; a.o = o;
    a.batch_stride = (int64_t)S * H * D; a.seq_stride = H * D; a.head_stride = D;
    a.seq_len = S; a.n_heads = H; a.n_bh = B * H; a.n_q_blocks = S / 256; a.n_kv_blocks = S / 64; a.causal = 0;
    a.trace = tr;
#ifndef TRACE_ABL
#define TRACE_ABL 0
#endif
#ifndef TRACE_SPEC
#define TRACE_SPEC 1
#endif
    auto kern = fa::fa_fwd_kernel64<15, false, TRACE_ABL, false, TRACE_SPEC != 0, TRACE_PSQ != 0>;
    CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 163840));
    // persistent kernel: workgroup w serves items w, w + 256, ...; trace the second item of two workgroups
    const int items[2] = {256 + 100, 256 + 203};
    const int visits[6] = {0, 1, 20, 61, 62, 63};
    const int grid = 256;
    for (int warm = 0; warm < 5; ++warm) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 163840, 0, a);
    for (int bi = 0; bi < 2; ++bi) for (int vi = 0; vi < 6; ++vi) {
        a.trace_block = items[bi]; a.trace_visit = visits[vi];
        CHECK(hipMemset(tr, 0, 4 * 24 * 8));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 163840, 0, a);
        CHECK(hipDeviceSynchronize());
        unsigned long long t[96];
        CHECK(hipMemcpy(t, tr, sizeof(t), hipMemcpyDeviceToHost));
        for (int w = 0; w < 4; ++w) {
            const unsigned long long *r = t + w * 24;
#if FA_TRACE == 2
            printf("item %4d visit %2d wave %d: total %5llu | gaps 48..63:", items[bi], visits[vi], w, r[18] - r[0]);
            for (int i = 2; i < 17; ++i) printf(" %3llu", r[i + 1] - r[i]);
            printf(" | 63->end %3llu", r[18] - r[17]);
#else
            printf("item %4d visit %2d wave %d: total %5llu | top %4llu | groups", items[bi], visits[vi], w, r[18] - r[0], r[2] - r[0]);
            for (int i = 2; i < 17; ++i) printf(" %3llu", r[i + 1] - r[i]);
            printf(" | last %3llu", r[18] - r[17]);
#endif
            if (vi == 5) printf(" | end-of-visit->epilogue %llu epilogue %llu reset %llu", r[21] - r[18], r[22] - r[21], r[23] - r[22]);
            printf("\n");
        }
    }
    return 0;
}
#endif
