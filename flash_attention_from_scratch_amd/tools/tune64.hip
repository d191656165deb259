// tune64.hip -- A/B timing of experiment knobs (ABL bits >= 256) of the 64-rows-per-wave pinned
// schedule on the headline shape, random data (the chip is power-limited: constant data
// clocks ~20 % higher and hides everything).
#include "../csrc/fa_fwd_kernel64.hpp"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

static uint16_t *q, *k, *v, *o, *q_scaled;  // q_scaled = q * c (bit 2048 variants: exp2 of the raw logit)
static const int B = 16, H = 16, D = 128;
static int S = 4096;
static std::vector<int> only_list;  // argv: only=<abl>[,<abl>...]
static int n_rep = 10;

struct Variant { const char *name; int abl; void (*launch)(const fa::KernelArgs &); double sum_ms; int n; float best; };
static std::vector<Variant> variants;
static hipEvent_t e0, e1;

// ABL bit 1 << 20: the speculative-softmax build (SPEC) of the same knobs; bit 23: the masked (causal-capable) build, run with
// nothing masked (causal = 0): what the masked form costs a caller that masks nothing
template <int ABL> void launch(const fa::KernelArgs &a) {
    auto kern = fa::fa_fwd_kernel64<15, ((ABL >> 23) & 1) != 0, (ABL & 0x7f4fffff), false, ((ABL >> 20) & 1) != 0, ((ABL >> 21) & 1) != 0>;  // bit 21: the pre-scaled Q; bits 24..28: the rotated plan
    static bool init = false;
    if (!init) { CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 163840)); init = true; }
    fa::KernelArgs b = a;
    if (ABL & 2048) b.q = q_scaled;
    hipLaunchKernelGGL(kern, dim3(b.n_bh * b.n_q_blocks < 256 ? b.n_bh * b.n_q_blocks : 256), dim3(256), 163840, 0, b);
}
// the previous round's kernel (tools/tune64_prev.hip), when the build links it: list entries -1 (speculative), -2 (lazy)
#ifdef TUNE64_PREV
namespace prev { void launch(bool spec, const void *q, const void *k, const void *v, void *o, long long bs, long long ss, long long hs,
                             int seq_len, int n_heads, int n_bh, int n_q_blocks, int n_kv_blocks); }
template <bool SPEC> void launch_prev(const fa::KernelArgs &a) {
    prev::launch(SPEC, a.q, a.k, a.v, a.o, a.batch_stride, a.seq_stride, a.head_stride, a.seq_len, a.n_heads, a.n_bh, a.n_q_blocks, a.n_kv_blocks);
}
#endif
// -3: the hand-placed ONE-Q-tile-per-wave form (QTP = 1: the reference's (128, 64, 4)+buffer shape, 128-row items); -4: the
// compiler-scheduled 32-rows-per-wave body the same configuration ran on through round 4 (two workgroups per CU);
// -5 / -6: the same two with the speculative softmax
template <bool SPEC> static void launch_qt1(const fa::KernelArgs &a) {
    auto kern = fa::fa_fwd_kernel64<15, false, 0, false, SPEC, false, 1>;
    static bool init = false;
    if (!init) { CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 163840)); init = true; }
    fa::KernelArgs b = a;
    b.n_q_blocks = a.seq_len / 128;
    hipLaunchKernelGGL(kern, dim3(b.n_bh * b.n_q_blocks < 256 ? b.n_bh * b.n_q_blocks : 256), dim3(256), 163840, 0, b);
}
// -7: the long-sequence form of the shipped speculative kernel (ALT: every second round of a head's Q blocks walks K / V
// [tile 0, then last-to-second]); differs from the shipped one only where the launcher would take it (S = 16384 here)
static void launch_alt(const fa::KernelArgs &a) {
    auto kern = fa::fa_fwd_kernel64<15, false, 0, false, true, false, 2, true>;
    static bool init = false;
    if (!init) { CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 163840)); init = true; }
    hipLaunchKernelGGL(kern, dim3(a.n_bh * a.n_q_blocks < 256 ? a.n_bh * a.n_q_blocks : 256), dim3(256), 163840, 0, a);
}
// -8 / -9: the EIGHT-wave ring form (round 6: the product's ring form of (128, 64, 4)+buffer): 256-row items, 512 threads
template <bool SPEC> static void launch_nw8(const fa::KernelArgs &a) {
    auto kern = fa::fa_fwd_kernel64<15, false, 0, false, SPEC, false, 1, false, 8>;
    static bool init = false;
    if (!init) { CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 163840)); init = true; }
    hipLaunchKernelGGL(kern, dim3(a.n_bh * a.n_q_blocks < 256 ? a.n_bh * a.n_q_blocks : 256), dim3(512), 163840, 0, a);
}
template <bool SPEC> static void launch_32row(const fa::KernelArgs &a) {
    using TR = fa::FwdTraits<15, 1, 4, 64, true, true, SPEC, true, true, false, 128>;
    auto kern = fa::fa_fwd_kernel<15, 1, 4, 64, true, true, SPEC, true, true, false, 128>;
    static bool init = false;
    if (!init) { CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, TR::kLdsBytes)); init = true; }
    fa::KernelArgs b = a;
    b.n_q_blocks = a.seq_len / 128;
    hipLaunchKernelGGL(kern, dim3(b.n_bh * b.n_q_blocks), dim3(256), TR::kLdsBytes, 0, b);
}
template <int ABL> void add(const char *name) {
    if (!only_list.empty() && std::find(only_list.begin(), only_list.end(), ABL) == only_list.end()) return;
    if constexpr (ABL == -3 || ABL == -5) {
        variants.push_back({name, ABL, launch_qt1<ABL == -5>, 0.0, 0, 1e9f});
    } else if constexpr (ABL == -8 || ABL == -9) {
        variants.push_back({name, ABL, launch_nw8<ABL == -9>, 0.0, 0, 1e9f});
    } else if constexpr (ABL == -7) {
        variants.push_back({name, ABL, launch_alt, 0.0, 0, 1e9f});
    } else if constexpr (ABL == -4 || ABL == -6) {
        variants.push_back({name, ABL, launch_32row<ABL == -6>, 0.0, 0, 1e9f});
    } else if constexpr (ABL < 0) {
#ifdef TUNE64_PREV
        variants.push_back({name, ABL, launch_prev<ABL == -1>, 0.0, 0, 1e9f});
#endif
    } else {
        variants.push_back({name, ABL, launch<ABL>, 0.0, 0, 1e9f});
    }
}
// interleaved timing: the chip's clock drifts with temperature / power state by a few percent
// over seconds, so every round runs every variant and the means are compared
static void time_all() {
    fa::KernelArgs a;
    const int Bx = S > 4096 ? 4 : B;
    a.q = q; a.k = k; a.v = v; a.o = o;
    a.batch_stride = (int64_t)S * H * D; a.seq_stride = H * D; a.head_stride = D;
    a.seq_len = S; a.n_heads = H; a.n_bh = Bx * H; a.n_q_blocks = S / 256; a.n_kv_blocks = S / 64; a.causal = 0;
    for (auto &v : variants) { v.sum_ms = 0; v.n = 0; v.best = 1e9f; }
    // (the order rotates from round to round: a variant's neighbour in time moves its mean by 1-2 % at S <= 1024)
    const size_t nv = variants.size();
    for (int round = -1; round < n_rep; ++round) {
        for (size_t vi = 0; vi < nv; ++vi) {
            auto &v = variants[(vi + (size_t)(round + 1) * 3) % nv];
            // a turn = 2 untimed launches (another variant ran just before: its code, not this one's, is in the instruction
            // caches -- at S = 512 a launch is 40 us and that shows) + 6 timed ones
            for (int i = 0; i < 8; ++i) {
                CHECK(hipEventRecord(e0));
                v.launch(a);
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (round >= 0 && i >= 2) { v.sum_ms += ms; v.n++; if (ms < v.best) v.best = ms; }
            }
        }
    }
    const double fl = 4.0 * Bx * H * (double)S * S * D;
    // a hash of each variant's output: variants that only move work around (not the arithmetic) must agree bit for bit
    const size_t n_o = (size_t)Bx * S * H * D;
    std::vector<uint16_t> ho(n_o), ho0;   // ho0: the first variant's output, the yardstick for the others' largest difference
    auto bf = [](uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; };
    for (auto &v : variants) {
        CHECK(hipMemset(o, 0xff, n_o * 2));
        v.launch(a);
        CHECK(hipMemcpy(ho.data(), o, n_o * 2, hipMemcpyDeviceToHost));
        unsigned long long hsh = 1469598103934665603ull;
        for (size_t i = 0; i < n_o; ++i) hsh = (hsh ^ ho[i]) * 1099511628211ull;
        double worst = 0.0; size_t n_diff = 0;
        if (ho0.empty()) ho0 = ho;
        else for (size_t i = 0; i < n_o; ++i) if (ho[i] != ho0[i]) { ++n_diff; const double d = fabs((double)bf(ho[i]) - (double)bf(ho0[i])) / (1.0 + fabs((double)bf(ho0[i]))); if (d > worst || d != d) worst = d != d ? 1e30 : d; }
        printf("%-52s abl=%8d S=%5d : mean %.4f ms  %7.1f TF   best %7.1f TF   out %016llx  vs first: %zu differ, worst %.2e\n", v.name, v.abl, S, v.sum_ms / v.n,
               fl / (v.sum_ms / v.n * 1e-3) / 1e12, fl / (v.best * 1e-3) / 1e12, hsh, n_diff, worst);
    }
}

int main(int argc, char **argv) {
    const size_t n = (size_t)B * 4096 * H * D;
    std::vector<uint16_t> h(n);
    CHECK(hipMalloc(&q, n * 2)); CHECK(hipMalloc(&q_scaled, n * 2)); CHECK(hipMalloc(&k, n * 2)); CHECK(hipMalloc(&v, n * 2)); CHECK(hipMalloc(&o, n * 2));
    srand(1);
    bool zeros = false;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "zeros")) zeros = true;
        if (!strncmp(argv[i], "only=", 5))  // only=a,b,c
            for (const char *p = argv[i] + 5; *p;) { only_list.push_back(atoi(p)); while (*p && *p != ',') ++p; if (*p) ++p; }
        if (!strncmp(argv[i], "reps=", 5)) n_rep = atoi(argv[i] + 5);
    }
    for (int t = 0; t < 3; ++t) {
        for (size_t i = 0; i < n; ++i) {
            float x = zeros ? 0.0f : ((rand() & 0xffff) / 65536.0f - 0.5f) * 3.4f;
            uint32_t u; memcpy(&u, &x, 4); h[i] = (uint16_t)(u >> 16);
        }
        CHECK(hipMemcpy(t == 0 ? q : t == 1 ? k : v, h.data(), n * 2, hipMemcpyHostToDevice));
        if (t == 0) {
            for (size_t i = 0; i < n; ++i) {
                uint32_t u = (uint32_t)h[i] << 16; float x; memcpy(&x, &u, 4);
                x *= 0.12751743f; memcpy(&u, &x, 4); h[i] = (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
            }
            CHECK(hipMemcpy(q_scaled, h.data(), n * 2, hipMemcpyHostToDevice));
        }
    }
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
#define X(abl, name) add<abl>(name);
#ifndef TUNE64_LIST
#define TUNE64_LIST "tune64_list.inc"
#endif
#include TUNE64_LIST
#undef X
    const int seqs[4] = {512, 1024, 4096, 16384};
    for (int pass = 0; pass < 4; ++pass) {
        S = seqs[pass];
        time_all();
    }
    return 0;
}
