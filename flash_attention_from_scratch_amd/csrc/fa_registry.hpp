// fa_registry.hpp -- table of device variants built into libfa_hip.so.
// The role of /root/reference/src/include/flash_kernels.cuh:14-186 (a generated
// std::map<config, fn>) is played by per-translation-unit tables of KernelEntry
// that fa_capi.cpp concatenates; which variants exist is decided by the
// FA_INSTANTIATE lines in fa_inst_*.hip.
#pragma once
#include <stdint.h>
#include "fa_fwd_kernel.hpp"
#include "fa_fwd_kernel64.hpp"

namespace fa {

typedef void (*kernel_fn)(const KernelArgs);

struct KernelEntry {
    int dtype;          // 5 / 15
    int rows_per_wave;  // 16, 32 or 64
    int n_waves;
    int B_c;
    int swizzled;
    int eager;
    int opt_softmax;    // the variant's OPT template flag: see softmax_mode for what it means there
    int pipelined;      // cfg.mma_double_buffer_loads
    int async_copy;     // 1: LDS-DMA transport, 0: register-staged
    int masked;         // 1: handles ragged seq_len and the causal mask; 2: the same through two device forms (fn, fn_ragged)
    int d_head;         // 128 (reference scope) or 64
    int threads;
    int lds_bytes;
    int persistent;     // 1: launch one workgroup per CU; the kernel walks the items itself
    kernel_fn fn;
    kernel_fn fn_ragged;  // masked == 2 only: the form for seq_len % B_r != 0 (any seq_len >= 64); else null
    int softmax_mode;     // fa_softmax_mode (include/fa_hip.h): 0 eager, 1 first block skips the rescale, 2 lazy, 3 speculative
    int prescaled_q;      // 1: logits from a 16-bit Q * c (fa_fwd_opts.prescaled_q); persistent kernel only
    // Round 5: the RING FORM of a 32-rows-per-wave configuration -- the hand-placed persistent kernel with one Q tile per
    // wave (fa_fwd_kernel64<..., QTP = 1>), which serves launches with seq_len % 256 == 0 (four ring stages = four tiles to
    // a group); `fn` serves the other multiples of B_r.  Null where none is built.  Its softmax: the lazy rescale for the
    // plain variant, the speculative schedule (softmax_mode 3) for the speculative one.
    kernel_fn fn_ring = nullptr;
    int ring_lds_bytes = 0;
    // Round 6: the 64-row speculative plain form whose first pass walks every second round of a long head's Q blocks
    // [tile 0, then last-to-second] (fa_fwd_kernel64<..., ALT = true>): the launcher takes it when a head's Q blocks fill an
    // even number of rounds of an XCD's workgroups, so that the K / V tail a round leaves in L2 is read again first.
    kernel_fn fn_alt = nullptr;
    // ... the ring form's own launch geometry (round 6: eight waves of one Q tile each -- two of the shape's 128-row workgroups
    // fused into one that shares the K / V rings: 512 threads, 256-row items); 0 = the entry's own threads / B_r
    int ring_threads = 0;
    int ring_rows = 0;
};

// What the OPT template flag selects in each kernel body (the device-side predicates are SPEC in
// fa_fwd_kernel.hpp / fa_fwd_kernel16.hpp and the SPEC parameter of fa_fwd_kernel64): the speculative softmax
// where it is built -- the persistent kernel (every form), the double-buffered LDS-DMA variants of the
// 32-rows-per-wave kernel without a mask, the double-buffered 16-rows-per-wave variants -- and the
// reference's first-block skip elsewhere.
constexpr int softmax_mode_of(bool persistent, bool opt, bool eager, bool dma, bool masked) {
    if (persistent) return opt ? 3 : 2;
    if (!opt) return 0;
    return (eager && dma && !masked) ? 3 : 1;
}

template <int DT, int QT, int NWAVES, int BC, bool SWZ, bool EAGER, bool OPT, bool PIPE, bool DMA,
          bool MASK = false, int D = 128>
constexpr KernelEntry make_entry() {
    using TR = FwdTraits<DT, QT, NWAVES, BC, SWZ, EAGER, OPT, PIPE, DMA, MASK, D>;
    if constexpr (TR::kPersistent) {  // (B_r 256, B_c 64, 4 waves) + buffer: fa_fwd_kernel64.hpp
        static_assert(NWAVES == 4 && BC == 64 && SWZ && EAGER && DMA && D == 128, "64-row pinned schedule");
        if constexpr (MASK)
            return KernelEntry{DT, 64, 4, 64, 1, 1, OPT, 1, 1, 2, 128, TR::kThreads, TR::kLdsBytes, 1,
                               (kernel_fn)&fa_fwd_kernel64<DT, true, 0, false, OPT>,   // causal form (OPT: speculative softmax)
                               (kernel_fn)&fa_fwd_kernel64<DT, true, 0, true, OPT>,    // ragged form
                               softmax_mode_of(true, OPT, true, true, true), 0};
        else
            return KernelEntry{DT, 64, 4, 64, 1, 1, OPT, 1, 1, 0, 128, TR::kThreads, TR::kLdsBytes, 1,
                               (kernel_fn)&fa_fwd_kernel64<DT, false, 0, false, OPT>, nullptr,  // OPT: speculative softmax
                               softmax_mode_of(true, OPT, true, true, false), 0, nullptr, 0,
                               OPT ? (kernel_fn)&fa_fwd_kernel64<DT, false, 0, false, OPT, false, 2, OPT> : nullptr};
    } else if constexpr (QT == 1 && NWAVES == 4 && BC == 64 && SWZ && EAGER && PIPE && DMA && !MASK && D == 128) {
        // the reference's winning tile shape, (B_r 128, B_c 64, 4 warps) + buffer (kernel_sass/16_A100.asm:5): the
        // compiler-scheduled body, and the hand-placed ring form for seq_len % 256 == 0.  OPT = true is the speculative
        // softmax (asked for through fa_fwd_opts only) in both.
        // Round 6: the ring form is the EIGHT-wave one (fa_fwd_kernel64<..., QTP = 1, ALT = false, NW = 8>): 32 rows per wave as
        // the config asks, two waves per SIMD, 256-row items -- two of the shape's 128-row workgroups fused into one that
        // shares one set of K / V rings; per 32-row tile the arithmetic of the four-wave ring form of round 5 and of the 64-row
        // lazy kernel, bit for bit (tools/check_nw8.hip, tools/check_qt1.hip).
        return KernelEntry{DT, 32, 4, 64, 1, 1, OPT, 1, 1, 0, 128, TR::kThreads, TR::kLdsBytes, 0,
                           (kernel_fn)&fa_fwd_kernel<DT, QT, NWAVES, BC, SWZ, EAGER, OPT, PIPE, DMA, MASK, D>, nullptr,
                           softmax_mode_of(false, OPT, EAGER, DMA, MASK), 0,
                           (kernel_fn)&fa_fwd_kernel64<DT, false, 0, false, OPT, false, 1, false, 8>, RingTraits<1, 8>::kLdsBytes,
                           nullptr, RingTraits<1, 8>::kThreads, RingTraits<1, 8>::kBr};
    } else {
        return KernelEntry{DT, 32 * QT, NWAVES, BC, SWZ, EAGER, OPT, PIPE, DMA, MASK ? 1 : 0, D, TR::kThreads,
                           TR::kLdsBytes, 0,
                           (kernel_fn)&fa_fwd_kernel<DT, QT, NWAVES, BC, SWZ, EAGER, OPT, PIPE, DMA, MASK, D>, nullptr,
                           softmax_mode_of(false, OPT, EAGER, DMA, MASK), 0};
    }
}

// The persistent kernel with the pre-scaled Q (fa_fwd_opts.prescaled_q): plain form only.
template <int DT, bool OPT>
constexpr KernelEntry make_entry_psq() {
    using TR = FwdTraits<DT, 2, 4, 64, true, true, OPT, true, true, false, 128>;
    static_assert(TR::kPersistent, "the 64-row pinned schedule");
    return KernelEntry{DT, 64, 4, 64, 1, 1, OPT, 1, 1, 0, 128, TR::kThreads, TR::kLdsBytes, 1,
                       (kernel_fn)&fa_fwd_kernel64<DT, false, 0, false, OPT, true>, nullptr,
                       softmax_mode_of(true, OPT, true, true, false), 1};
}

// (B_r 64, B_c 64, 4 waves): the key-split form of the 32-rows-per-wave kernel (KSPLIT = 2 in
// fa_fwd_kernel.hpp); registered under B_r / n_warps = 16 rows per wave, which is how configs find it
template <int DT, bool SWZ, bool EAGER, bool OPT, bool PIPE>
constexpr KernelEntry make_entry_ks() {
    using TR = FwdTraits<DT, 1, 4, 64, SWZ, EAGER, OPT, PIPE, true, false, 128, 2>;
    static_assert(TR::kBr == 64, "two row groups of 32");
    return KernelEntry{DT, 16, 4, 64, SWZ, EAGER, OPT, PIPE, 1, 0, 128, TR::kThreads, TR::kLdsBytes, 0,
                       (kernel_fn)&fa_fwd_kernel<DT, 1, 4, 64, SWZ, EAGER, OPT, PIPE, true, false, 128, 0, 2>, nullptr,
                       softmax_mode_of(false, OPT, EAGER, true, false), 0};
}

struct KernelTable {
    const KernelEntry *entries;
    int count;
};

}  // namespace fa
