// fa_fwd_kernel.hpp -- Flash-Attention-2 forward for CDNA4 / gfx950 (MI355X).
//
// Hand-written HIP; not a translation of the reference's CUDA.  It computes what
// /root/reference/src/include/forward_kernel.cuh:85-204 (flash_forward_kernel) and
// :19-83 (process_kv_block) compute -- O = softmax(Q K^T / sqrt(d)) V for one
// (batch, head, Q block) per workgroup, KV blocks visited last-to-first, raw-logit
// running max, base-2 exponent with c = rsqrt(d)*log2(e), P rounded RNE to the
// 16-bit type before P.V, l summed from fp32 P with the cross-lane reduction
// deferred to the epilogue (softmax.cuh:13-128) -- but is organised for wave64 MFMA:
//
//  * Both products are computed TRANSPOSED so every softmax statistic is lane-local:
//      S^T = K  Q^T   v_mfma_f32_32x32x16 (A = K tile rows from LDS, B = Q^T in VGPRs)
//      O^T = V^T P^T  v_mfma_f32_32x32x16 (A = V^T via ds_read_b64_tr_b16, B = P^T)
//    In the 32x32 C layout (col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5))
//    lane L then owns ONE query (column L&31) and 16 keys per 32-key tile; the other
//    16 keys sit in lane L^32.  Row max = in-register max + one v_permlane32_swap;
//    the rescale factor, m and l are per-lane scalars.
//  * The P^T B-operand needs, per lane, keys {8*(lane>>5) + j}; the C layout hands
//    the lane keys {4*(lane>>5) + (j&3) + 8*(j>>2)}.  Since a contraction index may
//    be permuted freely if both operands agree, V^T's A-operand is fetched with the
//    SAME key permutation (two transpose-reads per operand pick keys 4hi+0..3 and
//    8+4hi+0..3).  P therefore goes from the softmax registers straight into the
//    MFMA with just a v_cvt_pk -- no LDS round trip, no cross-lane shuffle.
//  * Q lives in VGPRs for the whole kernel (it IS the B operand layout: 16 B per
//    lane per 16-wide k step), loaded once straight from global memory.
//  * K and V tiles are DMA'd global->LDS (global_load_lds_dwordx4, 1 KiB per
//    wave-instruction), double-buffered, one barrier per KV tile.  The DMA writes
//    LDS lane-linearly, so both LDS images are produced by permuting the per-lane
//    SOURCE address:
//      K image  [key][128 d], 256 B rows, 16-B chunk index XOR (key & 15)  ->
//               conflict-free ds_read_b128 for the A operand (16-lane groups hit 16
//               distinct slots of the 256-B bank row);
//      V image  [key/8][d/32][8 keys][32 d] 512-B subtiles -> every
//               ds_read_b64_tr_b16 wave-instruction reads 512 contiguous bytes, all
//               32 reads of a tile are one base VGPR + immediate offsets.
//  * Workgroup ids are remapped so all Q blocks of one (batch, head) run on the
//    same XCD (block id mod 8) and share that XCD's L2 copy of the head's K/V.
//
// Template parameters select the device variant behind the reference's 13-field
// config (fa_capi.hip maps config -> variant; tools/generate_kernel_instantiations.py lists them):
//   DT      5 = fp16, 15 = bf16 (torch ScalarType codes)
//   QT      32-row Q tiles per wave (rows per wave = 32*QT)
//   NWAVES  wave64 wavefronts per workgroup  (B_r = 32*QT*NWAVES)
//   BC      keys per LDS tile (B_c)
//   SWZ     XOR-swizzled K image             (cfg.swizzled)
//   EAGER   prefetch next tile, 2 LDS buffers (cfg.eager_load_blocks)
//   OPT     first KV block skips the rescale  (cfg.optimized_softmax)
//   DMA     K/V tiles by global->LDS DMA (cfg.async_copy = 1, the cp.async analogue) or,
//           DMA = false, through registers: coalesced global_load_dwordx4 issued a visit
//           ahead, written to LDS with ds_write_b128 after the barrier that frees the stage
//   MASK    scope widener beyond the reference: seq_len need not be a multiple
//           of the tiles (keys >= seq_len masked, rows >= seq_len not stored) and an optional
//           causal mask (KV tiles above the diagonal are never visited)
//   PIPE    software-pipelined loop with two S accumulators: while the matrix pipe forms
//           S(i+1) = K(i+1) Q^T and then O += V(i) P(i), the vector ALU turns the finished
//           S(i) into P(i); every MFMA has ~4-5 VALU ops and 1-2 LDS operand reads pinned
//           beside it (cfg.mma_double_buffer_loads; 32 rows per wave, B_c <= 64)
//   ABL     0 in every shipped variant; tools/ablate.hip removes one cost at a time
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fa {

struct KernelArgs {
    const void *q;
    const void *k;
    const void *v;
    void *o;
    int64_t batch_stride;  // elements
    int64_t seq_stride;
    int64_t head_stride;
    int32_t seq_len;
    int32_t n_heads;
    int32_t n_bh;          // batch * heads
    int32_t n_q_blocks;
    int32_t n_kv_blocks;
    int32_t causal;        // MASK variants only: key j contributes to query i iff j <= i
#ifdef FA_TRACE
    unsigned long long *trace;  // tools/segment_timer.hip only: [wave][visit][8] s_memtime stamps
    int32_t trace_block;
    int32_t trace_visit;        // tools/trace64.hip: the one visit whose gap stamps are recorded
#endif
};

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define FA_LDS(type) __attribute__((address_space(3))) type
#define FA_DEV __device__ __forceinline__

template <bool B> struct BoolTag { static constexpr bool value = B; };
template <int I> struct IntTag { static constexpr int value = I; };
// f(IntTag<I>{}) for I = BEGIN .. END-1: unrolled by construction (a `#pragma unroll` loop whose
// unrolled size passes LLVM's pragma threshold is silently left rolled, and every register array
// it indexes then lives in scratch)
template <int BEGIN, int END, class F> __device__ __forceinline__ void static_for(F &&f) {
    if constexpr (BEGIN < END) {
        f(IntTag<BEGIN>{});
        static_for<BEGIN + 1, END>(f);
    }
}

template <int DT> struct Elem;

template <> struct Elem<15> {  // bf16
    typedef bf16x8 vec8;
    static FA_DEV f32x16 mfma(vec8 a, vec8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    static FA_DEV f32x4 mfma16(vec8 a, vec8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
    // Inline-asm MFMA forms with the register FILE of each operand chosen by hand (64-rows-per-
    // wave schedule): accumulator in AGPRs ("a") or VGPRs ("v"), B operand Q resident in AGPRs.
    // hipcc neither schedules nor hazard-pads these: see the call sites for the wait states.
    static FA_DEV void mfma_acc_a_q(f32x16 &acc, vec8 a, vec8 q) {  // acc(AGPR) += a * q(AGPR)
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "a"(q));
    }
    static FA_DEV void mfma_acc_a_q0(f32x16 &acc, vec8 a, vec8 q) {  // acc(AGPR) = a * q(AGPR)
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=a"(acc) : "v"(a), "a"(q));
    }
    static FA_DEV void mfma_acc_v_q(f32x16 &acc, vec8 a, vec8 q) {  // acc(VGPR) += a * q(AGPR)
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(q));
    }
    static FA_DEV void mfma_acc_v_q0(f32x16 &acc, vec8 a, vec8 q) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(acc) : "v"(a), "a"(q));
    }
    static FA_DEV void mfma_acc_a_p(f32x16 &acc, vec8 a, u32x4 p) {  // acc(AGPR) += a * p(VGPR, packed >= 1 step ago)
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(p));
    }
    static FA_DEV unsigned pack2(float x, float y) {  // RNE, low half = x
        typedef __bf16 pair_t __attribute__((ext_vector_type(2)));
        pair_t r;
        r[0] = (__bf16)x;
        r[1] = (__bf16)y;
        return __builtin_bit_cast(unsigned, r);
    }
    // RNE fp32 -> bf16 (v_cvt_pk_bf16_f32), load_store.cuh:345-349 semantics
    static FA_DEV vec8 pack8(const float *p) {
        vec8 r;
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = (__bf16)p[j];
        return r;
    }
};

template <> struct Elem<5> {  // fp16
    typedef f16x8 vec8;
    static FA_DEV f32x16 mfma(vec8 a, vec8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
    static FA_DEV f32x4 mfma16(vec8 a, vec8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
    // Inline-asm MFMA forms with the register FILE of each operand chosen by hand (64-rows-per-
    // wave schedule): accumulator in AGPRs ("a") or VGPRs ("v"), B operand Q resident in AGPRs.
    // hipcc neither schedules nor hazard-pads these: see the call sites for the wait states.
    static FA_DEV void mfma_acc_a_q(f32x16 &acc, vec8 a, vec8 q) {  // acc(AGPR) += a * q(AGPR)
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "a"(q));
    }
    static FA_DEV void mfma_acc_a_q0(f32x16 &acc, vec8 a, vec8 q) {  // acc(AGPR) = a * q(AGPR)
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=a"(acc) : "v"(a), "a"(q));
    }
    static FA_DEV void mfma_acc_v_q(f32x16 &acc, vec8 a, vec8 q) {  // acc(VGPR) += a * q(AGPR)
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(q));
    }
    static FA_DEV void mfma_acc_v_q0(f32x16 &acc, vec8 a, vec8 q) {
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=v"(acc) : "v"(a), "a"(q));
    }
    static FA_DEV void mfma_acc_a_p(f32x16 &acc, vec8 a, u32x4 p) {  // acc(AGPR) += a * p(VGPR, packed >= 1 step ago)
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(p));
    }
    static FA_DEV unsigned pack2(float x, float y) {  // RNE, low half = x
        typedef _Float16 pair_t __attribute__((ext_vector_type(2)));
        pair_t r;
        r[0] = (_Float16)x;
        r[1] = (_Float16)y;
        return __builtin_bit_cast(unsigned, r);
    }
    static FA_DEV vec8 pack8(const float *p) {
        vec8 r;
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = (_Float16)p[j];
        return r;
    }
};

// Single-instruction VALU forms for the hand-placed schedule: plain C++ lets hipcc SLP-pack adjacent
// f32 adds into v_pk_add_f32 and canonicalise fmaxf inputs, both slower beside MFMAs.
static FA_DEV float vmax3(float a, float b, float c) {
    float d;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
static FA_DEV float vmax2(float a, float b) {
    float d;
    asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
static FA_DEV float vadd(float a, float b) {
    float d;
    asm("v_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}

// max over the two lanes that share a query column (lane, lane^32).
static FA_DEV float pair_max(float x) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
static FA_DEV float pair_sum(float x) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// 16-byte-per-lane global -> LDS DMA (1 KiB per wave-instruction).  Written as inline
// asm on purpose: hipcc tracks the builtin form as an LDS write and puts a full
// `s_waitcnt vmcnt(0)` in front of the next ds_read of the same __shared__ array,
// which would serialise every prefetch with the compute it is meant to hide under.
// The kernel counts these loads itself: each wave executes dma_wait_all() before the
// barrier that publishes a stage (cdna_hip_programming.md 5.7 item 1).
// LDS destination = M0 (wave-uniform byte address) + lane * 16.
static FA_DEV void glds16(const void *gsrc, unsigned lds_dst_wave_uniform) {
    unsigned keep;  // M0 is compiler-reserved: save / restore it inside the statement
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst_wave_uniform)
                 : "memory");
}
// Same with the address split as SGPR base (wave-uniform: tensor + head + tile offset,
// computed on the scalar ALU) + 32-bit per-lane byte offset: no vector ALU work per piece.
static FA_DEV void glds16_sv(const void *base_wave_uniform, unsigned lane_byte_off,
                             unsigned lds_dst_wave_uniform) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(lane_byte_off), "s"(base_wave_uniform), "s"(lds_dst_wave_uniform)
                 : "memory");
}
// Variant that leaves M0 changed (3 instructions instead of 5).  Only for kernels in which hipcc
// itself never needs M0 (no LDS-DMA builtin, no s_movrel / sendmsg): the 64-row pinned schedule.
static FA_DEV void glds16_sv_m0(const void *base_wave_uniform, unsigned lane_byte_off,
                                unsigned lds_dst_wave_uniform) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :
                 : "v"(lane_byte_off), "s"(base_wave_uniform), "s"(lds_dst_wave_uniform)
                 : "memory");
}
static FA_DEV void dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// workgroup barrier that the compiler may not move LDS traffic across and that does
// not drain VMEM (in-flight DMA survives it)
static FA_DEV void wg_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#ifdef FA_TRACE
#define FA_STAMP(visit, k)                                                                  \
    do {                                                                                    \
        if (blockIdx.x == (unsigned)args.trace_block && visit < 64) {                       \
            unsigned long long t_;                                                          \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");      \
            if (lane == 0) args.trace[(wave * 64 + (visit)) * 8 + (k)] = t_;                \
        }                                                                                   \
    } while (0)
#else
#define FA_STAMP(visit, k) do { } while (0)
#endif

// Instruction-group pins for the LLVM scheduler (mask: 0x8 MFMA, 0x100 DS read).
// Left alone, hipcc feeds each MFMA from an LDS read issued one or two instructions
// earlier, so every MFMA eats the LDS latency (measured 57 cycles per MFMA instead of
// 32).  These sequences keep `depth` operand reads in flight ahead of the matrix pipe.
template <int N_MFMA, int READS_PER_MFMA, int DEPTH>
static FA_DEV void sched_mfma_fed_from_lds() {
    constexpr int kReads = N_MFMA * READS_PER_MFMA;
    constexpr int kPre = DEPTH < kReads ? DEPTH : kReads;
    __builtin_amdgcn_sched_group_barrier(0x100, kPre, 0);
#pragma unroll
    for (int i = 0; i < N_MFMA; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
        if (kPre + (i + 1) * READS_PER_MFMA <= kReads)
            __builtin_amdgcn_sched_group_barrier(0x100, READS_PER_MFMA, 0);
    }
}

static FA_DEV unsigned lds_addr(const char *p) {
    return (unsigned)(unsigned long long)(FA_LDS(const char) *)p;
}

template <int DT, int QT, int NWAVES, int BC, bool SWZ, bool EAGER, bool OPT, bool PIPE, bool DMA = true,
          bool MASK = false, int D = 128>
struct FwdTraits {
    static_assert(!PIPE || EAGER, "the pipelined loop needs both LDS stages");
    static_assert(DMA || EAGER, "register-staged tiles are only built double-buffered");
    static_assert(D == 128 || D == 64, "d_head 128 (the reference's scope) or 64 (widener)");
    static_assert(DMA || D == 128, "the register-staged transport is only built for d_head 128");
    static constexpr int kRowsPerWave = 32 * QT;
    static constexpr int kBr = kRowsPerWave * NWAVES;
    static constexpr int kBc = BC;
    static constexpr int kThreads = NWAVES * 64;
    static constexpr int kTileBytes = BC * 2 * D;               // one K or V tile
    // LDS ring depth per tensor: 4 for the 64-rows-per-wave schedule (tiles fetched 3 visits ahead)
    static constexpr int kStages = (QT == 2 && PIPE) ? 4 : (EAGER ? 2 : 1);
    static constexpr int kKvBytes = 2 * kStages * kTileBytes;   // K + V, all stages
    static constexpr int kOutBytes = kBr * 2 * D;               // O tile staged for the epilogue
    // The 64-rows-per-wave schedule is a PERSISTENT kernel: one workgroup per CU walks the
    // (batch*head, Q block) items and keeps the K/V tile stream running across item seams, so its
    // O staging (8 KB per wave, one 32-row Q tile at a time) lives beside the rings, not in them.
    static constexpr bool kPersistent = (QT == 2 && PIPE);
    static constexpr int kLdsBytes = kPersistent ? kKvBytes + NWAVES * 32 * 2 * D
                                                 : (kKvBytes > kOutBytes ? kKvBytes : kOutBytes);
};

// Filler plan of the 64-rows-per-wave schedule: what rides in the gap after MFMA g (g = 0..63 of a
// visit; 0..31 = QK^T of tile it+1, 32..63 = P.V of tile it).  Built at compile time so that
// placements can be compared by changing one function (tools/tune64.hip).
struct Plan64 {
    signed char exp_first[64], exp_n[64];  // softmax units (2 elements each), 32 per visit, in P.V order
    signed char max_first[64], max_n[64];  // row-max units over S(it+1), 32 per visit
    signed char dma[64];                   // DMA piece 0..7 (even = K, odd = V) or -1
    signed char tail[64];                  // end-of-visit chain step 1.. or 0
    signed char barrier[64];               // 1: the visit's counted DMA wait + workgroup barrier
};
// variant bits (tools/tune64.hip): 1 barrier at the visit top instead of inside the MFMA stream,
// 2 DMA pieces late in phase 2 instead of early in phase 1
constexpr Plan64 make_plan64(int variant, int n_phase1) {
    Plan64 p{};
    const bool bar_top = variant & 1, dma_late = variant & 2, masked = variant & 4;
    // masked: gaps 32..35 of a diagonal visit rewrite S(it+1) (causal mask) before its row max is
    // taken, so the 32 row-max units start 4 gaps later and the end-of-visit chain runs in 5 steps
    const int m0 = masked ? 4 : 0, odd0 = masked ? 11 : 9, mend = masked ? 27 : 24;
    int e = 0, m = 0, d = 0;
    for (int g = 0; g < 64; ++g) {
        const int h = g - 32;
        int ne = 0, nm = 0, dm = -1, tl = 0;
        if (g < 32) {
            // gaps g % 4 == 0 carry the operand wait + two K reads; gap 2 the barrier; the DMA
            // pieces follow it, one per four gaps
            if ((g & 3) == 0) {
                if (!dma_late && (bar_top || g >= 4)) dm = d++;
            } else if (e < n_phase1 && (bar_top || g != 2)) {
                const int slot = (g >> 2) * 3 + (g & 3) - 1;          // 0..23 over the gaps with g % 4 != 0
                if ((slot + 1) * n_phase1 / 24 > slot * n_phase1 / 24) ne = 1;
            }
        } else {
            if (!dma_late && d < 8 && h == 0) dm = d++;               // the eighth early piece
            if ((h & 1) && h <= 21 && e < 32) {                       // odd gaps up to 53: the rest of the units
                const int gaps_left = (21 - h) / 2 + 1;
                ne = (32 - e + gaps_left - 1) / gaps_left;            // 1, or 2 while behind
            }
            if (h >= m0 && h < mend) {
                if (!(h & 1)) nm = 2;                                 // even gaps (with the V reads): 2
                else if (h >= odd0) nm = 1;                           // late odd gaps: 1  -> 24 + 8 = 32
            }
            if (dma_late && h >= 24) dm = d++;
            if (masked) { if (h >= 27) tl = 10 + (h - 27); }          // merged chain steps 10..14
            else if (h >= 24) tl = h - 23;                            // chain steps 1..8
        }
        p.exp_first[g] = (signed char)e; p.exp_n[g] = (signed char)ne; e += ne;
        p.max_first[g] = (signed char)m; p.max_n[g] = (signed char)nm; m += nm;
        p.dma[g] = (signed char)dm; p.tail[g] = (signed char)tl;
        p.barrier[g] = (!bar_top && g == 2) ? 1 : 0;
    }
    return p;
}
constexpr bool plan64_ok(const Plan64 &p) {
    int e = 0, m = 0, d = 0, bar = -1;
    for (int g = 0; g < 64; ++g) {
        // P of 16-key slice s16 is consumed from gap 32 + 8 s16 on: its units must be >= 2 gaps older
        for (int u = p.exp_first[g]; u < p.exp_first[g] + p.exp_n[g]; ++u)
            if (g + 2 > 32 + 8 * (u >> 3)) return false;
        // S(it+1) tiles: nt = 0 last written at gap 29, nt = 1 at gap 31; read >= 2 MFMAs later
        for (int u = p.max_first[g]; u < p.max_first[g] + p.max_n[g]; ++u)
            if (g < ((u >> 4) ? 34 : 32)) return false;
        if ((p.tail[g] == 1 || p.tail[g] == 10) && m < 32) return false;
        if (p.barrier[g]) bar = g;
        if (p.dma[g] >= 0 && bar >= 0 && g <= bar) return false;      // DMA overwrites what the barrier frees
        if (p.barrier[g] && g >= 28) return false;                    // V(it+1) is first read at gap 30
        e += p.exp_n[g]; m += p.max_n[g]; d += p.dma[g] >= 0;
    }
    return e == 32 && m == 32 && d == 8;
}

// ---------------------------------------------------------------------------------
// The kernel.  d_head = 128 is the reference's scope (README.md:7-15); D = 64 is a widener.
// ---------------------------------------------------------------------------------
// ABL (tools/ablate.hip only; 0 in every shipped variant) removes one cost at a time to
// attribute cycles: 1 no v_exp, 2 no softmax VALU at all, 4 no LDS operand reads,
// 8 no barriers / DMA waits, 16 no DMA, 32 plain loads instead of DMA (data discarded);
// experiments: 64 s_setprio(1) around the MFMA clusters (-3 %), 128 operand prefetch 12 deep
// (no change).  Results are wrong by construction when ABL & 63 != 0.
template <int DT, int QT, int NWAVES, int BC, bool SWZ, bool EAGER, bool OPT, bool PIPE, bool DMA = true,
          bool MASK = false, int D = 128, int ABL = 0>
__global__ void
__launch_bounds__(NWAVES * 64, (QT == 1) ? 2 : 1)
fa_fwd_kernel(const KernelArgs args) {
    using E = Elem<DT>;
    using vec8 = typename E::vec8;
    using TR = FwdTraits<DT, QT, NWAVES, BC, SWZ, EAGER, OPT, PIPE, DMA, MASK, D>;
    constexpr int ROWB = 2 * D;              // bytes per K / V / O row (256, or 128 at d_head 64)
    constexpr int CPR = D / 8;               // 16-B chunks per row (16 / 8)
    constexpr int RPP = 64 / CPR;            // tile rows per 1-KiB DMA piece (4 / 8)
    constexpr int DSUB = D / 32;             // 32-wide d subtiles per key group of the V image
    // row -> XOR mask of the 16-B chunk index: 16 consecutive rows must land on 16 distinct
    // 16-B slots of the 256-B LDS bank row (d_head 64: two rows share a bank row)
    auto swz_of = [](int row) { return D == 128 ? (row & 15) : ((row >> 1) & 7); };
    constexpr int NT = BC / 32;              // 32-key tiles per LDS tile
    constexpr int KS = D / 16;               // k steps of the QK^T contraction
    constexpr int DTILES = D / 32;           // 32-wide d tiles of O^T
    constexpr int TILE = TR::kTileBytes;
    constexpr int N_DMA = BC / RPP;          // 1-KiB DMA pieces per K (or V) tile
    static_assert(N_DMA % NWAVES == 0, "tile/wave split");
    constexpr int DMA_PER_WAVE = N_DMA / NWAVES;
#ifdef FA_NO_SCHED
    constexpr bool SCHED = false;
#else
    constexpr bool SCHED = (QT == 1);
#endif

    extern __shared__ __attribute__((aligned(16))) char smem[];
    // LDS carve: K stage 0 | K stage 1 | V stage 0 | V stage 1
    constexpr int V_BASE = TR::kStages * TILE;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r31 = lane & 31;
    const int hi = lane >> 5;

    // ---- workgroup -> (batch*head, Q block); XCD-aware when n_bh % 8 == 0 --------
    const int nq = args.n_q_blocks;
    // item -> (batch*head, Q block).  Workgroups are dealt round-robin over the 8 XCDs, so items
    // congruent mod 8 share an L2: give each XCD whole heads (all Q blocks of a head read the same
    // K / V).  The persistent variant walks items blockIdx.x, + gridDim.x, ... with gridDim.x % 8 == 0.
    auto item_coords = [&](int bid, int &bh_out, int &qb_out) {
        if ((args.n_bh & 7) == 0) {
            const int xcd = bid & 7, local = bid >> 3;
            bh_out = (local / nq) * 8 + xcd;
            qb_out = local % nq;
        } else {
            bh_out = bid / nq;
            qb_out = bid % nq;
        }
    };
    int bh, qb;
    item_coords(blockIdx.x, bh, qb);
    if (MASK && args.causal && !TR::kPersistent) qb = nq - 1 - qb;  // longest rows first
    const int b = bh / args.n_heads, h = bh % args.n_heads;
    const int64_t ss = args.seq_stride;
    const int64_t head_off = (int64_t)b * args.batch_stride + (int64_t)h * args.head_stride;
    const uint16_t *Qg = (const uint16_t *)args.q + head_off;
    const uint16_t *Kg = (const uint16_t *)args.k + head_off;
    const uint16_t *Vg = (const uint16_t *)args.v + head_off;
    uint16_t *Og = (uint16_t *)args.o + head_off;

    // ---- per-lane DMA source offsets (elements), invariant over tiles ------------
    // piece i (wave-uniform) covers LDS chunks [64 i, 64 i + 64) of a tile.
    //   K: chunk p -> key p>>4, 16-B chunk (p&15) ^ (key&15)
    //   V: chunk p -> subtile p>>5 = (key>>3)*4 + (d>>5); inside: key&7 = (p&31)>>2,
    //      d&31 = (p&3)*8
    const int k_row_in_piece = lane / CPR;                                 // 0..RPP-1
    // swizzle of tile row RPP*i + k_row_in_piece; i = wave + NWAVES*j and RPP*NWAVES % 16 == 0
    const int k_swz = SWZ ? swz_of(RPP * wave + k_row_in_piece) : 0;
    const int64_t k_lane_off = (int64_t)k_row_in_piece * ss + (((lane & (CPR - 1)) ^ k_swz) << 3);
    const int v_sub_in_piece = lane >> 5;                                  // 0..1
    const int v_w = lane & 31;
    const int64_t v_lane_row = (v_w >> 2);                                 // key & 7
    const int v_lane_d = (v_w & 3) * 8;

    // KV blocks are visited last-to-first (forward_kernel.cuh:142,175-184): visit
    // index `it` is sequence block n_kv-1-it.
    // KV tiles this workgroup visits: all of them, or up to its last row's diagonal
    const int S_len = args.seq_len;
    const int wg_row0 = qb * TR::kBr;
    int n_kv_ = args.n_kv_blocks;
    if (MASK && args.causal) {
        const int last_row = (wg_row0 + TR::kBr < S_len ? wg_row0 + TR::kBr : S_len) - 1;
        const int need = last_row / BC + 1;
        n_kv_ = need < n_kv_ ? need : n_kv_;
    }
    const int n_kv = n_kv_;
    const unsigned smem_base = __builtin_amdgcn_readfirstlane(lds_addr(smem));
    // DMA addressing: SGPR base = head base + tile offset (scalar ALU), VGPR = 32-bit
    // per-lane byte offset of this wave's piece inside a tile (invariant over tiles).
    unsigned k_off[DMA_PER_WAVE], v_off[DMA_PER_WAVE];
#pragma unroll
    for (int j = 0; j < DMA_PER_WAVE; ++j) {
        const int i = wave + NWAVES * j;  // piece index, wave-uniform; keys 4i .. 4i+3
        k_off[j] = (unsigned)(((int64_t)(RPP * i) * ss + k_lane_off) * 2);
        const int sub = 2 * i + v_sub_in_piece;  // subtiles 2i, 2i+1
        v_off[j] = (unsigned)(((8 * (sub / DSUB) + v_lane_row) * ss + (sub % DSUB) * 32 + v_lane_d) * 2);
    }
    const int64_t tile_stride = (int64_t)BC * ss;  // elements between consecutive KV blocks
    f32x4 abl_dummy[2 * DMA_PER_WAVE];  // ABL & 32 only: landing registers of plain loads
    // MASK: rows of the last sequence block that lie beyond seq_len are fetched from the last
    // valid row instead (their logits are masked, their P is exactly 0).
    auto ragged_rows = [&](int it) -> int {  // valid rows of visit it's tile if it is ragged, else 0
        const int kv0 = (n_kv - 1 - it) * BC;
        return (MASK && kv0 + BC > S_len) ? S_len - kv0 : 0;
    };
    auto issue_k = [&](int it, int stage) {
        const uint16_t *base = Kg + (int64_t)(n_kv - 1 - it) * tile_stride;
        const unsigned kdst = smem_base + stage * TILE;
        if (ABL & 16) return;
        if (const int valid = ragged_rows(it)) {
#pragma unroll
            for (int j = 0; j < DMA_PER_WAVE; ++j) {
                int row = RPP * (wave + NWAVES * j) + k_row_in_piece;
                row = row < valid ? row : valid - 1;
                const unsigned off = (unsigned)(((int64_t)row * ss + (((lane & (CPR - 1)) ^ k_swz) << 3)) * 2);
                glds16_sv(base, off, kdst + (wave + NWAVES * j) * 1024);
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < DMA_PER_WAVE; ++j) {
            if (ABL & 32)
                asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(abl_dummy[j]) : "v"(k_off[j]), "s"(base) : "memory");
            else
                glds16_sv(base, k_off[j], kdst + (wave + NWAVES * j) * 1024);
        }
    };
    auto issue_v = [&](int it, int stage) {
        const uint16_t *base = Vg + (int64_t)(n_kv - 1 - it) * tile_stride;
        const unsigned vdst = smem_base + V_BASE + stage * TILE;
        if (ABL & 16) return;
        if (const int valid = ragged_rows(it)) {
#pragma unroll
            for (int j = 0; j < DMA_PER_WAVE; ++j) {
                const int sub = 2 * (wave + NWAVES * j) + v_sub_in_piece;
                int row = 8 * (sub / DSUB) + (int)v_lane_row;
                row = row < valid ? row : valid - 1;
                const unsigned off = (unsigned)(((int64_t)row * ss + (sub % DSUB) * 32 + v_lane_d) * 2);
                glds16_sv(base, off, vdst + (wave + NWAVES * j) * 1024);
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < DMA_PER_WAVE; ++j) {
            if (ABL & 32)
                asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(abl_dummy[DMA_PER_WAVE + j]) : "v"(v_off[j]), "s"(base) : "memory");
            else
                glds16_sv(base, v_off[j], vdst + (wave + NWAVES * j) * 1024);
        }
    };
    // Register-staged transport (DMA == false).  Piece i = 4 tile rows; lane L moves the
    // 16-B chunk (row 4i + L/16, chunk L%16): fully coalesced 256-B rows from global, and a
    // per-lane LDS address builds the same K / V images the DMA path builds by permuting
    // its source.  One set of landing registers per tile kind lives across a whole visit.
    f32x4 kreg[DMA_PER_WAVE], vreg[DMA_PER_WAVE];
    const int st_r = lane >> 4, st_c = lane & 15;
    const unsigned st_goff = (unsigned)((((int64_t)(4 * wave + st_r)) * ss + st_c * 8) * 2);
    const unsigned st_gstep = (unsigned)(((int64_t)(4 * NWAVES)) * ss * 2);  // bytes between a wave's pieces
    const int st_krow = 4 * (wave & 3) + st_r;                               // (tile row) & 15
    const unsigned st_kwr = wave * 1024 + st_r * 256 + ((st_c ^ (SWZ ? st_krow : 0)) << 4);
    const unsigned st_vwr = (wave >> 1) * 2048 + ((wave & 1) * 4 + st_r) * 64 + (st_c >> 2) * 512 + (st_c & 3) * 16;
    auto load_tile = [&](const uint16_t *tensor, int it, f32x4 (&reg)[DMA_PER_WAVE]) {
        const char *tile = (const char *)(tensor + (int64_t)(n_kv - 1 - it) * tile_stride);
        if (const int valid = ragged_rows(it)) {
#pragma unroll
            for (int j = 0; j < DMA_PER_WAVE; ++j) {
                int row = 4 * (wave + NWAVES * j) + st_r;
                row = row < valid ? row : valid - 1;
                reg[j] = *(const f32x4 *)(tile + ((int64_t)row * ss + st_c * 8) * 2);
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < DMA_PER_WAVE; ++j) reg[j] = *(const f32x4 *)(tile + st_goff + j * st_gstep);
    };
    auto load_k = [&](int it) { load_tile(Kg, it, kreg); };
    auto load_v = [&](int it) { load_tile(Vg, it, vreg); };
    auto store_k = [&](int stage) {
        char *dst = smem + stage * TILE + st_kwr;
#pragma unroll
        for (int j = 0; j < DMA_PER_WAVE; ++j) *(f32x4 *)(dst + j * NWAVES * 1024) = kreg[j];
    };
    auto store_v = [&](int stage) {
        char *dst = smem + V_BASE + stage * TILE + st_vwr;
#pragma unroll
        for (int j = 0; j < DMA_PER_WAVE; ++j) *(f32x4 *)(dst + j * (NWAVES / 2) * 2048) = vreg[j];
    };
    auto dma_wait = [&]() { if (DMA && !(ABL & 8)) dma_wait_all(); };
    auto barrier = [&]() { if (!(ABL & 8)) wg_barrier(); };
    auto wait_and_barrier = [&]() {
        dma_wait();
        barrier();
    };

#ifdef FA_TRACE
    if (blockIdx.x == (unsigned)args.trace_block && lane == 0)
        args.trace[(wave * 64 + 63) * 8 + 7] = __builtin_amdgcn_s_getreg((31 << 11) | 4);  // HW_REG_HW_ID
#endif
    // ---- prologue: first tiles in flight, then Q -> VGPRs --------------------------
    if (EAGER && DMA) {
        issue_k(0, 0);
        if (!TR::kPersistent) issue_v(0, 0);  // the persistent kernel asks for Q before V(0): see its prologue
    }
    if (!DMA) {
        load_k(0);
        load_v(0);
    }

    vec8 Qr[QT][KS];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        int64_t row = (int64_t)qb * TR::kBr + wave * TR::kRowsPerWave + qt * 32 + r31;
        if (MASK && row >= S_len) row = S_len - 1;  // rows past the end are computed, never stored
        const uint16_t *qp = Qg + row * ss + hi * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if constexpr (TR::kPersistent)  // straight into the accumulator file; waited for by hand below
                asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=a"(Qr[qt][ks]) : "v"(qp), "i"(ks * 32) : "memory");
            else
                Qr[qt][ks] = *(const vec8 *)(qp + ks * 16);
        }
    }

    // forward_kernel.cuh:150-151 (fp32 product of rsqrt(d) and log2 e)
    const float c = (float)((double)(1.0f / __builtin_sqrtf((float)D)) * 1.4426950408889634074);

    f32x16 O[QT][DTILES];
    float m[QT], l[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        m[qt] = -__builtin_inff();
        l[qt] = 0.0f;
#pragma unroll
        for (int t = 0; t < DTILES; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) O[qt][t][r] = 0.0f;
    }

    // per-lane LDS read offsets
    //   K A-operand: row 32*nt + r31, chunk (2*ks + hi) ^ (r31 & 15)
    const int ka_swz = SWZ ? swz_of(r31) : 0;
    const int ka_base = r31 * ROWB;
    //   V^T A-operand (transpose read): see header comment
    const int li = lane & 15, lg = lane >> 4;
    const int va_base = (4 * (lg >> 1) + (li >> 2)) * 64 + (lg & 1) * 32 + (li & 3) * 8;

    // ---- MASK: logits of keys >= seq_len, or above the causal diagonal, become -inf ---------
    const int wave_row0 = wg_row0 + wave * TR::kRowsPerWave;
    auto tile_needs_mask = [&](int it) -> bool {  // wave-uniform
        const int kv0 = (n_kv - 1 - it) * BC;
        return MASK && (kv0 + BC > S_len || (args.causal && kv0 + BC - 1 > wave_row0));
    };
    auto mask_S = [&](f32x16 (&S)[QT][NT], int it) {
        const int kv0 = (n_kv - 1 - it) * BC;
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            const int q_row = wave_row0 + qt * 32 + r31;
            const int limit = args.causal ? (q_row < S_len - 1 ? q_row : S_len - 1) : S_len - 1;  // last live key
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kv0 + 32 * nt + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    S[qt][nt][r] = key > limit ? -__builtin_inff() : S[qt][nt][r];
                }
        }
    };
    // a row whose keys were all masked so far has m = -inf: exponentiate against 0 instead
    auto finite_or_zero = [&](float mval) { return (MASK && mval == -__builtin_inff()) ? 0.0f : mval; };

    // ---- S^T = K Q^T ------------------------------------------------------------
    auto qk = [&](int stage, f32x16 (&S)[QT][NT]) {
        const char *kt = smem + stage * TILE;
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) S[qt][nt][r] = 0.0f;
        // ks outer / key-tile inner: consecutive MFMAs accumulate into different tiles
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int off = nt * 32 * ROWB + ka_base + (((2 * ks + hi) ^ ka_swz) << 4);
                const vec8 a = (ABL & 4) ? Qr[0][(ks + nt) % KS] : *(const vec8 *)(kt + off);
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) S[qt][nt] = E::mfma(a, Qr[qt][ks], S[qt][nt]);
            }
        }
        if (SCHED && !PIPE) sched_mfma_fed_from_lds<NT * KS * QT, 1, (ABL & 128) ? 12 : 8>();
    };

    // ---- online softmax, lane-local (softmax.cuh:85-105) --------------------------
    // Updates m, l; returns P (16-bit, MFMA B-operand order) and the rescale factor
    // alpha = exp2((m_prev - m_new) c) that (l, O) must be multiplied by BEFORE P.V
    // of this tile is accumulated (scale_l_O, softmax.cuh:36-49).  l is rescaled here;
    // O by rescale_O() -- skipped when alpha == 1 in every lane (multiplying by 1.0f is
    // the identity, so the skip is bit-exact).
    auto softmax = [&](f32x16 (&S)[QT][NT], vec8 (&Pb)[QT][NT][2], float (&alpha)[QT], auto first_tag) {
        constexpr bool FIRST = decltype(first_tag)::value;
        if (ABL & 2) {
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                alpha[qt] = 1.0f;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    float p[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) p[r] = S[qt][nt][r];
                    Pb[qt][nt][0] = E::pack8(p);
                    Pb[qt][nt][1] = E::pack8(p + 8);
                }
            }
            return;
        }
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            float mx = S[qt][0][0];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, S[qt][nt][r]);
            mx = pair_max(mx);
            float m_new;
            if (FIRST && OPT) {
                m_new = mx;
                alpha[qt] = 1.0f;
            } else {
                m_new = fmaxf(m[qt], mx);
                alpha[qt] = __builtin_amdgcn_exp2f((m[qt] - finite_or_zero(m_new)) * c);
                l[qt] *= alpha[qt];
            }
            m[qt] = m_new;
            const float neg_msc = -(finite_or_zero(m_new) * c);
            float rowsum = 0.0f;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                float p[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    // exponentiate_tensor, softmax.cuh:51-64: exp2(s*c - m*c)
                    p[r] = __builtin_fmaf(S[qt][nt][r], c, neg_msc);
                    if (!(ABL & 1)) p[r] = __builtin_amdgcn_exp2f(p[r]);
                    rowsum += p[r];  // fp32 P, before rounding (softmax.cuh:66-83)
                }
                Pb[qt][nt][0] = E::pack8(p);
                Pb[qt][nt][1] = E::pack8(p + 8);
            }
            l[qt] = (FIRST && OPT) ? rowsum : l[qt] + rowsum;
        }
    };

    auto rescale_O = [&](const float (&alpha)[QT]) {
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            if (__all(alpha[qt] == 1.0f)) continue;  // wave-uniform
#pragma unroll
            for (int t = 0; t < DTILES; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) O[qt][t][r] *= alpha[qt];
        }
    };

    // ---- O^T += V^T P^T ------------------------------------------------------------
    auto pv = [&](int stage, const vec8 (&Pb)[QT][NT][2]) {
        const char *vt = smem + V_BASE + stage * TILE;
        // 16-key slice outer / d tile inner: consecutive MFMAs hit different accumulators
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
#pragma unroll
                for (int t = 0; t < DTILES; ++t) {
                    const int s16 = 2 * nt + half;
                    const char *vp = vt + va_base + s16 * (DSUB * 1024) + t * 512;
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((FA_LDS(s16x4) *)(vp));
                    const s16x4 up =
                        __builtin_amdgcn_ds_read_tr16_b64_v4i16((FA_LDS(s16x4) *)(vp + DSUB * 512));
                    s16x8 av;
                    av.lo = lo;
                    av.hi = up;
                    const vec8 a = (ABL & 4) ? Qr[0][(t + s16) % KS] : __builtin_bit_cast(vec8, av);
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt)
                        O[qt][t] = E::mfma(a, Pb[qt][nt][half], O[qt][t]);
                }
            }
        }
        if (SCHED && !PIPE) sched_mfma_fed_from_lds<DTILES * NT * 2 * QT, 2, (ABL & 128) ? 12 : 8>();
    };

    using TrueTag = BoolTag<true>;
    using FalseTag = BoolTag<false>;

    if constexpr (QT == 2 && PIPE) {
        // ---- 64 rows per wave, one wave per SIMD, hand-placed registers and order ----------
        // Each K / V operand read from LDS feeds TWO MFMAs (the wave's two 32-row Q tiles), which
        // halves LDS traffic, DMA issue and barriers per MFMA.  512 registers per lane, by file:
        //   AGPR  O (128) | Q (64)
        //   VGPR  two S tiles (128) | P (32) | operand ring (16) | softmax temporaries
        // All MFMAs are inline asm so that O and Q never leave the accumulator file; hipcc does
        // not schedule or hazard-pad them, so the stream is pinned gap by gap (one MFMA + its
        // fillers, then sched_barrier(0)) and the wait states are kept by distance:
        //   * S(it+1) is accumulated in phase 1 and first read (row max) >= 2 MFMAs later;
        //   * a packed P operand is consumed >= 2 gaps after its v_cvt_pk;
        //   * O is read by VALU only in the rare rescale and in the epilogue, behind s_nop pads.
        // One wave per SIMD hides about five single-issue instructions per 32-cycle MFMA
        // (MI355X_MICROARCH.md, per-instruction constants), so the softmax of tile `it` is cut
        // into 32 two-element units {2 fma, 2 exp2, 2 add, 1 pack} and dealt over the gaps by
        // a compile-time plan (Plan64):
        //   phase 1 (32 MFMAs, S(it+1) = K(it+1) Q^T): K operand reads, most of the units
        //   phase 2 (32 MFMAs, O += V(it) P(it)):      V operand reads, the other units, the row max
        //            of S(it+1), the 8 DMA pieces of the tiles three visits ahead, the m / rescale test
        // K and V each ring through 4 LDS stages (128 KB; the 512-register waves allow one
        // workgroup per CU anyway).  A tile is requested three visits before it is read and must have
        // landed two visits after the request: the wait in front of the per-visit barrier is
        // COUNTED (vmcnt(8): the youngest visit's pieces stay in flight), so an HBM-latency fetch
        // does not stall the matrix pipe, and the barrier publishes a tile one visit early, which
        // lets the last gaps of a visit prefetch the next visit's first operands.
        //
        // Rescaling is lazy: O and l stay relative to a reference max m that is only moved (and
        // O, l multiplied by 2^((m_old - m_new) c)) when some row's max rose by more than
        // TAU / c logit units, so P <= 2^TAU.  The result is the same real number as the
        // reference's eager rescale (softmax.cuh:36-49); only the rounding point of P differs,
        // with the same relative error.  With O in the accumulator file a rescale costs ~200
        // issue slots per Q tile, and for random data some row of 32 finds a new max in most tiles.
        static_assert(DMA && D == 128 && BC == 64 && NT == 2 && NWAVES == 4, "64-row pinned schedule");
        static_assert(TR::kStages == 4, "ring depth");
        constexpr float TAU = 8.0f;
        constexpr Plan64 plan = make_plan64(((ABL >> 8) & 3) | (MASK ? 4 : 0), (ABL & 1024) ? 20 : 22);
        static_assert(plan64_ok(plan), "filler plan violates a wait-state distance");
        f32x16 Sa[2][NT], Sb[2][NT];
        u32x4 Pw[2][4] = {};     // P[qt][16-key slice]: B operand of O^T += V^T P^T
        float neg_msc[2];        // -(m c)
        float m_pend[2];         // candidate reference max found during the previous visit
        unsigned resc_any = 0;   // bit qt: Q tile qt moves its reference max at the next visit's top
        auto k_frag = [&](const char *kt, int step) -> vec8 {  // step = 2*ks + nt
            const int ks = step >> 1, nt = step & 1;
            return *(const vec8 *)(kt + nt * 32 * ROWB + ka_base + (((2 * ks + hi) ^ ka_swz) << 4));
        };
        auto v_frag = [&](const char *vt, int step) -> vec8 {  // step = 4*s16 + t
            const int s16 = step >> 2, t = step & 3;
            const char *vp = vt + va_base + s16 * (DSUB * 1024) + t * 512;
            s16x8 av;
            av.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((FA_LDS(s16x4) *)(vp));
            av.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((FA_LDS(s16x4) *)(vp + DSUB * 512));
            return __builtin_bit_cast(vec8, av);
        };
        auto qk_mfma = [&](auto &S, int step, int qt, vec8 a) {
            const int ks = step >> 1, nt = step & 1;
            if (ks == 0) E::mfma_acc_v_q0(S[qt][nt], a, Qr[qt][ks]);
            else E::mfma_acc_v_q(S[qt][nt], a, Qr[qt][ks]);
        };
        // ---- persistent walk over items; the K / V tile stream runs on across item seams --------
        // This workgroup serves items blockIdx.x, + gridDim.x, ...  Tiles are numbered along the
        // walk: visit index j of the current item for j < n_kv, visit index j - n_kv of the NEXT item
        // beyond (n_kv % 4 == 0, so a tile's ring stage is j & 3 either way).  The last visits of an
        // item therefore request the next item's first tiles, its last visit forms the next item's
        // S(0) with the next item's Q (loaded straight into the Q AGPRs one visit earlier), and a
        // seam costs the O epilogue only.  After the last item the "next" item is the item itself:
        // the re-fetched tiles land in stages nobody reads.
        const int n_items = args.n_bh * nq;
        int item = blockIdx.x;
        const uint16_t *Kc = Kg, *Vc = Vg;   // current item
        uint16_t *Oc = Og;
        int qb_c = qb;
        const uint16_t *Kn = Kg, *Vn = Vg, *Qn = Qg;  // next item (set per item below)
        uint16_t *On = Og;
        int qb_n = qb;
        bool has_next = false;
        // causal (MASK variants): an item visits the tiles up to its diagonal only, 4 (qb + 1) of them
        // -- still a multiple of the ring depth, so the stage arithmetic along the walk holds
        const bool causal = MASK && args.causal;
        int nkc = n_kv, nkn = n_kv;  // tiles of the current / next item
        auto tile_g = [&](const uint16_t *cur, const uint16_t *nxt, int j) {
            return j < nkc ? cur + (int64_t)(nkc - 1 - j) * tile_stride
                           : nxt + (int64_t)(nkn - 1 - (j - nkc)) * tile_stride;
        };
        // MASK: logits above the causal diagonal become -inf.  `tile` counts from the start of the
        // sequence, `qb_rows` is the Q block whose rows the S tile belongs to.  A wave's 64 rows meet
        // the diagonal in exactly one 64-key tile; tiles beyond it are masked whole.
        auto mask_tile = [&](auto &S, int tile, int qb_rows) {
            if constexpr (MASK) {
                const int d = tile - (4 * qb_rows + wave);
                if (causal && d >= 0) {  // wave-uniform
#pragma unroll
                    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                const int key = 32 * nt + (r & 3) + 8 * (r >> 2) + 4 * hi, row = 32 * qt + r31;
                                S[qt][nt][r] = (d > 0 || key > row) ? -__builtin_inff() : S[qt][nt][r];
                            }
                }
            }
        };
        auto dma_k = [&](const uint16_t *src, int stage) {
#pragma unroll
            for (int j = 0; j < DMA_PER_WAVE; ++j)
                glds16_sv_m0(MASK ? src + j * (16 * ss) : src, k_off[MASK ? 0 : j],
                             smem_base + stage * TILE + (wave + NWAVES * j) * 1024);
        };
        auto dma_v = [&](const uint16_t *src, int stage) {
#pragma unroll
            for (int j = 0; j < DMA_PER_WAVE; ++j)
                glds16_sv_m0(MASK ? src + j * (16 * ss) : src, v_off[MASK ? 0 : j],
                             smem_base + V_BASE + stage * TILE + (wave + NWAVES * j) * 1024);
        };
        const uint16_t *kq = nullptr, *vq = nullptr;  // next K / V tile to request (set per item below)
        vec8 ring[4];  // operand ring: slot u % 4, rewritten two steps after the MFMAs that read it
        vec8 Qr2[2][KS];  // the next item's Q (AGPRs), requested during the item's first visit
        float mraw[2];   // row max of the S tile formed by the last visit (the next item's S(0))
        bool seam = false;  // the first two visits after a seam: the epilogue's stores are in flight
        // next item's Q rows -> the Q AGPRs.  Plain asm loads: hipcc does not count them; the wait
        // is the vmcnt(0) at the top of the item's last visit.
        auto load_q_next = [&](auto piece_tag, vec8 &dst) {  // piece = 8*qt + ks, dst = Qr[qt][ks]
            constexpr int piece = decltype(piece_tag)::value, qt = piece >> 3, ks = piece & 7;
            const int64_t row = (int64_t)qb_n * TR::kBr + wave * TR::kRowsPerWave + qt * 32 + r31;
            const uint16_t *qp = Qn + row * ss + hi * 8;
            asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=a"(dst) : "v"(qp), "i"(ks * 32) : "memory");
        };
        auto visit = [&](int it, auto &S_cur, auto &S_nxt, auto r_tag) {
            constexpr int R = decltype(r_tag)::value;  // it & 3
#ifdef FA_TRACE
            unsigned long long ts[20];
            asm volatile("s_memtime %0" : "=s"(ts[0]));
#endif
            // the visit's synchronisation point: K(it+2), V(it+1) landed (requested two visits ago;
            // K(it+1), V(it) were published by the previous barrier), the 8 youngest pieces may fly
            // on; behind it every wave has finished visit it-1, whose K / V stages the DMA of this
            // visit overwrites.  In the default plan it sits two MFMAs into the visit, after the
            // gap-0 lgkmcnt(0) that retires this wave's last LDS reads of visit it-1.
            auto sync_point = [&]() {
                if (ABL & 8) return;
                // what may still be in flight behind the pieces this barrier publishes: this visit's
                // predecessor's 8 pieces, plus -- early in an item -- the 16 stores of the previous
                // item's epilogue and the 16 loads of the next item's Q
                int allow = 8;
                if constexpr (R == 0) allow = (it == 0 && seam) ? 24 : 8;
                if constexpr (R == 1) allow = (it == 1) ? 8 + (seam ? 16 : 0) + (has_next ? 16 : 0) : 8;
                if constexpr (R == 2) allow = (it == 2 && has_next) ? 24 : 8;
                if (allow == 8) asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
                else if (allow == 24) asm volatile("s_waitcnt vmcnt(24)\n\ts_barrier" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(40)\n\ts_barrier" ::: "memory");
            };
            if constexpr (R == 3) {
                // last visit of an item forms the next item's S(0): swap the next item's Q in
                if (it + 1 == nkc && has_next) {
                    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");  // the Q loads (visit 0) are older than 16 pieces
#pragma unroll
                    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
                        for (int ks = 0; ks < KS; ++ks) {
                            asm volatile("" : "+a"(Qr2[qt][ks]));  // value defined by the asm loads
                            Qr[qt][ks] = Qr2[qt][ks];
                        }
                    asm volatile("s_nop 3" ::: "memory");  // accvgpr write -> MFMA operand
                }
            }
            if constexpr (plan.barrier[2] == 0) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                sync_point();
            }
            const unsigned kdst = smem_base + R * TILE + wave * 1024;                        // K(it+4) -> stage of K(it)
            const unsigned vdst = smem_base + V_BASE + ((R + 3) & 3) * TILE + wave * 1024;   // V(it+3) -> stage of V(it-1)
            if (resc_any) {  // wave-uniform, rare: move the reference max of one or both Q tiles
                asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");  // MFMA D (O) -> VALU read
#pragma unroll
                for (int qt = 0; qt < 2; ++qt) {
                    if (!(resc_any & (1u << qt))) continue;
                    const float alpha = __builtin_amdgcn_exp2f((m[qt] - m_pend[qt]) * c);
                    m[qt] = m_pend[qt];
                    neg_msc[qt] = -(finite_or_zero(m[qt]) * c);
                    l[qt] *= alpha;
#pragma unroll
                    for (int t = 0; t < DTILES; ++t)
#pragma unroll
                        for (int r = 0; r < 16; ++r) O[qt][t][r] *= alpha;
                }
            }
            const char *kt = smem + ((R + 1) & 3) * TILE;
            const char *vt = smem + V_BASE + R * TILE;
            float rs[2][2] = {{0.0f, 0.0f}, {0.0f, 0.0f}};
            float vm[2][2], m_new[2];
            unsigned any01 = 0;
            auto exp_unit = [&](int u) {  // u = 8*s16 + 2*j + qt: in the order P.V consumes P
                if constexpr (ABL & 2) return;
                const int qt = u & 1, j = (u >> 1) & 3, s16 = u >> 3, r = 8 * (s16 & 1) + 2 * j;
                // exp2(s c - m c), softmax.cuh:51-64.  Scalar f32 forms on purpose: v_pk_fma_f32 /
                // v_pk_add_f32 here measured -6 % / -12 %; splitting the unit into stages over three
                // gaps (no dependent pair inside a gap) measured -1.5 %.
                float p0 = __builtin_fmaf(S_cur[qt][s16 >> 1][r], c, neg_msc[qt]);
                float p1 = __builtin_fmaf(S_cur[qt][s16 >> 1][r + 1], c, neg_msc[qt]);
                if (!(ABL & 1)) {
                    p0 = __builtin_amdgcn_exp2f(p0);
                    p1 = __builtin_amdgcn_exp2f(p1);
                }
                rs[qt][0] += p0;  // fp32 P, before rounding (softmax.cuh:66-83)
                rs[qt][1] += p1;
                // pin the adds to this gap: hipcc otherwise sinks the whole row-sum chain (and keeps
                // every p alive) to the first use of l, behind the next visit's barrier
                asm volatile("" : "+v"(rs[qt][0]), "+v"(rs[qt][1]));
                unsigned pk = E::pack2(p0, p1);
                // ... and the pack: sunk below a branch of the stream it would sit right in front of the
                // MFMA that reads it, which hipcc does not pad (the MFMAs are opaque asm)
                asm volatile("" : "+v"(pk));
                Pw[qt][s16][j] = pk;
            };
            auto max_unit = [&](int u) {  // u = 0..31: tile (nt = u>>4, qt = (u>>3)&1), elements 2(u&7), +1
                const int nt = u >> 4, qt = (u >> 3) & 1, e = 2 * (u & 7), a = u & 1;  // two chains per Q tile
                if constexpr (ABL & 2) { vm[qt][a] = 0.0f; return; }
                // asm forms: fmaxf() on MFMA results makes hipcc canonicalise both inputs first
                if ((u & 7) < 2 && nt == 0) vm[qt][a] = vmax2(S_nxt[qt][nt][e], S_nxt[qt][nt][e + 1]);
                else vm[qt][a] = vmax3(vm[qt][a], S_nxt[qt][nt][e], S_nxt[qt][nt][e + 1]);
                asm volatile("" : "+v"(vm[qt][a]));  // pinned to its gap (S_nxt is rewritten next visit)
            };
            auto lane_pair_max = [&](float x) {
                auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
                return vmax2(__uint_as_float(r[0]), __uint_as_float(r[1]));
            };
            auto tail_unit = [&](int k) {  // end-of-visit chain, one step per gap (pinned by volatile asm)
                if (k == 1) {
                    vm[0][0] = vmax2(vm[0][0], vm[0][1]);
                    vm[1][0] = vmax2(vm[1][0], vm[1][1]);
                    asm volatile("" : "+v"(vm[0][0]), "+v"(vm[1][0]));
                }
                if (k == 2) { vm[0][0] = lane_pair_max(vm[0][0]); asm volatile("" : "+v"(vm[0][0])); }
                if (k == 3) { vm[1][0] = lane_pair_max(vm[1][0]); asm volatile("" : "+v"(vm[1][0])); }
                if (k == 4) {  // candidate reference max
                    mraw[0] = vm[0][0];
                    mraw[1] = vm[1][0];
                    m_new[0] = vmax2(m[0], vm[0][0]);
                    m_new[1] = vmax2(m[1], vm[1][0]);
                    asm volatile("" : "+v"(m_new[0]), "+v"(m_new[1]));
                }
                if (k == 5) {
                    l[0] += rs[0][0] + rs[0][1];
                    l[1] += rs[1][0] + rs[1][1];
                    asm volatile("" : "+v"(l[0]), "+v"(l[1]));
                }
                if (k == 6 || k == 7) {  // did some row's max rise by more than TAU / c?
                    const int qt = k - 6;
                    m_pend[qt] = m_new[qt];
                    float rise = (m_new[qt] - m[qt]) * c;
                    asm volatile("" : "+v"(rise));  // (an "s" pin would make hipcc treat the flag as divergent)
                    any01 |= (__ballot(rise > TAU) != 0 ? 1u : 0u) << qt;
                }
                if (k == 8) {  // next tiles to request (scalar ALU)
                    resc_any = any01;
                    // visit it+1 requests K(it+5), V(it+4); the stream wraps into the next item
                    kq = (it + 5 == nkc) ? Kn + (int64_t)(nkn - 1) * tile_stride : kq - tile_stride;
                    vq = (it + 4 == nkc) ? Vn + (int64_t)(nkn - 1) * tile_stride : vq - tile_stride;
                    if constexpr (R == 1) seam = false;

                }
            };
            auto tail_step = [&](int k) {  // plan step: 1..8 one unit each; 10..14 the masked plan's merged steps
                if (k < 10) tail_unit(k);
                if (k == 10) tail_unit(1);
                if (k == 11) { tail_unit(2); tail_unit(3); }
                if (k == 12) { tail_unit(4); tail_unit(5); }
                if (k == 13) { tail_unit(6); tail_unit(7); }
                if (k == 14) tail_unit(8);
            };
            // operand u of the visit: 16 K fragments, 16 V fragments, then the first two K fragments
            // of the NEXT visit (its tile was published by this visit's barrier), so that no LDS
            // latency is exposed at the visit seam
            const char *kt_next = smem + ((R + 2) & 3) * TILE;
            auto operand = [&](int u) -> vec8 {
                if constexpr (ABL & 4) return __builtin_bit_cast(vec8, Pw[u & 1][(u >> 1) & 3]);
                return u < 16 ? k_frag(kt, u) : (u < 32 ? v_frag(vt, u - 16) : k_frag(kt_next, u - 32));
            };
            static_for<0, 64>([&](auto gap_tag) {
                constexpr int g = decltype(gap_tag)::value;
                constexpr int step = g >> 1, qt = g & 1;
                // An MFMA reads its A / B registers for a few cycles after it issues, and hipcc -- to
                // which the MFMAs are opaque asm -- is free to hand a register that just died to the very
                // next VALU instruction (seen: the pair-max temporary landing in the A operand of the
                // MFMA in front of it; one-ulp run-to-run differences that an s_nop 7 behind every MFMA
                // removed).  So every operand is kept alive until the NEXT MFMA has issued: an empty asm
                // that names it, placed behind that MFMA (volatile asm statements keep their order).
                vec8 prev_a = ring[(step + 3) % 4];  // A operand of the previous step (its slot is reloaded below)
                if constexpr (qt == 0 && (step & 1) == 0) {  // operands in pairs: one counted wait per two steps
                    __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0): operands step, step+1 landed
#ifdef FA_TRACE
                    asm volatile("s_memtime %0" : "=s"(ts[2 + step / 2]));
#endif
                    ring[(step + 2) % 4] = operand(step + 2);
                    ring[(step + 3) % 4] = operand(step + 3);
                }
                if constexpr (g < 32) {
                    qk_mfma(S_nxt, step, qt, ring[step % 4]);
                } else {
                    constexpr int s2 = step - 16, s16 = s2 >> 2, t = s2 & 3;
                    E::mfma_acc_a_p(O[qt][t], ring[step % 4], Pw[qt][s16]);
                }
                if constexpr (qt == 0) asm volatile("" ::"v"(prev_a));
                if constexpr (g >= 33) {
                    constexpr int pg = g - 1, ps2 = (pg >> 1) - 16;
                    asm volatile("" ::"v"(Pw[pg & 1][ps2 >> 2]));  // B operand of the previous P.V MFMA
                }
                if constexpr (g == 0) asm volatile("" ::"v"(Pw[1][3]));  // ... of the previous visit's last one
                if constexpr (plan.barrier[g] != 0) sync_point();
                if constexpr (MASK && g == 34) {
                    // S(it+1) is complete (last written at gap 31): causal mask, before its row max.
                    // The last visit's S tile is the NEXT item's S(0).
                    if (it + 1 < nkc) mask_tile(S_nxt, nkc - 2 - it, qb_c);
                    else mask_tile(S_nxt, nkn - 1, qb_n);
                }
                if constexpr (R == 0 && g == 33) {
                    // first visit of an item: request the NEXT item's Q rows into the spare Q set
                    // (64 of the AGPRs are otherwise unused); they are swapped in at the top of
                    // this item's last visit, several visits after they have landed
                    if (it == 0 && has_next) {
                        static_for<0, 16>([&](auto i) {
                            constexpr int pc = decltype(i)::value;
                            load_q_next(IntTag<pc>{}, Qr2[pc >> 3][pc & 7]);
                        });
                    }
                }
                if constexpr (plan.dma[g] >= 0 && !(ABL & 16)) {  // one 1-KiB DMA piece
                    constexpr int j = plan.dma[g] >> 1;
                    // per-piece lane offsets: 6 more VGPRs than one offset + a scalar piece stride (piece j
                    // of a wave starts 16 rows below piece j-1), but 16 fewer SALU instructions per
                    // visit (+0.5 %).  The masked variant has no VGPRs to spare and takes the stride.
                    if constexpr (MASK) {
                        const int64_t piece_stride = 16 * ss;
                        if constexpr ((plan.dma[g] & 1) == 0) glds16_sv_m0(kq + j * piece_stride, k_off[0], kdst + NWAVES * j * 1024);
                        else glds16_sv_m0(vq + j * piece_stride, v_off[0], vdst + NWAVES * j * 1024);
                    } else {
                        if constexpr ((plan.dma[g] & 1) == 0) glds16_sv_m0(kq, k_off[j], kdst + NWAVES * j * 1024);
                        else glds16_sv_m0(vq, v_off[j], vdst + NWAVES * j * 1024);
                    }
                }
                static_for<0, plan.exp_n[g]>([&](auto i) { exp_unit(plan.exp_first[g] + decltype(i)::value); });
                static_for<0, plan.max_n[g]>([&](auto i) { max_unit(plan.max_first[g] + decltype(i)::value); });
                if constexpr (plan.tail[g] > 0) tail_step(plan.tail[g]);
                __builtin_amdgcn_sched_barrier(0);
            });
#ifdef FA_TRACE
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts[18])::"memory");
            if (item == args.trace_block && it == args.trace_visit && lane == 0) {
#pragma unroll
                for (int i = 0; i < 19; ++i) args.trace[wave * 24 + i] = ts[i];
            }
#endif
        };
        // ---- first item: prologue -------------------------------------------------------------
        // causal: an item costs ~(qb + 1), and along the walk a workgroup would meet the same Q-block
        // position of a head again and again (round r: slot w + G r of the XCD's item list).  So
        // odd rounds run their G-slot window of a head (or their whole heads, if a head is shorter than
        // the window) in reverse: still every Q block of every head exactly once, and two consecutive
        // rounds sum to the same work for every workgroup.  Needs windows and rounds to line up
        // (Q blocks per head and workgroups per XCD both powers of two, as a rule); otherwise the walk
        // stays in order -- correct, just less balanced.
        auto walk_qb = [&](int it_, int pos) {
            const int G = (args.n_bh & 7) == 0 ? (int)gridDim.x >> 3 : (int)gridDim.x;
            const int W = nq < G ? nq : G;
            if (W <= 0 || G % W != 0 || nq % W != 0) return pos;
            const int in_w = pos % W;
            return (pos / W) * W + (((it_ / (int)gridDim.x) & 1) ? W - 1 - in_w : in_w);
        };
        auto set_next = [&]() {  // coordinates of the item after `item` (or `item` again)
            const int nitem = item + (int)gridDim.x;
            has_next = nitem < n_items;
            int bh_n;
            item_coords(has_next ? nitem : item, bh_n, qb_n);
            if (causal) {
                qb_n = walk_qb(has_next ? nitem : item, qb_n);
                nkn = 4 * (qb_n + 1);
            }
            const int b_n = bh_n / args.n_heads, h_n = bh_n % args.n_heads;
            const int64_t off_n = (int64_t)b_n * args.batch_stride + (int64_t)h_n * args.head_stride;
            Qn = (const uint16_t *)args.q + off_n;
            Kn = (const uint16_t *)args.k + off_n;
            Vn = (const uint16_t *)args.v + off_n;
            On = (uint16_t *)args.o + off_n;
        };
        set_next();
        // Every CU starts at once and the first requests (176 KB per workgroup) return at ~11 B/cycle
        // per CU, in issue order: K(0) and Q -- all S(0) needs -- were asked for first (common code);
        // then K(1), V(0) | K(2), V(1) | K(3), V(2) in the order the counted waits assume.
        dma_k(tile_g(Kc, Kn, 1), 1);
        dma_v(tile_g(Vc, Vn, 0), 0);
        dma_k(tile_g(Kc, Kn, 2), 2);
        dma_v(tile_g(Vc, Vn, 1), 1);
        dma_k(tile_g(Kc, Kn, 3), 3);
        dma_v(tile_g(Vc, Vn, 2), 2);
        kq = tile_g(Kc, Kn, 4);
        vq = tile_g(Vc, Vn, 3);
        if (!(ABL & 8)) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");  // K(0), Q landed
        barrier();
        {
            // S(0) and its row max, which becomes the first reference max (O = l = 0)
            const char *kt = smem;
            vec8 a_all[16];  // every operand stays allocated until the last MFMA has issued (see visit())
#pragma unroll
            for (int step = 0; step < 16; ++step) a_all[step] = k_frag(kt, step);
            static_for<0, 16>([&](auto step_tag) {
                constexpr int step = decltype(step_tag)::value;
                qk_mfma(Sa, step, 0, a_all[step]);
                qk_mfma(Sa, step, 1, a_all[step]);
            });
            asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // MFMA D -> VALU read
#pragma unroll
            for (int step = 0; step < 16; ++step) asm volatile("" ::"v"(a_all[step]));
            mask_tile(Sa, nkc - 1, qb_c);
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                float v = Sa[qt][0][0];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) v = fmaxf(v, Sa[qt][nt][r]);
                m[qt] = pair_max(v);
                neg_msc[qt] = -(finite_or_zero(m[qt]) * c);
                m_pend[qt] = m[qt];
            }
            if (!(ABL & 8)) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");  // K(1) landed (under S(0))
            barrier();
            ring[0] = k_frag(smem + TILE, 0);  // first operands of visit 0: K(1)
            ring[1] = k_frag(smem + TILE, 1);
        }
        // O of one item: finish l, normalise, RNE to 16 bit (final_softmax_normalization
        // softmax.cuh:107-128; forward_kernel.cuh:186-203), through this wave's 8-KB LDS staging
        // area one 32-row Q tile at a time so that the global stores are whole 256-B rows (16 B per
        // lane, 4 rows per wave-instruction).  Wave-private: no barrier.  The 16-B chunk index is
        // XORed with (row & 15) so the 8-B writes and the 16-B reads are bank-conflict free.
        auto store_item = [&]() {
            asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // last P.V -> VALU reads of O
            char *stage_o = smem + 2 * TR::kStages * TILE + wave * (32 * ROWB);
            // lane-derived indices recomputed here from a volatile v_mbcnt: values derived from
            // threadIdx at kernel entry would stay live (and get spilled) across the whole item loop
            int lane;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
            const int r31 = lane & 31, hi = lane >> 5;
            const int rsub = lane / CPR, chunk = lane & (CPR - 1);
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                const float inv = 1.0f / pair_sum(l[qt]);
                char *wp = stage_o + r31 * ROWB + hi * 8;
#pragma unroll
                for (int t = 0; t < DTILES; ++t) {
                    float o[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[r] = O[qt][t][r] * inv;
                    // regs 4rq..4rq+3 : d = 32t + 8rq + 4hi + 0..3  -> chunk 4t + rq, half hi
                    const s16x8 lo_s = __builtin_bit_cast(s16x8, E::pack8(o));
                    const s16x8 up_s = __builtin_bit_cast(s16x8, E::pack8(o + 8));
                    *(s16x4 *)(wp + (((4 * t + 0) ^ swz_of(r31)) << 4)) = lo_s.lo;
                    *(s16x4 *)(wp + (((4 * t + 1) ^ swz_of(r31)) << 4)) = lo_s.hi;
                    *(s16x4 *)(wp + (((4 * t + 2) ^ swz_of(r31)) << 4)) = up_s.lo;
                    *(s16x4 *)(wp + (((4 * t + 3) ^ swz_of(r31)) << 4)) = up_s.hi;
                    __builtin_amdgcn_sched_barrier(0);  // one d tile at a time: S(0) of the next item is live
                }
                const int64_t row0 = (int64_t)qb_c * TR::kBr + wave * TR::kRowsPerWave + qt * 32;
#pragma unroll
                for (int i = 0; i < 32 / RPP; ++i) {
                    const int row = RPP * i + rsub;
                    const s16x8 v = *(const s16x8 *)(stage_o + row * ROWB + ((chunk ^ swz_of(row)) << 4));
                    // non-temporal: O is written once and not read again by this kernel (+1.3...2.8 % at
                    // seq_len <= 1024, where the store-issue-bound epilogue is a visible share)
                    __builtin_nontemporal_store(v, (s16x8 *)(Oc + (row0 + row) * ss + chunk * 8));
                }
            }
        };
        // seq_len is a multiple of B_r = 256, so n_kv = seq_len / 64 is a multiple of 4 = ring depth
        for (;;) {
            for (int it = 0; it < nkc; it += 4) {
                visit(it, Sa, Sb, IntTag<0>{});
                visit(it + 1, Sb, Sa, IntTag<1>{});
                visit(it + 2, Sa, Sb, IntTag<2>{});
                visit(it + 3, Sb, Sa, IntTag<3>{});
            }
#ifdef FA_TRACE
            unsigned long long te0, te1, te2;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(te0)::"memory");
#endif
            store_item();
#ifdef FA_TRACE
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(te1)::"memory");
#endif
            if (!has_next) break;
            // ---- seam: the last visit left the next item's S(0) in Sa and its row max in mraw; its
            // first tiles are landed or in flight, its first operands sit in the ring
            item += (int)gridDim.x;
            Kc = Kn; Vc = Vn; Oc = On; qb_c = qb_n;
            nkc = nkn;
            set_next();
            kq = tile_g(Kc, Kn, 4);  // visit 0 requests K(4), V(3) (for n_kv == 4 that is already the item after)
            vq = tile_g(Vc, Vn, 3);
            seam = true;
            resc_any = 0;
#pragma unroll
            for (int qt = 0; qt < 2; ++qt) {
                m[qt] = mraw[qt];
                neg_msc[qt] = -(finite_or_zero(m[qt]) * c);
                m_pend[qt] = m[qt];
                l[qt] = 0.0f;
#pragma unroll
                for (int t = 0; t < DTILES; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) O[qt][t][r] = 0.0f;
            }
#ifdef FA_TRACE
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(te2)::"memory");
            if (item - (int)gridDim.x == args.trace_block && lane == 0) {
                args.trace[wave * 24 + 21] = te0;
                args.trace[wave * 24 + 22] = te1;
                args.trace[wave * 24 + 23] = te2;
            }
#endif
        }
        dma_wait();  // nothing may still be landing in the LDS when the workgroup retires
        return;
    } else if (PIPE) {
        // In-wave software pipeline with two S accumulators.  While the matrix pipe forms
        // S(it+1) = K(it+1) Q^T and then O += V(it) P(it), the VALU turns the finished S(it)
        // into P(it): every MFMA of the visit has ~4-5 independent VALU ops and 1-2 LDS
        // operand reads scheduled beside it (sched_group_barrier pins the interleave), so a
        // wave keeps the matrix pipe and the vector ALU busy together regardless of what
        // its SIMD partner is doing.  Per visit `it` (S_cur = S(it), complete):
        //   top    : barrier (K(it+1), V(it) landed; stages of K(it), V(it-1) free);
        //            DMA K(it+2), V(it+1); m/alpha/l update from the row max found last
        //            visit; O *= alpha only if some lane's max moved
        //   MFMA   : 8*NT x QK^T(it+1)  ->  4*NT x P.V(it) keys 0-31  ->  ... keys 32-63 ...
        //   VALU   : P = exp2(S_cur*c - m*c), row sums, 16-bit pack, tile by tile, then the
        //            row max of S(it+1)
        // O is only ever rescaled while no P.V is in flight (start of a visit), so the
        // factor applies to exactly the terms accumulated so far (guide T13 hazard).
        f32x16 Sa[QT][NT], Sb[QT][NT];
        float mx[QT];     // row max of the S tile that becomes S_cur next
        auto row_max = [&](f32x16 (&S)[QT][NT]) {
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                float v = S[qt][0][0];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) v = fmaxf(v, S[qt][nt][r]);
                mx[qt] = pair_max(v);
            }
        };
        auto visit = [&](int it, f32x16 (&S_cur)[QT][NT], f32x16 (&S_nxt)[QT][NT], auto last_tag) {
            constexpr bool LAST = decltype(last_tag)::value;
            FA_STAMP(it, 0);
            wait_and_barrier();
            FA_STAMP(it, 1);
            if (DMA) {
                if (it + 2 < n_kv) issue_k(it + 2, it & 1);
                if (it + 1 < n_kv) issue_v(it + 1, (it + 1) & 1);
            } else {
                // registers hold K(it+2), V(it+1) (loaded during the previous visit); their
                // LDS stages were freed by the barrier above.  Then start K(it+3), V(it+2).
                if (it + 2 < n_kv) store_k(it & 1);
                if (it + 1 < n_kv) store_v((it + 1) & 1);
                if (it + 3 < n_kv) load_k(it + 3);
                if (it + 2 < n_kv) load_v(it + 2);
            }
            // scale_l_O (softmax.cuh:36-49) with the new running max
            float neg_msc[QT], rowsum[QT];
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                const float m_new = fmaxf(m[qt], mx[qt]);
                const float alpha = __builtin_amdgcn_exp2f((m[qt] - finite_or_zero(m_new)) * c);
                l[qt] *= alpha;
                m[qt] = m_new;
                neg_msc[qt] = -(finite_or_zero(m_new) * c);
                rowsum[qt] = 0.0f;
                if (!(OPT && it == 0) && !__all(alpha == 1.0f)) {
#pragma unroll
                    for (int t = 0; t < DTILES; ++t)
#pragma unroll
                        for (int r = 0; r < 16; ++r) O[qt][t][r] *= alpha;
                }
            }
            FA_STAMP(it, 2);
            // ---- matrix stream 1: S_nxt = K(it+1) Q^T ---------------------------------
            if (!LAST) {
                qk((it + 1) & 1, S_nxt);
                if (MASK && tile_needs_mask(it + 1)) mask_S(S_nxt, it + 1);
            }
            // ---- vector stream: P = exp2(S_cur c - m c) (softmax.cuh:51-83) -------------
            vec8 P[QT][NT][2];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) {
                    float p[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        p[r] = __builtin_fmaf(S_cur[qt][nt][r], c, neg_msc[qt]);
                        if (!(ABL & 1)) p[r] = __builtin_amdgcn_exp2f(p[r]);
                        rowsum[qt] += p[r];
                    }
                    P[qt][nt][0] = E::pack8(p);
                    P[qt][nt][1] = E::pack8(p + 8);
                }
            }
            // ---- matrix stream 2: O += V(it) P --------------------------------------------
            pv(it & 1, P);
            if (!LAST) row_max(S_nxt);
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) l[qt] += rowsum[qt];
            if (SCHED) {
                // interleave: 8 operand reads ahead, then per MFMA 1-2 reads + 5 VALU/TRANS
                constexpr int N_QK = LAST ? 0 : NT * KS * QT, N_PV = DTILES * NT * 2 * QT;
                __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
                for (int i = 0; i < N_QK; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x402, 5, 0);
                }
#pragma unroll
                for (int i = 0; i < N_PV; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x402, 5, 0);
                }
            }
            FA_STAMP(it, 3);
        };
        if (DMA) {
            wait_and_barrier();  // K(0), V(0) landed
            if (n_kv > 1) issue_k(1, 1);
            qk(0, Sa);
            if (MASK && tile_needs_mask(0)) mask_S(Sa, 0);
            row_max(Sa);
        } else {
            // registers: K(0), V(0) -> LDS; then K(1), V(1) in flight; after S(0): K(1) -> LDS,
            // K(2) in flight.  Invariant at the top of visit `it`: LDS holds K(it+1), V(it)
            // (published by that visit's barrier), registers hold K(it+2), V(it+1).
            store_k(0);
            store_v(0);
            if (n_kv > 1) { load_k(1); load_v(1); }
            barrier();
            qk(0, Sa);
            if (MASK && tile_needs_mask(0)) mask_S(Sa, 0);
            row_max(Sa);
            if (n_kv > 1) store_k(1);
            if (n_kv > 2) load_k(2);
        }
        int it = 0;
        for (; it + 2 < n_kv; it += 2) {
            visit(it, Sa, Sb, FalseTag{});
            visit(it + 1, Sb, Sa, FalseTag{});
        }
        if (it + 2 == n_kv) {
            visit(it, Sa, Sb, FalseTag{});
            visit(it + 1, Sb, Sa, TrueTag{});
        } else {
            visit(it, Sa, Sb, TrueTag{});
        }
    } else if (EAGER) {
        // tile `it` lives in stage it&1; its DMA was issued one visit earlier.
        auto visit = [&](int it, auto first_tag) {
            const int stage = it & 1;
            FA_STAMP(it, 0);
            wait_and_barrier();  // tile `it` landed for every wave; stage^1 free again
            FA_STAMP(it, 1);
            if (DMA) {
                if (it + 1 < n_kv) {
                    issue_k(it + 1, stage ^ 1);
                    issue_v(it + 1, stage ^ 1);
                }
            } else {
                // registers hold tile it+1 (loaded during the previous visit)
                if (it + 1 < n_kv) { store_k(stage ^ 1); store_v(stage ^ 1); }
                if (it + 2 < n_kv) { load_k(it + 2); load_v(it + 2); }
            }
            f32x16 S[QT][NT];
            vec8 P[QT][NT][2];
            float alpha[QT];
            if (ABL & 64) __builtin_amdgcn_s_setprio(1);
            qk(stage, S);
            if (ABL & 64) __builtin_amdgcn_s_setprio(0);
            if (MASK && tile_needs_mask(it)) mask_S(S, it);
            FA_STAMP(it, 2);
            softmax(S, P, alpha, first_tag);
            if (!decltype(first_tag)::value) rescale_O(alpha);
            FA_STAMP(it, 3);
            if (ABL & 64) __builtin_amdgcn_s_setprio(1);
            pv(stage, P);
            if (ABL & 64) __builtin_amdgcn_s_setprio(0);
            FA_STAMP(it, 4);
        };
        if (!DMA) {
            store_k(0);
            store_v(0);
            if (n_kv > 1) { load_k(1); load_v(1); }
        }
        if (OPT) visit(0, TrueTag{}); else visit(0, FalseTag{});
        for (int it = 1; it < n_kv; ++it) visit(it, FalseTag{});
    } else {
        for (int it = 0; it < n_kv; ++it) {
            if (it > 0) barrier();  // everyone done reading the single stage
            issue_k(it, 0);
            issue_v(it, 0);
            wait_and_barrier();
            f32x16 S[QT][NT];
            vec8 P[QT][NT][2];
            float alpha[QT];
            qk(0, S);
            if (MASK && tile_needs_mask(it)) mask_S(S, it);
            if (OPT && it == 0) {
                softmax(S, P, alpha, TrueTag{});
            } else {
                softmax(S, P, alpha, FalseTag{});
                rescale_O(alpha);
            }
            pv(0, P);
        }
    }

    if ((ABL & 32) && args.seq_len < 0) {  // never true: keeps the landing registers allocated
#pragma unroll
        for (int j = 0; j < 2 * DMA_PER_WAVE; ++j) *(f32x4 *)(Og + j * 8 + lane * 64) = abl_dummy[j];
    }
    // ---- epilogue: finish l, normalise, RNE to 16 bit, store ----------------------
    // (final_softmax_normalization softmax.cuh:107-128; forward_kernel.cuh:186-203)
    // Like the reference, O goes through shared memory so that global stores are whole
    // 256-B rows (16 B per lane, 4 rows per wave-instruction) instead of 8-B pieces at a
    // row stride.  Each wave stages only its own rows; the 16-B chunk index is XORed with
    // (row & 15) so both the 8-B writes and the 16-B reads are bank-conflict free.
    barrier();  // every wave is done with the K/V stages
    {
        char *stage_o = smem + wave * (TR::kRowsPerWave * ROWB);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            const float inv = 1.0f / pair_sum(l[qt]);
            const int row = qt * 32 + r31;
            char *wp = stage_o + row * ROWB + hi * 8;
#pragma unroll
            for (int t = 0; t < DTILES; ++t) {
                float o[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) o[r] = O[qt][t][r] * inv;
                // regs 4rq..4rq+3 : d = 32t + 8rq + 4hi + 0..3  -> chunk 4t + rq, half hi
                const s16x8 lo_s = __builtin_bit_cast(s16x8, E::pack8(o));
                const s16x8 up_s = __builtin_bit_cast(s16x8, E::pack8(o + 8));
                *(s16x4 *)(wp + (((4 * t + 0) ^ swz_of(row)) << 4)) = lo_s.lo;
                *(s16x4 *)(wp + (((4 * t + 1) ^ swz_of(row)) << 4)) = lo_s.hi;
                *(s16x4 *)(wp + (((4 * t + 2) ^ swz_of(row)) << 4)) = up_s.lo;
                *(s16x4 *)(wp + (((4 * t + 3) ^ swz_of(row)) << 4)) = up_s.hi;
            }
        }
        const int64_t row0 = (int64_t)qb * TR::kBr + wave * TR::kRowsPerWave;
        const int rsub = lane / CPR, chunk = lane & (CPR - 1);
#pragma unroll
        for (int i = 0; i < TR::kRowsPerWave / RPP; ++i) {
            const int row = RPP * i + rsub;
            const s16x8 v = *(const s16x8 *)(stage_o + row * ROWB + ((chunk ^ swz_of(row)) << 4));
            if (!MASK || row0 + row < S_len) *(s16x8 *)(Og + (row0 + row) * ss + chunk * 8) = v;
        }
    }
}

}  // namespace fa
