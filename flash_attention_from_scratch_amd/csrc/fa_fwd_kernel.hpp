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
//   OPT     the reference's cfg.optimized_softmax -- first KV block skips the rescale -- on the single-stage, register-staged
//           and masked variants; on the double-buffered LDS-DMA variants the speculative softmax (SPEC below), which only
//           fa_fwd_opts.speculative selects (fa_registry.hpp: softmax_mode_of)
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
    uint32_t *stats = nullptr;  // nullptr, or two device counters the kernel adds to (fa_fwd_stats: items, items_redone)
    // adaptive speculative softmax (fa_fwd_opts.speculative == 2): nullptr, or one word of pinned HOST memory into which a
    // workgroup that had to compute an item twice stores this launch's sequence number -- written only then, so a
    // launch without failures costs nothing; the library reads it, without waiting, at its next launches
    uint32_t *redo_flag = nullptr;
    uint32_t redo_seq = 0;
#ifdef FA_TRACE
    unsigned long long *trace;  // tools/segment_timer.hip only: [wave][visit][8] s_memtime stamps
    int32_t trace_block;
    int32_t trace_visit;        // tools/trace64.hip: the one visit whose gap stamps are recorded
#endif
};

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define FA_LDS(type) __attribute__((address_space(3))) type
#define FA_DEV __device__ __forceinline__

// (see KernelArgs::redo_flag)  Every writer of a launch stores the same value: a plain system-scope store, no atomic
FA_DEV void report_redo(const KernelArgs &args) {
    if (args.redo_flag) __hip_atomic_store(args.redo_flag, args.redo_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

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

// Speculative softmax (fa_fwd_opts.speculative): a row of the first pass is accepted when its l = sum of P stays below this.
// Every P of the row is <= l, so below the limit fp32 exp2 did not overflow, the 16-bit P is in range (fp16: P < 65504)
// and the fp32 accumulators hold |O| <= l max|V| -- finite for every fp16 V; for bf16 (|V| up to 2^127) the verdict also
// looks at the accumulators themselves.  bf16: 2^64 (~44 nats above the first visited tile's max), fp16: 2^15 (~10 nats).
// !(l < limit) is also true for NaN and +inf.
template <int DT> __device__ constexpr float spec_limit() { return DT == 5 ? 32768.0f : 18446744073709551616.0f; }
// The persistent kernel's guard (fa_fwd_kernel64.hpp): every four visits a wave looks at its running row sums; a row
// whose sum has passed this threshold (bf16 2^32; fp16 2^13) is brought back to l in [1, 2) -- O and l multiplied by an exact
// power of two, the row's reference moved by as many binades -- so that slowly or moderately rising logits never reach the
// limit at all; only a jump beyond the limit within four visits (256 keys) still fails the item.  fp16: N(0, 1) rows
// reach l ~ 0.15 n on their own (2^11.3 at n = 16384 keys, unlucky rows 2.5x that), and a rescue costs the wave ~700
// instructions: with round 3's 2^11 every wave took it near the end of a 16384-key item and the speculative softmax gained
// nothing there (C3: 1228 vs 1226 lazy); 2^13 leaves the rescues to rows that need them and still two binades below the
// limit 2^15 (a jump of more than 4x inside 256 keys fails either way: profiles/r04/fp16_guard_threshold.txt).
// With the guard the persistent kernel checks O for inf / NaN itself whenever a row sum has passed the threshold, so
// its limit only has to keep P = 2^x finite in fp32 and in the 16-bit type: bf16 2^120 (a single jump of ~83 nats inside
// four visits), fp16 2^15 as above.
template <int DT> __device__ constexpr float spec_limit64() { return DT == 5 ? 32768.0f : 1.329227995784916e36f; }
#ifndef FA_FP16_GUARD
#define FA_FP16_GUARD 8192.0f
#endif
template <int DT> __device__ constexpr float spec_guard() { return DT == 5 ? FA_FP16_GUARD : 4294967296.0f; }

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
    static FA_DEV void mfma_acc_v_qc(f32x16 &acc, vec8 a, vec8 q, const f32x16 &cin) {  // acc(VGPR) = a * q(AGPR) + cin(VGPR)
        // "=&v": a 32x32 MFMA needs D and C identical or fully disjoint; without the early clobber the allocator may overlap
        // them partially once cin is dead behind the statement
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(acc) : "v"(a), "a"(q), "v"(cin));
    }
    static FA_DEV void mfma_acc_a_p(f32x16 &acc, vec8 a, u32x4 p) {  // acc(AGPR) += a * p(VGPR, packed >= 1 step ago)
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(p));
    }
    static FA_DEV void mfma_zero_a(f32x16 &acc, vec8 z) {  // acc(AGPR) = z * z + 0 with z = 0: 16 registers cleared by one instruction
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %1, 0" : "=a"(acc) : "v"(z));
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
    static FA_DEV void mfma_acc_v_qc(f32x16 &acc, vec8 a, vec8 q, const f32x16 &cin) {  // acc(VGPR) = a * q(AGPR) + cin(VGPR)
        // "=&v": a 32x32 MFMA needs D and C identical or fully disjoint; without the early clobber the allocator may overlap
        // them partially once cin is dead behind the statement
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(acc) : "v"(a), "a"(q), "v"(cin));
    }
    static FA_DEV void mfma_acc_a_p(f32x16 &acc, vec8 a, u32x4 p) {  // acc(AGPR) += a * p(VGPR, packed >= 1 step ago)
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(p));
    }
    static FA_DEV void mfma_zero_a(f32x16 &acc, vec8 z) {  // acc(AGPR) = z * z + 0 with z = 0: 16 registers cleared by one instruction
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %1, 0" : "=a"(acc) : "v"(z));
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
// s_nop 3: five wait states between whatever hipcc put in front and the load -- the scalar base may just have come
// back from a VGPR lane (v_readlane: hipcc parks scalars there), and a vector-memory instruction must not read an
// SGPR a vector instruction wrote fewer than 5 wait states earlier; hipcc pads that for its own loads only
// (tools/isa_lint64.py, finding SGPRVM).  Used at the prologue and the item seams, not in the steady-state visits.
static FA_DEV void glds16_sv_m0(const void *base_wave_uniform, unsigned lane_byte_off,
                                unsigned lds_dst_wave_uniform) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 3\n\tglobal_load_lds_dwordx4 %0, %1"
                 :
                 : "v"(lane_byte_off), "s"(base_wave_uniform), "s"(lds_dst_wave_uniform)
                 : "memory");
}
// ... and with the M0 write split off (the caller wrote M0 at least one instruction earlier)
static FA_DEV void glds16_issue(const void *base_wave_uniform, unsigned lane_byte_off) {
    asm volatile("global_load_lds_dwordx4 %0, %1" : : "v"(lane_byte_off), "s"(base_wave_uniform) : "memory");
}
// The same three with an IMMEDIATE offset (round 6).  The instruction adds it to the global address AND to the LDS
// destination (measured: M0 = base + 2048, offset:1024 lands at base + 3072 what lies 1024 bytes behind the lane's
// address), so the pieces of one tile that a wave requests can share ONE M0 value when they are neighbours in the LDS
// image: destination = M0 + 1024 j, the lane offsets carry -1024 j to compensate (fa_fwd_kernel64.hpp).
template <int OFF> static FA_DEV void glds16_sv_off(const void *base_wave_uniform, unsigned lane_byte_off, unsigned lds_dst_wave_uniform) {
    static_assert(OFF >= 0 && OFF < 4096, "13-bit signed immediate");
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:%4\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(lane_byte_off), "s"(base_wave_uniform), "s"(lds_dst_wave_uniform), "n"(OFF)
                 : "memory");
}
template <int OFF> static FA_DEV void glds16_sv_m0_off(const void *base_wave_uniform, unsigned lane_byte_off, unsigned lds_dst_wave_uniform) {
    static_assert(OFF >= 0 && OFF < 4096, "13-bit signed immediate");
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 3\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3"
                 :
                 : "v"(lane_byte_off), "s"(base_wave_uniform), "s"(lds_dst_wave_uniform), "n"(OFF)
                 : "memory");
}
template <int OFF> static FA_DEV void glds16_issue_off(const void *base_wave_uniform, unsigned lane_byte_off) {
    static_assert(OFF >= 0 && OFF < 4096, "13-bit signed immediate");
    asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" : : "v"(lane_byte_off), "s"(base_wave_uniform), "n"(OFF) : "memory");
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
          bool MASK = false, int D = 128, int KSPLIT = 1>
struct FwdTraits {
    static_assert(!PIPE || EAGER, "the pipelined loop needs both LDS stages");
    static_assert(DMA || EAGER, "register-staged tiles are only built double-buffered");
    static_assert(D == 128 || D == 64, "d_head 128 (the reference's scope) or 64 (widener)");
    static_assert(DMA || D == 128, "the register-staged transport is only built for d_head 128");
    static constexpr int kRowsPerWave = 32 * QT;
    static constexpr int kBr = kRowsPerWave * NWAVES / KSPLIT;  // KSPLIT waves share a row group (key split)
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
    // key split: 16 KB of 16-bit row staging + per row group 16 KB of fp32 O and (m, l) handed over in the epilogue
    static constexpr int kMergeBytes = KSPLIT == 2 ? 2 * kRowsPerWave * 2 * D + 2 * (kRowsPerWave * D * 4 + 2 * 64 * 4) : 0;
    static constexpr int kPlainBytes = kKvBytes > kOutBytes ? kKvBytes : kOutBytes;
    static constexpr int kLdsBytes = kPersistent ? kKvBytes + NWAVES * 32 * 2 * D
                                                 : (kPlainBytes > kMergeBytes ? kPlainBytes : kMergeBytes);
};

// ---------------------------------------------------------------------------------
// The kernel.  d_head = 128 is the reference's scope (README.md:7-15); D = 64 is a widener.
// ---------------------------------------------------------------------------------
// ABL (tools/ablate.hip only; 0 in every shipped variant) removes one cost at a time to
// attribute cycles: 1 no v_exp, 2 no softmax VALU at all, 4 no LDS operand reads,
// 8 no barriers / DMA waits, 16 no DMA, 32 plain loads instead of DMA (data discarded);
// experiments: 64 s_setprio(1) around the MFMA clusters (-3 %), 128 operand prefetch 12 deep
// (no change).  Results are wrong by construction when ABL & 63 != 0.
// KSPLIT = 2 (the reference's (B_r 64, B_c 64, 4 warps) configs): B_r / n_warps = 16 rows per wave would
// mean 16x16x32 MFMAs whose K / V operands feed ONE 16-cycle MFMA each -- 256 B/clk of LDS reads per CU
// at full matrix rate, the whole LDS bandwidth.  So the four waves are two row groups of 32 rows x two KEY
// groups instead: wave w works on rows 32 (w & 1) .. +31 and on the 32-key half (w >> 1) of every 64-key
// LDS tile with 32x32x16 MFMAs (half the LDS bytes per flop), keeps its own (m, l, O) over its half of
// the keys, and the two partial results of a row group are merged once, in the epilogue, through LDS
// (the in-workgroup form of a split-KV reduction: m = max, l and O rescaled to it and added).
template <int DT, int QT, int NWAVES, int BC, bool SWZ, bool EAGER, bool OPT, bool PIPE, bool DMA = true,
          bool MASK = false, int D = 128, int ABL = 0, int KSPLIT = 1>
__global__ void
__launch_bounds__(NWAVES * 64, (QT == 1) ? 2 : 1)
fa_fwd_kernel(const KernelArgs args) {
    using E = Elem<DT>;
    using vec8 = typename E::vec8;
    using TR = FwdTraits<DT, QT, NWAVES, BC, SWZ, EAGER, OPT, PIPE, DMA, MASK, D, KSPLIT>;
    static_assert(KSPLIT == 1 || (KSPLIT == 2 && NWAVES == 4 && QT == 1 && BC == 64 && DMA && !MASK && D == 128),
                  "the key split is built for (B_r 64, B_c 64, 4 waves)");
    constexpr int ROWB = 2 * D;              // bytes per K / V / O row (256, or 128 at d_head 64)
    constexpr int CPR = D / 8;               // 16-B chunks per row (16 / 8)
    constexpr int RPP = 64 / CPR;            // tile rows per 1-KiB DMA piece (4 / 8)
    constexpr int DSUB = D / 32;             // 32-wide d subtiles per key group of the V image
    // row -> XOR mask of the 16-B chunk index: 16 consecutive rows must land on 16 distinct
    // 16-B slots of the 256-B LDS bank row (d_head 64: two rows share a bank row)
    auto swz_of = [](int row) { return D == 128 ? (row & 15) : ((row >> 1) & 7); };
    constexpr int NT = BC / 32;              // 32-key tiles per LDS tile
    constexpr int NTW = NT / KSPLIT;         // ... of which this wave works on NTW, from tile NT0 on
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
    const int wave_r = KSPLIT == 2 ? (wave & 1) : wave;   // row group of this wave
    const int NT0 = KSPLIT == 2 ? (wave >> 1) * NTW : 0;  // first 32-key tile (of an LDS tile) of its key group

    // ---- workgroup -> (batch*head, Q block); XCD-aware when n_bh % 8 == 0 --------
    const int nq = args.n_q_blocks;
    // item -> (batch*head, Q block).  Workgroups are dealt round-robin over the 8 XCDs, so items
    // congruent mod 8 share an L2: give each XCD whole heads (all Q blocks of a head read the same
    // K / V).
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
    if (MASK && args.causal) qb = nq - 1 - qb;  // longest rows first
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
    auto first_requests = [&]() {
        if (EAGER && DMA) {
            issue_k(0, 0);
            issue_v(0, 0);
        }
        if (!DMA) {
            load_k(0);
            load_v(0);
        }
    };
    first_requests();

    vec8 Qr[QT][KS];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        int64_t row = (int64_t)qb * TR::kBr + wave_r * TR::kRowsPerWave + qt * 32 + r31;
        if (MASK && row >= S_len) row = S_len - 1;  // rows past the end are computed, never stored
        const uint16_t *qp = Qg + row * ss + hi * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) Qr[qt][ks] = *(const vec8 *)(qp + ks * 16);
    }

    // forward_kernel.cuh:150-151 (fp32 product of rsqrt(d) and log2 e)
    const float c = (float)((double)(1.0f / __builtin_sqrtf((float)D)) * 1.4426950408889634074);

    f32x16 O[QT][DTILES];
    float m[QT], l[QT];
    auto reset_state = [&]() {
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            m[qt] = -__builtin_inff();
            l[qt] = 0.0f;
#pragma unroll
            for (int t = 0; t < DTILES; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) O[qt][t][r] = 0.0f;
        }
    };
    reset_state();

    // per-lane LDS read offsets
    //   K A-operand: row 32*nt + r31, chunk (2*ks + hi) ^ (r31 & 15)
    const int ka_swz = SWZ ? swz_of(r31) : 0;
    const int ka_base = r31 * ROWB;
    //   V^T A-operand (transpose read): see header comment
    const int li = lane & 15, lg = lane >> 4;
    const int va_base = (4 * (lg >> 1) + (li >> 2)) * 64 + (lg & 1) * 32 + (li & 3) * 8;

    // ---- MASK: logits of keys >= seq_len, or above the causal diagonal, become -inf ---------
    const int wave_row0 = wg_row0 + wave_r * TR::kRowsPerWave;
    auto tile_needs_mask = [&](int it) -> bool {  // wave-uniform
        const int kv0 = (n_kv - 1 - it) * BC;
        return MASK && (kv0 + BC > S_len || (args.causal && kv0 + BC - 1 > wave_row0));
    };
    auto mask_S = [&](f32x16 (&S)[QT][NTW], int it) {
        const int kv0 = (n_kv - 1 - it) * BC;
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            const int q_row = wave_row0 + qt * 32 + r31;
            const int limit = args.causal ? (q_row < S_len - 1 ? q_row : S_len - 1) : S_len - 1;  // last live key
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kv0 + 32 * (NT0 + nt) + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    S[qt][nt][r] = key > limit ? -__builtin_inff() : S[qt][nt][r];
                }
        }
    };
    // a row whose keys were all masked so far has m = -inf: exponentiate against 0 instead
    auto finite_or_zero = [&](float mval) { return (MASK && mval == -__builtin_inff()) ? 0.0f : mval; };

    // ---- S^T = K Q^T ------------------------------------------------------------
    auto qk = [&](int stage, f32x16 (&S)[QT][NTW]) {
        const char *kt = smem + stage * TILE;
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) S[qt][nt][r] = 0.0f;
        // ks outer / key-tile inner: consecutive MFMAs accumulate into different tiles
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
                const int off = (NT0 + nt) * 32 * ROWB + ka_base + (((2 * ks + hi) ^ ka_swz) << 4);
                const vec8 a = (ABL & 4) ? Qr[0][(ks + nt) % KS] : *(const vec8 *)(kt + off);
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) S[qt][nt] = E::mfma(a, Qr[qt][ks], S[qt][nt]);
            }
        }
        if (SCHED && !PIPE) sched_mfma_fed_from_lds<NTW * KS * QT, 1, (ABL & 128) ? 12 : 8>();
    };

    // ---- online softmax, lane-local (softmax.cuh:85-105) --------------------------
    // Updates m, l; returns P (16-bit, MFMA B-operand order) and the rescale factor
    // alpha = exp2((m_prev - m_new) c) that (l, O) must be multiplied by BEFORE P.V
    // of this tile is accumulated (scale_l_O, softmax.cuh:36-49).  l is rescaled here;
    // O by rescale_O() -- skipped when alpha == 1 in every lane (multiplying by 1.0f is
    // the identity, so the skip is bit-exact).
    auto softmax = [&](f32x16 (&S)[QT][NTW], vec8 (&Pb)[QT][NTW][2], float (&alpha)[QT], auto first_tag, auto fast_tag) {
        constexpr bool FIRST = decltype(first_tag)::value;
        constexpr bool FAST = decltype(fast_tag)::value;  // speculative: m stays the first tile's row max
        if (ABL & 2) {
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                alpha[qt] = 1.0f;
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) {
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
            if constexpr (!FAST || FIRST) {
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, S[qt][nt][r]);
                mx = pair_max(mx);
            }
            float m_new;
            if (FAST && !FIRST) {
                m_new = m[qt];
                alpha[qt] = 1.0f;
            } else if (FIRST && OPT) {
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
            for (int nt = 0; nt < NTW; ++nt) {
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
    auto pv = [&](int stage, const vec8 (&Pb)[QT][NTW][2]) {
        const char *vt = smem + V_BASE + stage * TILE;
        // 16-key slice outer / d tile inner: consecutive MFMAs hit different accumulators
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
#pragma unroll
                for (int t = 0; t < DTILES; ++t) {
                    const int s16 = 2 * (NT0 + nt) + half;
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
        if (SCHED && !PIPE) sched_mfma_fed_from_lds<DTILES * NTW * 2 * QT, 2, (ABL & 128) ? 12 : 8>();
    };

    using TrueTag = BoolTag<true>;
    using FalseTag = BoolTag<false>;

    static_assert(!(QT == 2 && PIPE), "the 64-rows-per-wave pipelined schedule lives in fa_fwd_kernel64.hpp");
    // SPEC (the OPT build of the double-buffered plain variants; reached through fa_fwd_opts.speculative): the speculative softmax of
    // fa_fwd_kernel64.hpp on a one-item workgroup.  attempt<FAST> keeps the row max of the
    // FIRST visited tile as the reference for every tile -- no row max, no rescale factor, no O rescale --
    // and checks the row sums l >= every P against the overflow limit at the end; if any wave of the
    // workgroup fails, all of them run the item again with the running max (attempt<SAFE>): the K / V
    // stream simply starts over.
    // (not on the register-staged transport: with its landing registers live across both attempts two
    // of those variants spill)
    constexpr bool SPEC = OPT && EAGER && !MASK && DMA;
    auto attempt = [&](auto fast_tag) -> bool {
    constexpr bool FAST = decltype(fast_tag)::value;
    if constexpr (PIPE) {
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
        f32x16 Sa[QT][NTW], Sb[QT][NTW];
        float mx[QT];     // row max of the S tile that becomes S_cur next
        auto row_max = [&](f32x16 (&S)[QT][NTW]) {
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                float v = S[qt][0][0];
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) v = fmaxf(v, S[qt][nt][r]);
                mx[qt] = pair_max(v);
            }
        };
        auto visit = [&](int it, f32x16 (&S_cur)[QT][NTW], f32x16 (&S_nxt)[QT][NTW], auto last_tag) {
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
                if constexpr (FAST) {  // m = the first tile's row max (set once, below): nothing moves
                    neg_msc[qt] = -(m[qt] * c);
                    rowsum[qt] = 0.0f;
                    continue;
                }
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
            vec8 P[QT][NTW][2];
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) {
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
            if (!LAST && !FAST) row_max(S_nxt);
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) l[qt] += rowsum[qt];
            if (SCHED) {
                // interleave: 8 operand reads ahead, then per MFMA 1-2 reads + 5 VALU/TRANS
                constexpr int N_QK = LAST ? 0 : NTW * KS * QT, N_PV = DTILES * NTW * 2 * QT;
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
            if constexpr (FAST) {
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) m[qt] = mx[qt];
            }
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
            if constexpr (FAST) {
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) m[qt] = mx[qt];
            }
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
            f32x16 S[QT][NTW];
            vec8 P[QT][NTW][2];
            float alpha[QT];
            if (ABL & 64) __builtin_amdgcn_s_setprio(1);
            qk(stage, S);
            if (ABL & 64) __builtin_amdgcn_s_setprio(0);
            if (MASK && tile_needs_mask(it)) mask_S(S, it);
            FA_STAMP(it, 2);
            softmax(S, P, alpha, first_tag, fast_tag);
            if (!decltype(first_tag)::value && !FAST) rescale_O(alpha);
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
            f32x16 S[QT][NTW];
            vec8 P[QT][NTW][2];
            float alpha[QT];
            qk(0, S);
            if (MASK && tile_needs_mask(it)) mask_S(S, it);
            if (OPT && it == 0) {
                softmax(S, P, alpha, TrueTag{}, FalseTag{});
            } else {
                softmax(S, P, alpha, FalseTag{}, FalseTag{});
                rescale_O(alpha);
            }
            pv(0, P);
        }
    }

    barrier();  // every wave is done with the K/V stages
    if constexpr (FAST) {
        // every P of a row is <= its l: below the limit nothing overflowed (fp32 exp2, the 16-bit P, fp32 O);
        // NaN fails the compare too.  One verdict per workgroup, through the idle LDS.
        constexpr float kLimit = spec_limit<DT>();
        bool bad = false;
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) bad |= !(pair_sum(l[qt]) < kLimit);
        if constexpr (DT == 15) {
            // bf16: |O| <= l max|V| stays finite below the limit only for |V| < 2^63 -- so look at what the accumulators
            // hold (o - o is 0 for a finite o, NaN for inf / NaN) instead of bounding it.  fp16: 2^15 * 65504 is finite.
            float nonfinite = 0.0f;
#pragma unroll
            for (int qt = 0; qt < QT; ++qt)
#pragma unroll
                for (int t = 0; t < DTILES; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) nonfinite += O[qt][t][r] - O[qt][t][r];
            bad |= !(nonfinite == 0.0f);
        }
        const int wave_bad = __ballot(bad) != 0 ? 1 : 0;
        if (lane == 0) *(int *)(smem + wave * 4) = wave_bad;
        barrier();
        int any = 0;
#pragma unroll
        for (int w = 0; w < NWAVES; ++w) any |= *(const int *)(smem + w * 4);
        any = __builtin_amdgcn_readfirstlane(any);
        barrier();  // (flags read before anything is staged or requested over them)
        return any == 0;
    }
    return true;
    };  // attempt
    bool done = false;
    if constexpr (SPEC) done = attempt(TrueTag{});
    if (SPEC && !done && threadIdx.x == 0) report_redo(args);
    if (args.stats && threadIdx.x == 0) {  // fa_fwd_stats: one item per workgroup; redone = the speculative pass failed
        atomicAdd(args.stats, 1u);
        if (SPEC && !done) atomicAdd(args.stats + 1, 1u);
    }
    if (!done) {
        if constexpr (SPEC) {  // start over with the running max
            reset_state();
            first_requests();
        }
        attempt(FalseTag{});
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
    if constexpr (KSPLIT == 2) {
        // Merge the two key groups of a row group (see the template comment): the wave of key group 1 hands
        // its (m, l, O) -- per-lane values in the same layout as its partner's -- over through LDS, lane-linear
        // (16-B chunk j of lane L at (64 j + L) * 16: conflict-free both ways), behind the 16 KB the two storing
        // waves stage their 16-bit rows in.  m = max(m0, m1); l and O are brought to it and added
        // (exp2((m_i - m) c), the factor scale_l_O applies, softmax.cuh:36-49) -- per lane: the factors are the
        // same in the two lanes that share a row, so the deferred lane-pair sum of l commutes with the merge.
        static_assert(QT == 1, "one 32-row tile per wave");
        char *xo = smem + 2 * TR::kRowsPerWave * ROWB + wave_r * (DTILES * 4 * 64 * 16 + 2 * 64 * 4);
        float *xml = (float *)(xo + DTILES * 4 * 64 * 16);
        if (NT0 != 0) {
#pragma unroll
            for (int t = 0; t < DTILES; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    *(f32x4 *)(xo + ((4 * t + j) * 64 + lane) * 16) =
                        f32x4{O[0][t][4 * j], O[0][t][4 * j + 1], O[0][t][4 * j + 2], O[0][t][4 * j + 3]};
            xml[lane] = m[0];
            xml[64 + lane] = l[0];
        }
        barrier();
        if (NT0 != 0) return;  // (no barrier follows: the storing waves work on wave-private LDS from here on)
        const float m1 = xml[lane], l1 = xml[64 + lane];
        const float m_all = fmaxf(m[0], m1);
        const float a0 = __builtin_amdgcn_exp2f((m[0] - m_all) * c), a1 = __builtin_amdgcn_exp2f((m1 - m_all) * c);
        l[0] = l[0] * a0 + l1 * a1;
#pragma unroll
        for (int t = 0; t < DTILES; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 o1 = *(const f32x4 *)(xo + ((4 * t + j) * 64 + lane) * 16);
#pragma unroll
                for (int e = 0; e < 4; ++e) O[0][t][4 * j + e] = O[0][t][4 * j + e] * a0 + o1[e] * a1;
            }
    }
    {
        char *stage_o = smem + wave_r * (TR::kRowsPerWave * ROWB);
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
        const int64_t row0 = (int64_t)qb * TR::kBr + wave_r * TR::kRowsPerWave;
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
