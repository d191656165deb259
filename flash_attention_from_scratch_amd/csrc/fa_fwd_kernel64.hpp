// fa_fwd_kernel64.hpp -- the persistent, hand-placed 64-rows-per-wave Flash-Attention-2 forward
// (DESIGN.md 3.5): config (B_r 256, B_c 64, 4 waves) + mma_double_buffer_loads, d_head 128.
// One wave per SIMD owns the whole 512-register file; one workgroup per CU walks the
// (batch*head, Q block) items.  Same arithmetic contract as fa_fwd_kernel.hpp (reference:
// src/include/forward_kernel.cuh:85-204, softmax.cuh) except for the lazy rescale documented
// below; shares that file's element traits, DMA helpers and LDS images.
#pragma once
#include "fa_fwd_kernel.hpp"
#include "fa_plan64.hpp"

namespace fa {

#ifdef FA_JITTER
#define FA_JIT(rare) jitter(BoolTag<rare>{})
#else
#define FA_JIT(rare) ((void)0)
#endif
#ifndef FA_ROT_DEFAULT
#define FA_ROT_DEFAULT 4   // softmax units of the next tile carried in a visit's last gaps (speculative plain forms; fa_plan64.hpp)
#endif

// Geometry of the persistent ring kernel (its own traits: FwdTraits describes fa_fwd_kernel.hpp's kernels, whose QT = 1
// forms have two LDS stages): 4 waves, 64-key tiles, d_head 128, four K and four V stages + 8 KB of staging per wave.
// NW = 8 (round 6, experimental: tools/check_nw8.hip): EIGHT waves of one 32-row Q tile each share the same rings -- two waves
// per SIMD; 4 KB of staging per wave, so a wave's Q / O tile passes through it in two 16-row halves.
template <int QT, int NW = 4> struct RingTraits {
    static constexpr int kRowsPerWave = 32 * QT;
    static constexpr int kBr = kRowsPerWave * NW;
    static constexpr int kTileBytes = 64 * 2 * 128;
    static constexpr int kStages = 4;
    static constexpr int kThreads = 64 * NW;
    static constexpr int kStageBytes = NW == 8 ? 4096 : 8192;   // O / Q staging per wave
    static constexpr int kLdsBytes = 2 * kStages * kTileBytes + NW * kStageBytes;
};

// (the reference's meaning of optimized_softmax -- the first tile skips the rescale -- holds here by
// construction, so the flag changes nothing on this kernel; SPEC and PSQ below are asked for through fa_fwd_opts)
// ABL: 0 in the product -- a translation unit built without -DFA_TUNE cannot instantiate anything else (static_assert
// below), and every experiment bit is dead code there.  tools/tune64.hip / trace64.hip (-DFA_TUNE) instantiate the kernel
// with TIMING-ONLY bits, so that a measured claim in profiles/ can be re-run against the shipped code; all but 128 and
// 262144 compute WRONG results on purpose: 1 no exp2, 2 no softmax vector work at all, 4 no LDS operand reads, 8 no waits
// and barriers, 16 no DMA, 32 no row sums, 2048 no multiply-add in front of exp2, 4096 no per-tile row max; 128 the round-4
// item loop (no hot region: every visit carries the seam handling), 262144 no guard.  (The knobs HISTORY.md 0 marks
// "no" -- barrier at the visit's top, DMA pieces late, units spread evenly, the eight-slot operand ring, the guard behind the
// visit, two d tiles per epilogue step, the prologue that requests less, the store cache policies, rot_k / chain-gap
// sweeps -- were deleted in round 6 with the measurements they produced kept in profiles/r01 .. r05.)
// RAG (a second MASK variant): any seq_len >= 64.  The host rounds the Q blocks up and passes
// n_kv_blocks = 4 n_q_blocks (the ring arithmetic wants a multiple of four tiles); a tile that would reach
// beyond the sequence is fetched as the window of its last 64 keys instead (always inside the tensor, no
// per-lane clamp in the request path) and the keys in front of the tile's own first key are masked, which
// masks a tile that lies beyond the sequence whole; Q rows beyond the sequence are fetched from its last
// row and not stored.
// SPEC (fa_fwd_opts.speculative / NativeKernelConfig.speculative_softmax; every form): speculative softmax.  The per-tile row max exists
// only to keep P = 2^((s - m) c) in range -- any reference m gives the same real result -- and its
// end-of-visit chain (32 v_max3, a lane-pair exchange, two ballots) costs 15-20 % of the kernel
// (tools/tune64.hip, knob 4096).  So an item is first run with m fixed at the row max of its FIRST tile,
// no row max and no rescale at all (walk<FAST>); its epilogue checks the row sums l >= every P against
// a limit that proves nothing overflowed (fp32 exp2, the 16-bit P, the accumulators), and an item that
// fails -- a row whose logits rise ~100 (bf16) / ~15 (fp16) binades above its first tile's max -- is
// run again by the lazy-rescale schedule (walk<SAFE>, the whole of the non-SPEC kernel) after the walk.
// PSQ (fa_fwd_opts.prescaled_q; NOT the reference's arithmetic, which multiplies every logit by
// c = log2(e) / sqrt(d) in fp32, softmax.cuh:51-64): Q is multiplied by c and rounded to 16 bit once, on its way from LDS
// into the accumulator file, so the MFMA delivers s c directly.  In the speculative first pass the item's reference
// -(m c) rides in the C operand of each S tile's first MFMA (a 16-register splat per Q tile, constant over the item)
// and the softmax unit is just exp2, row sum, pack: 160 instead of 224 vector instructions per visit.  The second pass
// (running max, which moves) adds -(m c) with one v_add per logit where the exact kernel has its v_fma.
// QTP (round 5): 32-row Q tiles per wave.  2 = the kernel described above, (B_r 256, B_c 64, 4 waves).  1 = the same
// machinery -- rings, counted waits, persistent walk, three-region item loop, hand-placed gaps -- for the reference's own
// winning tile shape (B_r 128, B_c 64, 4 warps) + buffer (kernel_sass/16_A100.asm:5, kernel_configs.py:389-423): one
// 32-row Q tile per wave, 128-row items, 32 MFMAs per visit, every K / V operand read feeds ONE MFMA (1.5 LDS operand
// reads per MFMA instead of 0.75).  Built plain: with the running max (lazy rescale), what a reference user's 13-field
// config asks for, and with the speculative schedule (SPEC: no rotated units there, so its row sums add up in the lazy
// schedule's order; the guard's checkpoint rides in gaps 27 / 29 of every fourth visit; an item given up on is redone
// by the lazy schedule like everywhere).  Needs seq_len % 256 == 0 like the 64-row form (four ring stages = four tiles
// to a group); the compiler-scheduled 32-rows-per-wave body of fa_fwd_kernel.hpp serves the other multiples of 128.
// NW (round 6): waves per workgroup.  4 everywhere but in the product's ring form of (128, 64, 4) + buffer, which runs QTP = 1 with
// NW = 8: eight waves of one 32-row Q tile each share the four-stage rings (256-row items: two of the configuration's Q blocks;
// two waves per SIMD, <= 256 registers per lane) -- a tile's sixteen DMA pieces are dealt two per wave, and a wave's staging area
// is 4 KiB, so its Q and O tiles pass through in two 16-row halves (HSUB).  Per 32-row tile the four-wave form's arithmetic, bit
// for bit (tools/check_nw8.hip).
// ALT (round 6; the 64-row speculative plain form, chosen by the launcher for long sequences): when a head's Q blocks
// take an even number (>= 2) of rounds of an XCD's workgroups -- n_q_blocks % (2 gridDim.x / 8) == 0, seq_len >= 16384 on
// 256 CUs -- every second round walks the head's K / V as [tile 0, then last-to-second]: the tail of the 2 x 8 MiB stream
// the round before left in the XCD's 4 MiB L2 is read again first instead of being evicted by a walk that starts over
// (C3: 1.50 x the algorithmic HBM bytes without).  Tile 0 stays first -- the first pass's reference, and where attention
// sinks sit (FWD below).  Which way an item walks is a function of its own Q block and the device's CU count, never of
// the batch it sits in: (qb / (gridDim.x / 8)) & 1.  Same tiles, same arithmetic per tile; the fp32 sums add up in
// the other order.  Everything else (the second pass, the other forms) keeps its order.
template <int DT, bool MASK = false, int ABL = 0, bool RAG = false, bool SPEC = false, bool PSQ = false, int QTP = 2, bool ALT = false, int NW = 4>
__global__ void
__launch_bounds__(64 * NW, 1)
fa_fwd_kernel64(const KernelArgs args) {
    static_assert(NW == 4 || (NW == 8 && QTP == 1 && !MASK && !PSQ && !ALT && ABL == 0), "eight waves: the one-Q-tile-per-wave plain forms");
    static_assert(!RAG || MASK, "the ragged form is a masked variant");
    static_assert(!PSQ || !MASK, "the pre-scaled Q is built for the plain form");
    static_assert(QTP == 2 || (QTP == 1 && !MASK && !RAG && !PSQ), "one Q tile per wave: the plain forms (lazy / speculative)");
    static_assert(!ALT || (SPEC && !MASK && !PSQ && QTP == 2), "alternating K / V direction: the 64-row speculative plain form");
#ifdef FA_TUNE
    constexpr int TUNE = ABL;
#else
    static_assert(ABL == 0, "experiment / timing-only variants exist in tools built with -DFA_TUNE only");
    constexpr int TUNE = 0;
#endif
    constexpr int QT = QTP, NWAVES = NW, BC = 64, D = 128;
    constexpr bool HSUB = NW == 8;   // Q / O tiles pass through a wave's staging area in two 16-row halves
    constexpr bool SWZ = true, EAGER = true, PIPE = true, DMA = true;

    using E = Elem<DT>;
    using vec8 = typename E::vec8;
    using TR = RingTraits<QT, NW>;
#define FA_TRACE64_MACROS
#include "fa_trace64.inc"
#undef FA_TRACE64_MACROS
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
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // LDS carve: K stages 0..3 | V stages 0..3 | O staging (8 KiB per wave)
    constexpr int V_BASE = TR::kStages * TILE;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r31 = lane & 31;
    const int hi = lane >> 5;
#ifdef FA_JITTER
#include "fa_jitter64.inc"   // jit_state + jitter(): every wave sleeps 0 .. 7 x 64 cycles in front of each protocol step
#endif
#define FA_TRACE64_HELPERS
#include "fa_trace64.inc"
#undef FA_TRACE64_HELPERS
    // ---- workgroup -> (batch*head, Q block); XCD-aware when n_bh % 8 == 0 --------
    const int nq = args.n_q_blocks;
    // item -> (batch*head, Q block).  Workgroups are dealt round-robin over the 8 XCDs, so items
    // congruent mod 8 share an L2: give each XCD whole heads (all Q blocks of a head read the same
    // K / V).  The walk is items blockIdx.x, + gridDim.x, ... with gridDim.x % 8 == 0.
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
    // causal: an item costs ~(qb + 1), and along the walk a workgroup would meet the same Q-block
    // position of a head again and again (round r: slot w + G r of the XCD's item list).  So
    // odd rounds run their G-slot window of a head (or their whole heads, if a head is shorter than
    // the window) in reverse: still every Q block of every head exactly once, and two consecutive
    // rounds sum to the same work for every workgroup.  Needs windows and rounds to line up
    // (Q blocks per head and workgroups per XCD both powers of two, as a rule); otherwise the walk
    // stays in order -- correct, just less balanced.  EVERY item's Q block goes through this, also the first of a
    // walk: the second pass of the speculative softmax may start at an odd round.
    auto walk_qb = [&](int it_, int pos) {
        const int G = (args.n_bh & 7) == 0 ? (int)gridDim.x >> 3 : (int)gridDim.x;
        const int W = nq < G ? nq : G;
        if (W <= 0 || G % W != 0 || nq % W != 0) return pos;
        const int in_w = pos % W;
        return (pos / W) * W + (((it_ / (int)gridDim.x) & 1) ? W - 1 - in_w : in_w);
    };
    const int64_t ss = args.seq_stride;

    // ---- per-lane DMA source offsets (elements), invariant over tiles ------------
    // piece i (wave-uniform) covers LDS chunks [64 i, 64 i + 64) of a tile.
    //   K: chunk p -> key p>>4, 16-B chunk (p&15) ^ (key&15)
    //   V: chunk p -> subtile p>>5 = (key>>3)*4 + (d>>5); inside: key&7 = (p&31)>>2,
    //      d&31 = (p&3)*8
    const int k_row_in_piece = lane / CPR;                                 // 0..RPP-1
    const int v_sub_in_piece = lane >> 5;                                  // 0..1
    const int v_w = lane & 31;
    const int64_t v_lane_row = (v_w >> 2);                                 // key & 7
    const int v_lane_d = (v_w & 3) * 8;

    const unsigned smem_base = __builtin_amdgcn_readfirstlane(lds_addr(smem));
    // DMA addressing: SGPR base = head base + tile offset (scalar ALU), VGPR = 32-bit
    // per-lane byte offset of this wave's piece inside a tile (invariant over tiles).
    // Round 6: a wave's pieces are NEIGHBOURS in the LDS image (piece i = 4 wave + j), and piece j is issued with the
    // immediate offset 1024 j, which the instruction adds to the LDS destination as well as to the global address
    // (fa_fwd_kernel.hpp, glds16_issue_off): ONE M0 per tile and wave (= stage + 4096 wave) instead of one per piece --
    // six scalar instructions fewer per visit.  The lane offsets carry the -1024 j; so that they stay unsigned, the K / V
    // head pointers of this kernel sit DMA_BIAS bytes in front of the tensors' and the offsets DMA_BIAS behind.
    constexpr unsigned DMA_BIAS = (DMA_PER_WAVE - 1) * 1024;
    unsigned k_off[DMA_PER_WAVE], v_off[DMA_PER_WAVE];
#pragma unroll
    for (int j = 0; j < DMA_PER_WAVE; ++j) {
        const int i = DMA_PER_WAVE * wave + j;  // piece index, wave-uniform; keys 4i .. 4i+3
        const int k_row = RPP * i + k_row_in_piece;
        const int k_swz = SWZ ? swz_of(k_row) : 0;
        k_off[j] = (unsigned)(((int64_t)k_row * ss + (((lane & (CPR - 1)) ^ k_swz) << 3)) * 2) + DMA_BIAS - 1024u * j;
        const int sub = 2 * i + v_sub_in_piece;  // subtiles 2i, 2i+1
        v_off[j] = (unsigned)(((8 * (sub / DSUB) + v_lane_row) * ss + (sub % DSUB) * 32 + v_lane_d) * 2) + DMA_BIAS - 1024u * j;
    }
    const int64_t tile_stride = (int64_t)BC * ss;  // elements between consecutive KV blocks
    auto tile_at = [&](const uint16_t *head, int t) {  // first row of tile t of a head's K or V
        if constexpr (RAG) {
            const int r0 = 64 * t < args.seq_len - 64 ? 64 * t : args.seq_len - 64;  // (see the note at the top)
            return head + (int64_t)r0 * ss;
        } else {
            return head + (int64_t)t * tile_stride;
        }
    };
    auto dma_wait = [&]() { if (DMA && !(TUNE & 8)) dma_wait_all(); };
    auto barrier = [&]() { if (!(TUNE & 8)) { FA_JIT(false); wg_barrier(); } };
    // forward_kernel.cuh:150-151 (fp32 product of rsqrt(d) and log2 e)
    const float c = (float)((double)(1.0f / __builtin_sqrtf((float)D)) * 1.4426950408889634074);
    const float cs = PSQ ? 1.0f : c;  // what turns an S element into a base-2 exponent (PSQ: the scale already sits in Q)

    // a row whose keys were all masked so far has m = -inf: exponentiate against 0 instead
    auto finite_or_zero = [&](float mval) { return (MASK && mval == -__builtin_inff()) ? 0.0f : mval; };

    // ---- one walk over this workgroup's items: ordinals o (item = blockIdx.x + o gridDim.x) whose bit
    // min(o, 63) is set in `todo`.  FAST: the speculative schedule (see SPEC above); returns the ordinals
    // (same encoding) of the items whose check failed in any row of this wave.
    // qt_tag: 32-row Q tiles per wave IN THIS WALK.  = QTP, except for the second pass of the 64-row speculative plain form
    // (round 6, HALF below): a failed item is redone as its failed 128-row HALVES, one 32-row tile per wave -- the
    // machinery of the QTP = 1 kernel inside this one.  A walk over half items takes half the time of a walk over whole
    // ones, and a failure is a few rows as a rule: the launch's tail behind a failed item halves.  Per 32-row tile the
    // one-tile walk performs exactly the arithmetic of the 64-row lazy walk (tools/check_qt1.hip: bit for bit), so a
    // redone row is what it was when whole items were redone: the lazy variant's.  Half h of ordinal o is this walk's
    // ordinal 2 o + h; `todo` holds the ordinals whose half 0 failed (waves 0, 1 of the first pass), `todo_hi` half 1.
    auto walk = [&](auto fast_tag, auto qt_tag, const unsigned long long todo, const unsigned long long todo_hi) -> unsigned long long {
        constexpr bool FAST = decltype(fast_tag)::value;
        constexpr int QT = decltype(qt_tag)::value;   // (shadows the kernel's: everything below is this walk's geometry)
        using TR = RingTraits<QT, NW>;
        constexpr bool HALF = QT != QTP;
        static_assert(!HALF || (QTP == 2 && QT == 1 && SPEC && !FAST && !MASK && !PSQ), "half items: the second pass of the 64-row speculative plain form");
        // FWD (round 6): the speculative first pass of the plain forms visits an item's K / V tiles FIRST-TO-LAST.  Its
        // reference is the row max of the first tile it visits, and attention-sink keys sit at the START of a sequence: walked
        // last-to-first (forward_kernel.cuh:142) they arrived last, ~17 binades above a reference taken at the other end --
        // beyond fp16's 2^15, every item redone.  The order is the only difference (same tiles, same arithmetic per tile; the
        // fp32 sums add up in the other order).  The second pass (running max) and the masked forms (a causal item's diagonal
        // tiles are its first) keep the reference's order.
        constexpr bool FWD = FAST && !MASK;
        constexpr bool ALTW = ALT && FWD;   // this walk alternates its K / V direction by rounds (see ALT above)
        auto rev_of = [&](int qb_) { return ALTW && (((qb_ / ((int)gridDim.x >> 3)) & 1) != 0); };
        // tile visited j-th by an item of nk tiles: j, or for a reversed item 0, nk - 1, nk - 2, ..., 1
        auto tord = [&](bool rev, int nk, int j) { return (ALTW && rev) ? (j == 0 ? 0 : nk - j) : j; };
        unsigned long long failed = 0;  // FAST: the ordinals whose check failed; the second pass: the number of items it computed
        const int n_items = args.n_bh * nq;
        auto parent = [&](int o) { return HALF ? (o >> 1) : o; };   // the workgroup's ordinal of the (whole) item
        auto next_ord = [&](int o) {  // next ordinal of this pass behind o, or -1 (scalar; item seams only)
            for (;;) {
                ++o;
                const int op = parent(o);
                if ((long long)blockIdx.x + (long long)op * (long long)gridDim.x >= (long long)n_items) return -1;
                const unsigned long long mask = (HALF && (o & 1)) ? todo_hi : todo;
                if (!SPEC || ((mask >> (op < 63 ? op : 63)) & 1ull)) return o;
            }
        };
        auto coords_of = [&](int o, int &bh_out, int &qb_out) {  // (batch * head, Q block in this walk's units) of ordinal o
            item_coords((int)blockIdx.x + parent(o) * (int)gridDim.x, bh_out, qb_out);
            if constexpr (HALF) qb_out = 2 * qb_out + (o & 1);
        };
        int ord = next_ord(-1);
        if (ord < 0) return failed;  // (second pass only: nothing of this workgroup's failed)
        int bh, qb;
        coords_of(ord, bh, qb);
        if (MASK && args.causal) qb = walk_qb((int)blockIdx.x + ord * (int)gridDim.x, qb);
        const int b = bh / args.n_heads, h = bh % args.n_heads;
        const int64_t head_off = (int64_t)b * args.batch_stride + (int64_t)h * args.head_stride;
        const uint16_t *Qg = (const uint16_t *)args.q + head_off;
        const uint16_t *Kg = (const uint16_t *)args.k + head_off - DMA_BIAS / 2;   // (see DMA_BIAS: only ever a DMA base)
        const uint16_t *Vg = (const uint16_t *)args.v + head_off - DMA_BIAS / 2;
        uint16_t *Og = (uint16_t *)args.o + head_off;
        // KV blocks are visited last-to-first (forward_kernel.cuh:142,175-184): visit index `it` is
        // sequence block n_kv-1-it.  Causal: only the 4 (qb + 1) tiles up to the item's diagonal.
        const int n_kv = (MASK && args.causal) ? 4 * (qb + 1) : args.n_kv_blocks;
        // ---- first requests of the walk: K(0), then Q (all S(0) needs); the rest follows in the prologue
        if (!(TUNE & 16)) {
            static_for<0, DMA_PER_WAVE>([&](auto j_) {
                constexpr int j = decltype(j_)::value;
                glds16_sv_off<1024 * j>(tile_at(Kg, FWD ? 0 : n_kv - 1), k_off[j], smem_base + wave * (DMA_PER_WAVE * 1024));
            });
        }

        vec8 Qr[QT][KS];  // Q of the current item (AGPRs), filled through LDS (request_q / read_q below)

        f32x16 O[QT][DTILES];
        float m[QT];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) m[qt] = -__builtin_inff();
        // O = 0 by eight MFMAs on a zero operand (16 registers apiece; 128 v_accvgpr_write otherwise)
        auto zero_o = [&]() {
            typename E::vec8 zz = __builtin_bit_cast(typename E::vec8, u32x4{0u, 0u, 0u, 0u});
            asm volatile("s_nop 3" : "+v"(zz));  // VALU write -> MFMA operand read
#pragma unroll
            for (int qt = 0; qt < QT; ++qt)
#pragma unroll
                for (int t = 0; t < DTILES; ++t) E::mfma_zero_a(O[qt][t], zz);
            asm volatile("s_nop 7" ::"v"(zz));  // operand registers stay allocated until the last one has read them
        };
        zero_o();

        {
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
            static_assert(DMA && D == 128 && BC == 64 && NT == 2 && (NWAVES == 4 || NWAVES == 8), "pinned schedule");
            static_assert(TR::kStages == 4, "ring depth");
            constexpr float TAU = 8.0f;
            // rotated units (fa_plan64.hpp): the speculative plain form of the 64-row kernel carries the next tile's first units
            constexpr int ROT_K = (FAST && !MASK && QT == 2) ? FA_ROT_DEFAULT : 0;
            constexpr Plan64 plan = QT == 1 ? make_plan32(FAST, 2 * DMA_PER_WAVE) : make_plan64(MASK, FAST, 22 - ROT_K, ROT_K);
            static_assert(QT == 1 ? plan32_ok(plan, FAST, 2 * DMA_PER_WAVE) : plan64_ok(plan, ROT_K), "filler plan violates a wait-state distance");
            constexpr int GAPS = 32 * QT, PH2 = 16 * QT;   // MFMAs of a visit; the first one of phase 2 (O += V P)
            static_assert(plan_barrier_gap(plan, GAPS) >= 0, "the sync point rides inside the stream");
            f32x16 Sa[QT][NT], Sb[QT][NT];
            u32x4 Pw[QT][4] = {};     // P[qt][16-key slice]: B operand of O^T += V^T P^T
            float neg_msc[QT];       // -(m c)
            float thr[QT];           // m + TAU / c: a row max above it moves the reference max
            // running row sums (fp32 P, before rounding: softmax.cuh:66-83), two chains per Q tile; l = their
            // sum, taken in the epilogue (the reference adds a per-tile sum to l: same terms, other order)
            float rs[QT][2] = {};
            float m_pend[QT];        // candidate reference max found during the previous visit
            unsigned resc_any = 0;   // bit qt: Q tile qt moves its reference max at the next visit's top
            // per-lane LDS read offsets, from a volatile lane id PER WALK: derived from threadIdx at kernel entry, the ones only
            // a walk's prologue needs stayed live across the whole first walk for the second one (round 6: one of them was
            // the pre-scaled-Q build's 257th register)
            //   K A-operand: row 32*nt + r31, chunk (2*ks + hi) ^ (r31 & 15)
            //   V^T A-operand (transpose read): see fa_fwd_kernel.hpp's header comment
            int ka_swz, ka_base, ka_hi, va_base;
            {
                int l_;
                asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l_));
                const int r31_ = l_ & 31, li = l_ & 15, lg = l_ >> 4;
                ka_swz = SWZ ? swz_of(r31_) : 0;
                ka_base = r31_ * ROWB;
                ka_hi = l_ >> 5;
                va_base = (4 * (lg >> 1) + (li >> 2)) * 64 + (lg & 1) * 32 + (li & 3) * 8;
            }
            auto k_frag = [&](const char *kt, int step) -> vec8 {  // step = 2*ks + nt
                const int ks = step >> 1, nt = step & 1;
                return *(const vec8 *)(kt + nt * 32 * ROWB + ka_base + (((2 * ks + ka_hi) ^ ka_swz) << 4));
            };
            auto v_frag = [&](const char *vt, int step) -> vec8 {  // step = 4*s16 + t
                const int s16 = step >> 2, t = step & 3;
                const char *vp = vt + va_base + s16 * (DSUB * 1024) + t * 512;
                s16x8 av;
                av.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((FA_LDS(s16x4) *)(vp));
                av.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((FA_LDS(s16x4) *)(vp + DSUB * 512));
                return __builtin_bit_cast(vec8, av);
            };
            // PSQ, speculative first pass: C operand of the first MFMA of every S tile = -(m c) of the row this lane
            // owns, 16 equal registers per Q tile, constant over an item.  Zero while the NEXT item's S(0) is formed (an
            // item's last visit, and the prologue): its reference is its own row max, subtracted once it is known.
            f32x16 Cinit[QT];
            if constexpr (PSQ && FAST) {
#pragma unroll
                for (int qt = 0; qt < QT; ++qt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) Cinit[qt][r] = 0.0f;
            }
            auto zero_cinit = [&]() {
                if constexpr (PSQ && FAST) {
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) Cinit[qt][r] = 0.0f;
                    asm volatile("s_nop 1" : "+v"(Cinit[0]), "+v"(Cinit[1]));  // VALU write -> MFMA C read
                }
            };
            auto set_cinit = [&](auto &S0) {  // S0 = the item's S(0), formed against C = 0: bring it to the reference too
                if constexpr (PSQ && FAST) {
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt) {
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                            for (int r = 0; r < 16; ++r) S0[qt][nt][r] = vadd(S0[qt][nt][r], neg_msc[qt]);
#pragma unroll
                        for (int r = 0; r < 16; ++r) Cinit[qt][r] = neg_msc[qt];
                    }
                    asm volatile("s_nop 1" : "+v"(Cinit[0]), "+v"(Cinit[1]));
                }
            };
            auto qk_mfma = [&](auto &S, int step, int qt, vec8 a) {
                const int ks = step >> 1, nt = step & 1;
                if (ks == 0) {
                    if constexpr (PSQ && FAST) E::mfma_acc_v_qc(S[qt][nt], a, Qr[qt][ks], Cinit[qt]);
                    else E::mfma_acc_v_q0(S[qt][nt], a, Qr[qt][ks]);
                } else {
                    E::mfma_acc_v_q(S[qt][nt], a, Qr[qt][ks]);
                }
            };
            // ---- persistent walk over items; the K / V tile stream runs on across item seams --------
            // This workgroup serves items blockIdx.x, + gridDim.x, ...  Tiles are numbered along the
            // walk: visit index j of the current item for j < n_kv, visit index j - n_kv of the NEXT item
            // beyond (n_kv % 4 == 0, so a tile's ring stage is j & 3 either way).  The last visits of an
            // item therefore request the next item's first tiles, its last visit forms the next item's
            // S(0) with the next item's Q (brought into the spare Q set during the item's first visits), and a
            // seam costs the O epilogue and the reset of the item state only.  After the last item the "next" item is the item itself:
            // the re-fetched tiles land in stages nobody reads.
            int item = (int)blockIdx.x + parent(ord) * (int)gridDim.x;   // (the whole item's id: traces, the causal walk)
            int ord_n = -1;  // ordinal of the next item of this pass, or -1
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
            bool rev_c = rev_of(qb), rev_n = rev_c;          // ALT: the current / next item walks [0, last .. 1]
            int64_t dstr = rev_c ? -tile_stride : tile_stride;  // ... and its request pointers' step in the hot visits
            (void)rev_n; (void)dstr;
            auto tile_g = [&](const uint16_t *cur, const uint16_t *nxt, int j) {
                if constexpr (FWD) return j < nkc ? tile_at(cur, tord(rev_c, nkc, j)) : tile_at(nxt, tord(rev_n, nkn, j - nkc));
                return j < nkc ? tile_at(cur, nkc - 1 - j) : tile_at(nxt, nkn - 1 - (j - nkc));
            };
            auto lane_now = [&]() {  // volatile: anything derived from threadIdx would be kept live across the walk
                int l_;
                asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l_));
                return l_;
            };
#include "fa_mask64.inc"   // mask_tile(S, tile, qb_rows, only_nt): logits above the causal diagonal / beyond a ragged end become -inf
            auto dma_k = [&](const uint16_t *src, int stage) {
                static_for<0, DMA_PER_WAVE>([&](auto j_) {
                    constexpr int j = decltype(j_)::value;
                    FA_JIT(false);
                    glds16_sv_m0_off<1024 * j>(src, k_off[j], smem_base + stage * TILE + wave * (DMA_PER_WAVE * 1024));
                });
            };
            auto dma_v = [&](const uint16_t *src, int stage) {
                static_for<0, DMA_PER_WAVE>([&](auto j_) {
                    constexpr int j = decltype(j_)::value;
                    FA_JIT(false);
                    glds16_sv_m0_off<1024 * j>(src, v_off[j], smem_base + V_BASE + stage * TILE + wave * (DMA_PER_WAVE * 1024));
                });
            };
            const uint16_t *kq = nullptr, *vq = nullptr;  // next K / V tile to request (set per item below)
            // operand ring: slot u % RS holds operand u; the loads of operands step + LA, step + LA + 1 are issued
            // at (even) step `step`, into the slots of the two operands whose MFMAs have just issued: two steps (4 MFMAs,
            // ~170 cycles) between a load and its use.  (Eight slots -- six steps of read-ahead, also with one counted wait
            // per four steps -- measured +0.1 ... +0.3 % in rounds 1, 4 and 5: the stream is issue bound, not latency bound;
            // profiles/r05/tune64_ring8_wait4.txt.)
            constexpr int RS = 4, LA = RS - 2;
            vec8 ring[RS];
            vec8 Qr2[QT][KS];  // the next item's Q (AGPRs), requested during the item's first visit
            float mraw[QT];  // row max of the S tile formed by the last visit (the next item's S(0))
            // the first two visits after a seam: the epilogue's row stores are in flight in front of the pieces
            // the counted waits allow -- 16 of them, or fewer (RAG: a wave whose rows reach beyond the sequence
            // skips stores; counted down to a multiple of 8, which only waits for more)
            int seam_st = 0;
            // the guard's common path (see guard() below) rides in the last gaps of every fourth visit, where the vector
            // stream has room (all 32 softmax units have issued by gap 53): two adds, a max, a compare.
            constexpr bool GUARD_IN_VISIT = FAST && !(TUNE & 262144);
            float g_l0 = 0.0f, g_l1 = 0.0f, g_lm = 0.0f;
            (void)g_l0; (void)g_l1;
            bool guard_hit = false;
            // ---- the next item's Q, through LDS -------------------------------------------------------
            // The MFMA wants a lane to hold one Q row's 16-byte chunk; fetched like that from global memory
            // a wave-instruction touches 32 rows (32 cache lines for 1 KiB), and 16 of them in a burst cost
            // each wave 0.7-3 k cycles of queueing in the address path (tools/trace64.hip).  So a 32-row Q
            // tile travels like a K tile: 8 coalesced 1-KiB LDS-DMA pieces (4 whole rows each) into this
            // wave's O staging area (idle between seams), XOR-swizzled, then 8 conflict-free ds_read_b128
            // straight into the spare Q set.  One tile per round: tile 0 is requested at the seam (or at
            // the end of the prologue), read behind the barrier of visit 1, where tile 1 is requested,
            // which is read behind the barrier of visit 2 -- all on the slow path that the sync point of
            // an item's first three visits takes anyway, so the steady state carries none of it.
            const unsigned q_stage = smem_base + 2 * TR::kStages * TILE + wave * TR::kStageBytes;
            // half (eight waves: 4 KB of staging): -1 = the whole tile (pieces 0 .. 7), 0 / 1 = its rows 0 .. 15 / 16 .. 31 (four pieces)
            auto request_q = [&](const uint16_t *Qh, int qblk, int qt, unsigned stage, int half = -1) {  // rows 32 qt .. 32 qt + 31 of this wave's rows
                const int l_ = lane_now();
                // piece i: rows 4i .. 4i+3; this lane: row 4i + l/16, chunk (l % 16) ^ (row & 15)
                //   = ((l % 16) ^ (l / 16)) ^ 4 (i & 3): one lane offset, 64 (i & 3) XORed in per piece
                const unsigned off = (unsigned)(((int64_t)(l_ >> 4) * ss) * 2) + ((((unsigned)l_ & 15) ^ ((unsigned)l_ >> 4)) << 4);
                if constexpr (RAG) {
                    const int row_b = qblk * TR::kBr + wave * TR::kRowsPerWave + 32 * qt;
                    if (row_b + 31 >= args.seq_len) {  // wave-uniform: rows beyond the sequence are fetched from its last row
                        const unsigned chunk_b = (((unsigned)l_ & 15) ^ ((unsigned)l_ >> 4)) << 4;
                        const int row_s = row_b < args.seq_len - 32 ? row_b : args.seq_len - 32;  // scalar base row (seq_len >= 64)
                        const uint16_t *rows_s = Qh + (int64_t)row_s * ss;
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            int rg = row_b + 4 * i + (l_ >> 4);
                            rg = rg < args.seq_len - 1 ? rg : args.seq_len - 1;
                            glds16_sv_m0(rows_s, (unsigned)(rg - row_s) * (unsigned)(ss * 2) + (chunk_b ^ (64u * (i & 3))), stage + i * 1024);
                        }
                        return;
                    }
                }
                const uint16_t *rows0 = Qh + ((int64_t)qblk * TR::kBr + wave * TR::kRowsPerWave + 32 * qt) * ss;
                if constexpr (HSUB) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        glds16_sv_m0(rows0, (off ^ (64u * (i & 3))) + (unsigned)(4 * (i + 4 * half)) * (unsigned)ss * 2u, stage + i * 1024);
                    return;
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    FA_JIT(false);
                    glds16_sv_m0(rows0, (off ^ (64u * (i & 3))) + (unsigned)(4 * i) * (unsigned)ss * 2u, stage + i * 1024);
                }
            };
            auto read_q = [&](vec8 (&dst)[KS], unsigned stage, int half = -1) {  // this lane's chunks: row l % 32, chunk (2 ks + l/32) ^ (row & 15)
                const int l_ = lane_now();
                if constexpr (HSUB) {
                    // the lanes whose row lies in this half (rows 0 .. 15: lanes 0-15, 32-47) read it from the 4-KB image; the others
                    // keep what they hold (EXEC narrowed inside the asm: no divergent control flow around an accumulator-file write)
                    const unsigned base_h = stage + (l_ & 15) * 256, x_h = (unsigned)((l_ >> 5) ^ (l_ & 15));
                    const unsigned long long mask = half ? 0xffff0000ffff0000ull : 0x0000ffff0000ffffull;
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        unsigned long long save_;
                        asm volatile("s_mov_b64 %1, exec\n\ts_and_b64 exec, exec, %3\n\tds_read_b128 %0, %2\n\ts_mov_b64 exec, %1"
                                     : "+a"(dst[ks]), "=&s"(save_) : "v"(base_h + ((x_h ^ (2 * ks)) << 4)), "s"(mask) : "memory", "scc");
                    }
                    return;
                }
                const unsigned base = stage + (l_ & 31) * 256, x = (unsigned)((l_ >> 5) ^ (l_ & 15));
                if constexpr (PSQ) {
                    // through VGPRs: Q * c in fp32, RNE to 16 bit (the one rounding this option adds), then into the
                    // accumulator file.  Two chunks at a time (register pressure: S(0) of the next item may be live).
#pragma unroll
                    for (int ks = 0; ks < KS; ks += 2) {
                        vec8 t0, t1;
                        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3\n\ts_waitcnt lgkmcnt(0)"
                                     : "=&v"(t0), "=&v"(t1)
                                     : "v"(base + ((x ^ (2 * ks)) << 4)), "v"(base + ((x ^ (2 * ks + 2)) << 4))
                                     : "memory");
                        float f0[8], f1[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            f0[j] = (float)t0[j] * c;
                            f1[j] = (float)t1[j] * c;
                        }
                        dst[ks] = E::pack8(f0);
                        dst[ks + 1] = E::pack8(f1);
                    }
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+a"(dst[ks]));  // in the accumulator file from here on
                    asm volatile("s_nop 3" ::: "memory");  // accvgpr write -> MFMA operand
                    return;
                }
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
                    asm volatile("ds_read_b128 %0, %1" : "=a"(dst[ks]) : "v"(base + ((x ^ (2 * ks)) << 4)) : "memory");
            };
            // sub: the wave's Q tile `sub` (four waves), or half `sub` of its one tile (eight waves)
            auto request_next_q = [&](int sub) { if constexpr (HSUB) request_q(Qn, qb_n, 0, q_stage, sub); else request_q(Qn, qb_n, sub, q_stage); };
            auto read_next_q = [&](vec8 (&dst)[KS], int half = -1) { read_q(dst, q_stage, half); };
            // one softmax unit = two elements of a tile's P: u = 8*s16 + 2*j + qt, in the order P.V consumes P.  Where its two
            // values go: SUM_RS the running row sums; SUM_EARLY the side sums rs_e of the rotated plan's early units (SET_EARLY:
            // the first unit of a Q tile starts them: no add), folded into rs by the next visit; SUM_NONE nowhere (the guard's
            // rare path forming the early units' packed P again)
            constexpr int SUM_RS = 0, SUM_EARLY = 1, SET_EARLY = 2, SUM_NONE = 3;
            float rs_e[QT][2] = {};
            auto exp_unit_on = [&](auto &S_src, int u, auto sum_tag) {
                constexpr int SUM = decltype(sum_tag)::value;
                if constexpr (TUNE & 2) return;
                const int qt = u % QT, j = (u / QT) & 3, s16 = u / (4 * QT), r = 8 * (s16 & 1) + 2 * j;   // QT = 2: u = 8 s16 + 2 j + qt
                // exp2(s c - m c), softmax.cuh:51-64.  Scalar f32 forms on purpose: v_pk_fma_f32 /
                // v_pk_add_f32 here measured -6 % / -12 %; splitting the unit into stages over three
                // gaps (no dependent pair inside a gap) measured -1.5 %.
                float p0, p1;
                if constexpr (TUNE & 2048) {  // (timing only: what a pre-scaled Q with -m c fed through the MFMA's C operand would save)
                    p0 = S_src[qt][s16 >> 1][r];
                    p1 = S_src[qt][s16 >> 1][r + 1];
                } else if constexpr (PSQ && FAST) {  // -(m c) came in through the C operand of the tile's first MFMA
                    p0 = S_src[qt][s16 >> 1][r];
                    p1 = S_src[qt][s16 >> 1][r + 1];
                } else if constexpr (PSQ) {
                    p0 = vadd(S_src[qt][s16 >> 1][r], neg_msc[qt]);
                    p1 = vadd(S_src[qt][s16 >> 1][r + 1], neg_msc[qt]);
                } else {
                    p0 = __builtin_fmaf(S_src[qt][s16 >> 1][r], c, neg_msc[qt]);
                    p1 = __builtin_fmaf(S_src[qt][s16 >> 1][r + 1], c, neg_msc[qt]);
                }
                if (!(TUNE & 1)) {
                    p0 = __builtin_amdgcn_exp2f(p0);
                    p1 = __builtin_amdgcn_exp2f(p1);
                }
                if constexpr (SUM == SUM_RS && !(TUNE & 32)) {  // (TUNE & 32, tools/tune64.hip, TIMING ONLY: no row sums)
                    rs[qt][0] += p0;  // fp32 P, before rounding (softmax.cuh:66-83)
                    rs[qt][1] += p1;
                    // pin the adds to this gap: hipcc otherwise sinks the whole row-sum chain (and keeps
                    // every p alive) to the first use of l, behind the next visit's barrier
                    asm volatile("" : "+v"(rs[qt][0]), "+v"(rs[qt][1]));
                }
                if constexpr (SUM == SUM_EARLY && !(TUNE & 32)) {
                    rs_e[qt][0] += p0;
                    rs_e[qt][1] += p1;
                    asm volatile("" : "+v"(rs_e[qt][0]), "+v"(rs_e[qt][1]));
                }
                if constexpr (SUM == SET_EARLY) {
                    // the pack FIRST: p0 / p1 then die into the side sums, which take over their registers (set behind the
                    // pack they cost a v_mov apiece, four per visit)
                    unsigned pk0 = E::pack2(p0, p1);
                    asm volatile("" : "+v"(pk0));
                    Pw[qt][s16][j] = pk0;
                    rs_e[qt][0] = p0;
                    rs_e[qt][1] = p1;
                    asm volatile("" : "+v"(rs_e[qt][0]), "+v"(rs_e[qt][1]));
                    return;
                }
                unsigned pk = E::pack2(p0, p1);
                // ... and the pack: sunk below a branch of the stream it would sit right in front of the
                // MFMA that reads it, which hipcc does not pad (the MFMAs are opaque asm)
                asm volatile("" : "+v"(pk));
                Pw[qt][s16][j] = pk;
            };
            // rotated plan: units 0 .. ROT_K-1 of an item's FIRST tile, which no visit's last gaps could carry (the item's
            // reference was not known yet): straight code behind the prologue / the seam, once per item
            auto head_units = [&](auto &S0) {
                                if constexpr (ROT_K > 0)
                    static_for<0, ROT_K>([&](auto i) { exp_unit_on(S0, decltype(i)::value, IntTag<(decltype(i)::value < 2 ? SET_EARLY : SUM_EARLY)>{}); });
            };
            // hot_tag (round 5): what a visit does for an item's two ENDS is decided at compile time.  Past the item's first
            // four visits (>= 1): no slow path at the sync point (the first three visits: Q tiles, stores in flight) and no mask
            // (a causal item's diagonal tiles and a ragged sequence's last tiles are the FIRST three it forms); before its last
            // eight as well (2): no Q swap (the last visit) and no change of direction in the request pointers (K of the next
            // item from the fifth visit before the end, V from the fourth).  The item loop below runs [first group: 0 | middle
            // groups: 2 | last two groups: 1].  With `it` a run-time value in every visit the steady state carried a
            // compare-and-branch in its sync point and six scalar selects in its pointer chain, and hipcc cut every visit into
            // blocks at them.
            auto visit = [&](int it, auto &S_cur, auto &S_nxt, auto r_tag, auto hot_tag) {
                constexpr int R = decltype(r_tag)::value;  // it & 3
                // 0: general; 1: not one of the item's first four (no slow sync path, no mask); 2: nor one of its last eight
                constexpr bool HOTB = decltype(hot_tag)::value >= 1, HOT = decltype(hot_tag)::value >= 2;
#if defined(FA_TRACE) && FA_TRACE < 4
                unsigned long long ts[20];
                asm volatile("s_memtime %0" : "=s"(ts[0]));
#endif
                FA_TL();
#if defined(FA_TRACE) && FA_TRACE >= 4
                if (it == 0) tl_at(1 + (ord < 60 ? ord : 60));
                if constexpr (FA_TRACE == 5) {
                    if (ord == 1 && it >= 1 && it <= 4) tl_at(55 + it);            // slots 56 .. 59: tops of visits 1 .. 4 behind the first seam
                    if (ord == 0 && it + 2 >= nkc) tl_at(48 + (it + 2 - nkc));    // slots 48, 49: tops of the first item's last two visits
                }
#endif
                // the visit's synchronisation point: K(it+2), V(it+1) landed (requested two visits ago;
                // K(it+1), V(it) were published by the previous barrier), the 8 youngest pieces may fly
                // on; behind it every wave has finished visit it-1, whose K / V stages the DMA of this
                // visit overwrites.  In the default plan it sits two MFMAs into the visit, after the
                // gap-0 lgkmcnt(0) that retires this wave's last LDS reads of visit it-1.
                auto sync_point = [&]() {
                    if (TUNE & 8) return;
                    FA_JIT(false);
                    // One compare and one branch on the common path.  The first three visits of an item take
                    // the slow path: more may be in flight behind the pieces the barrier publishes -- in issue
                    // order: [pieces(last visit of the previous item) | 16 epilogue stores] [Q tile 0: 8]
                    // pieces(0) [Q tile 1: 8] pieces(1) pieces(2) -- and the next item's Q tiles are moved
                    // into the spare Q set here (see request_next_q).
                    if constexpr (HSUB) {   // eight waves: four pieces per visit and wave, Q in halves of four pieces
                        int allow = 4;
                        if (!HOTB && it < 3) allow = (it < 2) ? 4 + seam_st + (has_next ? 4 : 0) : 4 + (has_next ? 4 : 0);
                        if (allow == 4) asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
                        else if (allow == 8) asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
                        else if (allow == 12) asm volatile("s_waitcnt vmcnt(12)\n\ts_barrier" ::: "memory");
                        else asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory");
                        if constexpr (!HOTB && (R == 1 || R == 2)) {
                            if (it == R && has_next) {   // half R - 1 of the next item's Q landed (only this visit's pieces are younger)
                                asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                                read_next_q(Qr2[0], R - 1);
                                if constexpr (R == 1) {
                                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                                    request_next_q(1);
                                }
                            }
                        }
                        return;
                    }
                    if (HOTB || it >= 3) {
                        asm volatile("s_waitcnt vmcnt(" FA_VM8 ")\n\ts_barrier" ::: "memory");
                        return;
                    }
                    const int q8 = has_next ? 8 : 0;
                    // (it == 2: the pieces of Q tile 1, requested at visit 1 -- the 64-row form only)
                    const int allow = (it < 2) ? 8 + seam_st + q8 : 8 + (QT == 2 ? q8 : 0);
                    if (allow == 8) asm volatile("s_waitcnt vmcnt(" FA_VM8 ")\n\ts_barrier" ::: "memory");
                    else if (allow == 16) asm volatile("s_waitcnt vmcnt(" FA_VM16 ")\n\ts_barrier" ::: "memory");
                    else if (allow == 24) asm volatile("s_waitcnt vmcnt(" FA_VM24 ")\n\ts_barrier" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(" FA_VM32 ")\n\ts_barrier" ::: "memory");
                    if constexpr (R >= 1 && R <= QT) {   // (Q tile R - 1 of the next item: the wave has QT of them)
                        if (it == R && has_next) {
                            // Q tile R-1 landed (only pieces(R-1) are younger)
                            asm volatile("s_waitcnt vmcnt(" FA_VM8 ")" ::: "memory");
                            read_next_q(Qr2[R - 1]);
                            if constexpr (R == 1 && QT == 2) {
                                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // ... and read: its image may be overwritten
                                request_next_q(1);
                            }
                        }
                    }
                };
                if constexpr (R == 3 && !HOT) {
                    // last visit of an item forms the next item's S(0): swap the next item's Q in
                    if (it + 1 == nkc && has_next) {
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the LDS reads into the spare set (visits 1, 2)
#pragma unroll
                        for (int qt = 0; qt < QT; ++qt)
#pragma unroll
                            for (int ks = 0; ks < KS; ++ks) {
                                asm volatile("" : "+a"(Qr2[qt][ks]));  // value defined by the asm loads
                                Qr[qt][ks] = Qr2[qt][ks];
                            }
                        asm volatile("s_nop 3" ::: "memory");  // accvgpr write -> MFMA operand
                        zero_cinit();  // this visit forms the NEXT item's S(0): no reference yet
                    }
                }
                const unsigned kdst = smem_base + R * TILE + wave * (DMA_PER_WAVE * 1024);                        // K(it+4) -> stage of K(it)
                const unsigned vdst = smem_base + V_BASE + ((R + 3) & 3) * TILE + wave * (DMA_PER_WAVE * 1024);   // V(it+3) -> stage of V(it-1)
                if (!FAST && resc_any) {  // wave-uniform, rare: move the reference max of one or both Q tiles
                    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");  // MFMA D (O) -> VALU read
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt) {
                        if (!(resc_any & (1u << qt))) continue;
                        const float m_new = fmaxf(m[qt], m_pend[qt]);
                        const float alpha = __builtin_amdgcn_exp2f((m[qt] - m_new) * cs);
                        m[qt] = m_new;
                        neg_msc[qt] = -(finite_or_zero(m_new) * cs);
                        thr[qt] = m_new + TAU / cs;
                        rs[qt][0] *= alpha;
                        rs[qt][1] *= alpha;
#pragma unroll
                        for (int t = 0; t < DTILES; ++t) {
                            // RAG: the accumulator copies start here, behind the pads above, and end behind the
                            // multiply: in that variant (and in trace builds) hipcc otherwise hoists the reads out of
                            // this branch to the end of the previous visit, right behind the MFMAs that write the
                            // tiles (tools/isa_lint64.py, finding AGPR); so it does in the second pass of the speculative
                            // build, and -- since the statistics tail of round 3 -- in the plain lazy build too: pinned
                            // in every variant (the lint checks the result; 65 fewer VGPRs, 0.3 % on the lazy build).
                            asm volatile("" : "+a"(O[qt][t]));
#pragma unroll
                            for (int r = 0; r < 16; ++r) O[qt][t][r] *= alpha;
                            asm volatile("" : "+a"(O[qt][t]));
                        }
                    }
                }
                const char *kt = smem + ((R + 1) & 3) * TILE;
                const char *vt = smem + V_BASE + R * TILE;
                float vm[QT][2];
                unsigned any01 = 0;
                auto max_unit = [&](int u) {  // u = 0..31: tile (nt = u>>4, qt = (u>>3)&1), elements 2(u&7), +1
                    const int nt = u / (8 * QT), qt = (u >> 3) % QT, e = 2 * (u & 7), a = u & 1;  // two chains per Q tile
                    if constexpr (TUNE & 2) { vm[qt][a] = 0.0f; return; }
                    if constexpr (TUNE & 4096) return;  // (timing only: no per-tile row max)
                    // asm forms: fmaxf() on MFMA results makes hipcc canonicalise both inputs first
                    // volatile: pinned to its gap (S_nxt is rewritten next visit).  Not an empty "+v" asm behind
                    // it: hipcc pads an asm that reads what the asm right before it wrote with an s_nop
                    if ((u & 7) < 2 && nt == 0)
                        asm volatile("v_max_f32 %0, %1, %2" : "=v"(vm[qt][a]) : "v"(S_nxt[qt][nt][e]), "v"(S_nxt[qt][nt][e + 1]));
                    else
                        asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(vm[qt][a]) : "v"(S_nxt[qt][nt][e]), "v"(S_nxt[qt][nt][e + 1]));
                };
                auto lane_pair_max = [&](float x) {
                    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
                    float d;
                    asm volatile("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(__uint_as_float(r[0])), "v"(__uint_as_float(r[1])));
                    return d;
                };
                auto tail_unit = [&](int k) {
                    if constexpr (FAST || (TUNE & 4096)) { if (k < 8) return; }
                    if (k == 1) {
                        asm volatile("v_max_f32 %0, %0, %1" : "+v"(vm[0][0]) : "v"(vm[0][1]));
                        if constexpr (QT == 2) asm volatile("v_max_f32 %0, %0, %1" : "+v"(vm[QT - 1][0]) : "v"(vm[QT - 1][1]));
                    }
                    if (k == 2) vm[0][0] = lane_pair_max(vm[0][0]);
                    if (k == 3 && QT == 2) vm[QT - 1][0] = lane_pair_max(vm[QT - 1][0]);
                    if (k == 4) {
                        mraw[0] = m_pend[0] = vm[0][0];
                        if constexpr (QT == 2) {
                            mraw[QT - 1] = m_pend[QT - 1] = vm[QT - 1][0];
                            asm volatile("" : "+v"(m_pend[0]), "+v"(m_pend[QT - 1]));
                        } else {
                            asm volatile("" : "+v"(m_pend[0]));
                        }
                    }
                    if (k == 6 || k == 7) {
                        const int qt = k - 6;
                        if (qt < QT) any01 |= (__ballot(m_pend[qt < QT ? qt : 0] > thr[qt < QT ? qt : 0]) != 0 ? 1u : 0u) << qt;
                    }
                    if (k == 8) {
                        if constexpr (!FAST) resc_any = any01;
                        if constexpr (RAG) {  // (a window per tile: no pointer chain)
                            kq = tile_g(Kc, Kn, it + 5);
                            vq = tile_g(Vc, Vn, it + 4);
                        } else if constexpr (HOT) {  // (it + 5 < n_kv: the next requests are this item's next tiles down -- FWD: up)
                            kq += ALTW ? dstr : (FWD ? tile_stride : -tile_stride);
                            vq += ALTW ? dstr : (FWD ? tile_stride : -tile_stride);
                        } else if constexpr (ALTW) {  // (an item's ends: by tile index -- either item may walk either way)
                            kq = tile_g(Kc, Kn, it + 5);
                            vq = tile_g(Vc, Vn, it + 4);
                        } else if constexpr (FWD) {
                            kq = (it + 5 == nkc) ? Kn : kq + tile_stride;
                            vq = (it + 4 == nkc) ? Vn : vq + tile_stride;
                        } else {
                            kq = (it + 5 == nkc) ? Kn + (int64_t)(nkn - 1) * tile_stride : kq - tile_stride;
                            vq = (it + 4 == nkc) ? Vn + (int64_t)(nkn - 1) * tile_stride : vq - tile_stride;
                        }
                        if constexpr (R == 1 && !HOTB) seam_st = 0;
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
                // operand u of the visit: 16 K fragments, 16 V fragments, then the first LA K fragments
                // of the NEXT visit (its tile was published by this visit's barrier), so that no LDS
                // latency is exposed at the visit seam
                const char *kt_next = smem + ((R + 2) & 3) * TILE;
                auto operand = [&](int u) -> vec8 {
                    if constexpr (TUNE & 4) return __builtin_bit_cast(vec8, Pw[u % QT][(u >> 1) & 3]);
                    return u < 16 ? k_frag(kt, u) : (u < 32 ? v_frag(vt, u - 16) : k_frag(kt_next, u - 32));
                };
                auto gap_body = [&](auto gap_tag) {
                    constexpr int g = decltype(gap_tag)::value;
                    constexpr int step = g / QT, qt = g % QT;
                    // An MFMA reads its A / B registers for a few cycles after it issues, and hipcc -- to
                    // which the MFMAs are opaque asm -- is free to hand a register that just died to the very
                    // next VALU instruction (seen: the pair-max temporary landing in the A operand of the
                    // MFMA in front of it; one-ulp run-to-run differences that an s_nop 7 behind every MFMA
                    // removed).  So every operand is kept alive until the NEXT MFMA has issued: an empty asm
                    // that names it, placed behind that MFMA (volatile asm statements keep their order).
                    vec8 prev_a = ring[(step + RS - 1) % RS];  // A operand of the previous step (its slot is reloaded below)
                    if constexpr (qt == 0 && (step & 1) == 0) {  // operands in pairs
                        // the counted wait: operands step, step + 1 landed; the LDS reads of the operands behind them up to
                        // step + LA - 1 (one per K fragment, two per V fragment; LDS returns in order) may still fly
                        constexpr int fly = [] { int n = 0; for (int u = step + 2; u < step + LA; ++u) n += (u >= 16 && u < 32) ? 2 : 1; return n; }();
                        FA_JIT(true);
#if defined(FA_TRACE) && FA_TRACE < 4
                        __builtin_amdgcn_s_waitcnt(0xC07F);      // (s_memtime returns out of order: no counting)
#else
                        __builtin_amdgcn_s_waitcnt(0xC07F | (fly << 8));
#endif
#if defined(FA_TRACE) && FA_TRACE == 1
                        asm volatile("s_memtime %0" : "=s"(ts[2 + step / 2]));
#endif
                        ring[(step + LA) % RS] = operand(step + LA);
                        ring[(step + LA + 1) % RS] = operand(step + LA + 1);
                    }
                    if constexpr (g < PH2) {
                        qk_mfma(S_nxt, step, qt, ring[step % RS]);
                    } else {
                        constexpr int s2 = step - 16, s16 = s2 >> 2, t = s2 & 3;
                        E::mfma_acc_a_p(O[qt][t], ring[step % RS], Pw[qt][s16]);
                    }
#if defined(FA_TRACE) && FA_TRACE == 2
                    if constexpr (g >= 48) asm volatile("s_memtime %0" : "=s"(ts[2 + g - 48]));  // fine trace of the visit's last 16 gaps
#endif
                    if constexpr (qt == 0) asm volatile("" ::"v"(prev_a));
                    if constexpr (g >= PH2 + 1) {
                        constexpr int pg = g - 1, ps2 = (pg / QT) - 16;
                        asm volatile("" ::"v"(Pw[pg % QT][ps2 >> 2]));  // B operand of the previous P.V MFMA
                    }
                    if constexpr (g == 0) asm volatile("" ::"v"(Pw[QT - 1][3]));  // ... of the previous visit's last one
                    if constexpr (plan.barrier[g] != 0) sync_point();
                    if constexpr (MASK && (g == 34 || g == 35) && (!HOTB || (R == 3 && !HOT))) {
                        // S(it+1) is complete (last written at gap 31): causal mask, before its row max (whose units start at
                        // gap 36 in the masked plan), one 32-key half per gap.  The tiles a wave's rows meet the diagonal in,
                        // and a ragged sequence's last tiles, are S(0) .. S(3) of an item: S(1) .. S(3) are formed by its
                        // first three visits, S(0) by the LAST visit of the item before it (or the prologue) -- no other
                        // visit carries mask code.
                        if (it + 1 < nkc) {
                            if constexpr (!HOTB) mask_tile(S_nxt, nkc - 2 - it, qb_c, g - 34);
                        } else {
                            mask_tile(S_nxt, nkn - 1, qb_n, g - 34);
                        }
                    }
                    if constexpr (g < GAPS - 1 && plan.dma[g + 1] >= 0 && (plan.dma[g + 1] % DMA_PER_WAVE) == 0 && !(TUNE & 16)) {
                        // M0 (LDS destination) of the tile whose first piece rides in the NEXT gap (pieces 0..3: K, 4..7: V;
                        // the immediate offset of a piece moves its destination on): the write needs one instruction between
                        // it and the DMA, and that gap's MFMA is one (hipcc itself never touches M0 here)
                        asm volatile("s_mov_b32 m0, %0" ::"s"(plan.dma[g + 1] == 0 ? kdst : vdst));
                    }
                    if constexpr (plan.dma[g] >= 0 && !(TUNE & 16)) {  // one 1-KiB DMA piece (its tile's M0 was set a gap or more ago)
                        constexpr int j = plan.dma[g] % DMA_PER_WAVE;
                        static_assert(g > 0 && plan.dma[g - 1] < 0, "a tile's first DMA piece needs the gap before it for its M0");
                        // per-piece lane offsets: 6 more VGPRs than one offset + a scalar piece stride, but 16 fewer SALU
                        // instructions per visit (+0.5 %, round 1)
                        FA_JIT(false);
                        if constexpr (plan.dma[g] < DMA_PER_WAVE) glds16_issue_off<1024 * j>(kq, k_off[j]);
                        else glds16_issue_off<1024 * j>(vq, v_off[j]);
                    }
                    static_for<0, plan.exp_n[g]>([&](auto i) { exp_unit_on(S_cur, plan.exp_first[g] + decltype(i)::value, IntTag<SUM_RS>{}); });
                    if constexpr (ROT_K > 0) {
                        // (rotated plan) the side sums of this tile's early units -- formed in the previous visit's last gaps,
                        // or by head_units -- join the row sums, in the last gaps before the next early units start them again
                        if constexpr (g == 52 && !(TUNE & 32)) { rs[0][0] += rs_e[0][0]; rs[0][1] += rs_e[0][1]; asm volatile("" : "+v"(rs[0][0]), "+v"(rs[0][1])); }
                        if constexpr (g == 53 && !(TUNE & 32)) { rs[1][0] += rs_e[1][0]; rs[1][1] += rs_e[1][1]; asm volatile("" : "+v"(rs[1][0]), "+v"(rs[1][1])); }
                        // ... and the next tile's first units, against the same reference.  In an item's LAST visit that tile
                        // is the next item's S(0), whose reference is not known yet: what the units leave then (packed P, side
                        // sums) is formed again by head_units behind the seam, nothing else is touched
                        static_for<0, plan.early_n[g]>([&](auto i) {
                            constexpr int u = plan.early_first[g] + decltype(i)::value;
                            exp_unit_on(S_nxt, u, IntTag<(u < 2 ? SET_EARLY : SUM_EARLY)>{});
                        });
                    }
                    static_for<0, plan.max_n[g]>([&](auto i) { max_unit(plan.max_first[g] + decltype(i)::value); });
                    if constexpr (plan.tail[g] > 0) tail_step(plan.tail[g]);
                    if constexpr (GUARD_IN_VISIT && R == 3 && QT == 2) {
                        if constexpr (g == 56) g_l0 = vadd(rs[0][0], rs[0][1]);
                        if constexpr (g == 57) g_l1 = vadd(rs[QT - 1][0], rs[QT - 1][1]);
                        if constexpr (g == 58) g_lm = vmax2(g_l0, g_l1);
                        if constexpr (g == 60) guard_hit = __ballot(!(g_lm <= spec_guard<DT>())) != 0;
                    }
                    if constexpr (GUARD_IN_VISIT && R == 3 && QT == 1) {  // (all 16 units have issued by gap 25)
                        if constexpr (g == 27) g_lm = vadd(rs[0][0], rs[0][1]);
                        if constexpr (g == 29) guard_hit = __ballot(!(g_lm <= spec_guard<DT>())) != 0;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                };
                static_for<0, GAPS>([&](auto gap_tag) { gap_body(gap_tag); });
                if constexpr (FAST && MASK && !HOTB) {
                    // masked forms, speculative: a wave's reference is the row max of the FIRST tile it visits that is
                    // not masked whole -- causal: its diagonal tile 4 qb + wave; ragged: the last tile that holds keys
                    // of the sequence (the rounded-up tiles beyond it are masked whole); both: the earlier of the two.
                    // Until then S = -inf gives P = 0 against any finite reference, and m = -inf stands for "0".
                    // When that tile is the S tile this visit formed (masked at gap 34) its row max is taken here;
                    // when it is the item's S(0), the prologue / the seam took it.  Once per item and wave, behind
                    // the stream.
                    const int t_last = RAG ? (args.seq_len + 63) / 64 - 1 : nkc - 1;
                    const int t_diag = 4 * qb_c + wave;
                    const int t_ref = (causal && t_diag < t_last) ? t_diag : t_last;
                    if (it + 1 < nkc && nkc - 2 - it == t_ref) {
                        asm volatile("s_nop 7" ::: "memory");
#pragma unroll
                        for (int qt = 0; qt < QT; ++qt) {
                            float v0 = vmax2(S_nxt[qt][0][0], S_nxt[qt][0][1]), v1 = vmax2(S_nxt[qt][1][0], S_nxt[qt][1][1]);
#pragma unroll
                            for (int r = 2; r < 16; r += 2) {
                                v0 = vmax3(v0, S_nxt[qt][0][r], S_nxt[qt][0][r + 1]);
                                v1 = vmax3(v1, S_nxt[qt][1][r], S_nxt[qt][1][r + 1]);
                            }
                            m[qt] = pair_max(vmax2(v0, v1));
                            neg_msc[qt] = -(finite_or_zero(m[qt]) * c);
                        }
                    }
                }
#if defined(FA_TRACE) && FA_TRACE < 4
                asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts[18])::"memory");
                if (item == args.trace_block && it == args.trace_visit && lane == 0) {
#pragma unroll
                    for (int i = 0; i < 19; ++i) args.trace[wave * 24 + i] = ts[i];
                }
#endif
            };
            // ---- the speculative first pass's guard (see spec_guard in fa_fwd_kernel.hpp) -------------------------------
            // Called between groups of four visits (outside the pinned stream) and once behind an item's last visit.
            // Common path: two adds, a max, a compare, a ballot.  Rare path (some row of this wave has l above the
            // threshold): each such row is brought down by an exact power of two -- O, the row sums and the row's reference --
            // and the wave's O is looked at while it passes through the vector registers: a non-finite element, or a row
            // sum at or beyond the limit, marks the item for the second pass (item_bad).  So a first pass is accepted only if
            // at every checkpoint l was below the limit and, wherever l had risen above the threshold, O was finite.
            bool item_bad = false;
            auto guard = [&](auto &S_cur, const bool behind_last_visit) {
                if constexpr (GUARD_IN_VISIT) {  // (= FAST, but for the tool that times the kernel without the guard)
                    constexpr float kResc = spec_guard<DT>(), kLimit = spec_limit64<DT>();
                    if (__builtin_expect(!guard_hit, 1)) return;  // (decided inside the visit that just ended)
                    float l_q[QT];
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt) l_q[qt] = vadd(rs[qt][0], rs[qt][1]);
                    const float lm = QT == 2 ? vmax2(l_q[0], l_q[QT - 1]) : l_q[0];
                    if (__builtin_expect(__ballot(!(lm <= kResc)) == 0, 1)) return;
                    // ---- rare ----  (per ROW: a row that does not need it keeps its scale -- dragged along by its neighbours'
                    // rescues it would sink towards l = 0 -- and a row that does is brought to l in [1, 2) in one step)
                    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");  // the last P.V MFMAs -> VALU reads of O
                    // NaN-propagating maximum of |o| over the wave's O (v_maximum3_f32: IEEE 754-2019 maximum -- a NaN operand gives
                    // NaN): finite while every element of O is.  (Round 6: 64 instructions per Q tile where "sum of o - o" took 256,
                    // and the multiply in pairs -- heavy-tailed keys send every item of a head through here a few times.)
                    float o_max = 0.0f;
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt) {
                        const float lq = pair_max(l_q[qt]);           // the two lanes of a row decide together
                        if (__ballot(!(lq < kLimit)) != 0) item_bad = true;
                        const bool big = !(lq <= kResc);
                        int e = big ? __builtin_amdgcn_frexp_expf(lq) - 1 : 0;   // lq in [2^e, 2^(e+1))
                        e = e < 0 ? 0 : (e > 126 ? 126 : e);                      // (inf / NaN: the item is marked already)
                        const float sc = __builtin_ldexpf(1.0f, -e);              // a power of two: exact
                        rs[qt][0] *= sc;
                        rs[qt][1] *= sc;
                        if constexpr (ROT_K > 0) { rs_e[qt][0] *= sc; rs_e[qt][1] *= sc; }
                        neg_msc[qt] -= (float)e;
#pragma unroll
                        for (int t = 0; t < DTILES; ++t) {
                            asm volatile("" : "+a"(O[qt][t]));  // (the copies start behind the pads above and end behind the multiply)
#pragma unroll
                            for (int r = 0; r < 16; r += 2) {
                                typedef float f32x2 __attribute__((ext_vector_type(2)));
                                const f32x2 p2 = f32x2{O[qt][t][r], O[qt][t][r + 1]} * sc;  // v_pk_mul_f32
                                O[qt][t][r] = p2[0];
                                O[qt][t][r + 1] = p2[1];
                                asm volatile("v_maximum3_f32 %0, |%1|, |%2|, %0" : "+v"(o_max) : "v"(p2[0]), "v"(p2[1]));
                            }
                            asm volatile("" : "+a"(O[qt][t]));
                        }
                        if constexpr (PSQ) {
                            if (!behind_last_visit) {  // S(it) was formed against the old reference, and so would the next tiles be
#pragma unroll
                                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                                    for (int r = 0; r < 16; ++r) S_cur[qt][nt][r] = vadd(S_cur[qt][nt][r], -(float)e);
#pragma unroll
                                for (int r = 0; r < 16; ++r) Cinit[qt][r] = neg_msc[qt];
                            }
                        }
                    }
                    if constexpr (PSQ) asm volatile("s_nop 1" : "+v"(Cinit[0]), "+v"(Cinit[1]));
                    // rotated plan: the next visit's first ROT_K units were exponentiated against the old references in the last
                    // gaps of the visit that just ended -- their share of the row sums was scaled with the rest above, their packed
                    // P is formed again against the new ones (everything at the old scale moves exactly once: guide T13)
                    if constexpr (ROT_K > 0) {
                        if (!behind_last_visit) static_for<0, ROT_K>([&](auto i) { exp_unit_on(S_cur, decltype(i)::value, IntTag<SUM_NONE>{}); });
                    }
                    if (__ballot(!(o_max < __builtin_inff())) != 0) item_bad = true;  // inf or NaN somewhere in O
                    asm volatile("s_nop 3" ::: "memory");  // accvgpr write -> MFMA accumulator
                }
            };
            // ---- first item: prologue -------------------------------------------------------------
            auto set_next = [&]() {  // coordinates of the item after `item` (or `item` again)
                ord_n = next_ord(ord);
                has_next = ord_n >= 0;
                const int nitem = (int)blockIdx.x + parent(ord_n) * (int)gridDim.x;
                int bh_n;
                coords_of(has_next ? ord_n : ord, bh_n, qb_n);
                rev_n = rev_of(qb_n);
                if (causal) {
                    qb_n = walk_qb(has_next ? nitem : item, qb_n);
                    nkn = 4 * (qb_n + 1);
                }
                const int b_n = bh_n / args.n_heads, h_n = bh_n % args.n_heads;
                const int64_t off_n = (int64_t)b_n * args.batch_stride + (int64_t)h_n * args.head_stride;
                Qn = (const uint16_t *)args.q + off_n;
                Kn = (const uint16_t *)args.k + off_n - DMA_BIAS / 2;
                Vn = (const uint16_t *)args.v + off_n - DMA_BIAS / 2;
                On = (uint16_t *)args.o + off_n;
            };
            set_next();
            // Every CU starts at once and the first requests return at ~11 B/cycle per CU, in issue order:
            // K(0) (common code above) and Q -- all S(0) needs -- go first.  Q travels like the next item's Q
            // does later (coalesced LDS-DMA pieces, then ds_read_b128 into the Q registers; fetched row-per-lane
            // a wave-instruction touches 32 cache lines and the 16 of them took ~9 k cycles to issue): tile 0
            // through this wave's staging area, tile 1 through its quarter of K stage 3 / V stage 3, which
            // are idle until K(3) / V(2) are requested behind the barrier below.  Then K(1), V(0) here and
            // K(2), V(1) | K(3), V(2) under S(0), in the order the counted waits assume.
            FA_TLP(0);  // K(0) requested, next item known
            // (eight waves: the tile's second half through this wave's eighth of K stage 3 / V stage 3, like the second tile of the 64-row form)
            const unsigned q_alt = HSUB ? smem_base + (wave < 4 ? 3 * TILE : V_BASE + 3 * TILE) + (wave & 3) * 4096
                                        : smem_base + (wave < 2 ? 3 * TILE : V_BASE + 3 * TILE) + (wave & 1) * 8192;
            if constexpr (HSUB) {
                request_q(Qg, qb, 0, q_stage, 0);
                request_q(Qg, qb, 0, q_alt, 1);
            } else {
                request_q(Qg, qb, 0, q_stage);
            }
            if constexpr (QT == 2) request_q(Qg, qb, 1, q_alt);
            // (TUNE & 256, tools/tune64.hip: every second workgroup of an XCD asks for V(0) before K(1) -- does the launch's first
            // burst, 256 CUs asking for the same kind of tile at once, go faster out of step?  profiles/r06/s512_floor.txt)
            const bool v_first = (TUNE & 256) != 0 && (((int)blockIdx.x >> 3) & 1) != 0;
            if (v_first) dma_v(tile_g(Vc, Vn, 0), 0);
            dma_k(tile_g(Kc, Kn, 1), 1);
            if (!v_first) dma_v(tile_g(Vc, Vn, 0), 0);
            FA_TLP(1);  // Q, K(1), V(0) requested
            // S(0) of the wave's first 32 rows needs only K(0) and Q tile 0, which land ~1.5 k cycles before Q tile 1
            // (the requests return in issue order at the CU's start-up rate): start on them, take tile 1 when it is in
            if (!(TUNE & 8)) {  // K(0), Q tile 0 landed: [Q tile 1,] K(1), V(0) fly on
                if constexpr (QT == 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                else if constexpr (HSUB) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // (both halves: K(1), V(0) = 2 + 2 pieces fly on)
                else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            }
            FA_TLP(2);  // K(0), Q tile 0 landed
            if constexpr (HSUB) {
                read_q(Qr[0], q_stage, 0);
                read_q(Qr[0], q_alt, 1);
            } else {
                read_q(Qr[0], q_stage);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            barrier();  // K(0) is visible
            FA_TLP(3);  // Q tile 0 read, barrier passed
            {
                // S(0) and its row max, which becomes the first reference max (O = l = 0)
                const char *kt = smem;
                vec8 a_all[16];  // every operand stays allocated until the last MFMA has issued (see visit())
#pragma unroll
                for (int step = 0; step < 16; ++step) a_all[step] = k_frag(kt, step);
                static_for<0, 16>([&](auto step_tag) { qk_mfma(Sa, decltype(step_tag)::value, 0, a_all[decltype(step_tag)::value]); });
                if constexpr (QT == 2) {
                    if (!(TUNE & 8)) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // Q tile 1 landed: K(1), V(0) fly on
                    read_q(Qr[QT - 1], q_alt);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    static_for<0, 16>([&](auto step_tag) { qk_mfma(Sa, decltype(step_tag)::value, QT - 1, a_all[decltype(step_tag)::value]); });
                }
                // the rest of the first requests, issued while the matrix pipe works through S(0): a CU keeps
                // only ~32 KB of requests in flight, so asking for all 176 KB up front held the waves at the
                // issue of the last pieces (~11 k cycles) long after K(0) and Q had landed
                dma_k(tile_g(Kc, Kn, 2), 2);
                dma_v(tile_g(Vc, Vn, 1), 1);
                barrier();  // every wave has read its Q tile 1 out of stages 3, which K(3) now overwrites
                dma_k(tile_g(Kc, Kn, 3), 3);
                dma_v(tile_g(Vc, Vn, 2), 2);
                kq = tile_g(Kc, Kn, 4);
                vq = tile_g(Vc, Vn, 3);
                if (has_next) request_next_q(0);
                FA_TLP(4);  // S(0) MFMAs and the remaining requests issued
                asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // MFMA D -> VALU read
#pragma unroll
                for (int step = 0; step < 16; ++step) asm volatile("" ::"v"(a_all[step]));
                mask_tile(Sa, nkc - 1, qb_c);
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) {
                    float v = Sa[qt][0][0];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) v = fmaxf(v, Sa[qt][nt][r]);
                    m[qt] = pair_max(v);
                    neg_msc[qt] = -(finite_or_zero(m[qt]) * cs);
                    thr[qt] = m[qt] + TAU / cs;
                    m_pend[qt] = m[qt];
                }
                set_cinit(Sa);
                head_units(Sa);
                if (!(TUNE & 8)) {  // K(1) landed (under S(0)); younger: V(0), K(2), V(1) [, K(3), V(2) [, the next Q tile 0]]
                    if (v_first) {  // (V(0) is older than K(1) here)
                        if (has_next) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
                        else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                    } else if constexpr (HSUB) {   // (ten pieces of five tiles [, the next Q's first half: four])
                        if (has_next) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
                        else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
                    } else if (has_next) asm volatile("s_waitcnt vmcnt(28)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
                }
                FA_TLP(5);  // row max done, K(1) landed
                barrier();
                FA_TLF();
                FA_TL();  // S(0) formed, K(1) landed
                FA_TL4(62);
#pragma unroll
                for (int u = 0; u < LA; ++u) ring[u] = k_frag(smem + TILE, u);  // first operands of visit 0: K(1)
            }
#include "fa_epilogue64.inc"   // store_item(Oc, qb_c, ord, zero_behind): the item's O through the staging area to global memory
            // seq_len is a multiple of B_r = 256, so n_kv = seq_len / 64 is a multiple of 4 = ring depth.
            // The item loop, in groups of four visits (one per ring stage): an item's first group and its last two run the
            // GENERAL visits (whatever an item's ends need, decided at run time), the groups in between the HOT ones (see
            // visit()).  Straight-line by construction -- [first group] [hot loop] [loop over the last two groups] [epilogue
            // + seam] -- because alternative visit bodies that join behind a branch (or a loop both reach) made hipcc
            // shuffle the accumulator tiles between them and spill.  TUNE & 128 (tools/tune64.hip): no hot region, as round 4 ran.
            constexpr bool HOT_LOOP = !(TUNE & 128);
            for (;;) {
                int it = 0;
                if constexpr (HOT_LOOP) {
                    visit(0, Sa, Sb, IntTag<0>{}, IntTag<0>{});
                    visit(1, Sb, Sa, IntTag<1>{}, IntTag<0>{});
                    visit(2, Sa, Sb, IntTag<2>{}, IntTag<0>{});
                    visit(3, Sb, Sa, IntTag<3>{}, IntTag<0>{});
                    for (it = 4; it + 12 <= nkc; it += 4) {
                        guard(Sa, false);
                        visit(it, Sa, Sb, IntTag<0>{}, IntTag<2>{});
                        visit(it + 1, Sb, Sa, IntTag<1>{}, IntTag<2>{});
                        visit(it + 2, Sa, Sb, IntTag<2>{}, IntTag<2>{});
                        visit(it + 3, Sb, Sa, IntTag<3>{}, IntTag<2>{});
                    }
                    for (; it < nkc; it += 4) {  // the last two groups (or the last one: n_kv = 8)
                        guard(Sa, false);
                        visit(it, Sa, Sb, IntTag<0>{}, IntTag<1>{});
                        visit(it + 1, Sb, Sa, IntTag<1>{}, IntTag<1>{});
                        visit(it + 2, Sa, Sb, IntTag<2>{}, IntTag<1>{});
                        visit(it + 3, Sb, Sa, IntTag<3>{}, IntTag<1>{});
                    }
                } else {
                    for (; it < nkc; it += 4) {
                        if (it) guard(Sa, false);
                        visit(it, Sa, Sb, IntTag<0>{}, IntTag<0>{});
                        visit(it + 1, Sb, Sa, IntTag<1>{}, IntTag<0>{});
                        visit(it + 2, Sa, Sb, IntTag<2>{}, IntTag<0>{});
                        visit(it + 3, Sb, Sa, IntTag<3>{}, IntTag<0>{});
                    }
                }
#if defined(FA_TRACE) && FA_TRACE < 4
                unsigned long long te0, te1, te2;
                asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(te0)::"memory");
#endif
#if defined(FA_TRACE) && FA_TRACE >= 4
                const bool tl_seam = FA_TRACE == 5 && ord == 0;  // fine stamps of the first seam: slots 50 .. 55
                if (tl_seam) tl_at(50);  // last visit done
#endif
                guard(Sa, true);
                const int qb_st = qb_c;  // the item being stored
                (void)qb_st;
                store_item(Oc, qb_c, ord, has_next && !(TUNE & 128));
#if defined(FA_TRACE) && FA_TRACE >= 4
                if (tl_seam) tl_at(51);  // epilogue issued
#endif
#if defined(FA_TRACE) && FA_TRACE < 4
                asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(te1)::"memory");
#endif
                if (!has_next) break;
                // ---- seam: the last visit left the next item's S(0) in Sa and its row max in mraw; its
                // first tiles are landed or in flight, its first operands sit in the ring
                ord = ord_n;
                item = (int)blockIdx.x + parent(ord) * (int)gridDim.x;
                Kc = Kn; Vc = Vn; Oc = On; qb_c = qb_n;
                nkc = nkn;
                rev_c = rev_n;
                dstr = rev_c ? -tile_stride : tile_stride;
                set_next();
#if defined(FA_TRACE) && FA_TRACE >= 4
                if (tl_seam) tl_at(53);  // coordinates of the item after
#endif
                kq = tile_g(Kc, Kn, 4);  // visit 0 requests K(4), V(3) (for n_kv == 4 that is already the item after)
                vq = tile_g(Vc, Vn, 3);
                if (has_next) request_next_q(0);  // (the staging area is free again: store_item's reads have retired)
                seam_st = 8 * QT;
                if constexpr (RAG) {  // stores the epilogue above issued: one per four rows inside the sequence (16 per 64 rows)
                    const int rows_in = args.seq_len - (qb_st * TR::kBr + wave * TR::kRowsPerWave);
                    seam_st = rows_in >= 64 ? 16 : (rows_in >= 32 ? 8 : 0);
                }
                resc_any = 0;
#if defined(FA_TRACE) && FA_TRACE >= 4
                if (tl_seam) tl_at(54);  // next Q tile 0 requested
#endif
                if constexpr (FAST) {
                    // the row max of the S tile the last visit formed (the next item's S(0)): the speculative
                    // schedule has no row-max units, this is the only one an item needs (behind store_item's pads)
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt) {
                        float v0 = vmax2(Sa[qt][0][0], Sa[qt][0][1]), v1 = vmax2(Sa[qt][1][0], Sa[qt][1][1]);
#pragma unroll
                        for (int r = 2; r < 16; r += 2) {
                            v0 = vmax3(v0, Sa[qt][0][r], Sa[qt][0][r + 1]);
                            v1 = vmax3(v1, Sa[qt][1][r], Sa[qt][1][r + 1]);
                        }
                        mraw[qt] = pair_max(vmax2(v0, v1));
                    }
                }
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) {
                    m[qt] = mraw[qt];
                    neg_msc[qt] = -(finite_or_zero(m[qt]) * cs);
                    thr[qt] = m[qt] + TAU / cs;
                    m_pend[qt] = m[qt];
                    rs[qt][0] = rs[qt][1] = 0.0f;
                }
                item_bad = false;
                set_cinit(Sa);
                head_units(Sa);
#if defined(FA_TRACE) && FA_TRACE >= 4
                if (tl_seam) tl_at(55);  // row max of S(0), softmax state reset
#endif
                if constexpr ((TUNE & 128) != 0) zero_o();  // (otherwise cleared inside store_item)
#if defined(FA_TRACE) && FA_TRACE >= 4
                if (tl_seam) tl_at(52);  // O = 0 issued
#endif
#if defined(FA_TRACE) && FA_TRACE < 4
                asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(te2)::"memory");
                if (item == args.trace_block + (int)gridDim.x && lane == 0) {
                    args.trace[wave * 24 + 21] = te0;
                    args.trace[wave * 24 + 22] = te1;
                    args.trace[wave * 24 + 23] = te2;
                }
#endif
            }
            dma_wait();  // nothing may still be landing in the LDS when the workgroup retires (or the next pass starts)
            FA_TL();
#if defined(FA_TRACE) && FA_TRACE >= 4
            if constexpr (FAST || !SPEC) {
                tl_at(63);
                tl_real(47);
                ((unsigned *)args.trace)[(wave * 256 + (int)blockIdx.x) * 64 + lane] = tlv;
            }
#endif
            return failed;
        }
    };  // walk

    if constexpr (SPEC) {
        unsigned long long failed = walk(BoolTag<true>{}, IntTag<QTP>{}, ~0ull, 0ull);
        // every wave's failures -> workgroup-uniform masks.  Each wave leaves its mask in its OWN O staging
        // area (beside the rings: no K / V piece ever lands there, and its own epilogue reads have retired),
        // so one barrier publishes all four.
        char *slot = smem + 2 * TR::kStages * TILE;
        if (lane == 0) *(unsigned long long *)(slot + wave * TR::kStageBytes) = failed;
        barrier();
        auto uniform64 = [&](unsigned long long x) {
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)x);
            const unsigned hi32 = __builtin_amdgcn_readfirstlane((unsigned)(x >> 32));
            return ((unsigned long long)hi32 << 32) | lo;
        };
        // the 64-row speculative plain form redoes a failed item as its failed HALVES (see walk): waves 0 and 1 own an item's
        // rows 0 .. 127, waves 2 and 3 its rows 128 .. 255
        constexpr bool HALVES = QTP == 2 && !MASK && !PSQ;
        unsigned long long lo_half = 0, hi_half = 0;
#pragma unroll
        for (int w = 0; w < NWAVES; ++w) {
            const unsigned long long f_w = *(const unsigned long long *)(slot + w * TR::kStageBytes);
            if (HALVES && w >= NWAVES / 2) hi_half |= f_w;
            else lo_half |= f_w;
        }
        lo_half = uniform64(lo_half);
        hi_half = uniform64(hi_half);
        unsigned redone = 0;
        if (lo_half | hi_half) {
            barrier();  // (all four slots read before the second pass requests its first Q tile into one of them)
            if constexpr (HALVES) redone = (unsigned)walk(BoolTag<false>{}, IntTag<1>{}, lo_half, hi_half);
            else redone = (unsigned)walk(BoolTag<false>{}, IntTag<QTP>{}, lo_half, 0ull);
        }
        if (redone && threadIdx.x == 0) report_redo(args);
        // (eight waves: a 256-row item is TWO Q blocks of the (128, 64, 4) configuration it serves -- counted as such)
        constexpr unsigned PER_ITEM = NW == 8 ? 2 : 1;
        if (args.stats && threadIdx.x == 0) {  // fa_fwd_stats: this workgroup's items, and how many of them ran twice
            const long long n_items = (long long)args.n_bh * args.n_q_blocks;
            atomicAdd(args.stats, PER_ITEM * (unsigned)((n_items - (long long)blockIdx.x + (long long)gridDim.x - 1) / (long long)gridDim.x));
            if (redone) atomicAdd(args.stats + 1, PER_ITEM * redone);
        }
    } else {
        walk(BoolTag<false>{}, IntTag<QTP>{}, ~0ull, 0ull);
        if (args.stats && threadIdx.x == 0) {
            const long long n_items = (long long)args.n_bh * args.n_q_blocks;
            atomicAdd(args.stats, (NW == 8 ? 2u : 1u) * (unsigned)((n_items - (long long)blockIdx.x + (long long)gridDim.x - 1) / (long long)gridDim.x));
        }
    }
}

}  // namespace fa
