// fa_fwd_kernel16.hpp -- 16-Q-rows-per-wave variant of the gfx950 forward kernel.
//
// Same algorithm and LDS K image as fa_fwd_kernel.hpp, built on
// v_mfma_f32_16x16x32_{bf16,f16} so that a 4-wave workgroup covers B_r = 64 -- the
// reference's (B_r = 64, n_warps = 4) configurations (kernel_configs.py:389-423).
// In the 16x16 C layout (col = lane&15, row = 4*(lane>>4) + reg) a query column is
// shared by FOUR lanes (lane, ^16, ^32, ^48): the row max / final row sum take two
// cross-lane steps instead of one.  P^T again feeds the PV MFMA without a shuffle:
// a 32-key contraction chunk pairs 16-key tiles 2u and 2u+1, lane group g owning
// keys {32u + 4g + 0..3} u {32u + 16 + 4g + 0..3}, and V^T's A operand is fetched
// with that same key order by two ds_read_b64_tr_b16.
//   V image  [key/16][d/16][16 keys][16 d]  512-B subtiles: each transpose-read
//            wave-instruction covers 512 contiguous bytes.
#pragma once
#include "fa_fwd_kernel.hpp"
#include "fa_registry.hpp"

namespace fa {

template <int DT, int NWAVES, int BC, bool SWZ, bool EAGER, bool OPT>
struct FwdTraits16 {
    static constexpr int kRowsPerWave = 16;
    static constexpr int kBr = 16 * NWAVES;
    static constexpr int kThreads = NWAVES * 64;
    static constexpr int kTileBytes = BC * 256;
    static constexpr int kStages = EAGER ? 2 : 1;
    static constexpr int kKvBytes = 2 * kStages * kTileBytes;
    static constexpr int kOutBytes = kBr * 256;
    static constexpr int kLdsBytes = kKvBytes > kOutBytes ? kKvBytes : kOutBytes;
};

// Reductions over the four lanes that share a query column (lane, ^16, ^32, ^48) by the gfx950 row
// exchanges, in the vector ALU: v_permlane16_swap trades the odd 16-lane rows of its first operand for
// the even rows of its second, v_permlane32_swap the upper half for the lower half, so with the same
// value in both operands the two results are {own pair's even, own pair's odd} row / {lower, upper}
// half.  (__shfl_xor lowers to ds_bpermute: an LDS-path round trip on the row max -> rescale -> exp2
// critical path of every tile.)
static FA_DEV float quad_max(float x) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    x = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
static FA_DEV float quad_sum(float x) {
    auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    x = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

template <int DT, int NWAVES, int BC, bool SWZ, bool EAGER, bool OPT>
__global__ void
__launch_bounds__(NWAVES * 64, 2)
fa_fwd_kernel16(const KernelArgs args) {
    using E = Elem<DT>;
    using vec8 = typename E::vec8;
    using TR = FwdTraits16<DT, NWAVES, BC, SWZ, EAGER, OPT>;
    constexpr int D = 128;
    constexpr int KT = BC / 16;      // 16-key tiles per LDS tile
    constexpr int KU = BC / 32;      // 32-key contraction chunks
    constexpr int KS = D / 32;       // k steps of QK^T
    constexpr int DT16 = D / 16;     // 16-wide d tiles of O^T
    constexpr int TILE = TR::kTileBytes;
    constexpr int N_DMA = BC / 4;
    static_assert(N_DMA % NWAVES == 0, "tile/wave split");
    constexpr int DMA_PER_WAVE = N_DMA / NWAVES;
    constexpr int V_BASE = TR::kStages * TILE;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r15 = lane & 15;
    const int g = lane >> 4;

    const int nq = args.n_q_blocks;
    int bh, qb;
    {
        const int bid = blockIdx.x;
        if ((args.n_bh & 7) == 0) {
            const int xcd = bid & 7, local = bid >> 3;
            bh = (local / nq) * 8 + xcd;
            qb = local % nq;
        } else {
            bh = bid / nq;
            qb = bid % nq;
        }
    }
    const int b = bh / args.n_heads, h = bh % args.n_heads;
    const int64_t ss = args.seq_stride;
    const int64_t head_off = (int64_t)b * args.batch_stride + (int64_t)h * args.head_stride;
    const uint16_t *Qg = (const uint16_t *)args.q + head_off;
    const uint16_t *Kg = (const uint16_t *)args.k + head_off;
    const uint16_t *Vg = (const uint16_t *)args.v + head_off;
    uint16_t *Og = (uint16_t *)args.o + head_off;

    // DMA source offsets.  K image as in fa_fwd_kernel.hpp.  V: chunk p -> subtile
    // p>>5 = (key>>4)*8 + (d>>4); inside: key&15 = (p&31)>>1, d&15 = (p&1)*8.
    const int k_row_in_piece = lane >> 4;
    const int k_swz = SWZ ? (((wave & 3) * 4 + k_row_in_piece) & 15) : 0;
    const int64_t k_lane_off = (int64_t)k_row_in_piece * ss + (((lane & 15) ^ k_swz) << 3);
    const int v_w = lane & 31;

    const unsigned smem_base = __builtin_amdgcn_readfirstlane(lds_addr(smem));
    // SGPR base (head + tile, scalar ALU) + 32-bit per-lane byte offset, invariant over tiles
    unsigned k_off[DMA_PER_WAVE], v_off[DMA_PER_WAVE];
#pragma unroll
    for (int j = 0; j < DMA_PER_WAVE; ++j) {
        const int i = wave + NWAVES * j;
        k_off[j] = (unsigned)(((int64_t)(4 * i) * ss + k_lane_off) * 2);
        const int sub = 2 * i + (lane >> 5);
        v_off[j] = (unsigned)((((int64_t)(16 * (sub >> 3) + (v_w >> 1))) * ss + (sub & 7) * 16 + (v_w & 1) * 8) * 2);
    }
    auto issue_tile = [&](int kv_block, int stage) {
        const int64_t kv0 = (int64_t)kv_block * BC;
        const unsigned kdst = smem_base + stage * TILE;
        const unsigned vdst = smem_base + V_BASE + stage * TILE;
#pragma unroll
        for (int j = 0; j < DMA_PER_WAVE; ++j) {
            glds16_sv(Kg + kv0 * ss, k_off[j], kdst + (wave + NWAVES * j) * 1024);
            glds16_sv(Vg + kv0 * ss, v_off[j], vdst + (wave + NWAVES * j) * 1024);
        }
    };

    int kv_block = args.n_kv_blocks - 1;
    if (EAGER) issue_tile(kv_block, 0);
    // SPEC (the OPT build of the double-buffered variants; reached through fa_fwd_opts.speculative): the speculative softmax (see
    // fa_fwd_kernel.hpp): attempt<FAST> keeps the first visited tile's row max as the reference for all
    // tiles -- no quad reduction, no rescale -- and checks l at the end; a workgroup that fails starts over.
    constexpr bool SPEC = OPT && EAGER;

    vec8 Qr[KS];
    {
        const int64_t row = (int64_t)qb * TR::kBr + wave * 16 + r15;
        const uint16_t *qp = Qg + row * ss + g * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) Qr[ks] = *(const vec8 *)(qp + ks * 32);
    }

    const float c = (float)((double)(1.0f / __builtin_sqrtf((float)D)) * 1.4426950408889634074);

    f32x4 O[DT16];
    float m, l;
    auto reset_state = [&]() {
#pragma unroll
        for (int t = 0; t < DT16; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) O[t][r] = 0.0f;
        m = -__builtin_inff();
        l = 0.0f;
    };
    reset_state();

    const int ka_swz = SWZ ? r15 : 0;
    const int ka_base = r15 * 256;
    const int li = lane & 15;
    const int va_base = (4 * g + (li >> 2)) * 32 + (li & 3) * 8;

    auto compute_tile = [&](int stage, auto first_tag, auto fast_tag) {
        constexpr bool FIRST = decltype(first_tag)::value;
        constexpr bool FAST = decltype(fast_tag)::value;
        const char *kt_ptr = smem + stage * TILE;
        const char *vt_ptr = smem + V_BASE + stage * TILE;

        f32x4 S[KT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) S[kt][r] = 0.0f;
        // k step outer / key tile inner: consecutive MFMAs accumulate into different tiles;
        // operand reads kept 8 ahead of the matrix pipe (see fa_fwd_kernel.hpp 3.2)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                const int off = kt * 16 * 256 + ka_base + (((4 * ks + g) ^ ka_swz) << 4);
                const vec8 a = *(const vec8 *)(kt_ptr + off);
                S[kt] = E::mfma16(a, Qr[ks], S[kt]);
            }
        }
        sched_mfma_fed_from_lds<KT * KS, 1, 8>();

        float mx = S[0][0];
        if constexpr (!FAST || FIRST) {
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, S[kt][r]);
            mx = quad_max(mx);
        }
        float m_new;
        if (FAST && !FIRST) {
            m_new = m;
        } else if (FIRST && OPT) {
            m_new = mx;
        } else {
            m_new = fmaxf(m, mx);
            const float alpha = __builtin_amdgcn_exp2f((m - m_new) * c);
            l *= alpha;
            if (!__all(alpha == 1.0f)) {  // multiplying by 1.0f is the identity: bit-exact skip
#pragma unroll
                for (int t = 0; t < DT16; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) O[t][r] *= alpha;
            }
        }
        m = m_new;
        const float neg_msc = -(m_new * c);
        float rowsum = 0.0f;
        vec8 Pb[KU];
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            float p[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                p[j] = __builtin_amdgcn_exp2f(__builtin_fmaf(S[2 * u + (j >> 2)][j & 3], c, neg_msc));
                rowsum += p[j];
            }
            Pb[u] = E::pack8(p);
        }
        l = (FIRST && OPT) ? rowsum : l + rowsum;

#pragma unroll
        for (int u = 0; u < KU; ++u) {
#pragma unroll
            for (int t = 0; t < DT16; ++t) {
                const char *vp = vt_ptr + va_base + u * 8192 + t * 512;
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((FA_LDS(s16x4) *)(vp));
                const s16x4 up = __builtin_amdgcn_ds_read_tr16_b64_v4i16((FA_LDS(s16x4) *)(vp + 4096));
                s16x8 av;
                av.lo = lo;
                av.hi = up;
                O[t] = E::mfma16(__builtin_bit_cast(vec8, av), Pb[u], O[t]);
            }
        }
        sched_mfma_fed_from_lds<DT16 * KU, 2, 8>();
    };

    using TrueTag = BoolTag<true>;
    using FalseTag = BoolTag<false>;
    const int n_kv = args.n_kv_blocks;
    auto attempt = [&](auto fast_tag) -> bool {
        constexpr bool FAST = decltype(fast_tag)::value;
        if (EAGER) {
            dma_wait_all();
            wg_barrier();
            if (n_kv > 1) issue_tile(kv_block - 1, 1);
            if (OPT) compute_tile(0, TrueTag{}, fast_tag); else compute_tile(0, FalseTag{}, fast_tag);
            for (int it = 1; it < n_kv; ++it) {
                const int stage = it & 1;
                dma_wait_all();
                wg_barrier();
                if (it + 1 < n_kv) issue_tile(kv_block - it - 1, stage ^ 1);
                compute_tile(stage, FalseTag{}, fast_tag);
            }
        } else {
            for (int it = 0; it < n_kv; ++it) {
                if (it > 0) wg_barrier();
                issue_tile(kv_block - it, 0);
                dma_wait_all();
                wg_barrier();
                if (OPT && it == 0) compute_tile(0, TrueTag{}, fast_tag); else compute_tile(0, FalseTag{}, fast_tag);
            }
        }
        wg_barrier();  // every wave is done with the K/V stages
        if constexpr (FAST) {  // l >= every P of the row: below the limit nothing overflowed; one verdict per workgroup
            constexpr float kLimit = spec_limit<DT>();
            bool bad = !(quad_sum(l) < kLimit);
            if constexpr (DT == 15) {  // bf16: the accumulators themselves (see spec_limit)
                float nonfinite = 0.0f;
#pragma unroll
                for (int t = 0; t < DT16; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) nonfinite += O[t][r] - O[t][r];
                bad |= !(nonfinite == 0.0f);
            }
            const int wave_bad = __ballot(bad) != 0 ? 1 : 0;
            if (lane == 0) *(int *)(smem + wave * 4) = wave_bad;
            wg_barrier();
            int any = 0;
#pragma unroll
            for (int w = 0; w < NWAVES; ++w) any |= *(const int *)(smem + w * 4);
            any = __builtin_amdgcn_readfirstlane(any);
            wg_barrier();
            return any == 0;
        }
        return true;
    };
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
            issue_tile(kv_block, 0);
        }
        attempt(FalseTag{});
    }

    // epilogue: O staged through LDS (each wave its own 16 rows), stored as whole rows
    {
        const float inv = 1.0f / quad_sum(l);
        char *stage_o = smem + wave * (16 * 256);
        char *wp = stage_o + r15 * 256 + (g & 1) * 8;
#pragma unroll
        for (int t = 0; t < DT16; t += 2) {
            float o[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                o[r] = O[t][r] * inv;          // d = 16t + 4g + r      -> chunk 2t + (g>>1), half g&1
                o[4 + r] = O[t + 1][r] * inv;  // d = 16(t+1) + 4g + r
            }
            const s16x8 packed = __builtin_bit_cast(s16x8, E::pack8(o));
            *(s16x4 *)(wp + (((2 * t + (g >> 1)) ^ r15) << 4)) = packed.lo;
            *(s16x4 *)(wp + (((2 * (t + 1) + (g >> 1)) ^ r15) << 4)) = packed.hi;
        }
        const int64_t row0 = (int64_t)qb * TR::kBr + wave * 16;
        const int rsub = lane >> 4, chunk = lane & 15;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = 4 * i + rsub;
            const s16x8 v = *(const s16x8 *)(stage_o + row * 256 + ((chunk ^ (row & 15)) << 4));
            *(s16x8 *)(Og + (row0 + row) * ss + chunk * 8) = v;
        }
    }
}

template <int DT, int NWAVES, int BC, bool SWZ, bool EAGER, bool OPT>
constexpr KernelEntry make_entry16() {
    using TR = FwdTraits16<DT, NWAVES, BC, SWZ, EAGER, OPT>;
    return KernelEntry{DT, 16, NWAVES, BC, SWZ, EAGER, OPT, 0, 1, 0, 128, TR::kThreads, TR::kLdsBytes, 0,
                       (kernel_fn)&fa_fwd_kernel16<DT, NWAVES, BC, SWZ, EAGER, OPT>, nullptr,
                       softmax_mode_of(false, OPT, EAGER, true, false), 0};
}

}  // namespace fa
