/*
 * fa_oracle.c -- CPU restatement of the reference's Flash-Attention-2 forward.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (the package
 * flash_attention_from_scratch_amd/, libfa_hip.so) links, loads or calls this
 * file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * Parity status: PINNED.  The reference holds no golden vectors of its own
 * (SURVEY.md 8c); the restatement is checked against fixtures generated in the
 * build container by importing the reference's Python oracles
 * (py/flash_helpers/test/utils.py:137-162 py_flash_attention and
 * tools/debug/debug.py:40-153 block_flash_attention) -- see oracle/gen_golden.py
 * and tests/golden/.
 *
 * What is restated (all /root/reference paths):
 *   src/include/forward_kernel.cuh:19-83   process_kv_block   -> kv_block_step()
 *   src/include/forward_kernel.cuh:85-204  flash_forward_kernel -> q_block_forward()
 *   src/include/softmax.cuh:13-34          calc_row_max (raw-logit running max)
 *   src/include/softmax.cuh:36-49          scale_l_O    (exp2((m_prev-m)*c))
 *   src/include/softmax.cuh:51-64          exponentiate_tensor (exp2(s*c - m*c))
 *   src/include/softmax.cuh:66-83          update_row_exp_sum  (fp32 P, pre-rounding)
 *   src/include/softmax.cuh:107-128        final_softmax_normalization (O *= 1/l)
 *   src/include/load_store.cuh:314-353     convert_to_16_bit_dtype (RNE; P and O)
 *   src/flash_attention.cu:100-112         grid = (S/B_r, H, B), reverse KV order
 *
 * Invariants reproduced: KV blocks visited last-to-first; the running max is of
 * the UNSCALED logits and the scale c = rsqrt(d)*log2(e) is applied inside the
 * base-2 exponent; P is rounded to the 16-bit type before P.V while l sums the
 * fp32 P; O is multiplied by 1/l and rounded RNE to the 16-bit type.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define FA_ORACLE_FP16 5  /* torch ScalarType codes, kernel_configs.py:12-13 */
#define FA_ORACLE_BF16 15

/* ---------- 16-bit <-> fp32, round-to-nearest-even ---------- */
static inline float u32_as_f32(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f32_as_u32(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

static inline float bf16_to_f32(uint16_t h) { return u32_as_f32((uint32_t)h << 16); }
static inline uint16_t f32_to_bf16(float f) {
    uint32_t u = f32_as_u32(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40); /* NaN */
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

static inline float fp16_to_f32(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1fu;
    uint32_t man = h & 0x3ffu;
    if (exp == 0) {
        if (man == 0) return u32_as_f32(sign);
        float v = (float)man * 5.9604644775390625e-8f; /* 2^-24 */
        return sign ? -v : v;
    }
    if (exp == 31) return u32_as_f32(sign | 0x7f800000u | (man << 13));
    return u32_as_f32(sign | ((exp + 112u) << 23) | (man << 13));
}
static inline uint16_t f32_to_fp16(float f) {
    uint32_t u = f32_as_u32(f);
    uint16_t sign = (uint16_t)((u >> 16) & 0x8000u);
    uint32_t a = u & 0x7fffffffu;
    if (a > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);          /* NaN */
    if (a >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);          /* >= 65520 -> inf */
    if (a < 0x33000001u) return sign;                                 /* <= 2^-25 -> 0 */
    if (a < 0x38800000u) {                                            /* subnormal half */
        float v = u32_as_f32(a) * 16777216.0f;                        /* * 2^24, exact */
        float r = nearbyintf(v);                                      /* RNE (default mode) */
        return (uint16_t)(sign | (uint16_t)r);
    }
    uint32_t e = a - 0x38000000u;                                     /* rebias 127 -> 15 */
    e += 0xfffu + ((e >> 13) & 1u);
    return (uint16_t)(sign | (uint16_t)(e >> 13));
}

static inline float b16_to_f32(uint16_t h, int dtype) {
    return dtype == FA_ORACLE_BF16 ? bf16_to_f32(h) : fp16_to_f32(h);
}
static inline uint16_t f32_to_b16(float f, int dtype) {
    return dtype == FA_ORACLE_BF16 ? f32_to_bf16(f) : f32_to_fp16(f);
}
static inline float round_b16(float f, int dtype) { return b16_to_f32(f32_to_b16(f, dtype), dtype); }

/* exported so tests can pin the conversions against torch */
float fa_oracle_b16_to_f32(uint16_t h, int dtype) { return b16_to_f32(h, dtype); }
uint16_t fa_oracle_f32_to_b16(float f, int dtype) { return f32_to_b16(f, dtype); }

typedef struct {
    const uint16_t *q, *k, *v;
    uint16_t *o;
    int dtype;
    int64_t batch, seq, heads, d;
    int64_t bs, ss, hs; /* strides in elements, flash_attention.cu:84-86 */
    int prescale_q;     /* NOT the reference's arithmetic: logits from a 16-bit Q * c (the device's pre-scaled-Q option) */
    int kv_forward;     /* NOT the reference's order (forward_kernel.cuh:142 walks last-to-first): KV blocks first-to-last, as
                         * the device's speculative first pass walks them since round 6 (its reference is the row max of the
                         * first block it visits, and attention sinks sit at the first keys).  A value G >= 2: first-to-last
                         * for the Q blocks with (qb / G) even, [block 0, then last-to-second] for the others -- the device's
                         * alternating form for long sequences (fa_fwd_kernel64<..., ALT>; G = its workgroups per XCD) */
} tensors_t;

/*
 * One (batch, head, Q block): forward_kernel.cuh:85-204.  qf/kf/vf are this
 * head's rows upcast to fp32 (exact).  Scratch: S[B_r*B_c], O[B_r*d], m, l [B_r].
 * round_p = 1 reproduces the device (P -> 16 bit before PV); 0 keeps fp32 P
 * (the tools/debug/debug.py:block_flash_attention arithmetic).
 * If m_trace/l_trace are non-NULL they receive the final m (raw logits) and l.
 */
static void q_block_forward(const tensors_t *t, int64_t b, int64_t h, int64_t qb, int B_r,
                            int B_c, int round_p, int optimized_softmax, const float *kf,
                            const float *vf, float *S, float *O, float *m, float *l,
                            float *qrow, float *m_trace, float *l_trace, int causal,
                            float lazy_tau) {
    const int64_t d = t->d;
    /* Scope wideners (not in the reference): seq need not be a multiple of the
     * tiles, and an optional causal mask.  Masked logits are -inf; a row whose keys were all
     * masked so far keeps m = -inf and exponentiates against 0 (as the device kernel does). */
    const int64_t rows = (qb * B_r + B_r <= t->seq) ? B_r : t->seq - qb * B_r;
    int64_t n_kv = (t->seq + B_c - 1) / B_c;
    if (causal) {
        const int64_t need = (qb * B_r + rows - 1) / B_c + 1;
        if (need < n_kv) n_kv = need;
    }
    /* forward_kernel.cuh:150-151: rsqrt(d) * M_LOG2E evaluated in fp32 */
    const float c_ref = (float)((double)(1.0f / sqrtf((float)d)) * M_LOG2E);
    /* pre-scaled Q (fa_fwd_opts.prescaled_q): Q' = RNE16(Q * c) once, then the exponent is the raw dot
     * product K . Q' -- c applied BEFORE the 16-bit rounding of Q instead of in fp32 behind the dot product */
    const float c = t->prescale_q ? 1.0f : c_ref;
    for (int r = 0; r < B_r; ++r) { m[r] = -INFINITY; l[r] = 0.0f; }
    memset(O, 0, sizeof(float) * B_r * d);

    for (int64_t step = 0; step < n_kv; ++step) { /* forward_kernel.cuh:142,179-184: last to first (kv_forward: first to last) */
        int64_t blk = t->kv_forward ? step : n_kv - 1 - step;
        if (t->kv_forward >= 2 && ((qb / t->kv_forward) & 1)) blk = step == 0 ? 0 : n_kv - step;
        const int is_first = (step == 0);
        /* S = Q K^T, fp32 accumulate (gemm.cuh:45-87, mma f32 accum) */
        for (int r = 0; r < rows; ++r) {
            const int64_t qi = qb * B_r + r;
            const uint16_t *qp = t->q + b * t->bs + qi * t->ss + h * t->hs;
            for (int64_t x = 0; x < d; ++x) {
                qrow[x] = b16_to_f32(qp[x], t->dtype);
                if (t->prescale_q) qrow[x] = round_b16(qrow[x] * c_ref, t->dtype);
            }
            for (int cidx = 0; cidx < B_c; ++cidx) {
                const int64_t key = blk * B_c + cidx;
                if (key >= t->seq || (causal && key > qi)) { S[r * B_c + cidx] = -INFINITY; continue; }
                const float *kp = kf + key * d;
                float acc = 0.0f;
                for (int64_t x = 0; x < d; ++x) acc = fmaf(qrow[x], kp[x], acc);
                S[r * B_c + cidx] = acc;
            }
        }
        /* Lazy-rescale restatement (lazy_tau > 0; NOT the reference's arithmetic, which is the
         * branch below): the MI355X 64-rows-per-wave kernel keeps O and l relative to a
         * reference max m that moves only when, in a 32-row group (one wave's Q tile), some
         * row's max rose by more than lazy_tau in the base-2 exponent; then every row of the
         * group moves to its own running max.  Same real-valued result as softmax.cuh:36-49;
         * P is bounded by 2^lazy_tau instead of 1. */
        if (lazy_tau > 0.0f) {
            for (int g0 = 0; g0 < rows; g0 += 32) {
                const int g1 = g0 + 32 < rows ? g0 + 32 : (int)rows;
                int need = 0;
                for (int r = g0; r < g1; ++r) {
                    float *s = S + r * B_c;
                    float mx = s[0];
                    for (int cidx = 1; cidx < B_c; ++cidx) mx = fmaxf(mx, s[cidx]);
                    if (is_first) m[r] = mx;
                    else {
                        const float m_new = fmaxf(m[r], mx);
                        if ((m_new - m[r]) * c > lazy_tau) need = 1;
                    }
                }
                if (need) {
                    for (int r = g0; r < g1; ++r) {
                        float *s = S + r * B_c;
                        float mx = s[0];
                        for (int cidx = 1; cidx < B_c; ++cidx) mx = fmaxf(mx, s[cidx]);
                        const float m_new = fmaxf(m[r], mx);
                        const float scale = exp2f((m[r] - m_new) * c);
                        m[r] = m_new;
                        l[r] *= scale;
                        float *orow = O + r * d;
                        for (int64_t x = 0; x < d; ++x) orow[x] *= scale;
                    }
                }
                for (int r = g0; r < g1; ++r) {
                    float *s = S + r * B_c;
                    const float max_scaled = m[r] * c;
                    float rowsum = 0.0f;
                    for (int cidx = 0; cidx < B_c; ++cidx) {
                        const float p = exp2f(fmaf(s[cidx], c, -max_scaled));
                        rowsum += p;
                        s[cidx] = round_p ? round_b16(p, t->dtype) : p;
                    }
                    l[r] += rowsum;
                }
            }
        } else
        /* local_softmax, softmax.cuh:85-105 */
        for (int r = 0; r < rows; ++r) {
            float *s = S + r * B_c;
            const float m_prev = m[r];
            float mx = is_first ? s[0] : fmaxf(m_prev, s[0]);
            for (int cidx = 1; cidx < B_c; ++cidx) mx = fmaxf(mx, s[cidx]);
            m[r] = mx;
            if (mx == -INFINITY) mx = 0.0f; /* all keys masked so far (widened modes only) */
            if (!(is_first && optimized_softmax)) {
                /* scale_l_O: exp2f((m_prev - m_cur) * softmax_scale), softmax.cuh:42 */
                const float scale = exp2f((m_prev - mx) * c);
                l[r] *= scale;
                float *orow = O + r * d;
                for (int64_t x = 0; x < d; ++x) orow[x] *= scale;
            }
            const float max_scaled = mx * c; /* softmax.cuh:58 */
            float rowsum = 0.0f;
            for (int cidx = 0; cidx < B_c; ++cidx) {
                const float p = exp2f(fmaf(s[cidx], c, -max_scaled)); /* softmax.cuh:61 */
                rowsum += p;                                          /* fp32 P, :71-82 */
                s[cidx] = round_p ? round_b16(p, t->dtype) : p;       /* load_store.cuh:345 */
            }
            l[r] = (is_first && optimized_softmax) ? rowsum : l[r] + rowsum;
        }
        /* O += P V, fp32 accumulate */
        for (int r = 0; r < rows; ++r) {
            float *orow = O + r * d;
            for (int cidx = 0; cidx < B_c; ++cidx) {
                const float p = S[r * B_c + cidx];
                if (blk * B_c + cidx >= t->seq) continue; /* p == 0 there */
                const float *vp = vf + (blk * B_c + cidx) * d;
                for (int64_t x = 0; x < d; ++x) orow[x] = fmaf(p, vp[x], orow[x]);
            }
        }
    }
    /* final_softmax_normalization + convert + store, forward_kernel.cuh:186-203 */
    for (int r = 0; r < rows; ++r) {
        const int64_t qi = qb * B_r + r;
        const float inv = 1.0f / l[r];
        uint16_t *op = t->o + b * t->bs + qi * t->ss + h * t->hs;
        for (int64_t x = 0; x < d; ++x) op[x] = f32_to_b16(O[r * d + x] * inv, t->dtype);
        if (m_trace) m_trace[((b * t->heads + h) * t->seq) + qi] = m[r];
        if (l_trace) l_trace[((b * t->heads + h) * t->seq) + qi] = l[r];
    }
}

/*
 * Blockwise forward over the whole tensor.  Returns 0, or a negative code:
 * -1 bad dtype, -2 seq not a multiple of B_r/B_c (flash_attention.cu:79-82),
 * -3 allocation failure.
 */
static int blockwise_impl(const uint16_t *q, const uint16_t *k, const uint16_t *v,
                          uint16_t *o, int dtype, int64_t batch, int64_t seq,
                          int64_t heads, int64_t d_head, int64_t batch_stride,
                          int64_t seq_stride, int64_t head_stride, int B_r, int B_c,
                          int round_p, int optimized_softmax, float *m_trace,
                          float *l_trace, int n_threads, int masked, int causal, float lazy_tau,
                          int prescale_q, int kv_forward) {
    if (dtype != FA_ORACLE_FP16 && dtype != FA_ORACLE_BF16) return -1;
    if (B_r <= 0 || B_c <= 0 || seq <= 0) return -2;
    if (!masked && (seq % B_r != 0 || seq % B_c != 0)) return -2;
    tensors_t t = {q, k, v, o, dtype, batch, seq, heads, d_head,
                   batch_stride, seq_stride, head_stride, prescale_q, kv_forward};
    const int64_t n_heads_total = batch * heads;
    const int64_t n_q = (seq + B_r - 1) / B_r;
    int failed = 0;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#else
    (void)n_threads;
#endif
#pragma omp parallel
    {
        float *kf = (float *)malloc(sizeof(float) * seq * d_head);
        float *vf = (float *)malloc(sizeof(float) * seq * d_head);
        float *S = (float *)malloc(sizeof(float) * B_r * B_c);
        float *O = (float *)malloc(sizeof(float) * B_r * d_head);
        float *ml = (float *)malloc(sizeof(float) * (2 * B_r + d_head));
        if (!kf || !vf || !S || !O || !ml) {
#pragma omp atomic write
            failed = 1;
        } else {
#pragma omp for schedule(dynamic, 1)
            for (int64_t bh = 0; bh < n_heads_total; ++bh) {
                const int64_t b = bh / heads, h = bh % heads;
                for (int64_t s = 0; s < seq; ++s) {
                    const uint16_t *kp = k + b * batch_stride + s * seq_stride + h * head_stride;
                    const uint16_t *vp = v + b * batch_stride + s * seq_stride + h * head_stride;
                    for (int64_t x = 0; x < d_head; ++x) {
                        kf[s * d_head + x] = b16_to_f32(kp[x], dtype);
                        vf[s * d_head + x] = b16_to_f32(vp[x], dtype);
                    }
                }
                for (int64_t qb = 0; qb < n_q; ++qb)
                    q_block_forward(&t, b, h, qb, B_r, B_c, round_p, optimized_softmax, kf, vf,
                                    S, O, ml, ml + B_r, ml + 2 * B_r, m_trace, l_trace, causal,
                                    lazy_tau);
            }
        }
        free(kf); free(vf); free(S); free(O); free(ml);
    }
    return failed ? -3 : 0;
}

int fa_oracle_forward_blockwise(const uint16_t *q, const uint16_t *k, const uint16_t *v,
                                uint16_t *o, int dtype, int64_t batch, int64_t seq,
                                int64_t heads, int64_t d_head, int64_t batch_stride,
                                int64_t seq_stride, int64_t head_stride, int B_r, int B_c,
                                int round_p, int optimized_softmax, float *m_trace,
                                float *l_trace, int n_threads) {
    return blockwise_impl(q, k, v, o, dtype, batch, seq, heads, d_head, batch_stride, seq_stride,
                          head_stride, B_r, B_c, round_p, optimized_softmax, m_trace, l_trace,
                          n_threads, 0, 0, 0.0f, 0, 0);
}

/* The lazy-rescale restatement (see q_block_forward): pins the 64-rows-per-wave device variant. */
int fa_oracle_forward_blockwise_lazy(const uint16_t *q, const uint16_t *k, const uint16_t *v,
                                     uint16_t *o, int dtype, int64_t batch, int64_t seq,
                                     int64_t heads, int64_t d_head, int64_t batch_stride,
                                     int64_t seq_stride, int64_t head_stride, int B_r, int B_c,
                                     float tau, int n_threads) {
    return blockwise_impl(q, k, v, o, dtype, batch, seq, heads, d_head, batch_stride, seq_stride,
                          head_stride, B_r, B_c, 1, 0, NULL, NULL, n_threads, 0, 0, tau, 0, 0);
}

/* ... with the pre-scaled Q (tensors_t.prescale_q): the restatement of fa_fwd_opts.prescaled_q on the same kernel. */
int fa_oracle_forward_blockwise_lazy_psq(const uint16_t *q, const uint16_t *k, const uint16_t *v,
                                         uint16_t *o, int dtype, int64_t batch, int64_t seq,
                                         int64_t heads, int64_t d_head, int64_t batch_stride,
                                         int64_t seq_stride, int64_t head_stride, int B_r, int B_c,
                                         float tau, int n_threads) {
    return blockwise_impl(q, k, v, o, dtype, batch, seq, heads, d_head, batch_stride, seq_stride,
                          head_stride, B_r, B_c, 1, 0, NULL, NULL, n_threads, 0, 0, tau, 1, 0);
}

/* The speculative first pass of the persistent device kernel (plain forms, round 6): the reference of a row is the row max
 * of the FIRST block visited and never moves (tau = infinity), the blocks are visited first-to-last (kv_forward = 1; 0 gives
 * the masked forms' and rounds 2-5's order; G >= 2 the alternating order of tensors_t.kv_forward). */
int fa_oracle_forward_blockwise_spec(const uint16_t *q, const uint16_t *k, const uint16_t *v,
                                     uint16_t *o, int dtype, int64_t batch, int64_t seq,
                                     int64_t heads, int64_t d_head, int64_t batch_stride,
                                     int64_t seq_stride, int64_t head_stride, int B_r, int B_c,
                                     int kv_forward, int prescale_q, int n_threads) {
    return blockwise_impl(q, k, v, o, dtype, batch, seq, heads, d_head, batch_stride, seq_stride,
                          head_stride, B_r, B_c, 1, 0, NULL, NULL, n_threads, 0, 0, 1e30f, prescale_q, kv_forward);
}

/* Widened modes (causal mask, any seq): same arithmetic, see q_block_forward. */
int fa_oracle_forward_blockwise_masked(const uint16_t *q, const uint16_t *k, const uint16_t *v,
                                       uint16_t *o, int dtype, int64_t batch, int64_t seq,
                                       int64_t heads, int64_t d_head, int64_t batch_stride,
                                       int64_t seq_stride, int64_t head_stride, int B_r, int B_c,
                                       int optimized_softmax, int causal, int n_threads) {
    return blockwise_impl(q, k, v, o, dtype, batch, seq, heads, d_head, batch_stride, seq_stride,
                          head_stride, B_r, B_c, 1, optimized_softmax, NULL, NULL, n_threads, 1,
                          causal, 0.0f, 0, 0);
}

/*
 * Eager attention, the arithmetic of py_flash_attention (utils.py:137-162):
 *   S = einsum(q,k) / sqrt(d);  P = softmax(S);  O = einsum(P, v)
 * upcast = 1: fp32 everywhere, result rounded once to the 16-bit type.
 * upcast = 0: every intermediate tensor (S, S/sqrt(d), P, O) is rounded to the
 *             16-bit type, as eager torch does when the inputs stay 16-bit
 *             (dot products accumulate in fp32 inside the matmul).
 * out_f32 (optional, B*S*H*d fp32, contiguous (B,S,H,d)) receives the unrounded
 * fp32 result when upcast = 1.
 */
int fa_oracle_forward_eager(const uint16_t *q, const uint16_t *k, const uint16_t *v,
                            uint16_t *o, float *out_f32, int dtype, int64_t batch, int64_t seq,
                            int64_t heads, int64_t d_head, int64_t batch_stride,
                            int64_t seq_stride, int64_t head_stride, int upcast,
                            int n_threads) {
    if (dtype != FA_ORACLE_FP16 && dtype != FA_ORACLE_BF16) return -1;
    const int64_t n_heads_total = batch * heads;
    int failed = 0;
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#else
    (void)n_threads;
#endif
#pragma omp parallel
    {
        float *kf = (float *)malloc(sizeof(float) * seq * d_head);
        float *vf = (float *)malloc(sizeof(float) * seq * d_head);
        float *s = (float *)malloc(sizeof(float) * seq);
        float *qrow = (float *)malloc(sizeof(float) * d_head);
        double *acc = (double *)malloc(sizeof(double) * d_head);
        if (!kf || !vf || !s || !qrow || !acc) {
#pragma omp atomic write
            failed = 1;
        } else {
            const float sqrt_d = upcast ? sqrtf((float)d_head)
                                        : round_b16(sqrtf((float)d_head), dtype);
#pragma omp for schedule(dynamic, 1)
            for (int64_t bh = 0; bh < n_heads_total; ++bh) {
                const int64_t b = bh / heads, h = bh % heads;
                const int64_t base = b * batch_stride + h * head_stride;
                for (int64_t r = 0; r < seq; ++r)
                    for (int64_t x = 0; x < d_head; ++x) {
                        kf[r * d_head + x] = b16_to_f32(k[base + r * seq_stride + x], dtype);
                        vf[r * d_head + x] = b16_to_f32(v[base + r * seq_stride + x], dtype);
                    }
                for (int64_t qi = 0; qi < seq; ++qi) {
                    for (int64_t x = 0; x < d_head; ++x)
                        qrow[x] = b16_to_f32(q[base + qi * seq_stride + x], dtype);
                    float mx = -INFINITY;
                    for (int64_t j = 0; j < seq; ++j) {
                        double a = 0.0;
                        for (int64_t x = 0; x < d_head; ++x)
                            a += (double)qrow[x] * (double)kf[j * d_head + x];
                        float val = (float)a;
                        if (!upcast) val = round_b16(val, dtype);
                        val = val / sqrt_d;
                        if (!upcast) val = round_b16(val, dtype);
                        s[j] = val;
                        mx = fmaxf(mx, val);
                    }
                    double denom = 0.0;
                    for (int64_t j = 0; j < seq; ++j) {
                        s[j] = expf(s[j] - mx);
                        denom += (double)s[j];
                    }
                    const float inv = (float)(1.0 / denom);
                    for (int64_t x = 0; x < d_head; ++x) acc[x] = 0.0;
                    for (int64_t j = 0; j < seq; ++j) {
                        float p = s[j] * inv;
                        if (!upcast) p = round_b16(p, dtype);
                        for (int64_t x = 0; x < d_head; ++x)
                            acc[x] += (double)p * (double)vf[j * d_head + x];
                    }
                    for (int64_t x = 0; x < d_head; ++x) {
                        o[base + qi * seq_stride + x] = f32_to_b16((float)acc[x], dtype);
                        if (out_f32 && upcast)
                            out_f32[((b * seq + qi) * heads + h) * d_head + x] = (float)acc[x];
                    }
                }
            }
        }
        free(kf); free(vf); free(s); free(qrow); free(acc);
    }
    return failed ? -3 : 0;
}

int fa_oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
