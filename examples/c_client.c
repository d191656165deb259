/* c_client.c -- libfa_hip.so from plain C: no Python, no torch, no C++.
 *
 * Fills Q, K, V (bf16, (batch, seq, heads, 128) contiguous) with a small pseudo-random pattern, launches the forward through
 * fa_fwd_launch_ex with the device counters on, and checks the result against softmax(Q K^T / sqrt(d)) V computed here in
 * double precision on a sample of rows.  Exit code 0 = every sampled element within 2^-7 (1 + |ref|).
 *
 *   gcc -std=c99 -O2 -D__HIP_PLATFORM_AMD__ -I../include -I/opt/rocm/include c_client.c \
 *       -L../flash_attention_from_scratch_amd/lib -lfa_hip -L/opt/rocm/lib -lamdhip64 -lm -o c_client
 *   LD_LIBRARY_PATH=../flash_attention_from_scratch_amd/lib:/opt/rocm/lib ./c_client
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "fa_hip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

static uint16_t to_bf16(float x) {  /* round to nearest even */
    uint32_t u;
    memcpy(&u, &x, 4);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
static float from_bf16(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float x;
    memcpy(&x, &u, 4);
    return x;
}

int main(void) {
    const int64_t B = 2, S = 1024, H = 3, D = 128;
    const size_t n = (size_t)(B * S * H * D);
    uint16_t *hq = malloc(2 * n), *hk = malloc(2 * n), *hv = malloc(2 * n), *ho = malloc(2 * n);
    uint32_t seed = 12345u;
    for (size_t i = 0; i < 3 * n; ++i) {
        seed = seed * 1664525u + 1013904223u;
        float x = ((float)(seed >> 8) / 16777216.0f - 0.5f) * 3.0f;
        (i < n ? hq : i < 2 * n ? hk : hv)[i % n] = to_bf16(x);
    }
    void *dq, *dk, *dv, *dout;
    uint32_t *dstats;
    CHECK_HIP(hipMalloc(&dq, 2 * n)); CHECK_HIP(hipMalloc(&dk, 2 * n)); CHECK_HIP(hipMalloc(&dv, 2 * n)); CHECK_HIP(hipMalloc(&dout, 2 * n));
    CHECK_HIP(hipMalloc((void **)&dstats, sizeof(fa_fwd_stats)));
    CHECK_HIP(hipMemcpy(dq, hq, 2 * n, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(dk, hk, 2 * n, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemcpy(dv, hv, 2 * n, hipMemcpyHostToDevice));
    CHECK_HIP(hipMemset(dstats, 0, sizeof(fa_fwd_stats)));

    fa_fwd_args a;
    memset(&a, 0, sizeof(a));
    a.q = dq; a.k = dk; a.v = dv; a.o = dout;
    a.batch = B; a.seq_len = S; a.n_heads = H; a.d_head = D;
    a.batch_stride = S * H * D; a.seq_stride = H * D; a.head_stride = D;
    /* the 13-field key (flash_attention.cuh:34-52): the persistent (256, 64, 4)+buffer shape */
    a.cfg.dtype = FA_BF16; a.cfg.d_head = 128; a.cfg.B_r = 256; a.cfg.B_c = 64; a.cfg.n_warps = 4;
    a.cfg.async_copy = 1; a.cfg.eager_load_blocks = 1; a.cfg.swizzled = 1;
    a.cfg.mma_double_buffer_loads = 1; a.cfg.optimized_softmax = 0;

    fa_fwd_opts o;
    memset(&o, 0, sizeof(o));
    o.struct_size = (uint32_t)sizeof(o);
    o.speculative = FA_SPECULATIVE_ADAPTIVE;   /* speculative softmax, demoted per device while its launches report redone items */
    o.stats = (fa_fwd_stats *)dstats;
    float ms = 0.0f;
    o.ms = &ms;
    fa_kernel_info info;
    if (fa_fwd_query(&a.cfg, &o, &info) != FA_OK) { fprintf(stderr, "fa_fwd_query: %s\n", fa_last_error()); return 3; }
    if (fa_fwd_launch_ex(&a, &o, NULL) != FA_OK) { fprintf(stderr, "fa_fwd_launch_ex: %s\n", fa_last_error()); return 3; }
    CHECK_HIP(hipDeviceSynchronize());
    fa_fwd_stats st;
    CHECK_HIP(hipMemcpy(&st, dstats, sizeof(st), hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(ho, dout, 2 * n, hipMemcpyDeviceToHost));

    /* a sample of (batch, head, row): every 37th row */
    double worst = 0.0;
    int bad = 0, rows = 0;
    double *p = malloc(sizeof(double) * (size_t)S);
    for (int64_t b = 0; b < B; ++b)
        for (int64_t h = 0; h < H; ++h)
            for (int64_t i = (b * 7 + h * 3) % 37; i < S; i += 37) {
                const uint16_t *qi = hq + ((b * S + i) * H + h) * D;
                double m = -1e300, l = 0.0;
                for (int64_t j = 0; j < S; ++j) {
                    const uint16_t *kj = hk + ((b * S + j) * H + h) * D;
                    double s = 0.0;
                    for (int64_t d = 0; d < D; ++d) s += (double)from_bf16(qi[d]) * (double)from_bf16(kj[d]);
                    p[j] = s / sqrt((double)D);
                    if (p[j] > m) m = p[j];
                }
                for (int64_t j = 0; j < S; ++j) { p[j] = exp(p[j] - m); l += p[j]; }
                for (int64_t d = 0; d < D; ++d) {
                    double acc = 0.0;
                    for (int64_t j = 0; j < S; ++j) acc += p[j] * (double)from_bf16(hv[((b * S + j) * H + h) * D + d]);
                    const double ref = acc / l, got = (double)from_bf16(ho[((b * S + i) * H + h) * D + d]);
                    const double err = fabs(got - ref);
                    if (err > worst) worst = err;
                    if (!(err <= 0.0078125 * (1.0 + fabs(ref)))) ++bad;
                }
                ++rows;
            }
    fa_adaptive_info ad;
    int dev = 0;
    CHECK_HIP(hipGetDevice(&dev));
    if (fa_adaptive_state(dev, &ad) != FA_OK) { fprintf(stderr, "fa_adaptive_state: %s\n", fa_last_error()); return 3; }
    printf("%s (ABI %d) | softmax_mode %d, %d threads, %d B LDS | %.3f ms | items %u, computed twice %u | adaptive: %u launches, %u demoted, mode %u | "
           "%d rows checked, max |err| %.3e, %d outside 2^-7 (1 + |ref|)\n",
           fa_version(), fa_abi_version(), info.softmax_mode, info.threads, info.lds_bytes, ms, st.items, st.items_redone,
           ad.launches, ad.demoted, ad.mode, rows, worst, bad);
    return bad == 0 && st.items == (uint32_t)(B * H * (S / 256)) ? 0 : 1;
}
