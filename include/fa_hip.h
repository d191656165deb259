/*
 * fa_hip.h -- C ABI of libfa_hip.so, the MI355X (gfx950) Flash-Attention-2 forward.
 *
 * This is the drop-in boundary for the one hot path of
 * sonnyli/flash_attention_from_scratch: everything the reference's pybind module
 * `flash_attention_kernels` (src/flash_attention.cu) does below its tensor checks.
 * Plain pointers and sizes only -- no torch / ATen types -- so any host language
 * can bind it (ctypes stub: flash_attention_from_scratch_amd/_capi.py; see
 * INTEGRATION.md for the binding a reference maintainer would add).
 *
 * Reference interface each entry point replaces (paths under /root/reference):
 *   fa_fwd_config      <- FlashForwardKernelConfig, src/include/flash_attention.cuh:34-52
 *                         (13 fields, same order; Python side kernel_configs.py:106-120)
 *   fa_fwd_args        <- flash::ForwardKernelArgs, src/include/flash_attention.cuh:8-27
 *                         + the launcher locals of src/flash_attention.cu:58-108
 *   fa_fwd_supported   <- forward_kernels.contains(cfg), src/flash_attention.cu:60-61
 *   fa_fwd_lds_bytes   <- FlashForwardKernelConfig::smem_bytes(), flash_attention.cuh:54-56
 *   fa_fwd_launch      <- kernel<<<grid, block, smem, stream>>>(args), flash_attention.cu:110-126
 *   fa_fwd_launch_timed<- the benchmark=True event bracket, flash_attention.cu:119-132
 *   fa_fwd_launch_ex   <- the same launch with this library's NATIVE extensions (no reference counterpart):
 *                         causal / ragged seq_len, the speculative softmax, the pre-scaled Q, event timing,
 *                         device-side statistics -- one options struct (fa_fwd_opts)
 *   fa_init            <- PYBIND11_MODULE body (max dynamic smem opt-in), flash_attention.cu:142-149
 *   fa_num_kernels / fa_get_kernel <- iteration over the forward_kernels map,
 *                         src/include/flash_kernels.cuh:14-186
 *
 * Error behaviour: functions return 0 on success or a negative fa_status; the
 * message (the reference's TORCH_CHECK text where one exists) is available from
 * fa_last_error() on the calling thread.  Unlike the reference (which never calls
 * cudaGetLastError), launch failures are reported.
 *
 * Threading / streams: no internal threads.  Launches are asynchronous on the caller's
 * stream (graph-capturable); fa_fwd_launch_timed blocks on its stop event.  State:
 *   - fa_fwd_launch, fa_fwd_launch_timed, fa_fwd_launch_masked and fa_fwd_launch_ex with
 *     speculative = 0 or 1 are STATELESS, as the reference's launcher is
 *     (src/flash_attention.cu:42,118,126-131): nothing but the const kernel registry and
 *     the one-time per-device setup (fa_init) outlives a call, the same inputs give the
 *     same bits from any thread, process or stream.
 *   - fa_fwd_launch_ex with speculative = 2 (FA_SPECULATIVE_ADAPTIVE; what
 *     flash_helpers.kernel_configs.best_config() asks for) is NOT: it keeps, per device and
 *     per device variant, a small record (fa_adaptive_info) behind one mutex, one word of
 *     pinned host memory the kernels report into and one event; which of two valid variants
 *     serves a launch depends on that record, so the last bits of its output may differ from
 *     run to run on data that makes the speculative softmax fail.  The record is per process
 *     (two processes on one device do not see each other's).  See fa_speculative_mode.
 */
#ifndef FA_HIP_H
#define FA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* torch ScalarType codes, as the reference's DType enum (kernel_configs.py:12-13). */
typedef enum fa_dtype { FA_FP16 = 5, FA_BF16 = 15 } fa_dtype;

typedef enum fa_status {
    FA_OK = 0,
    FA_ERR_NULL = -1,        /* null pointer argument */
    FA_ERR_DTYPE = -2,       /* "Only fp16 and bf16 are supported" */
    FA_ERR_NO_KERNEL = -3,   /* "Kernel configuration was not found ..." */
    FA_ERR_SHAPE = -4,       /* seq_len not a multiple of B_r / B_c, bad d_head, ... */
    FA_ERR_ALIGN = -5,       /* pointers / strides not 16-byte compatible */
    FA_ERR_LAUNCH = -6,      /* HIP launch or event error */
    FA_ERR_DEVICE = -7       /* not a gfx950 device / no device */
} fa_status;

/* The reference's 13-field kernel key.  Booleans are 0/1 ints. */
typedef struct fa_fwd_config {
    int32_t dtype;                   /* fa_dtype */
    int32_t d_head;                  /* 128 */
    int32_t B_r;                     /* Q rows per workgroup */
    int32_t B_c;                     /* keys per LDS tile */
    int32_t n_warps;                 /* wave64 wavefronts per workgroup */
    int32_t async_copy;              /* K/V by direct global->LDS DMA */
    int32_t eager_load_blocks;       /* double-buffered K/V prefetch */
    int32_t swizzled;                /* XOR-swizzled K image in LDS */
    int32_t Q_mma_load_K_tiles;      /* operand-fetch hints: validated, see DESIGN.md */
    int32_t K_mma_load_K_tiles;
    int32_t V_mma_load_K_tiles;
    int32_t mma_double_buffer_loads;
    int32_t optimized_softmax;       /* the reference's meaning and nothing else: the first KV block skips the
                                        (l, O) rescale (softmax.cuh:85-105, forward_kernel.cuh:158-161).  The
                                        result is identical with or without it (the skipped factor is exp2(-inf)
                                        = 0 on l = O = 0), so device variants whose schedule has no first-block
                                        rescale to skip accept the flag and ignore it (fa_kernel_info.softmax_mode
                                        says what a variant does).  The speculative softmax is NOT selected by
                                        this field: fa_fwd_opts.speculative */
} fa_fwd_config;

/* How a device variant keeps exp2 in range (fa_kernel_info.softmax_mode).  All four give the same real
 * result; the first two are the reference's arithmetic bit for bit, the other two move the rounding
 * point of P and are checked against the same tolerances. */
typedef enum fa_softmax_mode {
    FA_SOFTMAX_EAGER = 0,            /* running row max, (l, O) rescaled at every tile (softmax.cuh:85-105) */
    FA_SOFTMAX_FIRST_BLOCK_SKIP = 1, /* the same, first KV block skips the rescale (the reference's optimized_softmax) */
    FA_SOFTMAX_LAZY = 2,             /* running row max, moved only when some row's max rose by > 8 binades */
    FA_SOFTMAX_SPECULATIVE = 3       /* reference = row max of the item's first visited tile, no per-tile max;
                                        the row sums are checked at the end and an item that fails is computed
                                        again with the running max (counted: fa_fwd_stats.items_redone) */
} fa_softmax_mode;

/*
 * One forward call.  q, k, v, o are DEVICE pointers to (batch, seq_len, n_heads,
 * d_head) tensors of the 16-bit dtype in cfg.dtype whose last dimension is
 * contiguous; the three strides are in ELEMENTS and shared by all four tensors
 * (flash_attention.cuh:14-19).  o may alias none of the inputs.  Strides must be positive multiples of 8;
 * seq_stride additionally <= (2^32 - 1) / (2 max(B_r, B_c)) (32-bit per-lane offsets inside a tile); the
 * batch and head strides are used in 64-bit arithmetic.
 */
typedef struct fa_fwd_args {
    const void *q;
    const void *k;
    const void *v;
    void *o;
    int64_t batch;
    int64_t seq_len;
    int64_t n_heads;
    int64_t d_head;
    int64_t batch_stride;
    int64_t seq_stride;
    int64_t head_stride;
    fa_fwd_config cfg;
} fa_fwd_args;

/* Resource report of one registered kernel (code-object metadata via HIP). */
typedef struct fa_kernel_info {
    fa_fwd_config cfg;       /* canonical config of the device variant */
    int32_t threads;         /* workgroup size */
    int32_t lds_bytes;       /* dynamic LDS per workgroup */
    int32_t num_regs;        /* VGPR+AGPR per lane (hipFuncAttributes.numRegs) */
    int32_t scratch_bytes;   /* per-thread scratch; 0 = no spills */
    int32_t rows_per_wave;   /* Q rows owned by one wavefront */
    int32_t masked;          /* 1: the causal / ragged-length variant of cfg; 2: the same for the persistent
                                (256, 64, 4) kernel, which needs seq_len >= B_c when seq_len % B_r != 0 */
    int32_t softmax_mode;    /* fa_softmax_mode of this device variant */
    int32_t prescaled_q;     /* 1: the variant folds the softmax scale into a 16-bit copy of Q (fa_fwd_opts.prescaled_q) */
    /* ABI 5: the RING FORM of a 32-rows-per-wave configuration -- (B_r 128, B_c 64, 4 warps) + buffer, the reference's own
     * winning tile shape: launches with seq_len % 256 == 0 are served by the hand-placed persistent kernel with one
     * 32-row Q tile per wave (the machinery of the (256, 64, 4) kernel: its lazy rescale for the plain variant,
     * bit-identical to that kernel's non-speculative form, and its speculative schedule for the speculative variant,
     * whose failed items are redone by the lazy one), the other multiples of B_r by the variant described above. */
    int32_t ring_form;           /* 1: such a form exists for this variant.  Round 6: it runs EIGHT waves of one Q tile each (512
                                    threads; two of the configuration's 128-row Q blocks make one 256-row item, counted as two in
                                    fa_fwd_stats) sharing one set of K / V rings -- same bits as the four-wave form of ABI 5 */
    int32_t ring_softmax_mode;   /* its fa_softmax_mode (FA_SOFTMAX_LAZY or FA_SOFTMAX_SPECULATIVE) */
    int32_t ring_num_regs;       /* VGPR+AGPR per lane of the ring form */
    int32_t ring_scratch_bytes;  /* 0 = no spills */
    /* ABI 6: what a launch of the ring form occupies -- lds_bytes / threads above describe the OTHER form */
    int32_t ring_lds_bytes;      /* dynamic LDS per workgroup of the ring form (160 KiB: one workgroup per CU); 0 without one */
    int32_t persistent;          /* 1: the variant (`fn`; a ring form always) is launched as one workgroup per CU that walks the
                                    (batch*head, Q block) items itself -- the grid is min(items, CUs rounded down to 8) */
    int32_t alt_form;            /* 1: long sequences run a second device form of this variant, whose speculative first pass
                                    walks every second round of a head's Q blocks [K / V tile 0, then last-to-second] so that
                                    the tail a round left in the XCD's L2 is read again first (taken when batch * heads is a
                                    multiple of 8 and seq_len / B_r is a multiple of 2 * grid / 8: seq_len 16384, 32768, ... on
                                    256 CUs).  Same tiles and arithmetic per tile; which way an item walks depends on its Q
                                    block and the CU count only, never on the batch.  (FA_HIP_NO_ALT in the environment, read
                                    once per process, sends such launches through the plain form: a MEASUREMENT switch for
                                    A/B runs -- profiles/r06/c3_alt_ab.txt, tools/l2_stride_probe.py -- that nothing sets) */
    int32_t ring_threads;        /* workgroup size of the ring form (512: eight waves, round 6); 0 without one */
} fa_kernel_info;

/* Device-side statistics (optional, fa_fwd_opts.stats): a DEVICE pointer to two 32-bit counters the
 * kernel ADDS to (zero them yourself).  items = work items (batch*head, Q block) computed;
 * items_redone = items the speculative softmax had to compute a second time because a row sum
 * exceeded its limit -- the cost of such an item is 2x.  0 for every other softmax mode.  The persistent kernel keeps a
 * workgroup's failures in a 64-bit mask of its walk ordinals, and ordinals >= 63 share the last bit: on a problem with
 * more than 63 items per workgroup (> ~16 000 items on a 256-CU device) a failure at ordinal >= 63 makes the workgroup
 * compute, and count, ALL its items from ordinal 63 on again -- items_redone is what ran twice, which there can exceed
 * what had to. */
typedef struct fa_fwd_stats {
    uint32_t items;
    uint32_t items_redone;
} fa_fwd_stats;

/* fa_fwd_opts.speculative.  ADAPTIVE: the speculative variant, except that the library watches the failure reports of its
 * speculative launches on the device (a failed item costs its workgroup a second item time, and a launch ends with its
 * slowest workgroup: one failing item in a thousand costs a quarter of a 4-items-per-CU launch, everything failing 2x):
 *   NORMAL   every launch speculative; a report -> the next `hold` adaptive launches take the NON-speculative variant of cfg
 *   DEMOTED  ... after which ONE launch probes the speculative variant again (an event is recorded behind it)
 *   PROBING  launches stay demoted until the probe has completed (hipEventQuery, never a wait); a probe that reported too
 *            doubles `hold` (32 ... 4096), one that did not returns the device to NORMAL.
 * The report is one word of pinned host memory a failing workgroup stores into; the library reads it when an adaptive
 * launch is enqueued and never waits for the device.  Both variants compute the same real result within the same
 * tolerance (they round P at different points), so WHICH of the two served a launch -- which depends on when a report
 * arrived -- is visible only in the last bits: ask for 0 or 1 where bit-reproducible output matters.  A launch made
 * during a stream capture is always speculative (nothing adaptive is recorded into a graph).  State per device AND per
 * device variant (ABI 5: a failing fp16 layer no longer demotes a benign bf16 one on the same device, and a probe's verdict
 * belongs to the configuration that probed): fa_adaptive_state (the device's records taken together),
 * fa_adaptive_state_for (one configuration's), fa_adaptive_reset (all of the device's). */
typedef enum fa_speculative_mode {
    FA_SPECULATIVE_OFF = 0,
    FA_SPECULATIVE_ALWAYS = 1,
    FA_SPECULATIVE_ADAPTIVE = 2
} fa_speculative_mode;

typedef struct fa_adaptive_info {
    uint32_t available;      /* 1: the pinned report word exists on this device (else ADAPTIVE behaves like ALWAYS) */
    uint32_t launches;       /* adaptive launches enqueued so far (fa_adaptive_state: summed over the device's records) */
    uint32_t demoted;        /* ... of which took the non-speculative variant */
    uint32_t reports;        /* distinct failure reports acted on */
    uint32_t hold;           /* current length of a demotion, in adaptive launches */
    uint32_t mode;           /* 0 NORMAL, 1 DEMOTED, 2 PROBING */
    uint32_t remaining;      /* DEMOTED: launches of the hold still to come */
    uint32_t last_report;    /* sequence number of the most recent launch that reported (0: none yet) */
} fa_adaptive_info;

/* Options of fa_fwd_launch_ex: this library's extensions beyond the reference's launch.  Zero-initialise,
 * set struct_size = sizeof(fa_fwd_opts), then set what you need. */
typedef struct fa_fwd_opts {
    uint32_t struct_size;    /* sizeof(fa_fwd_opts) of the caller's header */
    int32_t causal;          /* key j contributes to query i iff j <= i (masked variant of cfg); seq_len still has to be a
                                multiple of B_r and B_c unless allow_ragged is set too */
    int32_t allow_ragged;    /* accept seq_len that is not a multiple of B_r / B_c (masked variant of cfg) */
    int32_t speculative;     /* fa_speculative_mode.  1: the speculative-softmax variant of cfg (FA_SOFTMAX_SPECULATIVE); 2: the same,
                                adaptively -- see fa_speculative_mode below.  A row may rise, above the max of
                                its LAST 64 keys (visited first), by ~44 nats (bf16) / ~10 nats (fp16) before its item is
                                computed a second time; the persistent kernel also re-centres rising rows every four visits,
                                so there only a JUMP of ~83 nats (bf16) / ~1.4-10 nats (fp16) inside 256 keys fails.  The result is
                                right either way; fa_fwd_stats counts the items that ran twice.  See INTEGRATION.md 5 */
    int32_t prescaled_q;     /* 1: logits from a 16-bit Q * (log2 e / sqrt d) instead of an fp32 multiply per logit: a logit moves by
                                ~|k| |q c| 2^-9 (bf16) / 2^-12 (fp16) -- FA-3 / Triton practice, not the reference's arithmetic;
                                inside the reference's tolerance rule on benchmark-like data, outside it when keys of very
                                large norm (~30 sigma) appear; see DESIGN.md 3.7 */
    float *ms;               /* HOST pointer or NULL: bracket the launch with events, block, return elapsed ms */
    fa_fwd_stats *stats;     /* DEVICE pointer or NULL */
} fa_fwd_opts;

/* One-time setup for the CURRENT device (idempotent; also called lazily by every launch): arch check,
 * CU count, the > 48 KB dynamic-LDS opt-in of every kernel function.  State is kept per device ordinal,
 * so a host that drives several GPUs from one process initialises each at its first call there
 * (the reference: device guard src/flash_attention.cu:42 + module init :142-149).  fa_fwd_launch itself is
 * one plain kernel launch -- capturable into a hipGraph -- but this setup queries the device: call
 * fa_init() on the device BEFORE starting a stream capture there. */
int fa_init(void);

/* Introspection of that per-device state (tests): has `device` been initialised, with which status,
 * and how many CUs cap the persistent grid there.  Any out pointer may be NULL.  Safe to call from any
 * thread at any time: `inited` is published (release) after status and num_cus are final. */
int fa_device_state(int device, int *inited, int *status, int *num_cus);

/* 1 if a device kernel exists for cfg, else 0 (never negative). */
int fa_fwd_supported(const fa_fwd_config *cfg);

/* Dynamic LDS bytes the kernel for cfg uses, or a negative fa_status. */
int fa_fwd_lds_bytes(const fa_fwd_config *cfg);

/* Enqueue the forward on `stream` (a hipStream_t; NULL = default stream). */
int fa_fwd_launch(const fa_fwd_args *args, void *stream);

/* Same, bracketed by hipEvents on `stream`; blocks; *ms = elapsed milliseconds. */
int fa_fwd_launch_timed(const fa_fwd_args *args, void *stream, float *ms);

/*
 * Scope wideners beyond the reference (README.md:7-15 lists them as unsupported):
 * an optional causal mask (key j contributes to query i iff j <= i; equal Q/K lengths) and
 * seq_len that is NOT a multiple of B_r / B_c.  Same argument contract as fa_fwd_launch
 * otherwise.  ms == NULL: asynchronous; ms != NULL: timed like fa_fwd_launch_timed.
 * fa_fwd_masked_supported: 1 if a masked device variant exists for cfg.
 */
int fa_fwd_masked_supported(const fa_fwd_config *cfg);
int fa_fwd_launch_masked(const fa_fwd_args *args, int causal, void *stream, float *ms);

/* The general launch: fa_fwd_launch (opts == NULL or all zero), fa_fwd_launch_timed (opts->ms) and
 * fa_fwd_launch_masked (opts->causal / allow_ragged) are special cases of it.  fa_fwd_ex_supported: 1 if a
 * device variant exists for cfg with these options. */
int fa_fwd_ex_supported(const fa_fwd_config *cfg, const fa_fwd_opts *opts);
/* Which device variant would serve cfg with these options (opts may be NULL)?  Fills *out like fa_get_kernel
 * -- softmax_mode says what the variant does with the softmax -- or returns FA_ERR_NO_KERNEL. */
int fa_fwd_query(const fa_fwd_config *cfg, const fa_fwd_opts *opts, fa_kernel_info *out);
int fa_fwd_launch_ex(const fa_fwd_args *args, const fa_fwd_opts *opts, void *stream);

/* The adaptive speculative mode's record on `device` (fa_speculative_mode), and a reset of its demotion state (tests,
 * or a caller that knows its data has changed character). */
int fa_adaptive_state(int device, fa_adaptive_info *out);
/* ... of the ONE device variant that serves cfg with these options when it runs speculatively (opts may be NULL; its
 * `speculative` field is ignored).  FA_ERR_NO_KERNEL if cfg has no speculative variant. */
int fa_adaptive_state_for(int device, const fa_fwd_config *cfg, const fa_fwd_opts *opts, fa_adaptive_info *out);
int fa_adaptive_reset(int device);
/* The policy alone, on a fresh state and without a device (tests): launch i sees report_word[i] in the pinned word and
 * probe_state[i] for its outstanding probe (-1 none, 0 pending, 1 complete, 2 error); run[i] = 0 speculative, 1 the
 * non-speculative sibling, 2 the probe (speculative, an event recorded behind it). */
int fa_adaptive_simulate(int n, const uint32_t *report_word, const int32_t *probe_state, int32_t *run, fa_adaptive_info *final_state);

/* Registry enumeration: distinct device variants built into this library.  fa_get_kernel / fa_fwd_query write
 * sizeof(fa_kernel_info) bytes of THIS library's header; fa_kernel_info has grown (0.2 -> 0.3: softmax_mode,
 * prescaled_q), so a client that may run against a newer library than the header it was compiled with calls the _sized
 * forms, which write at most out_size bytes (whole leading fields; the struct only ever grows at its end), or checks
 * fa_abi_version() first.  ABI history: INTEGRATION.md 6. */
int fa_num_kernels(void);
int fa_get_kernel(int index, fa_kernel_info *out);
int fa_get_kernel_sized(int index, fa_kernel_info *out, uint32_t out_size);
int fa_fwd_query_sized(const fa_fwd_config *cfg, const fa_fwd_opts *opts, fa_kernel_info *out, uint32_t out_size);

/* Increases whenever a struct of this header grows or an entry point changes meaning (6 = this header: fa_kernel_info grew
 * by ring_lds_bytes, persistent, alt_form and ring_threads; the speculative first pass of the persistent kernel walks K / V first-to-last; 5: the
 * adaptive record per device variant, fa_adaptive_state_for, the four ring_* fields). */
#define FA_ABI_VERSION 6
int fa_abi_version(void);

/* Message for the last non-zero status on this thread ("" if none). */
const char *fa_last_error(void);

/* Library version string, e.g. "fa_hip 0.4 gfx950". */
const char *fa_version(void);

#ifdef __cplusplus
}
#endif
#endif /* FA_HIP_H */
