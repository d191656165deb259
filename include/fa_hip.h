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
 *   fa_init            <- PYBIND11_MODULE body (max dynamic smem opt-in), flash_attention.cu:142-149
 *   fa_num_kernels / fa_get_kernel <- iteration over the forward_kernels map,
 *                         src/include/flash_kernels.cuh:14-186
 *
 * Error behaviour: functions return 0 on success or a negative fa_status; the
 * message (the reference's TORCH_CHECK text where one exists) is available from
 * fa_last_error() on the calling thread.  Unlike the reference (which never calls
 * cudaGetLastError), launch failures are reported.
 *
 * Threading / streams: no internal threads, no global mutable state besides the
 * const kernel registry and the one-time per-device setup (fa_init).  Launches are
 * asynchronous on the caller's stream; fa_fwd_launch_timed blocks on its stop event.
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
    int32_t optimized_softmax;       /* first KV block skips the (l, O) rescale */
} fa_fwd_config;

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
} fa_kernel_info;

/* One-time setup for the CURRENT device (idempotent; also called lazily by every launch): arch check,
 * CU count, the > 48 KB dynamic-LDS opt-in of every kernel function.  State is kept per device ordinal,
 * so a host that drives several GPUs from one process initialises each at its first call there
 * (the reference: device guard src/flash_attention.cu:42 + module init :142-149).  fa_fwd_launch itself is
 * one plain kernel launch -- capturable into a hipGraph -- but this setup queries the device: call
 * fa_init() on the device BEFORE starting a stream capture there. */
int fa_init(void);

/* Introspection of that per-device state (tests): has `device` been initialised, with which status,
 * and how many CUs cap the persistent grid there.  Any out pointer may be NULL. */
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

/* Registry enumeration: distinct device variants built into this library. */
int fa_num_kernels(void);
int fa_get_kernel(int index, fa_kernel_info *out);

/* Message for the last non-zero status on this thread ("" if none). */
const char *fa_last_error(void);

/* Library version string, e.g. "fa_hip 0.1 gfx950". */
const char *fa_version(void);

#ifdef __cplusplus
}
#endif
#endif /* FA_HIP_H */
