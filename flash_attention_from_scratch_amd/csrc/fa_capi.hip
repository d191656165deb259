// fa_capi.hip -- C ABI of libfa_hip.so (declared in include/fa_hip.h).
//
// Host-side launcher for the gfx950 forward kernel: the work
// /root/reference/src/flash_attention.cu:58-134 does after its torch checks
// (config -> kernel lookup, shape checks, grid/block/LDS, optional event timing),
// with no torch types.  Tensor-level checks (device, contiguity, dtype equality,
// output allocation, current stream) stay in the Python shim
// flash_attention_from_scratch_amd/flash_attention_kernels.py.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <vector>

#include "../../include/fa_hip.h"
#include "fa_registry.hpp"

extern "C" {
fa::KernelTable fa_inst_table_dt15_qt1();
fa::KernelTable fa_inst_table_dt15_qt2();
fa::KernelTable fa_inst_table_dt5_qt1();
fa::KernelTable fa_inst_table_dt5_qt2();
fa::KernelTable fa_inst16_table_dt15();
fa::KernelTable fa_inst16_table_dt5();
}

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

std::vector<fa::KernelEntry> &registry() {
    static std::vector<fa::KernelEntry> all = [] {
        std::vector<fa::KernelEntry> v;
        const fa::KernelTable tables[] = {
            fa_inst_table_dt15_qt1(), fa_inst_table_dt15_qt2(), fa_inst_table_dt5_qt1(),
            fa_inst_table_dt5_qt2(),  fa_inst16_table_dt15(),   fa_inst16_table_dt5(),
        };
        for (const auto &t : tables)
            for (int i = 0; i < t.count; ++i) v.push_back(t.entries[i]);
        return v;
    }();
    return all;
}

bool valid_load_tiles(int v) { return v == 0 || v == 2 || v == 4 || v == 8; }

// Config -> device variant.  The operand-fetch hints (load_K_tiles, double
// buffer) are validated with the reference's rules
// (static_kernel_configuration.cuh:13-35) and then ignored: on CDNA4 the
// compiler schedules LDS->MFMA operand reads and all hint values share one variant.
// cfg.optimized_softmax keeps the reference's meaning (the first KV block skips the rescale): it selects the
// variant built that way where one exists (softmax_mode 1) and is ignored where the schedule has no first-block
// rescale to skip (the result is the same either way).  The speculative softmax and the pre-scaled Q are this
// library's extensions and are asked for explicitly (fa_fwd_opts).
struct Want {
    bool masked = false;       // the causal / ragged-length variant of the configuration
    bool ragged = false;       // ... and seq_len may be off the tile sizes (fa_fwd_opts.allow_ragged)
    bool speculative = false;
    bool prescaled_q = false;
};
const fa::KernelEntry *find_kernel(const fa_fwd_config *c, const char **why, const Want &w = Want()) {
    static const char *kNotFound = "Kernel configuration was not found in the libfa_hip.so registry";
    *why = kNotFound;
    if (c->d_head != 128 && c->d_head != 64) { *why = "Only d_head = 128 (and 64) is supported"; return nullptr; }
    if (c->n_warps <= 0 || c->B_r <= 0 || c->B_r % c->n_warps != 0) return nullptr;
    if (!valid_load_tiles(c->Q_mma_load_K_tiles) || !valid_load_tiles(c->K_mma_load_K_tiles) ||
        !valid_load_tiles(c->V_mma_load_K_tiles))
        return nullptr;
    if (c->Q_mma_load_K_tiles != 0 && c->Q_mma_load_K_tiles != c->K_mma_load_K_tiles)
        return nullptr;
    const int rows_per_wave = c->B_r / c->n_warps;
    // mma_double_buffer_loads selects the software-pipelined loop where one is built;
    // otherwise it is a hint like the load_K_tiles fields and the plain loop is used.
    // Preference among the variants of one shape: the pipelined flag as asked, then the softmax mode --
    // speculative if (and only if) asked for; else the first-block-skip build for optimized_softmax where it
    // exists; else the eager / lazy build.
    const fa::KernelEntry *best = nullptr;
    int best_score = -1;
    for (const auto &e : registry()) {
        if (!(e.d_head == c->d_head && e.dtype == c->dtype && e.rows_per_wave == rows_per_wave &&
              e.n_waves == c->n_warps && e.B_c == c->B_c && e.swizzled == (c->swizzled != 0) &&
              e.eager == (c->eager_load_blocks != 0) && e.async_copy == (c->async_copy != 0) &&
              (e.masked != 0) == w.masked && (e.prescaled_q != 0) == w.prescaled_q))
            continue;
        if ((e.softmax_mode == FA_SOFTMAX_SPECULATIVE) != w.speculative) continue;
        const bool pipe_match = e.pipelined == (c->mma_double_buffer_loads != 0);
        if (!pipe_match && e.pipelined) continue;  // the plain loop stands in for a pipelined one that is not built, never the reverse
        const bool fbs = e.softmax_mode == FA_SOFTMAX_FIRST_BLOCK_SKIP;
        if (fbs && !c->optimized_softmax) continue;
        const int score = (pipe_match ? 2 : 0) + (fbs ? 1 : 0);
        if (score > best_score) { best = &e; best_score = score; }
    }
    if (!best && (w.speculative || w.prescaled_q))
        *why = "Kernel configuration has no device variant with the requested options (speculative softmax / pre-scaled Q) "
               "in the libfa_hip.so registry";
    return best;
}

// One-time setup PER DEVICE (the reference's device guard + module init, src/flash_attention.cu:42,142-149):
// the arch check, the CU count that caps the persistent grid and the > 48 KB dynamic-LDS opt-in of every
// kernel function all belong to the device that is current at the call, so a process that drives several
// GPUs gets each of them initialised at its first launch there, and a failure on one device (not a gfx950)
// does not stick to the others.
// Adaptive speculative softmax (fa_fwd_opts.speculative == FA_SPECULATIVE_ADAPTIVE).  A speculative launch whose
// first pass fails some item computes that item twice, and a launch ends with its slowest workgroup: ONE failing item
// costs a whole item time, everything failing 2x.  The kernels report such a launch by storing its
// sequence number into one word of pinned host memory (KernelArgs::redo_flag; a launch without failures writes
// nothing).  The library reads that word -- a plain host read, it never waits for the device -- whenever an adaptive
// launch is enqueued:
//   NORMAL    every launch speculative.  A new report -> DEMOTED for `hold` launches (this one included).
//   DEMOTED   the non-speculative sibling of the configuration serves the launch; when the hold has run out the next
//             launch is a PROBE: speculative again, with an event recorded behind it.
//   PROBING   launches are demoted until the probe's event has completed (hipEventQuery: no wait); then either the probe
//             reported too -> DEMOTED with `hold` doubled (32 ... 4096), or it did not -> NORMAL, `hold` back to 32.
// So steady bad data costs one speculative launch per hold; the host running ahead of the device costs at most the
// launches enqueued before the first report lands.  Both variants give a valid result (they differ in the rounding point
// of P); which one served a given launch depends on when a report arrived, so bit-reproducible callers ask for
// speculative = 0 or 1 instead.  During a stream capture the mode behaves like ALWAYS (no event is recorded into a graph).
// The policy itself is plain arithmetic over (the report word as read now, what the probe's event says): kept apart from
// HIP so that the CPU tier can script it (fa_adaptive_simulate; tests/test_host_cpu.py).
struct AdaptivePolicy {
    enum { NORMAL = 0, DEMOTED = 1, PROBING = 2 };
    enum { PROBE_NA = -1, PROBE_PENDING = 0, PROBE_COMPLETE = 1, PROBE_ERROR = 2 };
    enum { RUN_SPECULATIVE = 0, RUN_DEMOTED = 1, RUN_PROBE = 2 };
    uint32_t seq = 0;        // adaptive launches enqueued on this device so far
    uint32_t seen = 0;       // the last report acted on
    uint32_t mode = NORMAL;
    uint32_t remaining = 0;  // DEMOTED: launches of the hold still to come
    uint32_t probe_seq = 0;
    uint32_t hold = 32;
    uint32_t demoted = 0;    // adaptive launches that took the non-speculative variant
    uint32_t reports = 0;    // distinct failure reports acted on

    // One adaptive launch.  `rep`: the report word, read AFTER the probe's event was queried (a probe that has completed
    // has its report in memory by then); `probe`: PROBE_* (only looked at while PROBING).
    int step(uint32_t rep, int probe) {
        ++seq;
        const bool fresh = rep != seen;   // a speculative launch that has completed since computed items twice
        if (fresh) { seen = rep; ++reports; }
        if (mode == PROBING) {
            if (fresh) {                            // the probe (or a straggler from before the hold) failed: a longer hold
                if (hold < 4096) hold *= 2;
                mode = DEMOTED;
                remaining = hold;
            } else if (probe == PROBE_COMPLETE) {   // the probe ran clean
                mode = NORMAL;
                hold = 32;
            } else if (probe == PROBE_ERROR) {
                mode = NORMAL;
            }
        } else if (fresh) {                         // NORMAL or DEMOTED: (another) failure reported
            mode = DEMOTED;
            remaining = hold;
        }
        if (mode == DEMOTED) {
            if (remaining > 0) { --remaining; ++demoted; return RUN_DEMOTED; }
            mode = PROBING;                         // the hold has run out: this launch is the probe
            probe_seq = seq;
            return RUN_PROBE;
        }
        if (mode == PROBING) { ++demoted; return RUN_DEMOTED; }
        return RUN_SPECULATIVE;
    }
    void reset(uint32_t rep_now) { seen = rep_now; mode = NORMAL; remaining = 0; hold = 32; demoted = 0; reports = 0; }
};
// Round 5: ONE RECORD PER DEVICE VARIANT (the registry index of the speculative entry that a launch asked for), each with
// its own report word and probe event: a failing workload demotes its own configuration only -- an fp16 attention-sink layer
// no longer drags a benign bf16 one down with it, and a clean probe of one configuration no longer resets another's hold
// (ADVICE r04).  What stays shared is the mutex (launches are enqueued one at a time per device anyway).
struct AdaptiveSlot {
    AdaptivePolicy p;
    hipEvent_t probe_done = nullptr;  // created at the slot's first probe
};
struct AdaptiveState {
    std::mutex mu;
    uint32_t *flag_host = nullptr;  // hipHostMalloc'ed (mapped, coherent) words, one per slot; null: no pinned memory -> always speculative
    uint32_t *flag_dev = nullptr;   // the same words as the device addresses them
    std::vector<AdaptiveSlot> slot; // registry().size() records, allocated with the report words at the device's init (ADVICE
                                    // r05: a fixed 512 x 64 static array put 1.3 MB of initialised data into the library)
    int n_slots() const { return (int)slot.size(); }
};
struct DeviceState {
    std::once_flag once;
    int status = FA_OK;
    char err[256] = "";
    int num_cus = 256;  // persistent variants launch one workgroup per CU
    AdaptiveState adaptive;
    std::atomic<int> inited{0};  // published LAST (release): a reader that sees 1 sees the final status and num_cus
};
constexpr int kMaxDevices = 64;
DeviceState g_dev[kMaxDevices];

void do_init_body(int dev, DeviceState *st) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        st->status = FA_ERR_DEVICE;
        snprintf(st->err, sizeof(st->err), "hipGetDeviceProperties failed for device %d", dev);
        return;
    }
    st->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        st->status = FA_ERR_DEVICE;
        snprintf(st->err, sizeof(st->err),
                 "Flash Attention (HIP) requires gfx950 / MI355X (device %d: %s)", dev, prop.gcnArchName);
        return;
    }
    // flash_attention.cu:142-149: opt in to large dynamic shared memory per kernel (a per-device
    // attribute of the function: applied with `dev` current)
    for (const auto &e : registry()) {
        // (each function by ITS OWN LDS size: the ring form's 160 KB does not depend on what the compiler-scheduled body of
        // the same entry uses -- ADVICE r05)
        const struct { fa::kernel_fn fn; int bytes; } fns[] = {{e.fn, e.lds_bytes}, {e.fn_ragged, e.lds_bytes}, {e.fn_ring, e.ring_lds_bytes},
                                                                 {e.fn_alt, e.lds_bytes}};
        for (const auto &f : fns) {
            if (!f.fn || f.bytes <= 48 * 1024) continue;
            const hipError_t rc = hipFuncSetAttribute((const void *)f.fn, hipFuncAttributeMaxDynamicSharedMemorySize, f.bytes);
            if (rc != hipSuccess) {
                st->status = FA_ERR_LAUNCH;
                snprintf(st->err, sizeof(st->err), "hipFuncSetAttribute(%d B LDS) on device %d: %s", f.bytes, dev,
                         hipGetErrorString(rc));
                return;
            }
        }
    }
}

void do_init(int dev, DeviceState *st) {
    do_init_body(dev, st);
    if (st->status == FA_OK) {  // the adaptive mode's report word (see AdaptiveState); without it the mode stays speculative
        void *h = nullptr, *d = nullptr;
        const size_t bytes = sizeof(uint32_t) * registry().size();
        if (hipHostMalloc(&h, bytes, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess && h &&
            hipHostGetDevicePointer(&d, h, 0) == hipSuccess && d) {
            memset(h, 0, bytes);
            st->adaptive.slot = std::vector<AdaptiveSlot>(registry().size());
            st->adaptive.flag_host = (uint32_t *)h;
            st->adaptive.flag_dev = (uint32_t *)d;
        } else {
            (void)hipGetLastError();
            if (h) (void)hipHostFree(h);
        }
    }
    st->inited.store(1, std::memory_order_release);
}

// State of the CURRENT device, initialised on first use; nullptr (and the thread's error set) if there is none.
DeviceState *current_device(int *status) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        (void)hipGetLastError();
        *status = fail(FA_ERR_DEVICE, "no HIP device available");
        return nullptr;
    }
    if (dev < 0 || dev >= kMaxDevices) {
        *status = fail(FA_ERR_DEVICE, "device ordinal %d out of range (max %d)", dev, kMaxDevices - 1);
        return nullptr;
    }
    DeviceState *st = &g_dev[dev];
    std::call_once(st->once, do_init, dev, st);
    *status = st->status == FA_OK ? FA_OK : fail(st->status, "%s", st->err);
    return st->status == FA_OK ? st : nullptr;
}

int validate(const fa_fwd_args *a, const fa::KernelEntry **out, const Want &want = Want()) {
    const bool masked = want.masked;
    if (!a || !a->q || !a->k || !a->v || !a->o) return fail(FA_ERR_NULL, "null pointer argument");
    if (a->cfg.dtype != FA_FP16 && a->cfg.dtype != FA_BF16)
        return fail(FA_ERR_DTYPE, "Only fp16 and bf16 are supported");
    const char *why = "";
    const fa::KernelEntry *e = find_kernel(&a->cfg, &why, want);
    if (!e)
        return fail(FA_ERR_NO_KERNEL, "%s%s", why,
                    masked ? " (no causal / ragged-length variant is built for this configuration)" : "");
    if (a->d_head != a->cfg.d_head)
        return fail(FA_ERR_SHAPE, "Tensor d_head (%lld) does not match kernel configuration d_head (%d)",
                    (long long)a->d_head, a->cfg.d_head);
    if (a->batch <= 0 || a->seq_len <= 0 || a->n_heads <= 0)
        return fail(FA_ERR_SHAPE, "batch, seq_len and n_heads must be positive");
    // masked == 2 (persistent kernel): a causal-only form for seq_len % B_r == 0 and a second form for every
    // other seq_len >= B_c, which fetches a tile that would reach beyond the sequence as its last B_c keys
    if (masked && want.ragged && e->masked == 2 && a->seq_len % a->cfg.B_r != 0 && (a->seq_len < a->cfg.B_c || !e->fn_ragged))
        return fail(FA_ERR_SHAPE, "the masked variant of this configuration needs seq_len >= B_c (%d) when "
                                  "seq_len is not a multiple of B_r", a->cfg.B_c);
    // (a causal launch that did not also ask for allow_ragged keeps the reference's divisibility rules)
    if (!(masked && want.ragged) && a->seq_len % a->cfg.B_r != 0)
        return fail(FA_ERR_SHAPE, "Only multiples of B_r are supported for seq_len Q currently");
    if (!(masked && want.ragged) && a->seq_len % a->cfg.B_c != 0)
        return fail(FA_ERR_SHAPE, "Only multiples of B_c are supported for seq_len K currently");
    if (a->seq_len > INT32_MAX - 1024 || a->batch * a->n_heads > INT32_MAX ||
        a->batch * a->n_heads * ((a->seq_len + a->cfg.B_r - 1) / a->cfg.B_r) > INT32_MAX)
        return fail(FA_ERR_SHAPE, "problem too large for a 1-D grid");
    // The kernels keep per-lane offsets inside a tile / Q block in 32 bits (row * seq_stride * 2 bytes for up
    // to max(B_r, B_c) rows); the batch and head strides only enter 64-bit scalar bases.  All four tensors
    // share the strides, so a zero or negative one would make rows of O alias.
    if (a->seq_stride <= 0 || a->batch_stride < 0 || a->head_stride < 0 ||
        (a->batch > 1 && a->batch_stride == 0) || (a->n_heads > 1 && a->head_stride == 0))
        return fail(FA_ERR_SHAPE, "strides must be positive (batch %lld, seq %lld, head %lld elements)",
                    (long long)a->batch_stride, (long long)a->seq_stride, (long long)a->head_stride);
    {
        const int64_t rows = a->cfg.B_r > a->cfg.B_c ? a->cfg.B_r : a->cfg.B_c;
        if (a->seq_stride > (int64_t)0xffffffffLL / (2 * rows))
            return fail(FA_ERR_SHAPE, "seq_stride %lld too large: %lld rows * seq_stride * 2 bytes must fit 32 bits",
                        (long long)a->seq_stride, (long long)rows);
    }
    // 16-byte vector accesses: element strides must be multiples of 8, pointers 16-B aligned.
    if ((a->batch_stride | a->seq_stride | a->head_stride) & 7)
        return fail(FA_ERR_ALIGN, "strides must be multiples of 8 elements (16 bytes)");
    if (((uintptr_t)a->q | (uintptr_t)a->k | (uintptr_t)a->v | (uintptr_t)a->o) & 15)
        return fail(FA_ERR_ALIGN, "q, k, v, o must be 16-byte aligned");
    *out = e;
    return FA_OK;
}

int launch(const fa_fwd_args *a, const fa::KernelEntry *e, const DeviceState *dev, hipStream_t stream, int causal = 0,
           fa_fwd_stats *stats = nullptr, uint32_t *redo_flag = nullptr, uint32_t redo_seq = 0) {
    fa::KernelArgs ka;
    ka.stats = (uint32_t *)stats;
    ka.redo_flag = redo_flag;
    ka.redo_seq = redo_seq;
    ka.q = a->q;
    ka.k = a->k;
    ka.v = a->v;
    ka.o = a->o;
    ka.batch_stride = a->batch_stride;
    ka.seq_stride = a->seq_stride;
    ka.head_stride = a->head_stride;
    ka.seq_len = (int32_t)a->seq_len;
    ka.n_heads = (int32_t)a->n_heads;
    ka.n_bh = (int32_t)(a->batch * a->n_heads);
    ka.n_q_blocks = (int32_t)((a->seq_len + a->cfg.B_r - 1) / a->cfg.B_r);   // exact unless masked
    ka.n_kv_blocks = (int32_t)((a->seq_len + a->cfg.B_c - 1) / a->cfg.B_c);
    ka.causal = causal;
    fa::kernel_fn fn = e->fn;
    if (e->masked == 2 && a->seq_len % a->cfg.B_r != 0) {  // ragged form: whole ring rounds of K / V tiles
        fn = e->fn_ragged;
        ka.n_kv_blocks = ka.n_q_blocks * (a->cfg.B_r / a->cfg.B_c);
    }
    // 1-D grid over (batch*head, Q block); the kernel un-maps it XCD-aware.
    // (reference: dim3(n_Q_blocks, n_heads, batch), flash_attention.cu:110-112)
    // Persistent variants: one workgroup per CU (a multiple of 8, so an item keeps its XCD) that
    // walks items blockIdx.x, + gridDim.x, ... itself.
    // the ring form of a 32-rows-per-wave configuration (KernelEntry::fn_ring): seq_len a multiple of 256
    bool persistent = e->persistent != 0;
    int lds_bytes = e->lds_bytes;
    unsigned threads = (unsigned)e->threads;
    if (e->fn_ring && a->seq_len % 256 == 0 && !causal) {
        fn = e->fn_ring;
        lds_bytes = e->ring_lds_bytes;
        persistent = true;
        if (e->ring_threads) threads = (unsigned)e->ring_threads;
        if (e->ring_rows) ka.n_q_blocks = (int32_t)(a->seq_len / e->ring_rows);   // (the ring form's own items: 256 rows)
    }
    unsigned n_wg = (unsigned)(ka.n_bh * ka.n_q_blocks);
    if (persistent) {
        const unsigned cap = (unsigned)(dev->num_cus & ~7) ? (unsigned)(dev->num_cus & ~7) : 8u;
        if (n_wg > cap) n_wg = cap;
    }
    // a long head (its Q blocks fill an even number of rounds of an XCD's n_wg / 8 workgroups): the form that walks every second
    // round [tile 0, then last-to-second] (KernelEntry::fn_alt) -- a function of the sequence length and the CU count alone
    // (FA_HIP_NO_ALT in the environment, read once per process: a MEASUREMENT switch -- the same launch through the plain form,
    // for the A/B of profiles/r06/c3_alt_ab.txt; nothing in the product sets it)
    static const bool no_alt = getenv("FA_HIP_NO_ALT") != nullptr;
    if (!no_alt && persistent && fn == e->fn && e->fn_alt && (ka.n_bh & 7) == 0 && n_wg >= 16 && ka.n_q_blocks % (2 * (int)(n_wg >> 3)) == 0)
        fn = e->fn_alt;
    const dim3 grid(n_wg);
    const dim3 block(threads);
    void *params[] = {&ka};
    hipError_t rc = hipLaunchKernel((const void *)fn, grid, block, params, (size_t)lds_bytes, stream);
    if (rc != hipSuccess) return fail(FA_ERR_LAUNCH, "hipLaunchKernel: %s", hipGetErrorString(rc));
    return FA_OK;
}

}  // namespace

extern "C" {

int fa_init(void) {
    int rc = FA_OK;
    (void)current_device(&rc);
    return rc;
}

int fa_device_state(int device, int *inited, int *status, int *num_cus) {
    if (device < 0 || device >= kMaxDevices) return fail(FA_ERR_DEVICE, "device ordinal %d out of range", device);
    const DeviceState &st = g_dev[device];
    const int done = st.inited.load(std::memory_order_acquire);  // 0: nothing below is final yet
    if (inited) *inited = done;
    if (status) *status = done ? st.status : FA_OK;
    if (num_cus) *num_cus = done ? st.num_cus : 0;
    return FA_OK;
}

int fa_fwd_supported(const fa_fwd_config *cfg) {
    if (!cfg) return 0;
    const char *why;
    return find_kernel(cfg, &why) != nullptr;
}

int fa_fwd_lds_bytes(const fa_fwd_config *cfg) {
    if (!cfg) return fail(FA_ERR_NULL, "null config");
    const char *why = "";
    const fa::KernelEntry *e = find_kernel(cfg, &why);
    if (!e) return fail(FA_ERR_NO_KERNEL, "%s", why);
    return e->lds_bytes;
}

// Enqueue (ms == nullptr) or enqueue between two events on the stream and wait for the second one
// (flash_attention.cu:119-132).  Whatever was created is destroyed on every path.
// `probe` (an adaptive probe launch): its event is recorded right behind the kernel, its policy told if that failed, and
// the device's adaptive lock -- held by the caller until then, so that no other thread queries an event that has not
// been recorded -- is released BEFORE the timed path waits for the stop event (a timed probe used to keep
// every other thread's adaptive launch on the device out for the kernel's whole duration).
struct ProbeHook {
    AdaptiveSlot *slot = nullptr;
    std::unique_lock<std::mutex> *lock = nullptr;
    void after_launch(int launch_rc, hipStream_t s) {
        if (!slot) return;
        if (launch_rc != FA_OK || hipEventRecord(slot->probe_done, s) != hipSuccess) {
            (void)hipGetLastError();
            slot->p.mode = AdaptivePolicy::NORMAL;  // (no event to wait for; a failing probe reports like any launch)
        }
        if (lock && lock->owns_lock()) lock->unlock();
        slot = nullptr;
    }
};
static int launch_maybe_timed(const fa_fwd_args *args, const fa::KernelEntry *e, const DeviceState *dev,
                              hipStream_t s, int causal, float *ms, fa_fwd_stats *stats = nullptr,
                              uint32_t *redo_flag = nullptr, uint32_t redo_seq = 0, ProbeHook probe = ProbeHook()) {
    if (!ms) {
        const int rc0 = launch(args, e, dev, s, causal, stats, redo_flag, redo_seq);
        probe.after_launch(rc0, s);
        return rc0;
    }
    hipEvent_t start = nullptr, stop = nullptr;
    hipError_t hrc = hipEventCreate(&start);
    if (hrc == hipSuccess) hrc = hipEventCreate(&stop);
    if (hrc == hipSuccess) hrc = hipEventRecord(start, s);
    int rc = FA_OK;
    if (hrc == hipSuccess) {
        rc = launch(args, e, dev, s, causal, stats, redo_flag, redo_seq);
        probe.after_launch(rc, s);
        hrc = hipEventRecord(stop, s);  // (recorded even if the launch failed: nothing is left pending)
        if (hrc == hipSuccess) hrc = hipEventSynchronize(stop);
    }
    float elapsed = 0.0f;
    if (hrc == hipSuccess && rc == FA_OK) hrc = hipEventElapsedTime(&elapsed, start, stop);
    probe.after_launch(FA_ERR_LAUNCH, s);  // (only if the events could not even be created: nothing was launched)
    if (start) (void)hipEventDestroy(start);
    if (stop) (void)hipEventDestroy(stop);
    if (rc != FA_OK) return rc;
    if (hrc != hipSuccess) return fail(FA_ERR_LAUNCH, "event timing / kernel execution: %s", hipGetErrorString(hrc));
    *ms = elapsed;
    return FA_OK;
}

int fa_fwd_launch(const fa_fwd_args *args, void *stream) {
    const fa::KernelEntry *e = nullptr;
    int rc = validate(args, &e);
    if (rc != FA_OK) return rc;
    const DeviceState *dev = current_device(&rc);
    if (!dev) return rc;
    return launch(args, e, dev, (hipStream_t)stream);
}

int fa_fwd_launch_timed(const fa_fwd_args *args, void *stream, float *ms) {
    if (!ms) return fail(FA_ERR_NULL, "null ms pointer");
    const fa::KernelEntry *e = nullptr;
    int rc = validate(args, &e);
    if (rc != FA_OK) return rc;
    const DeviceState *dev = current_device(&rc);
    if (!dev) return rc;
    return launch_maybe_timed(args, e, dev, (hipStream_t)stream, 0, ms);
}

int fa_fwd_masked_supported(const fa_fwd_config *cfg) {
    if (!cfg) return 0;
    const char *why;
    Want w;
    w.masked = true;
    return find_kernel(cfg, &why, w) != nullptr;
}

int fa_fwd_launch_masked(const fa_fwd_args *args, int causal, void *stream, float *ms) {
    fa_fwd_opts o;
    memset(&o, 0, sizeof(o));
    o.struct_size = (uint32_t)sizeof(o);
    o.causal = causal;
    o.allow_ragged = 1;
    o.ms = ms;
    return fa_fwd_launch_ex(args, &o, stream);
}

// fa_fwd_opts as this library understands it: a caller built against an older (shorter) header passes a
// smaller struct_size and gets zeros for the fields it does not know.
static int read_opts(const fa_fwd_opts *in, fa_fwd_opts *o) {
    memset(o, 0, sizeof(*o));
    if (!in) return FA_OK;
    if (in->struct_size < sizeof(uint32_t) || in->struct_size > 4096)
        return fail(FA_ERR_SHAPE, "fa_fwd_opts.struct_size (%u) is not set: zero the struct and set it to sizeof(fa_fwd_opts)",
                    in->struct_size);
    memcpy(o, in, in->struct_size < sizeof(*o) ? in->struct_size : sizeof(*o));
    // (checked HERE so that fa_fwd_ex_supported, fa_fwd_query and fa_fwd_launch_ex agree: ADVICE r04)
    if (o->speculative < 0 || o->speculative > FA_SPECULATIVE_ADAPTIVE)
        return fail(FA_ERR_SHAPE, "fa_fwd_opts.speculative must be 0, 1 (always) or 2 (adaptive), not %d", o->speculative);
    return FA_OK;
}

// hipEventQuery is not a capture-safe call: with another thread's stream in a global-mode capture it would invalidate
// that capture.  Queried in relaxed mode (this thread's view only); if the mode cannot be switched the probe counts as
// still pending -- the policy stays demoted, which is always valid.
static int query_probe(hipEvent_t ev) {
    hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
    if (hipThreadExchangeStreamCaptureMode(&mode) != hipSuccess) { (void)hipGetLastError(); return AdaptivePolicy::PROBE_PENDING; }
    const hipError_t q = hipEventQuery(ev);
    if (q != hipSuccess) (void)hipGetLastError();
    if (hipThreadExchangeStreamCaptureMode(&mode) != hipSuccess) (void)hipGetLastError();
    return q == hipSuccess ? AdaptivePolicy::PROBE_COMPLETE
                           : (q == hipErrorNotReady ? AdaptivePolicy::PROBE_PENDING : AdaptivePolicy::PROBE_ERROR);
}

int fa_fwd_ex_supported(const fa_fwd_config *cfg, const fa_fwd_opts *opts) {
    if (!cfg) return 0;
    fa_fwd_opts o;
    if (read_opts(opts, &o) != FA_OK) return 0;
    const char *why;
    Want w;
    w.masked = o.causal || o.allow_ragged;
    w.ragged = o.allow_ragged != 0;
    w.speculative = o.speculative != 0;
    w.prescaled_q = o.prescaled_q != 0;
    return find_kernel(cfg, &why, w) != nullptr;
}

int fa_fwd_launch_ex(const fa_fwd_args *args, const fa_fwd_opts *opts, void *stream) {
    fa_fwd_opts o;
    int rc = read_opts(opts, &o);
    if (rc != FA_OK) return rc;
    Want w;
    w.masked = o.causal || o.allow_ragged;
    w.ragged = o.allow_ragged != 0;
    w.speculative = o.speculative != 0;
    w.prescaled_q = o.prescaled_q != 0;
    const fa::KernelEntry *e = nullptr;
    rc = validate(args, &e, w);
    if (rc != FA_OK) return rc;
    if (o.stats && ((uintptr_t)o.stats & 3)) return fail(FA_ERR_ALIGN, "fa_fwd_opts.stats must be 4-byte aligned");
    DeviceState *dev = current_device(&rc);
    if (!dev) return rc;
    uint32_t *redo_flag = nullptr;
    uint32_t redo_seq = 0;
    ProbeHook hook;
    std::unique_lock<std::mutex> probe_lock;
    if (o.speculative == FA_SPECULATIVE_ADAPTIVE && dev->adaptive.flag_host) {
        // (the speculative variant exists: validated above.  Its non-speculative sibling serves a demoted launch)
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing((hipStream_t)stream, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
        AdaptiveState &ad = dev->adaptive;
        const int idx = (int)(e - registry().data());   // the record of THIS device variant
        bool demote = false;
        if (cap == hipStreamCaptureStatusNone && idx >= 0 && idx < ad.n_slots()) {
            AdaptiveSlot &sl = ad.slot[idx];
            probe_lock = std::unique_lock<std::mutex>(ad.mu);
            int probe = AdaptivePolicy::PROBE_NA;
            if (sl.p.mode == AdaptivePolicy::PROBING) probe = query_probe(sl.probe_done);   // has the probe finished?  (never a wait)
            const uint32_t rep = __atomic_load_n(ad.flag_host + idx, __ATOMIC_ACQUIRE);   // (behind the query: see step())
            int run = sl.p.step(rep, probe);
            if (run == AdaptivePolicy::RUN_PROBE && !sl.probe_done &&
                hipEventCreateWithFlags(&sl.probe_done, hipEventDisableTiming) != hipSuccess) {
                (void)hipGetLastError();
                sl.probe_done = nullptr;
                sl.p.mode = AdaptivePolicy::NORMAL;   // no event to probe with: speculative, reports still demote
                run = AdaptivePolicy::RUN_SPECULATIVE;
            }
            demote = run == AdaptivePolicy::RUN_DEMOTED;
            redo_flag = ad.flag_dev + idx;
            redo_seq = sl.p.seq;
            // (a probe keeps the lock until its event is recorded -- ProbeHook: another thread's launch must not query an
            // event that has not been recorded yet, it would read as complete)
            if (run == AdaptivePolicy::RUN_PROBE) { hook.slot = &sl; hook.lock = &probe_lock; }
            else probe_lock.unlock();
        }
        if (demote) {
            Want w2 = w;
            w2.speculative = false;
            const fa::KernelEntry *e2 = nullptr;
            if (validate(args, &e2, w2) == FA_OK) {
                e = e2;
                redo_flag = nullptr;
            }  // (no such sibling: the speculative variant stays)
        }
    }
    return launch_maybe_timed(args, e, dev, (hipStream_t)stream, o.causal != 0, o.ms, o.stats, redo_flag, redo_seq, hook);
}

static void add_slot(const AdaptiveState &ad, int idx, fa_adaptive_info *out) {
    const AdaptivePolicy &p = ad.slot[idx].p;
    out->launches += p.seq;
    out->demoted += p.demoted;
    out->reports += p.reports;
    if (p.mode != AdaptivePolicy::NORMAL || out->hold < p.hold) {   // the record that is (or was longest) demoted
        if (p.hold >= out->hold) { out->hold = p.hold; out->remaining = p.remaining; }
        if (p.mode > out->mode) out->mode = p.mode;
    }
    if (ad.flag_host) { const uint32_t r = __atomic_load_n(ad.flag_host + idx, __ATOMIC_RELAXED); if (r > out->last_report) out->last_report = r; }
}

int fa_adaptive_state(int device, fa_adaptive_info *out) {
    if (!out) return fail(FA_ERR_NULL, "null output");
    if (device < 0 || device >= kMaxDevices) return fail(FA_ERR_DEVICE, "device ordinal %d out of range", device);
    memset(out, 0, sizeof(*out));
    out->hold = 32;
    DeviceState &st = g_dev[device];
    if (!st.inited.load(std::memory_order_acquire)) return FA_OK;
    AdaptiveState &ad = st.adaptive;
    std::lock_guard<std::mutex> lock(ad.mu);
    // the device's records taken together: counters summed; mode / hold / remaining of the record that is demoted (or was
    // demoted longest)
    for (int i = 0; i < ad.n_slots(); ++i) add_slot(ad, i, out);
    out->available = ad.flag_host != nullptr;
    return FA_OK;
}

int fa_adaptive_state_for(int device, const fa_fwd_config *cfg, const fa_fwd_opts *opts, fa_adaptive_info *out) {
    if (!out || !cfg) return fail(FA_ERR_NULL, "null pointer argument");
    if (device < 0 || device >= kMaxDevices) return fail(FA_ERR_DEVICE, "device ordinal %d out of range", device);
    fa_fwd_opts o;
    int rc = read_opts(opts, &o);
    if (rc != FA_OK) return rc;
    Want w;
    w.masked = o.causal || o.allow_ragged;
    w.ragged = o.allow_ragged != 0;
    w.speculative = true;
    w.prescaled_q = o.prescaled_q != 0;
    const char *why = "";
    const fa::KernelEntry *e = find_kernel(cfg, &why, w);
    if (!e) return fail(FA_ERR_NO_KERNEL, "%s", why);
    memset(out, 0, sizeof(*out));
    out->hold = 32;
    DeviceState &st = g_dev[device];
    if (!st.inited.load(std::memory_order_acquire)) return FA_OK;
    AdaptiveState &ad = st.adaptive;
    std::lock_guard<std::mutex> lock(ad.mu);
    const int idx = (int)(e - registry().data());
    if (idx >= 0 && idx < ad.n_slots()) {
        const AdaptivePolicy &p = ad.slot[idx].p;
        out->launches = p.seq; out->demoted = p.demoted; out->reports = p.reports;
        out->hold = p.hold; out->mode = p.mode; out->remaining = p.remaining;
        out->last_report = ad.flag_host ? __atomic_load_n(ad.flag_host + idx, __ATOMIC_RELAXED) : 0u;
    }
    out->available = ad.flag_host != nullptr;
    return FA_OK;
}

int fa_adaptive_simulate(int n, const uint32_t *report_word, const int32_t *probe_state, int32_t *run, fa_adaptive_info *final_state) {
    if (n < 0 || (n > 0 && (!report_word || !probe_state || !run))) return fail(FA_ERR_NULL, "null pointer argument");
    AdaptivePolicy p;
    for (int i = 0; i < n; ++i) run[i] = p.step(report_word[i], probe_state[i]);
    if (final_state) {
        memset(final_state, 0, sizeof(*final_state));
        final_state->available = 1;
        final_state->launches = p.seq;
        final_state->demoted = p.demoted;
        final_state->reports = p.reports;
        final_state->hold = p.hold;
        final_state->mode = p.mode;
        final_state->remaining = p.remaining;
        final_state->last_report = n > 0 ? report_word[n - 1] : 0u;
    }
    return FA_OK;
}

int fa_adaptive_reset(int device) {
    if (device < 0 || device >= kMaxDevices) return fail(FA_ERR_DEVICE, "device ordinal %d out of range", device);
    DeviceState &st = g_dev[device];
    if (!st.inited.load(std::memory_order_acquire)) return FA_OK;
    AdaptiveState &ad = st.adaptive;
    std::lock_guard<std::mutex> lock(ad.mu);
    for (int i = 0; i < ad.n_slots(); ++i)
        ad.slot[i].p.reset(ad.flag_host ? __atomic_load_n(ad.flag_host + i, __ATOMIC_RELAXED) : 0u);
    return FA_OK;
}

int fa_num_kernels(void) { return (int)registry().size(); }

static void fill_info(const fa::KernelEntry &e, fa_kernel_info *out);

int fa_fwd_query(const fa_fwd_config *cfg, const fa_fwd_opts *opts, fa_kernel_info *out) {
    if (!cfg || !out) return fail(FA_ERR_NULL, "null pointer argument");
    fa_fwd_opts o;
    int rc = read_opts(opts, &o);
    if (rc != FA_OK) return rc;
    Want w;
    w.masked = o.causal || o.allow_ragged;
    w.ragged = o.allow_ragged != 0;
    w.speculative = o.speculative != 0;
    w.prescaled_q = o.prescaled_q != 0;
    const char *why = "";
    const fa::KernelEntry *e = find_kernel(cfg, &why, w);
    if (!e) return fail(FA_ERR_NO_KERNEL, "%s", why);
    fill_info(*e, out);
    return FA_OK;
}

int fa_get_kernel(int index, fa_kernel_info *out) {
    if (!out) return fail(FA_ERR_NULL, "null output");
    if (index < 0 || index >= (int)registry().size()) return fail(FA_ERR_SHAPE, "index out of range");
    fill_info(registry()[index], out);
    return FA_OK;
}

static void fill_info(const fa::KernelEntry &e, fa_kernel_info *out) {
    memset(out, 0, sizeof(*out));
    out->cfg.dtype = e.dtype;
    out->cfg.d_head = e.d_head;
    out->cfg.B_r = e.rows_per_wave * e.n_waves;
    out->cfg.B_c = e.B_c;
    out->cfg.n_warps = e.n_waves;
    out->cfg.async_copy = e.async_copy;
    out->cfg.eager_load_blocks = e.eager;
    out->cfg.swizzled = e.swizzled;
    out->cfg.optimized_softmax = e.softmax_mode == FA_SOFTMAX_FIRST_BLOCK_SKIP;  // (the reference's meaning only)
    out->cfg.mma_double_buffer_loads = e.pipelined;
    out->softmax_mode = e.softmax_mode;
    out->prescaled_q = e.prescaled_q;
    out->threads = e.threads;
    out->lds_bytes = e.lds_bytes;
    out->rows_per_wave = e.rows_per_wave;
    out->masked = e.masked;
    out->num_regs = -1;
    out->scratch_bytes = -1;
    hipFuncAttributes attr;
    if (hipFuncGetAttributes(&attr, (const void *)e.fn) == hipSuccess) {
        out->num_regs = attr.numRegs;
        out->scratch_bytes = (int32_t)attr.localSizeBytes;
    } else {
        (void)hipGetLastError();  // no device: resource fields stay -1
    }
    out->ring_form = e.fn_ring ? 1 : 0;
    out->ring_lds_bytes = e.fn_ring ? e.ring_lds_bytes : 0;
    out->persistent = e.persistent;
    out->alt_form = e.fn_alt ? 1 : 0;
    out->ring_threads = e.fn_ring ? (e.ring_threads ? e.ring_threads : e.threads) : 0;
    out->ring_softmax_mode = e.fn_ring ? (e.softmax_mode == FA_SOFTMAX_SPECULATIVE ? FA_SOFTMAX_SPECULATIVE : FA_SOFTMAX_LAZY) : 0;
    out->ring_num_regs = out->ring_scratch_bytes = e.fn_ring ? -1 : 0;
    if (e.fn_ring) {
        if (hipFuncGetAttributes(&attr, (const void *)e.fn_ring) == hipSuccess) {
            out->ring_num_regs = attr.numRegs;
            out->ring_scratch_bytes = (int32_t)attr.localSizeBytes;
        } else {
            (void)hipGetLastError();
        }
    }
}

int fa_get_kernel_sized(int index, fa_kernel_info *out, uint32_t out_size) {
    if (!out) return fail(FA_ERR_NULL, "null output");
    fa_kernel_info full;
    const int rc = fa_get_kernel(index, &full);
    if (rc != FA_OK) return rc;
    memcpy(out, &full, out_size < sizeof(full) ? out_size : sizeof(full));
    return FA_OK;
}

int fa_fwd_query_sized(const fa_fwd_config *cfg, const fa_fwd_opts *opts, fa_kernel_info *out, uint32_t out_size) {
    if (!out) return fail(FA_ERR_NULL, "null pointer argument");
    fa_kernel_info full;
    const int rc = fa_fwd_query(cfg, opts, &full);
    if (rc != FA_OK) return rc;
    memcpy(out, &full, out_size < sizeof(full) ? out_size : sizeof(full));
    return FA_OK;
}

int fa_abi_version(void) { return FA_ABI_VERSION; }

const char *fa_last_error(void) { return g_err; }

const char *fa_version(void) { return "fa_hip 0.6 gfx950"; }

}  // extern "C"
