// fa_redo_queue.hpp -- the speculative softmax's cheap second pass (round 6).
//
// An item whose speculative first pass failed used to be walked again by its OWN workgroup after that workgroup's walk:
// one failing item in a thousand cost a quarter of a four-items-per-workgroup launch, because a launch ends with its
// slowest workgroup.  Now the failing workgroup pushes the item into a device-side queue, cut along K / V into up to 16
// CHUNKS (whole groups of four 64-key tiles); every workgroup that reaches the end of its walk pops jobs until the queue
// is empty: a chunk job walks its tiles with the running-max (lazy) schedule and leaves partial (m c, l, O) -- fp32, O
// not normalised -- in the workspace; the workgroup that completes an item's last chunk pushes the item's 16 MERGE jobs
// (kBr / 16 rows each: the partials combined in chunk order, normalised, rounded, stored).  Which workgroup runs which
// job is timing; what a job computes is not: the bits of a redone item depend on its inputs only.
//
// Nobody ever waits for a workgroup that may not be resident: a pusher drains the queue itself after pushing, so every job
// is eventually run by a workgroup that is already running; helpers only shorten the tail.  (A grid barrier would hang two
// launches that share the device: tests/test_gpu_parity.py::test_two_streams_launching_at_once.)
//
// The workspace belongs to the device (allocated and zeroed once by fa_init); a launch owns it for as long as any of its
// workgroups is engaged with the queue: `state` = (ticket << 32) | engaged workgroups.  The last one out resets what the
// epoch dirtied and sets `state` back to 0.  A launch that finds another launch's ticket in `state` (two streams failing
// at once) -- or the item table full -- falls back to the in-workgroup second pass, which needs nothing shared.
//
// Inter-workgroup visibility follows /opt/skills/guides/cdna_hip_programming.md 6 G16: queue words are 4- / 8-byte
// agent-scope atomics (sc1), partials are written through (sc1 16-byte stores) and drained (s_waitcnt vmcnt(0) in every
// wave + workgroup barrier) before ONE lane bumps the item's counter; a merge job takes ONE agent-scope acquire behind its
// pop, then plain loads.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fa {

struct RedoQueue {
    static constexpr int kMaxItems = 32;    // failed items one epoch of the queue can hold (more: in-workgroup second pass)
    static constexpr int kMaxSplit = 16;    // chunks per item
    static constexpr int kMergeJobs = 16;   // merge jobs per item
    static constexpr int kMaxJobs = kMaxItems * (kMaxSplit + kMergeJobs);
    static constexpr int kRows = 256;       // rows of the largest item (B_r)
    static constexpr unsigned kResetting = 0xffffffffu;
    unsigned long long state;               // (ticket << 32) | engaged workgroups; low word kResetting: being reset; 0: free
    unsigned reserve;                       // job slots handed out
    unsigned head;                          // next job slot to pop
    unsigned n_items;                       // item slots handed out (may run past kMaxItems: only compared)
    unsigned pad_;
    unsigned done[kMaxItems];               // chunks of the item completed
    unsigned long long jobs[kMaxJobs];      // 0: not written (yet)
    float ml[kMaxItems][kMaxSplit][kRows][2];      // per row: reference max times c (base-2 exponent units), row sum
    float part[kMaxItems][kMaxSplit][kRows][128];  // per row: O relative to that reference, not normalised
};
// job word: bit 63 valid | bit 62 merge job | bits 48..55 chunk / merge index | bits 32..39 item slot | bits 0..31 item id
constexpr unsigned long long kJobValid = 1ull << 63, kJobMerge = 1ull << 62;
__device__ __forceinline__ unsigned long long rq_job(bool merge, unsigned sub, unsigned slot, unsigned item) {
    return kJobValid | (merge ? kJobMerge : 0ull) | ((unsigned long long)sub << 48) | ((unsigned long long)slot << 32) | item;
}
__device__ __forceinline__ unsigned rq_job_sub(unsigned long long j) { return (unsigned)(j >> 48) & 0xffu; }
__device__ __forceinline__ unsigned rq_job_slot(unsigned long long j) { return (unsigned)(j >> 32) & 0xffu; }
__device__ __forceinline__ unsigned rq_job_item(unsigned long long j) { return (unsigned)j; }

// chunks of an item of `n_kv` tiles (a multiple of four): whole groups of four tiles, as even as they come
__device__ __forceinline__ int rq_n_split(int n_kv) {
    const int groups = n_kv >> 2;
    return groups < RedoQueue::kMaxSplit ? groups : RedoQueue::kMaxSplit;
}
__device__ __forceinline__ void rq_chunk_range(int n_kv, int c, int &kv0, int &nk) {
    const int groups = n_kv >> 2, ns = rq_n_split(n_kv);
    const int g0 = c * groups / ns, g1 = (c + 1) * groups / ns;
    kv0 = 4 * g0;
    nk = 4 * (g1 - g0);
}

typedef __attribute__((address_space(1))) unsigned rq_gu32;
typedef __attribute__((address_space(1))) unsigned long long rq_gu64;
#define FA_RQ_RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
#define FA_RQ_CAS(p, expected, desired) \
    __hip_atomic_compare_exchange_strong(p, expected, desired, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)

// ---- ONE LANE of a workgroup calls these ----------------------------------------------------------------------------
// Engage with the queue on behalf of launch `ticket` (never 0).  claim: take a free queue (a pusher); otherwise only join
// what this launch already owns (a helper at the end of its walk).  false: not ours / nothing there / timed out.
__device__ __forceinline__ bool rq_join(RedoQueue *q, unsigned ticket, bool claim) {
    rq_gu64 *st = (rq_gu64 *)&q->state;
    for (int spin = 0; spin < (1 << 16); ++spin) {
        unsigned long long cur = __hip_atomic_load(st, FA_RQ_RLX);
        unsigned long long want;
        if (cur == 0) {
            if (!claim) return false;
            want = ((unsigned long long)ticket << 32) | 1ull;
        } else if ((unsigned)(cur >> 32) != ticket) {
            return false;                                   // another launch's epoch
        } else if ((unsigned)cur == RedoQueue::kResetting) {
            if (!claim) return false;                       // (its last workgroup is cleaning up: nothing left to help with)
            __builtin_amdgcn_s_sleep(16);
            continue;
        } else {
            want = cur + 1;
        }
        if (FA_RQ_CAS(st, &cur, want)) return true;
    }
    return false;
}
// Disengage.  true: this was the last engaged workgroup -- the caller's workgroup resets the queue (rq_reset_*).
__device__ __forceinline__ bool rq_leave(RedoQueue *q) {
    rq_gu64 *st = (rq_gu64 *)&q->state;
    for (;;) {
        unsigned long long cur = __hip_atomic_load(st, FA_RQ_RLX);
        const bool last = (unsigned)cur <= 1u;
        const unsigned long long want = last ? ((cur & 0xffffffff00000000ull) | RedoQueue::kResetting) : cur - 1;
        if (FA_RQ_CAS(st, &cur, want)) return last;
    }
}
// Queue one failed item as its chunk jobs.  false: the item table is full (the caller redoes the item itself).
__device__ __forceinline__ bool rq_push_item(RedoQueue *q, unsigned item, int n_split) {
    const unsigned slot = __hip_atomic_fetch_add((rq_gu32 *)&q->n_items, 1u, FA_RQ_RLX);
    if (slot >= (unsigned)RedoQueue::kMaxItems) return false;
    const unsigned j0 = __hip_atomic_fetch_add((rq_gu32 *)&q->reserve, (unsigned)n_split, FA_RQ_RLX);
    for (int c = 0; c < n_split; ++c)
        __hip_atomic_store((rq_gu64 *)&q->jobs[j0 + c], rq_job(false, (unsigned)c, slot, item), FA_RQ_RLX);
    return true;
}
__device__ __forceinline__ void rq_push_merge(RedoQueue *q, unsigned slot, unsigned item) {
    const unsigned j0 = __hip_atomic_fetch_add((rq_gu32 *)&q->reserve, (unsigned)RedoQueue::kMergeJobs, FA_RQ_RLX);
    for (int m = 0; m < RedoQueue::kMergeJobs; ++m)
        __hip_atomic_store((rq_gu64 *)&q->jobs[j0 + m], rq_job(true, (unsigned)m, slot, item), FA_RQ_RLX);
}
// Next job, or 0.  Jobs are taken in slot order and the pop stops at the first slot not written yet: whoever reserved it
// writes it within a few instructions and drains the queue itself afterwards.
__device__ __forceinline__ unsigned long long rq_pop(RedoQueue *q) {
    for (;;) {
        unsigned h = __hip_atomic_load((rq_gu32 *)&q->head, FA_RQ_RLX);
        if (h >= (unsigned)RedoQueue::kMaxJobs) return 0ull;
        const unsigned long long e = __hip_atomic_load((rq_gu64 *)&q->jobs[h], FA_RQ_RLX);
        if (!(e & kJobValid)) return 0ull;
        if (FA_RQ_CAS((rq_gu32 *)&q->head, &h, h + 1u)) return e;
    }
}
// A chunk's partials are in memory (written through, drained by every wave, workgroup barrier passed): count it.
// true: it was the item's last chunk.
__device__ __forceinline__ bool rq_chunk_done(RedoQueue *q, unsigned slot, int n_split) {
    return __hip_atomic_fetch_add((rq_gu32 *)&q->done[slot], 1u, FA_RQ_RLX) == (unsigned)(n_split - 1);
}
// ---- the last workgroup out: every thread (tid of n_threads) clears its share, then ONE lane frees the queue ------------
__device__ __forceinline__ void rq_reset_share(RedoQueue *q, int tid, int n_threads) {
    const unsigned n_jobs = __hip_atomic_load((rq_gu32 *)&q->reserve, FA_RQ_RLX);
    for (unsigned i = (unsigned)tid; i < n_jobs && i < (unsigned)RedoQueue::kMaxJobs; i += (unsigned)n_threads)
        __hip_atomic_store((rq_gu64 *)&q->jobs[i], 0ull, FA_RQ_RLX);
    for (int i = tid; i < RedoQueue::kMaxItems; i += n_threads) __hip_atomic_store((rq_gu32 *)&q->done[i], 0u, FA_RQ_RLX);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's clears are in memory (the caller's barrier follows)
}
__device__ __forceinline__ void rq_reset_finish(RedoQueue *q) {
    __hip_atomic_store((rq_gu32 *)&q->head, 0u, FA_RQ_RLX);
    __hip_atomic_store((rq_gu32 *)&q->n_items, 0u, FA_RQ_RLX);
    __hip_atomic_store((rq_gu32 *)&q->reserve, 0u, FA_RQ_RLX);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_store((rq_gu64 *)&q->state, 0ull, FA_RQ_RLX);
}

}  // namespace fa
