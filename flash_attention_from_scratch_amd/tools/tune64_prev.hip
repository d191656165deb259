// tune64_prev.hip -- the PREVIOUS round's persistent kernel, compiled into its own namespace so that tools/tune64.hip can
// time it in the same process, interleaved with this tree's (boxes differ by +-4 % and a lease's clock drifts: a
// round-over-round figure from two binaries run one after the other is mostly noise at seq_len <= 1024).
// Built by csrc/Makefile (target ../lib/tune64): the previous round's two kernel headers are taken from git
// (`git show $(PREV_REV):...` into build/prev/) -- nothing of them is kept in the tree.
#define fa fa_prev
#include "prev/fa_fwd_kernel64.hpp"
#undef fa
#include <hip/hip_runtime.h>

namespace prev {
template <bool SPEC> static void launch_t(const fa_prev::KernelArgs &a) {
    auto kern = fa_prev::fa_fwd_kernel64<15, false, 0, false, SPEC, false>;
    static bool init = false;
    if (!init) { (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 163840); init = true; }
    hipLaunchKernelGGL(kern, dim3(a.n_bh * a.n_q_blocks < 256 ? a.n_bh * a.n_q_blocks : 256), dim3(256), 163840, 0, a);
}
// plain C entry points: the argument block is passed by address and copied field by field (the two rounds' KernelArgs
// may differ in their tails)
void launch(bool spec, const void *q, const void *k, const void *v, void *o, long long bs, long long ss, long long hs,
            int seq_len, int n_heads, int n_bh, int n_q_blocks, int n_kv_blocks) {
    fa_prev::KernelArgs a{};
    a.q = q; a.k = k; a.v = v; a.o = o;
    a.batch_stride = bs; a.seq_stride = ss; a.head_stride = hs;
    a.seq_len = seq_len; a.n_heads = n_heads; a.n_bh = n_bh; a.n_q_blocks = n_q_blocks; a.n_kv_blocks = n_kv_blocks; a.causal = 0;
    if (spec) launch_t<true>(a); else launch_t<false>(a);
}
}  // namespace prev
