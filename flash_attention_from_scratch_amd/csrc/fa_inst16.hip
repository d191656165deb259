// fa_inst16.hip -- variant table of the 16-rows-per-wave kernel (B_r = 64 with 4
// waves: the reference's (64, *, 4) configs), -DFA_INST_DT=<5|15>.  List generated into
// fa_variants.inc (tools/generate_kernel_instantiations.py).
#include "fa_registry.hpp"
#include "fa_fwd_kernel16.hpp"

#ifndef FA_INST_DT
#error "define FA_INST_DT (5 = fp16, 15 = bf16)"
#endif
#define FA_INST_ROWS16 1

namespace fa {
namespace {
const KernelEntry kEntries[] = {
#include "fa_variants.inc"
};
}  // namespace

#define FA_CAT2(a, b) a##b
#define FA_CAT(a, b) FA_CAT2(a, b)
extern "C" KernelTable FA_CAT(fa_inst16_table_dt, FA_INST_DT)() {
    return KernelTable{kEntries, (int)(sizeof(kEntries) / sizeof(kEntries[0]))};
}
}  // namespace fa
