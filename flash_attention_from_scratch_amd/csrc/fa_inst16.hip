// fa_inst16.hip -- variant table of the 16-rows-per-wave kernel (B_r = 64 with 4
// waves: the reference's (64, *, 4) configs), -DFA_INST_DT=<5|15>.
#include "fa_registry.hpp"
#include "fa_fwd_kernel16.hpp"

#ifndef FA_INST_DT
#error "define FA_INST_DT (5 = fp16, 15 = bf16)"
#endif

namespace fa {
namespace {
#define E(NW, BC, SWZ, EAGER, OPT) make_entry16<FA_INST_DT, NW, BC, SWZ, EAGER, OPT>()
const KernelEntry kEntries[] = {
    E(4, 64, true, true, false), E(4, 64, true, true, true),
    E(4, 32, true, true, false), E(4, 32, true, true, true),
    // the reference's progression steps at (64, 64, 4)
    E(4, 64, false, false, false), E(4, 64, true, false, false),
};
#undef E
}  // namespace

#define FA_CAT2(a, b) a##b
#define FA_CAT(a, b) FA_CAT2(a, b)
extern "C" KernelTable FA_CAT(fa_inst16_table_dt, FA_INST_DT)() {
    return KernelTable{kEntries, (int)(sizeof(kEntries) / sizeof(kEntries[0]))};
}
}  // namespace fa
