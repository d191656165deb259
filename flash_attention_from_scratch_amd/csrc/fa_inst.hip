// fa_inst.hip -- one translation unit per (dtype, QT) slice of the variant table,
// selected with -DFA_INST_DT=<5|15> -DFA_INST_QT=<1|2> so the slices compile in
// parallel (see Makefile).  The variant list itself is generated from the Python config
// enumerations (tools/generate_kernel_instantiations.py -> fa_variants.inc), the
// counterpart of the reference's generated src/include/flash_kernels.cuh.
#include "fa_registry.hpp"

#ifndef FA_INST_DT
#error "define FA_INST_DT (5 = fp16, 15 = bf16)"
#endif
#ifndef FA_INST_QT
#error "define FA_INST_QT (1 or 2)"
#endif

namespace fa {
namespace {
const KernelEntry kEntries[] = {
#include "fa_variants.inc"
};
}  // namespace

#define FA_CAT2(a, b, c, d) a##b##c##d
#define FA_CAT(a, b, c, d) FA_CAT2(a, b, c, d)
extern "C" KernelTable FA_CAT(fa_inst_table_dt, FA_INST_DT, _qt, FA_INST_QT)() {
    return KernelTable{kEntries, (int)(sizeof(kEntries) / sizeof(kEntries[0]))};
}

}  // namespace fa
