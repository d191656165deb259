// fa_inst.hip -- one translation unit per (dtype, QT) slice of the variant table,
// selected with -DFA_INST_DT=<5|15> -DFA_INST_QT=<1|2> so the slices compile in
// parallel (see Makefile).  Counterpart of the reference's generated instantiation
// list (tools/build/generate_kernel_instantiations.py -> flash_kernels.cuh).
#include "fa_registry.hpp"

#ifndef FA_INST_DT
#error "define FA_INST_DT (5 = fp16, 15 = bf16)"
#endif
#ifndef FA_INST_QT
#error "define FA_INST_QT (1 or 2)"
#endif

namespace fa {
namespace {

#define E1(NW, BC, SWZ, EAGER, OPT) \
    make_entry<FA_INST_DT, FA_INST_QT, NW, BC, SWZ, EAGER, OPT, false>()
#define E(NW, BC, SWZ, EAGER, OPT) \
    make_entry<FA_INST_DT, FA_INST_QT, NW, BC, SWZ, EAGER, OPT, false>(), \
    make_entry<FA_INST_DT, FA_INST_QT, NW, BC, SWZ, EAGER, OPT, EAGER>()

const KernelEntry kEntries[] = {
#if FA_INST_QT == 1
    // 32 Q rows per wave: B_r = 128 (4 waves) or 256 (8 waves)
    E(4, 64, true, true, false),  E(4, 64, true, true, true),
    E(4, 32, true, true, false),  E(4, 32, true, true, true),
    E1(4, 128, true, true, false), E1(4, 128, true, true, true),  // pipelined loop spills at B_c=128
    E(8, 64, true, true, false),  E(8, 64, true, true, true),
    E(8, 32, true, true, false),  E(8, 32, true, true, true),
    E1(8, 128, true, true, false), E1(8, 128, true, true, true),
    // progression steps: no swizzle / no eager prefetch
    E1(4, 64, false, false, false), E1(4, 64, true, false, false),
#else
    // 64 Q rows per wave (one wave per SIMD, whole register file): B_r = 256 (4 waves)
    // (the pipelined loop and B_c = 128 do not fit 512 registers without spilling)
    E1(4, 64, true, true, false), E1(4, 64, true, true, true),
#endif
};
#undef E
#undef E1

}  // namespace

#define FA_CAT2(a, b, c, d) a##b##c##d
#define FA_CAT(a, b, c, d) FA_CAT2(a, b, c, d)
extern "C" KernelTable FA_CAT(fa_inst_table_dt, FA_INST_DT, _qt, FA_INST_QT)() {
    return KernelTable{kEntries, (int)(sizeof(kEntries) / sizeof(kEntries[0]))};
}

}  // namespace fa
