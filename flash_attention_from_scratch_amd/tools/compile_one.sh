#!/bin/bash
# compile_one.sh -- ONE device variant of the persistent kernel, alone, with the product's flags, its ISA kept and linted:
# a minute instead of the seven a whole slice takes (round 6: how the queued second pass was iterated -- the first form
# of it made hipcc spill 276 bytes in the default kernel, which a full build only told after seven minutes).
#   compile_one.sh "15, false, 0, false, true, false, 2" [out_dir]      (template arguments DT, MASK, ABL, RAG, SPEC, PSQ, QTP)
# Prints the register / scratch figures and the lint's verdict.
ARGS=${1:-"15, false, 0, false, true, false, 2"}
OUT=${2:-/tmp/fa_one}
HERE=$(cd "$(dirname "$0")" && pwd)
mkdir -p "$OUT" && cd "$OUT" || exit 1
cat > one.hip <<EOS
#include "$HERE/../csrc/fa_fwd_kernel64.hpp"
template __global__ void fa::fa_fwd_kernel64<$ARGS>(const fa::KernelArgs);
EOS
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fno-slp-vectorize \
    --cuda-device-only -save-temps=obj -c one.hip -o one.o 2>&1 | grep -E "error|Error" | head
S=one-hip-amdgcn-amd-amdhsa-gfx950.s
grep -E "; NumVgprs|; NumAgprs|ScratchSize|; Occupancy" $S | tr '\n' ' '; echo
python3 "$HERE/isa_lint64.py" $S --window 4 --raw 3 --only fa_fwd_kernel64 | tail -3
