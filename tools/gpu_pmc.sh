#!/bin/bash
# PMC collection for one or more kernels: separate rocprofv3 passes, counters only with
# --kernel-trace (never with sys/hip/hsa traces).  Usage: bash tools/gpu_pmc.sh tag "cfg1" "cfg2" ...
TAG=$1; shift
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export PYTHONPATH=$PWD:$PYTHONPATH
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
[ -f $OUT/counters.txt ] || rocprofv3 -L > $OUT/counters.txt 2>&1
RUN="python $REPO/flash_attention_from_scratch_amd/tools/run_kernels.py 4096 128 --n_runs 3 --batch 4 --kernels"
pass() { # name counters...
  local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_$name -o p -- $RUN "${CFGS[@]}" > $OUT/pmc_$name.log 2>&1
}
CFGS=("$@")
pass a SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM
pass b SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM
pass c SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_TRANS
pass d GRBM_GUI_ACTIVE GRBM_COUNT TCC_HIT_sum TCC_MISS_sum
pass e FETCH_SIZE
pass f WRITE_SIZE
python - <<PY
import csv,glob,collections,os
out="$OUT"
agg=collections.defaultdict(dict)
for f in sorted(glob.glob(out+"/pmc_*/**/*counter_collection.csv",recursive=True)):
    for r in csv.DictReader(open(f)):
        k=r.get("Kernel_Name","")
        if "fa_fwd" not in k: continue
        agg[k].setdefault(r["Counter_Name"],[]).append(float(r["Counter_Value"]))
with open(out+"/pmc_summary.txt","w") as fo:
    for k,d in agg.items():
        fo.write(k+"\n")
        for c,v in sorted(d.items()):
            fo.write(f"  {c:34s} mean {sum(v)/len(v):16.1f}  n={len(v)}\n")
print(open(out+"/pmc_summary.txt").read())
import json
tr={}
for k,d in agg.items():
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        f=sum(d["FETCH_SIZE"])/len(d["FETCH_SIZE"]); w=sum(d["WRITE_SIZE"])/len(d["WRITE_SIZE"])
        # rocprofv3 reports KiB; on gfx950 FETCH_SIZE counts 128-B requests as 64 B -> x2
        # (MI355X_MICROARCH.md, HBM section); WRITE_SIZE used as reported.
        tr[k]={"fetch_size_kib":f,"write_size_kib":w,"hbm_bytes_per_launch":(2*f+w)*1024,
               "workload":"c1: bf16 batch=4 heads=16 seq_len=4096 d_head=128",
               "algorithmic_bytes":268435456}
json.dump(tr,open(out+"/pmc_traffic.json","w"),indent=1)
PY
for n in a b c d e f; do tail -2 $OUT/pmc_$n.log | cut -c1-200; done
