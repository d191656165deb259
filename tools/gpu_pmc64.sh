#!/bin/bash
# PMC pass over one variant of tune64 (argument: abl value); counters in separate passes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ABL=${1:-1792}
$R/flash_attention_from_scratch_amd/lib/tune64 zeros 2>&1 | grep -v warm | head -8
for C in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM"; do
  rm -rf /tmp/pmc; rocprofv3 --pmc $C --kernel-trace -d /tmp/pmc -o p --output-format csv -- $R/flash_attention_from_scratch_amd/lib/tune64 only=$ABL reps=2 > /tmp/pmc.log 2>&1
  f=$(find /tmp/pmc -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv,sys,collections
if not sys.argv[1]: print("no csv"); sys.exit()
rows=list(csv.DictReader(open(sys.argv[1])))
# group by dispatch id; print the last dispatch at S=4096 (grid 4096 WGs*256) and S=16384
agg=collections.defaultdict(dict)
for r in rows:
    agg[(r['Dispatch_Id'],r['Grid_Size'])][r['Counter_Name']]=agg[(r['Dispatch_Id'],r['Grid_Size'])].get(r['Counter_Name'],0)+float(r['Counter_Value'])
seen={}
for (d,g),v in agg.items(): seen[g]=(d,v)
for g,(d,v) in seen.items(): print("grid",g,"dispatch",d,{k:int(x) for k,x in v.items()})
PY
done
