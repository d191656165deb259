#!/bin/bash
# wave cycles and durations of the default kernel on random vs zero inputs (same binary): is the difference
# in throughput cycles or clock?  One PMC pass + one plain kernel-trace pass.
export PYTHONPATH=$PWD:$PYTHONPATH
OUT=$PWD/gpurun_out/pmczero; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pz; (cd $R && rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d /tmp/pz/c -o p --output-format csv -- python tools/pmc_zero_vs_random.py) > /tmp/pz.log 2>&1
(cd $R && rocprofv3 --kernel-trace -d /tmp/pz/t -o p --output-format csv -- python tools/pmc_zero_vs_random.py) >> /tmp/pz.log 2>&1
python3 - <<'PY' | tee $OUT/pmc_zero_vs_random.txt
import csv, glob, collections
c = glob.glob('/tmp/pz/c/**/*counter_collection.csv', recursive=True)[0]
t = glob.glob('/tmp/pz/t/**/*kernel_trace.csv', recursive=True)[0]
agg = collections.OrderedDict()
for r in csv.DictReader(open(c)):
    if 'fa_fwd_kernel64' not in r['Kernel_Name']: continue
    agg.setdefault(int(r['Dispatch_Id']), {})[r['Counter_Name']] = agg.setdefault(int(r['Dispatch_Id']), {}).get(r['Counter_Name'], 0) + float(r['Counter_Value'])
ids = sorted(agg)
durs = [ (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in csv.DictReader(open(t)) if 'fa_fwd_kernel64' in r['Kernel_Name'] ]
n = len(ids) // 2
for name, sl, ds in (('random', ids[n - 10:n], durs[len(durs)//2 - 10:len(durs)//2]), ('zeros ', ids[-10:], durs[-10:])):
    m = {k: sum(agg[i][k] for i in sl) / len(sl) for k in agg[sl[0]]}
    d = sum(ds) / len(ds)
    print(f"{name}: SQ_WAVE_CYCLES {m['SQ_WAVE_CYCLES']/1e6:8.2f} M quad-cycles  SQ_BUSY_CYCLES {m['SQ_BUSY_CYCLES']/1e6:7.2f} M  GRBM_GUI_ACTIVE {m['GRBM_GUI_ACTIVE']/1e6:6.3f} M (sum of 8 XCDs) | un-profiled duration {d:7.1f} us -> {549755813888/d/1e6:7.1f} TFLOP/s, effective clock {m['GRBM_GUI_ACTIVE']/8/d/1e3:5.2f} GHz (GUI_ACTIVE per XCD / duration; the PMC pass itself runs a little slower)")
PY
tail -3 /tmp/pz.log | cut -c1-200
