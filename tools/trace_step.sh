#!/bin/bash
# kernel-trace of a short bench run; prints per-kernel per-grid average durations of the last step
R=${GRAFT_REPO_ROOT:-$PWD}; WL=${1:-256}
OUT=$R/gpurun_out/trace_tmp; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $R/bench.py --workload $WL --steps 6 --warmup 2 --no-cpu-baseline --only ${2:+--precision $2} > /dev/null 2> $OUT/err.txt
python3 - <<PY
import csv,glob,collections
f=glob.glob("$OUT/*/*_kernel_trace.csv")[0]
rows=[r for r in csv.DictReader(open(f)) if 'nqa::' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
agg=collections.OrderedDict()
for r in rows[len(rows)//2:]:
    name=r['Kernel_Name'].split('(')[0].replace('void nqa::','').replace('nqa::','')[:60]
    k=(name, r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z'], r['VGPR_Count'], r['LDS_Block_Size'])
    agg.setdefault(k,[]).append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in agg.items():
    print(f"{k[0]:62s} grid=({k[1]},{k[2]},{k[3]}) vgpr={k[4]} lds={k[5]} n={len(v)} avg={sum(v)/len(v):8.1f} us")
PY
