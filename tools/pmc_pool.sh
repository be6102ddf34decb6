#!/bin/bash
# FETCH_SIZE of the pool+statistics pass per tap (1080p B=8), to compare with the algorithmic bytes.
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/pmc_pool; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT -- python3 $R/bench.py --workload 1080p --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/err.txt
python3 - <<PY
import csv,glob
for f in glob.glob("$OUT/*/*_counter_collection.csv"):
    rows=[r for r in csv.DictReader(open(f)) if 'pool_stats' in r['Kernel_Name'] and r['Counter_Name']=='FETCH_SIZE']
    for r in rows[:4]:
        print(r['Grid_Size'], f"{float(r['Counter_Value'])*2*1024/1e6:.0f} MB (x2-corrected)", (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3,'us')
PY
