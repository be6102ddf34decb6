#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/adists_tr; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/gpu_adists_bench.py > $OUT/out.txt 2> $OUT/err.txt
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/*/*_kernel_stats.csv")
print(open("$OUT/out.txt").read())
for r in list(csv.DictReader(open(f[0])))[:16]:
    print(f"{r['Name'][:90]:90s} calls={r['Calls']:>5s} total={float(r['TotalDurationNs'])/1e6:9.2f} ms avg={float(r['AverageNs'])/1e3:9.1f} us {r['Percentage']}%")
PY
