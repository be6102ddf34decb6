#!/bin/bash
# kernel trace of one small-batch DISTS step (B from $1, default 1) at 256x256
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/trace_b1; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/bench.py --batch ${1:-1} --steps 50 --warmup 5 --no-cpu-baseline > $OUT/out.txt 2> $OUT/err.txt
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/*/*_kernel_stats.csv")
tot=0
for r in list(csv.DictReader(open(f[0])))[:14]:
    if 'nqa' in r['Name']:
        tot+=float(r['TotalDurationNs'])
        print(f"{r['Name'][10:90]:80s} calls={r['Calls']:>5s} avg={float(r['AverageNs'])/1e3:8.1f} us  per-step={float(r['TotalDurationNs'])/55/1e3:7.1f} us")
print("sum per step us", tot/55/1e3)
print(open("$OUT/out.txt").read()[:300])
PY
