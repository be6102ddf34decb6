R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/chk; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/gpu_layer_bench.py 256 > $OUT/out.txt 2> $OUT/err.txt
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/*/*_kernel_stats.csv")
for r in list(csv.DictReader(open(f[0])))[:12]:
    print(f"{r['Name'][:100]:100s} calls={r['Calls']:>5s} avg={float(r['AverageNs'])/1e3:9.1f} us")
PY
