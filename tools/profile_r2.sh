#!/bin/bash
# rocprofv3 passes of round 2 (GPU box, from the repo root through gpurun).  Kernel trace + stats and each PMC group
# are separate runs (never combined).  Summaries land in gpurun_out/prof_r2_summary/ -> copied to profiles/r02_*.
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_r2
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run_trace() {  # tag, bench args...
  local tag=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$tag -- python3 $R/bench.py "$@" --only --no-cpu-baseline > $OUT/bench_$tag.json 2> $OUT/trace_$tag.err
  echo "trace $tag done" >> $R/gpurun_out/prof_r2.progress
}
run_trace 1080p_f16 --steps 10 --warmup 3
run_trace 1080p_f32s --steps 5 --warmup 2 --precision f32s
run_trace 256_f16 --workload 256 --steps 20 --warmup 3
run_trace 256_f32s --workload 256 --steps 10 --warmup 3 --precision f32s
run_trace adists1080p_f32s --workload adists1080p --steps 5 --warmup 2
for C in FETCH_SIZE WRITE_SIZE; do
  for W in 1080p 256 adists1080p; do
    rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_${C}_$W -- python3 $R/bench.py --workload $W --steps 2 --warmup 1 --only --no-cpu-baseline > /dev/null 2> $OUT/pmc_${C}_$W.err
    echo "pmc $C $W done" >> $R/gpurun_out/prof_r2.progress
  done
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma_1080p -- python3 $R/bench.py --steps 2 --warmup 1 --only --no-cpu-baseline > /dev/null 2> $OUT/pmc_mfma_1080p.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma_256 -- python3 $R/bench.py --workload 256 --steps 3 --warmup 1 --only --no-cpu-baseline > /dev/null 2> $OUT/pmc_mfma_256.err
echo "pmc mfma done" >> $R/gpurun_out/prof_r2.progress
python3 $R/tools/summarize_profile_r2.py $OUT $R/gpurun_out/prof_r2_summary
