#!/bin/bash
# rocprofv3 passes of round 3 (GPU box, from the repo root through gpurun).  Kernel trace + stats and each PMC group
# are separate runs (never combined).  Summaries land in gpurun_out/prof_r3_summary/ -> copied to profiles/r03_*.
# usage: bash tools/profile_r3.sh [trace|pmc|all]
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_r3
WHAT=${1:-all}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run_trace() {  # tag, bench args...
  local tag=$1; shift
  rm -rf $OUT/trace_$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$tag -- python3 $R/bench.py "$@" --only --no-cpu-baseline > $OUT/bench_$tag.json 2> $OUT/trace_$tag.err
  echo "trace $tag done" >> $R/gpurun_out/prof_r3.progress
}
if [ "$WHAT" != "pmc" ]; then
  run_trace 1080p_auto --steps 6 --warmup 2              # the shipped default (calibrates the 1080p class; f16 on the gain-1.0 stand-in weights)
  run_trace 1080p_f16w --steps 6 --warmup 2 --precision f16w
  run_trace 1080p_f16 --steps 10 --warmup 3 --precision f16
  run_trace 1080p_f32m --steps 6 --warmup 2 --precision f32m
  run_trace 1080p_f32s --steps 5 --warmup 2 --precision f32s
  run_trace 256_f16 --workload 256 --steps 20 --warmup 3 --precision f16
  run_trace 256_f32m --workload 256 --steps 10 --warmup 3 --precision f32m
  run_trace adists1080p_f32s --workload adists1080p --steps 5 --warmup 2
fi
if [ "$WHAT" != "trace" ]; then
  for C in FETCH_SIZE WRITE_SIZE; do
    for W in 1080p:f16w 1080p:f16 1080p:f32m 1080p:f32s 256:f16w 256:f16 adists1080p:f32s; do
      wl=${W%%:*}; pr=${W##*:}
      rm -rf $OUT/pmc_${C}_${wl}_${pr}
      rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_${C}_${wl}_${pr} -- python3 $R/bench.py --workload $wl --precision $pr --steps 2 --warmup 1 --only --no-cpu-baseline > /dev/null 2> $OUT/pmc_${C}_${wl}_${pr}.err
      echo "pmc $C $wl $pr done" >> $R/gpurun_out/prof_r3.progress
    done
  done
  for W in 1080p:f16w 1080p:f16 1080p:f32m 1080p:f32s 256:f16w; do
    wl=${W%%:*}; pr=${W##*:}
    rm -rf $OUT/pmc_mfma_${wl}_${pr}
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma_${wl}_${pr} -- python3 $R/bench.py --workload $wl --precision $pr --steps 2 --warmup 1 --only --no-cpu-baseline > /dev/null 2> $OUT/pmc_mfma_${wl}_${pr}.err
    echo "pmc mfma $wl $pr done" >> $R/gpurun_out/prof_r3.progress
  done
fi
python3 $R/tools/summarize_profile_r3.py $OUT $R/gpurun_out/prof_r3_summary
