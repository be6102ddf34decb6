#!/bin/bash
# rocprofv3 passes of round 4 (GPU box, from the repo root through gpurun).  Kernel trace + stats and each PMC group
# are separate runs (never combined).  Summaries land in gpurun_out/prof_r4_summary/ -> copied to profiles/r04_*.
# usage: bash tools/profile_r4.sh [trace|pmc|mfma|all]
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_r4
WHAT=${1:-all}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export NQA_CAL_CACHE=$R/gpurun_out/prof_r4_calibration.json   # one calibration for all the passes of the `auto` lines
run_trace() {  # tag, bench args...
  local tag=$1; shift
  rm -rf $OUT/trace_$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$tag -- python3 $R/bench.py "$@" --only --no-cpu-baseline > $OUT/bench_$tag.json 2> $OUT/trace_$tag.err
  echo "trace $tag done" >> $R/gpurun_out/prof_r4.progress
}
if [ "$WHAT" = "trace" ] || [ "$WHAT" = "all" ]; then
  run_trace 1080p_auto --steps 10 --warmup 3            # the shipped default on the gain-1.0 stand-ins: f16, taps 1-2 fused
  run_trace 1080p_auto_g13 --steps 6 --warmup 2 --vgg synth:1234:1.3   # ... on the ImageNet-magnitude stand-ins: f32m / f32m4
  run_trace 1080p_f16 --steps 10 --warmup 3 --precision f16
  run_trace 1080p_f32m --steps 6 --warmup 2 --precision f32m
  run_trace 1080p_f32s --steps 5 --warmup 2 --precision f32s
  run_trace 256_f16 --workload 256 --steps 20 --warmup 3 --precision f16
  run_trace 256_auto --workload 256 --steps 10 --warmup 3          # the default below 0.9 Mpx: f32s
  run_trace adists1080p_f32s --workload adists1080p --steps 5 --warmup 2
fi
if [ "$WHAT" = "pmc" ] || [ "$WHAT" = "all" ]; then
  for C in FETCH_SIZE WRITE_SIZE; do
    for W in 1080p:f16 1080p:f32m 1080p:f32s 256:f16 256:f32s adists1080p:f32s; do
      wl=${W%%:*}; pr=${W##*:}
      rm -rf $OUT/pmc_${C}_${wl}_${pr}
      rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_${C}_${wl}_${pr} -- python3 $R/bench.py --workload $wl --precision $pr --steps 2 --warmup 1 --only --no-cpu-baseline > /dev/null 2> $OUT/pmc_${C}_${wl}_${pr}.err
      echo "pmc $C $wl $pr done" >> $R/gpurun_out/prof_r4.progress
    done
  done
fi
if [ "$WHAT" = "mfma" ] || [ "$WHAT" = "all" ]; then
  for W in 1080p:f16 1080p:f32m 1080p:f32s 256:f32s; do
    wl=${W%%:*}; pr=${W##*:}
    rm -rf $OUT/pmc_mfma_${wl}_${pr}
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma_${wl}_${pr} -- python3 $R/bench.py --workload $wl --precision $pr --steps 2 --warmup 1 --only --no-cpu-baseline > /dev/null 2> $OUT/pmc_mfma_${wl}_${pr}.err
    echo "pmc mfma $wl $pr done" >> $R/gpurun_out/prof_r4.progress
  done
fi
python3 $R/tools/summarize_profile_r3.py $OUT $R/gpurun_out/prof_r4_summary > $R/gpurun_out/prof_r4_summary.log 2>&1
