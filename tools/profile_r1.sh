#!/bin/bash
# rocprofv3 passes for the round profile (run on the GPU box through gpurun from the repo root).
# Kernel trace/stats and each PMC group are separate runs (never combined with --sys-trace).
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_r1
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace256 -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench256.json 2> $OUT/trace256.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace1080 -- python3 $R/bench.py --workload 1080p --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench1080.json 2> $OUT/trace1080.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/traceadists -- python3 $R/bench.py --workload adists1080p --steps 5 --warmup 2 --no-cpu-baseline > $OUT/benchadists.json 2> $OUT/traceadists.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --workload 1080p --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --workload 1080p --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_mfma.err
python3 $R/tools/summarize_profile.py $OUT $R/gpurun_out/prof_r1_summary
