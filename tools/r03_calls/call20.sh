#!/bin/bash
# round 3, call 20: kernel trace of a short bench run -- the prefill kernels (LDS-staged attention, library GEMMs)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03
mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/t20 -o trace -- python $R/bench.py --steps 8 --warmup 2 --profile-steps 0 --no-cpu-baseline --no-traffic --no-full-run --tune 0 > $OUT/c20_trace.log 2>&1
python $R/tools/rocpd_summary.py $OUT/t20/trace_results.db > $OUT/c20_kernel_trace_stats.txt 2>&1
rm -rf $OUT/t20
head -24 $OUT/c20_kernel_trace_stats.txt | cut -c1-150
