#!/bin/bash
# kernel-trace only pass over a short bench run: per-kernel durations (prefill + graph-replayed decode)
TAG=${1:-t}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/trace_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $R/bench.py --steps 64 --warmup 8 --profile-steps 0 --no-cpu-baseline > $OUT/trace.log 2>&1
python $R/tools/rocpd_summary.py $OUT/trace/trace_results.db > $OUT/kernel_trace_stats.txt 2>&1
grep '"metric"' $OUT/trace.log > $OUT/bench_line_under_rocprof.json
rm -rf $OUT/trace
head -40 $OUT/kernel_trace_stats.txt | cut -c1-150
