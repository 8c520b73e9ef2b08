#!/bin/bash
# round 4, call 33: refreshed evidence on the final tree: request-stream line (sessions TM_ASYNC_STEP = 1 then 0 on one engine: the
# default is the headline), rocprofv3 kernel trace of the decode steps by (kernel, grid)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/call33
mkdir -p $O
cd $R
timeout 300 python tools/bench_continuous.py --modes 1,0 > $O/continuous_batching_line.json 2> $O/cb.err
cat $O/continuous_batching_line.json | cut -c1-600
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 64 --warmup 8 --profile-steps 4 --no-cpu-baseline --no-traffic --no-full-run"
timeout 400 rocprofv3 --kernel-trace --stats -d $O/t_default -o trace -- $CMD > $O/trace_default.log 2>&1
python $R/tools/rocpd_summary.py $O/t_default/trace_results.db > $O/kernel_trace_stats_default.txt 2>&1
python $R/tools/rocpd_summary.py --by-grid $O/t_default/trace_results.db > $O/kernel_trace_by_grid_default.txt 2>&1
rm -rf $O/t_default
head -14 $O/kernel_trace_by_grid_default.txt | cut -c1-130
grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' $O/trace_default.log | head -4
