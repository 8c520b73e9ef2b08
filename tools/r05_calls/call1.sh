#!/bin/bash
# round 5, call 1: the per-launch fixed-cost table + the baseline driver-command line and kernel trace of this tree
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/call1
mkdir -p $O
cd $R
TM_GEMM_TUNE_VERBOSE=1 timeout 500 python tools/fixed_cost_table.py --per-layer > $O/fixed_cost_by_launch.txt 2> $O/fixed_cost.err
tail -5 $O/fixed_cost.err | cut -c1-300
cat $O/fixed_cost_by_launch.txt | cut -c1-260
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_line_driver_command.json 2> $O/bench.err
cut -c1-700 $O/bench_line_driver_command.json
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 64 --warmup 8 --profile-steps 4 --no-cpu-baseline --no-traffic --no-full-run"
timeout 400 rocprofv3 --kernel-trace --stats -d $O/t_default -o trace -- $CMD > $O/trace_default.log 2>&1
python $R/tools/rocpd_summary.py --by-grid $O/t_default/trace_results.db > $O/kernel_trace_by_grid_default.txt 2>&1
rm -rf $O/t_default
head -24 $O/kernel_trace_by_grid_default.txt | cut -c1-130
