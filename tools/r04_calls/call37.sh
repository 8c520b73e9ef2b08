#!/bin/bash
# round 4, call 37: rocprofv3 --kernel-trace --stats of the DRIVER's bench command (bench.py --steps 20 --warmup 5), summary by kernel and by
# (kernel, grid): the trace's mean duration of the decode attention launch against the bench line's HIP-event figure of the same run
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/call37
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/t -o trace -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic > $O/bench_line_under_rocprof.json 2> $O/err.txt
python $R/tools/rocpd_summary.py $O/t/trace_results.db > $O/kernel_trace_stats_driver_command.txt 2>&1
python $R/tools/rocpd_summary.py --by-grid $O/t/trace_results.db decode_attention > $O/attention_by_grid.txt 2>&1
rm -rf $O/t
head -12 $O/kernel_trace_stats_driver_command.txt | cut -c1-150
cat $O/attention_by_grid.txt | cut -c1-150
python -c "
import json; d=json.loads(open('$O/bench_line_under_rocprof.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['us_per_launch'], d['roofline']['frac'], d['roofline']['ctx'])"
