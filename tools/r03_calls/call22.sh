#!/bin/bash
# round 3, call 22: refresh the evidence the prefill changes touch: default + driver-command bench lines, request stream, kernel trace
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/profile_r03b
mkdir -p $OUT
B="python $R/bench.py"
timeout 600 $B > $OUT/bench_line_default.json 2> $OUT/bench_line_default.err
timeout 400 $B --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_line_driver_command.json 2>/dev/null
timeout 600 python $R/tools/bench_continuous.py > $OUT/continuous_batching_line.json 2>/dev/null
CMD="$B --steps 64 --warmup 8 --profile-steps 4 --no-cpu-baseline --no-traffic --no-full-run"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/t_default -o trace -- $CMD > $OUT/trace_default.log 2>&1
python $R/tools/rocpd_summary.py $OUT/t_default/trace_results.db > $OUT/kernel_trace_stats_default.txt 2>&1
python $R/tools/rocpd_summary.py --by-grid $OUT/t_default/trace_results.db > $OUT/kernel_trace_by_grid_default.txt 2>&1
rm -rf $OUT/t_default $OUT/*.log
head -12 $OUT/kernel_trace_stats_default.txt | cut -c1-140
python - <<PY
import json,glob
for f in sorted(glob.glob('$OUT/*line*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['value'], d.get('ms_per_step'), d.get('step_roofline',{}).get('frac'), d.get('value_full_run',{}).get('value'), d.get('ttft_p50_ms'), d.get('prefill_tokens_per_s'))
    except Exception as e: print(f, 'no json', e)
PY
