#!/bin/bash
# round 4, call 34: refreshed bench lines on the final tree: the default line (512 steps + the 1k-out continuation) and BASELINE config 5
# (Mixtral-8x7B e4m3 weights on one GPU: the general-kernel / grouped-GEMM tuner now runs for it; TM_GEMM_TUNE_VERBOSE shows its picks)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/call34
mkdir -p $O
cd $R
timeout 200 python bench.py > $O/bench_line_default.json 2> $O/default.err
python -c "
import json; d=json.loads(open('$O/bench_line_default.json').read().strip().splitlines()[-1]); print('default', d['value'], d['ms_per_step'], d['step_roofline']['frac'], d.get('value_1k_out'), d['ttft_p50_ms'], d['roofline']['frac'])"
TM_GEMM_TUNE_VERBOSE=1 timeout 330 python bench.py --model mixtral_8x7b --steps 128 --no-cpu-baseline --no-traffic --no-full-run > $O/bench_line_config5_mixtral_fp8_tp1.json 2> $O/mixtral.err
python -c "
import json; d=json.loads(open('$O/bench_line_config5_mixtral_fp8_tp1.json').read().strip().splitlines()[-1]); print('mixtral', d['value'], d['ms_per_step'], d['step_roofline']['frac'], d['config'].get('lm_head_tiling'))"
grep "general\|experts" $O/mixtral.err | cut -c1-200 | tail -40
