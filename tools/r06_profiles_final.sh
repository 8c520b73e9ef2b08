#!/bin/bash
# Round-6 evidence of the final tree: bench lines (default, driver command, the other configs, rank emulations), rocprofv3 kernel-trace stats of the
# driver command and by grid, PMC passes (FETCH_SIZE / SQ) of the decode kernels, the fixed-cost table.  Summaries -> gpurun_out/profile_r06/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/profile_r06
mkdir -p $OUT
B="python $R/bench.py"
timeout 700 $B > $OUT/bench_line_default.json 2> $OUT/bench_line_default.err
timeout 400 $B --steps 20 --warmup 5 > $OUT/bench_line_driver_command.json 2>/dev/null
timeout 400 $B --quant-policy 4 --steps 256 --no-cpu-baseline --no-full-run > $OUT/bench_line_llama3_8b_int4kv.json 2>/dev/null
timeout 400 $B --quant-policy 0 --steps 256 --no-cpu-baseline --no-full-run > $OUT/bench_line_config1_fp16kv.json 2>/dev/null
timeout 600 $B --model internlm2_20b --batch 128 --steps 128 --no-cpu-baseline --no-full-run > $OUT/bench_line_config2_internlm2_20b_b128.json 2>/dev/null
timeout 600 $B --model llama3_70b --quant-policy 4 --emulate-tp 8 --steps 128 --no-cpu-baseline --no-traffic --no-full-run > $OUT/bench_line_config3_llama3_70b_tp8_rank_emulation.json 2>/dev/null
timeout 600 $B --emulate-tp 8 --steps 128 --no-cpu-baseline --no-traffic --no-full-run > $OUT/bench_line_llama3_8b_tp8_rank_emulation.json 2>/dev/null
timeout 600 $B --emulate-tp 2 --steps 128 --no-cpu-baseline --no-traffic --no-full-run > $OUT/bench_line_llama3_8b_tp2_rank_emulation.json 2>/dev/null
timeout 900 $B --model mixtral_8x7b --steps 128 --no-cpu-baseline --no-traffic --no-full-run > $OUT/bench_line_config5_mixtral_fp8_tp1.json 2>/dev/null
timeout 300 $B --cpu-config0 > $OUT/bench_line_config0_cpu.json 2>/dev/null
timeout 600 python $R/tools/bench_continuous.py > $OUT/continuous_batching_line.json 2>/dev/null
CMD="$B --steps 20 --warmup 5 --no-cpu-baseline --no-traffic"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/t_driver -o trace -- $CMD > $OUT/trace_driver.log 2>&1
python $R/tools/rocpd_summary.py $OUT/t_driver/trace_results.db > $OUT/kernel_trace_stats_driver_command.txt 2>&1
rm -rf $OUT/t_driver
CMD="$B --steps 64 --warmup 8 --profile-steps 4 --no-cpu-baseline --no-traffic --no-full-run"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/t_default -o trace -- $CMD > $OUT/trace_default.log 2>&1
python $R/tools/rocpd_summary.py --by-grid $OUT/t_default/trace_results.db > $OUT/kernel_trace_by_grid_default.txt 2>&1
rm -rf $OUT/t_default
# PMC passes (their own runs, counters only): eager decode steps, 2 layers
CMDP="$B --steps 6 --warmup 1 --profile-steps 0 --no-cpu-baseline --no-traffic --no-full-run --no-graph --layers 2 --tune 0"
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $OUT/p_fetch -o pmc -- $CMDP > $OUT/pmc_fetch.log 2>&1
python $R/tools/rocpd_summary.py $OUT/p_fetch/pmc_results.db decode_attention > $OUT/pmc_fetch_attention.txt 2>&1
python $R/tools/rocpd_summary.py $OUT/p_fetch/pmc_results.db gemm_dec32 > $OUT/pmc_fetch_gemm_dec32.txt 2>&1
rm -rf $OUT/p_fetch
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $OUT/p_sq -o pmc -- $CMDP > $OUT/pmc_sq.log 2>&1
python $R/tools/rocpd_summary.py $OUT/p_sq/pmc_results.db gemm_dec32 > $OUT/pmc_sq_gemm_dec32.txt 2>&1
python $R/tools/rocpd_summary.py $OUT/p_sq/pmc_results.db decode_attention > $OUT/pmc_sq_attention.txt 2>&1
rm -rf $OUT/p_sq
cd $R && timeout 500 python tools/fixed_cost_table.py --attn-detail > $OUT/fixed_cost_by_launch_final_tree.txt 2> $OUT/fixed_cost.err
rm -f $OUT/trace_default.log $OUT/trace_driver.log $OUT/pmc_fetch.log $OUT/pmc_sq.log $OUT/fixed_cost.err
head -14 $OUT/kernel_trace_by_grid_default.txt | cut -c1-130
python - <<PY
import json,glob
for f in sorted(glob.glob('$OUT/bench_line_*.json')) + ['$OUT/continuous_batching_line.json']:
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['value'], d.get('ms_per_step'), d.get('step_roofline',{}).get('frac'), d.get('value_1k_out'), d.get('ttft_p50_ms'), d.get('roofline',{}).get('frac'), d.get('attention_roofline',{}).get('frac'))
    except Exception as e: print(f, 'no json', e)
PY
