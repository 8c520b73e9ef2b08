#!/bin/bash
# Round-3 evidence: bench lines + rocprofv3 kernel trace (by grid) + PMC passes (separate runs, no tracing domains mixed with --pmc).
# Summaries land in gpurun_out/profile_r03/ (copied to profiles/r03_* afterwards).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/profile_r03
mkdir -p $OUT
B="python $R/bench.py"
timeout 600 $B > $OUT/bench_line_default.json 2> $OUT/bench_line_default.err
timeout 400 $B --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_line_driver_command.json 2>/dev/null
timeout 400 $B --quant-policy 4 --steps 256 --no-cpu-baseline --no-full-run > $OUT/bench_line_llama3_8b_int4kv.json 2>/dev/null
timeout 400 $B --quant-policy 0 --steps 128 --no-cpu-baseline --no-full-run > $OUT/bench_line_config1_fp16kv.json 2>/dev/null
timeout 600 $B --model internlm2_20b --batch 128 --steps 128 --no-cpu-baseline --no-full-run > $OUT/bench_line_config2_internlm2_20b_b128.json 2>/dev/null
timeout 600 $B --model llama3_70b --quant-policy 4 --emulate-tp 8 --steps 128 --no-cpu-baseline --no-traffic --no-full-run > $OUT/bench_line_config3_llama3_70b_tp8_rank_emulation.json 2>/dev/null
timeout 900 $B --model mixtral_8x7b --steps 128 --no-cpu-baseline --no-traffic --no-full-run > $OUT/bench_line_config5_mixtral_fp8_tp1.json 2>/dev/null
timeout 600 python $R/tools/bench_continuous.py > $OUT/continuous_batching_line.json 2>/dev/null
CMD="$B --steps 64 --warmup 8 --profile-steps 4 --no-cpu-baseline --no-traffic --no-full-run"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/t_default -o trace -- $CMD > $OUT/trace_default.log 2>&1
python $R/tools/rocpd_summary.py $OUT/t_default/trace_results.db > $OUT/kernel_trace_stats_default.txt 2>&1
python $R/tools/rocpd_summary.py --by-grid $OUT/t_default/trace_results.db > $OUT/kernel_trace_by_grid_default.txt 2>&1
rm -rf $OUT/t_default
PMC="$B --steps 12 --warmup 2 --profile-steps 0 --no-cpu-baseline --no-traffic --no-full-run --no-graph"
pmc() { # name, counters...
  name=$1; shift
  timeout 600 rocprofv3 --pmc "$@" -d $OUT/p_$name -o pmc -- $PMC > $OUT/pmc_$name.log 2>&1
  python $R/tools/rocpd_summary.py $OUT/p_$name/pmc_results.db gemm_dec32 > $OUT/pmc_${name}_gemm_dec32.txt 2>&1
  python $R/tools/rocpd_summary.py $OUT/p_$name/pmc_results.db decode_attention > $OUT/pmc_${name}_attention.txt 2>&1
  rm -rf $OUT/p_$name
}
pmc fetch FETCH_SIZE
pmc sq SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES
rm -f $OUT/*.log
ls -la $OUT | head -40
head -14 $OUT/kernel_trace_by_grid_default.txt
python - <<PY
import json,glob
for f in sorted(glob.glob('$OUT/bench_line_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['value'], d['ms_per_step'], d['step_roofline']['frac'], d.get('value_full_run',{}).get('value'))
    except Exception as e: print(f, 'no json', e)
PY
