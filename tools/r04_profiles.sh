#!/bin/bash
# Round-4 evidence: bench lines (all BASELINE configs + TP-shard emulations) + rocprofv3 kernel trace (by grid) + PMC passes keyed by
# (kernel, workgroups).  PMC passes are separate runs, never combined with tracing domains.  Summaries -> gpurun_out/profile_r04/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/profile_r04
mkdir -p $OUT
B="python $R/bench.py"
if [ "$1" != "pmc-only" ]; then
timeout 600 $B > $OUT/bench_line_default.json 2> $OUT/bench_line_default.err
timeout 400 $B --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_line_driver_command.json 2>/dev/null
timeout 400 $B --quant-policy 4 --steps 256 --no-cpu-baseline --no-full-run > $OUT/bench_line_llama3_8b_int4kv.json 2>/dev/null
timeout 400 $B --quant-policy 0 --steps 128 --no-cpu-baseline --no-full-run > $OUT/bench_line_config1_fp16kv.json 2>/dev/null
timeout 600 $B --model internlm2_20b --batch 128 --steps 128 --no-cpu-baseline --no-full-run > $OUT/bench_line_config2_internlm2_20b_b128.json 2>/dev/null
timeout 600 $B --model llama3_70b --quant-policy 4 --emulate-tp 8 --steps 128 --no-cpu-baseline --no-traffic --no-full-run > $OUT/bench_line_config3_llama3_70b_tp8_rank_emulation.json 2>/dev/null
timeout 600 $B --emulate-tp 8 --steps 128 --no-cpu-baseline --no-traffic --no-full-run > $OUT/bench_line_llama3_8b_tp8_rank_emulation.json 2>/dev/null
timeout 600 $B --emulate-tp 2 --steps 128 --no-cpu-baseline --no-traffic --no-full-run > $OUT/bench_line_llama3_8b_tp2_rank_emulation.json 2>/dev/null
timeout 900 $B --model mixtral_8x7b --steps 128 --no-cpu-baseline --no-traffic --no-full-run > $OUT/bench_line_config5_mixtral_fp8_tp1.json 2>/dev/null
timeout 600 python $R/tools/bench_continuous.py > $OUT/continuous_batching_line.json 2>/dev/null
CMD="$B --steps 64 --warmup 8 --profile-steps 4 --no-cpu-baseline --no-traffic --no-full-run"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/t_default -o trace -- $CMD > $OUT/trace_default.log 2>&1
python $R/tools/rocpd_summary.py $OUT/t_default/trace_results.db > $OUT/kernel_trace_stats_default.txt 2>&1
python $R/tools/rocpd_summary.py --by-grid $OUT/t_default/trace_results.db > $OUT/kernel_trace_by_grid_default.txt 2>&1
rm -rf $OUT/t_default
fi
# PMC: eager decode steps, NO tuner (its candidate launches would pollute the per-template means), 4 layers
PMC="$B --steps 12 --warmup 2 --profile-steps 0 --no-cpu-baseline --no-traffic --no-full-run --no-graph --tune 0 --layers 4"
pmc() { # name, counters...
  name=$1; shift
  timeout 600 rocprofv3 --pmc "$@" -d $OUT/p_$name -o pmc -- $PMC > $OUT/pmc_$name.log 2>&1
  python $R/tools/rocpd_summary.py $OUT/p_$name/pmc_results.db gemm_dec > $OUT/pmc_${name}_gemm.txt 2>&1
  python $R/tools/rocpd_summary.py $OUT/p_$name/pmc_results.db decode_attention > $OUT/pmc_${name}_attention.txt 2>&1
  rm -rf $OUT/p_$name
}
pmc fetch FETCH_SIZE
pmc sq SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES
pmc lds SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
pmc tcc1 TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum
pmc tcc2 TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum TCC_REQ_sum TCC_BUSY_sum
pmc ta TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_READ_LDS_WAVEFRONTS_sum
pmc tcp TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_RFIFO_STALL_CYCLES_sum
grep -l "rror\|nvalid" $OUT/pmc_*.log 2>/dev/null | head
tail -3 $OUT/pmc_tcc1.log $OUT/pmc_ta.log 2>/dev/null | cut -c1-300
rm -f $OUT/trace_default.log
ls $OUT | head -60
head -16 $OUT/kernel_trace_by_grid_default.txt 2>/dev/null
python - <<PY
import json,glob
for f in sorted(glob.glob('$OUT/bench_line_*.json')) + ['$OUT/continuous_batching_line.json']:
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['value'], d.get('ms_per_step'), d.get('step_roofline',{}).get('frac'), d.get('value_1k_out'), d.get('ttft_p50_ms'), d.get('roofline',{}).get('frac'))
    except Exception as e: print(f, 'no json', e)
PY
