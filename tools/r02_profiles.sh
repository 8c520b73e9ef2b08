#!/bin/bash
# Round-2 evidence: bench lines of every BASELINE config + rocprofv3 kernel traces + PMC passes (separate runs, no tracing
# domains mixed with --pmc).  Summaries land in gpurun_out/profile_r02/ (copied to profiles/ afterwards).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/profile_r02
mkdir -p $OUT
B="python $R/bench.py"
# bench lines
timeout 600 $B > $OUT/bench_line_default.json 2> $OUT/bench_line_default.err
timeout 400 $B --quant-policy 4 --steps 256 --no-cpu-baseline > $OUT/bench_line_llama3_8b_int4kv.json 2>/dev/null
timeout 400 $B --quant-policy 0 --steps 128 --no-cpu-baseline > $OUT/bench_line_config1_fp16kv.json 2>/dev/null
timeout 600 $B --model internlm2_20b --batch 128 --steps 128 --no-cpu-baseline > $OUT/bench_line_config2_internlm2_20b_b128.json 2>/dev/null
timeout 600 $B --model llama3_70b --quant-policy 4 --emulate-tp 8 --steps 128 --no-cpu-baseline --no-traffic > $OUT/bench_line_config3_llama3_70b_tp8_rank_emulation.json 2>/dev/null
timeout 900 $B --model mixtral_8x7b --steps 128 --no-cpu-baseline --no-traffic > $OUT/bench_line_config5_mixtral_fp8_tp1.json 2>/dev/null
timeout 900 $B --model mixtral_8x7b --emulate-tp 2 --steps 128 --no-cpu-baseline --no-traffic > $OUT/bench_line_config5_mixtral_fp8_tp2_rank_emulation.json 2>/dev/null
# kernel traces
CMD="$B --steps 64 --warmup 8 --profile-steps 4 --no-cpu-baseline --no-traffic --no-full-run"
trace() { # name, extra args
  name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/t_$name -o trace -- $CMD "$@" > $OUT/trace_$name.log 2>&1
  python $R/tools/rocpd_summary.py $OUT/t_$name/trace_results.db > $OUT/kernel_trace_stats_$name.txt 2>&1
  rm -rf $OUT/t_$name
}
trace default
trace int4kv --quant-policy 4
trace config2_internlm2_20b_b128 --model internlm2_20b --batch 128
# PMC passes (eager, few steps)
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
pmc lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_UNALIGNED_STALL
rm -f $OUT/*.log
ls -la $OUT | head -40
head -12 $OUT/kernel_trace_stats_default.txt
