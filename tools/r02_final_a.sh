#!/bin/bash
# round-2 evidence, part A: default bench line, the driver's command, kernel trace (by name / by grid), PMC passes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/final_r02
mkdir -p $OUT
B="python $R/bench.py"
TM_GEMM_TUNE_VERBOSE=1 timeout 700 $B > $OUT/bench_line_default.json 2> $OUT/bench_default.err
grep "tm tune" $OUT/bench_default.err > $OUT/gemm_tune_llama3_8b.txt
timeout 300 $B --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_line_driver_command.json 2>/dev/null
CMD="$B --steps 64 --warmup 8 --profile-steps 4 --no-cpu-baseline --no-traffic --no-full-run"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/t -o trace -- $CMD > $OUT/trace.log 2>&1
python $R/tools/rocpd_summary.py $OUT/t/trace_results.db > $OUT/kernel_trace_stats_default.txt 2>&1
python $R/tools/rocpd_summary.py $OUT/t/trace_results.db --by-grid > $OUT/kernel_trace_by_grid_default.txt 2>&1
rm -rf $OUT/t
PMC="$B --steps 12 --warmup 2 --profile-steps 0 --no-cpu-baseline --no-traffic --no-full-run --no-graph"
pmc() { name=$1; shift
  timeout 600 rocprofv3 --pmc "$@" -d $OUT/p_$name -o pmc -- $PMC > $OUT/pmc_$name.log 2>&1
  python $R/tools/rocpd_summary.py $OUT/p_$name/pmc_results.db gemm_dec32 > $OUT/pmc_${name}_gemm_dec32.txt 2>&1
  python $R/tools/rocpd_summary.py $OUT/p_$name/pmc_results.db decode_attention > $OUT/pmc_${name}_attention.txt 2>&1
  rm -rf $OUT/p_$name
}
pmc fetch FETCH_SIZE
pmc sq SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES
rm -f $OUT/*.log
cut -c1-400 $OUT/bench_line_default.json | tail -1; cut -c1-300 $OUT/bench_line_driver_command.json | tail -1
head -14 $OUT/kernel_trace_by_grid_default.txt
