#!/bin/bash
# round-2 evidence, part C: kernel trace and PMC passes WITHOUT the start-up tuner in the process (its candidate launches would
# mix into the per-kernel means); the heuristic picks the same tilings the tuner chose on these shapes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/final_r02
mkdir -p $OUT
B="python $R/bench.py --tune 0"
CMD="$B --steps 64 --warmup 8 --profile-steps 4 --no-cpu-baseline --no-traffic --no-full-run"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/t -o trace -- $CMD > $OUT/trace.log 2>&1
python $R/tools/rocpd_summary.py $OUT/t/trace_results.db > $OUT/kernel_trace_stats_default.txt 2>&1
python $R/tools/rocpd_summary.py $OUT/t/trace_results.db --by-grid > $OUT/kernel_trace_by_grid_default.txt 2>&1
grep '"metric"' $OUT/trace.log | cut -c1-260 >> $OUT/kernel_trace_by_grid_default.txt
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
head -16 $OUT/kernel_trace_by_grid_default.txt | cut -c1-130; tail -1 $OUT/kernel_trace_by_grid_default.txt
grep -E "FETCH|MFMA_BUSY|WAVE_CYCLES|WAIT_INST" $OUT/pmc_fetch_gemm_dec32.txt $OUT/pmc_sq_gemm_dec32.txt $OUT/pmc_fetch_attention.txt | cut -c1-200
