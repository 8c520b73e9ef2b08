#!/bin/bash
# rocprofv3 passes over a short bench.py run; summaries land in gpurun_out/profile_$TAG/ (copy to profiles/).
# PMC passes are separate runs and never combined with tracing domains.
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/profile_$TAG
mkdir -p $OUT
CMD="python $R/bench.py --steps 64 --warmup 8 --profile-steps 4 --no-cpu-baseline"
PMC_CMD="python $R/bench.py --steps 16 --warmup 4 --profile-steps 0 --no-cpu-baseline --no-graph"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
python $R/tools/rocpd_summary.py $OUT/trace/trace_results.db > $OUT/kernel_trace_stats.txt 2>&1
grep '"metric"' $OUT/trace.log > $OUT/bench_line_under_rocprof.json
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- $PMC_CMD > $OUT/pmc_fetch.log 2>&1
python $R/tools/rocpd_summary.py $OUT/pmc_fetch/pmc_results.db decode_attention > $OUT/pmc_fetch_attention.txt 2>&1
python $R/tools/rocpd_summary.py $OUT/pmc_fetch/pmc_results.db gemm_kernel > $OUT/pmc_fetch_gemm.txt 2>&1
python $R/tools/attn_traffic.py $OUT/pmc_fetch/pmc_results.db 64 1024 4 16 > $OUT/attention_traffic.json 2>$OUT/attention_traffic.err
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES -d $OUT/pmc_sq -o pmc -- $PMC_CMD > $OUT/pmc_sq.log 2>&1
python $R/tools/rocpd_summary.py $OUT/pmc_sq/pmc_results.db decode_attention > $OUT/pmc_sq_attention.txt 2>&1
python $R/tools/rocpd_summary.py $OUT/pmc_sq/pmc_results.db gemm_kernel > $OUT/pmc_sq_gemm.txt 2>&1
rm -rf $OUT/trace $OUT/pmc_fetch $OUT/pmc_sq   # keep the text summaries only (the .db files are tens of MB)
head -30 $OUT/kernel_trace_stats.txt
