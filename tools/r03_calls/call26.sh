#!/bin/bash
# round 3, call 26: prefill attention with the leaner softmax: parity, then its time in a kernel trace of a short bench run
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03
mkdir -p $OUT
(cd $R; timeout 300 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "prefill_attention" 2>&1 | tail -1)
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/t26 -o trace -- python $R/bench.py --steps 4 --warmup 1 --profile-steps 0 --no-cpu-baseline --no-traffic --no-full-run --tune 0 > $OUT/c26_trace.log 2>&1
python $R/tools/rocpd_summary.py $OUT/t26/trace_results.db > $OUT/c26_kernel_trace_stats.txt 2>&1
rm -rf $OUT/t26
grep -E "prefill_attention|Cijk|silu_mul" $OUT/c26_kernel_trace_stats.txt | cut -c1-150
grep -o '"ttft_p50_ms": [0-9.]*\|"prefill_tokens_per_s": [0-9.]*' $OUT/c26_trace.log | head -2
