#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
{
timeout 900 python -m pytest tests/test_gpu_engine.py -x -q -m gpu -k "continuous or thread_serving or sampling_static or pipeline_continuous" 2>&1 | grep -E "AssertionError|passed|failed|FAILED|rror|assert" | head -12
for arm in 1 0; do
echo "== bench_continuous TM_MIXED_STEP=$arm"; TM_MIXED_STEP=$arm timeout 400 python tools/bench_continuous.py 2>&1 | grep '"metric"'
done
} > gpurun_out/call27.log 2>&1
cat gpurun_out/call27.log
