#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
{
for i in 1 2 3; do
  timeout 300 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "prefill or row_block" 2>&1 | grep -E "AssertionError|passed|failed|FAILED" | head -8
done
echo "== without the 512-column tile"
for i in 1 2; do
  TM_PRE64_MIN_M=100000 timeout 300 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "prefill or row_block" 2>&1 | grep -E "AssertionError|passed|failed|FAILED" | head -8
done
} > gpurun_out/flaky.log 2>&1
cat gpurun_out/flaky.log
