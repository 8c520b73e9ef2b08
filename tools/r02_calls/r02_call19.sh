#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
{
timeout 300 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "row_half" 2>&1 | grep -E "AssertionError|passed|failed|FAILED|rror" | head -8
echo "== shape 6 (32 rows x 64 cols, full K unless splits)"
for cfg in "4096 6144 64 0 6 1" "4096 4096 64 0 6 1" "14336 4096 64 0 6 1" "14336 4096 64 0 6 2" "4096 6144 64 0 6 2" "4096 4096 64 0 6 2"; do
    timeout 200 python tools/trace_boundary.py $cfg 2>&1 | tail -1
done
echo "== shape 3 (64 rows x 64 cols), shape 0 for reference"
for cfg in "4096 6144 64 0 3 1" "4096 4096 64 0 3 1" "4096 4096 64 0 3 2" "14336 4096 64 0 3 2" "4096 6144 64 0 0 4" "4096 4096 64 0 0 4" "14336 4096 64 0 0 7"; do
    timeout 200 python tools/trace_boundary.py $cfg 2>&1 | tail -1
done
} > gpurun_out/call19.log 2>&1
cat gpurun_out/call19.log
