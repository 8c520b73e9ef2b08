#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
{
timeout 900 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "rmsnorm or row_half or rectangular" 2>&1 | grep -E "AssertionError|passed|failed|FAILED|rror" | head -8
for arm in "A=1" "TM_D32_MIN_KB=8" "A=1" "TM_D32_MIN_KB=8"; do
echo "== bench $arm"; env $arm timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-full-run 2>&1 | grep '"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['us_per_launch'], d['kernel_ms_per_step'])"
done
} > gpurun_out/call25.log 2>&1
cat gpurun_out/call25.log
