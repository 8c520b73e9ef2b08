#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
{
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_engine.py -x -q -m gpu -k "row_half or gemm_tuning" 2>&1 | grep -E "AssertionError|passed|failed|FAILED|rror|assert" | head -12
echo "== tuned (verbose)"; TM_GEMM_TUNE_VERBOSE=1 timeout 500 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-full-run --profile-steps 0 --tune 1 > gpurun_out/call30_tuned.out 2> gpurun_out/call30_tuned.err
grep "tm tune" gpurun_out/call30_tuned.err | grep -E "shape [6789]|->"
grep '"metric"' gpurun_out/call30_tuned.out | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
for arm in 0 1; do
echo "== bench --tune $arm"; timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-full-run --profile-steps 0 --tune $arm 2>/dev/null | grep '"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
} > gpurun_out/call30.log 2>&1
cat gpurun_out/call30.log
