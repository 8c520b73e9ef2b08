#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
{
timeout 600 python -m pytest tests/test_gpu_engine.py -x -q -m gpu -k "gemm_tuning" 2>&1 | grep -E "AssertionError|passed|failed|FAILED|rror|assert" | head -12
echo "== config 2 tuned (verbose)"; TM_GEMM_TUNE_VERBOSE=1 timeout 600 python bench.py --model internlm2_20b --batch 128 --steps 64 --no-cpu-baseline --no-traffic --no-full-run --profile-steps 0 --tune 1 > gpurun_out/call33_tuned.out 2> gpurun_out/call33_tuned.err
grep "tm tune" gpurun_out/call33_tuned.err | grep -E "\->|heuristic"
grep '"metric"' gpurun_out/call33_tuned.out | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['step_roofline']['frac'], d['sample_tokens'])"
echo "== config 2 --tune 0"; timeout 600 python bench.py --model internlm2_20b --batch 128 --steps 64 --no-cpu-baseline --no-traffic --no-full-run --profile-steps 0 --tune 0 2>/dev/null | grep '"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['step_roofline']['frac'], d['sample_tokens'])"
} > gpurun_out/call33.log 2>&1
cat gpurun_out/call33.log
