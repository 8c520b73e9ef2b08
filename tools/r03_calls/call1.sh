#!/bin/bash
# round 3, call 1: the in-launch residual-norm consumer -- parity tests, then the bench with and without it
mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "residual_norm or in_launch or rmsnorm or row_half" > gpurun_out/r03/c1_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r03/c1_tests.log
tail -5 gpurun_out/r03/c1_tests.log
for tail in 1 0; do
  TM_GEMM_TAIL=$tail timeout 600 python bench.py --steps 64 --warmup 8 --no-traffic --no-cpu-baseline --no-full-run > gpurun_out/r03/c1_bench_tail$tail.json 2> gpurun_out/r03/c1_bench_tail$tail.err
  echo "bench tail=$tail rc=$?"; python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r03/c1_bench_tail$tail.json').read().strip().splitlines()[-1])
    print(d['value'], d['ms_per_step'], d['step_roofline']['frac'], d.get('kernel_ms_per_step'))
except Exception as e: print('no json', e)
PY
done
