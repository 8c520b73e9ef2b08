#!/bin/bash
# round 3, call 2: where the in-launch consumer's time goes (stamps), then the bench A/B
mkdir -p gpurun_out/r03
{
timeout 300 python tools/trace_tail.py 4096 4096 64 6 2
timeout 300 python tools/trace_tail.py 4096 4096 64 6 1
timeout 300 python tools/trace_tail.py 14336 4096 64 3 4
timeout 300 python tools/trace_tail.py 14336 4096 64 3 2
} > gpurun_out/r03/c2_trace_tail.txt 2>&1
cat gpurun_out/r03/c2_trace_tail.txt
timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "residual_norm or in_launch" 2>&1 | tail -3
for tail in 1 0; do
  TM_GEMM_TAIL=$tail timeout 600 python bench.py --steps 64 --warmup 8 --no-traffic --no-cpu-baseline --no-full-run > gpurun_out/r03/c2_bench_tail$tail.json 2> gpurun_out/r03/c2_bench_tail$tail.err
  echo "bench tail=$tail rc=$?"; python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r03/c2_bench_tail$tail.json').read().strip().splitlines()[-1])
    print(d['value'], d['ms_per_step'], d['step_roofline']['frac'], d.get('kernel_ms_per_step'))
except Exception as e: print('no json', e)
PY
done
