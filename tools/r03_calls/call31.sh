#!/bin/bash
# round 3, call 31: decode attention with four blocks in flight per wave at one workgroup per CU (NS = 4): parity, kernel time, step time
mkdir -p gpurun_out/r03
TM_ATTN_DEEP=1 timeout 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py -q -m gpu -x -k "decode_attention" 2>&1 | tail -1
for d in 0 1 0 1; do
  echo -n "deep=$d: "; TM_ATTN_DEEP=$d timeout 100 python tools/bench_attention.py --ctx 1088 --layers 32 --splits 1 --iters 20 2>&1 | grep "ctx=" | tail -1
done
for d in 0 -1; do
  TM_ATTN_DEEP=$d timeout 300 python bench.py --steps 64 --warmup 8 --tune 0 --no-cpu-baseline --no-traffic --no-full-run --profile-steps 4 > gpurun_out/r03/c31_deep$d.json 2>/dev/null
  python - <<PY
import json
d = json.loads(open('gpurun_out/r03/c31_deep$d.json').read().strip().splitlines()[-1])
k = d.get('kernel_ms_per_step', {})
print('deep=$d', d['value'], d['ms_per_step'], d['step_roofline']['frac'], 'attention', k.get('attention'), d['roofline']['frac'])
PY
done
