#!/bin/bash
# round 3, call 7: Infinity-Cache prefetch of the next layer's newest KV blocks beside the GEMMs -- A/B over the block count
mkdir -p gpurun_out/r03
TM_KV_PREFETCH=8 timeout 600 python -m pytest tests/test_gpu_engine.py -q -m gpu -x -k "matches_oracle or other_config" 2>&1 | tail -3
for n in 0 4 8 17 0; do
  TM_KV_PREFETCH=$n timeout 600 python bench.py --steps 64 --warmup 8 --no-traffic --no-cpu-baseline --no-full-run --profile-steps 0 > gpurun_out/r03/c7_bench_pf$n.json 2> gpurun_out/r03/c7_bench_pf$n.err
  echo "prefetch blocks=$n rc=$?"; python - <<PY
import json
try:
    d=json.loads(open('gpurun_out/r03/c7_bench_pf$n.json').read().strip().splitlines()[-1])
    print(d['value'], d['ms_per_step'], d['step_roofline']['frac'])
except Exception as e: print('no json', e); print(open('gpurun_out/r03/c7_bench_pf$n.err').read()[-1500:])
PY
done
