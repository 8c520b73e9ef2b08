#!/bin/bash
# round 4, call 11: in-model A/B of w1w3 on the loader / consumer kernel (shape 11) vs the 16-wave tile (shape 0), same box, twice each
mkdir -p gpurun_out/r04
printf '4096 28672 64 11 1\n' > /tmp/lc.txt
printf '4096 28672 64 0 1\n' > /tmp/d0.txt
for rep in 1 2; do
for t in d0 lc; do
  TM_GEMM_IMPORT=/tmp/$t.txt timeout 300 python bench.py --steps 64 --warmup 8 --tune 0 --no-cpu-baseline --no-traffic --no-full-run --profile-steps 4 > gpurun_out/r04/c11_$t.json 2>/dev/null
  python - <<PY
import json
d = json.loads(open('gpurun_out/r04/c11_$t.json').read().strip().splitlines()[-1])
k = d.get('kernel_ms_per_step', {})
print('$t', d['value'], d['ms_per_step'], d['config']['gemm_tilings']['w1w3'], 'gate_up', k.get('gemm_gate_up'), 'ttft', d['ttft_p50_ms'], d['config']['prefill_gemm_tilings'])
PY
done
done
