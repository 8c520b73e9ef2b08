#!/bin/bash
# round 4, call 17: KV stream policy sc1 (TM_ATTN_POL=2) vs non-temporal (0) vs sc0 sc1 (3) in the model, same box, interleaved
mkdir -p gpurun_out/r04
for pol in 0 2 3 0 2; do
  TM_ATTN_POL=$pol timeout 300 python bench.py --steps 64 --warmup 8 --tune 0 --no-cpu-baseline --no-traffic --no-full-run --profile-steps 4 > gpurun_out/r04/c17_pol$pol.json 2>/dev/null
  python - <<PY
import json
d = json.loads(open('gpurun_out/r04/c17_pol$pol.json').read().strip().splitlines()[-1])
k = d.get('kernel_ms_per_step', {})
print('pol=$pol', d['value'], d['ms_per_step'], d['step_roofline']['frac'], 'attention', k.get('attention'), d['roofline']['frac'], d['roofline']['us_per_launch'])
PY
done
