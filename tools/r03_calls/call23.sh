#!/bin/bash
# round 3, call 23: prefill attention, query heads per wave 4 (spills 25 registers) vs 2 (none): prefill rate through bench.py
mkdir -p gpurun_out/r03
for g in 4 2 4 2; do
  TM_PREFILL_ATTN_G=$g timeout 300 python bench.py --steps 8 --warmup 2 --tune 0 --no-cpu-baseline --no-traffic --no-full-run --profile-steps 0 > gpurun_out/r03/c23_g$g.json 2>/dev/null
  python - <<PY
import json
d = json.loads(open('gpurun_out/r03/c23_g$g.json').read().strip().splitlines()[-1])
print('G=$g', d['ttft_p50_ms'], d['prefill_tokens_per_s'])
PY
done
