#!/bin/bash
# round 4, call 6: the driver command with the start-up tuner (shape 11 among its candidates) -- which tilings win in the model
mkdir -p gpurun_out/r04
TM_GEMM_TUNE_VERBOSE=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04/c6_driver.json 2> gpurun_out/r04/c6_driver.err
python - <<PY
import json
d = json.loads(open('gpurun_out/r04/c6_driver.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['step_roofline'], d['roofline']['frac'], d.get('ttft_p50_ms'))
print(d['config'].get('gemm_tilings'), d['config'].get('prefill_gemm_tilings'))
print(d.get('kernel_ms_per_step'))
PY
grep -i "tune\|shape 11\|winner" gpurun_out/r04/c6_driver.err | head -60
