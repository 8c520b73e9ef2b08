#!/bin/bash
# round 3, call 18: (a) the library path timed through the C-ABI next to torch's; (b) TTFT A/B with the tuner on non-zero activations
mkdir -p gpurun_out/r03
timeout 400 python tools/probes/prefill_vs_library.py --ms 2048,8192 --iters 8 > gpurun_out/r03/c18_prefill_vs_library.txt 2>&1
echo rc=$?; grep -v amdgpu.ids gpurun_out/r03/c18_prefill_vs_library.txt | tail -10
for arm in lib nolib resident; do
  case $arm in lib) envs="";; nolib) envs="TM_GEMM_F16_LIBRARY=0";; resident) envs="TM_PREFILL_F16_RESIDENT=1";; esac
  env $envs TM_GEMM_TUNE_VERBOSE=1 timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r03/c18_bench_$arm.json 2> gpurun_out/r03/c18_bench_$arm.err
  echo "bench $arm rc=$?"
  python - <<PY
import json
try:
    d = json.loads(open('gpurun_out/r03/c18_bench_$arm.json').read().strip().splitlines()[-1])
    print('$arm', d['value'], d['ttft_p50_ms'], d['prefill_tokens_per_s'], d['config'].get('prefill_gemm_tilings'), d['config'].get('gemm_tilings'))
except Exception as e:
    print('$arm: no line', e)
PY
  grep -E "M=(8192|64) ->" gpurun_out/r03/c18_bench_$arm.err | head -8
done
