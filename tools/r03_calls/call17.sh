#!/bin/bash
# round 3, call 17: the library path for prefill-sized forwards: parity tests, then TTFT A/B through bench.py
mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_engine.py -q -m gpu -x -k "f16_library or tuner_candidate or tuning_roundtrip" > gpurun_out/r03/c17_tests.log 2>&1
echo "tests rc=$? $(tail -1 gpurun_out/r03/c17_tests.log)"
for arm in lib nolib resident; do
  case $arm in lib) envs="";; nolib) envs="TM_GEMM_F16_LIBRARY=0";; resident) envs="TM_PREFILL_F16_RESIDENT=1";; esac
  env $envs TM_GEMM_TUNE_VERBOSE=1 timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r03/c17_bench_$arm.json 2> gpurun_out/r03/c17_bench_$arm.err
  echo "bench $arm rc=$?"
  python - <<PY
import json
try:
    d = json.loads(open('gpurun_out/r03/c17_bench_$arm.json').read().strip().splitlines()[-1])
    print('$arm', d['value'], d['ttft_p50_ms'], d['prefill_tokens_per_s'], d['config'].get('prefill_gemm_tilings'))
except Exception as e:
    print('$arm: no line', e)
PY
  grep "M=8192 ->" gpurun_out/r03/c17_bench_$arm.err | head -4
done
