#!/bin/bash
# round 3, call 19: LDS-staged prefill attention (parity), library-path tests, gated chunk size, TTFT
mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py tests/test_gpu_engine.py -q -m gpu -x -k "prefill_attention or f16_library or matches_oracle or continuous_batching_matches_oracle" > gpurun_out/r03/c19_tests.log 2>&1
echo "tests rc=$? $(tail -1 gpurun_out/r03/c19_tests.log)"
for mb in 96 256 512; do
  echo "gated chunk $mb MB:"; TM_F16_GATED_CHUNK_MB=$mb timeout 200 python tools/probes/prefill_vs_library.py --ms 8192 --iters 8 --only gate_up 2>&1 | grep gate_up | cut -c1-175
done
for arm in default nolib; do
  case $arm in default) envs="";; nolib) envs="TM_GEMM_F16_LIBRARY=0";; esac
  env $envs TM_GEMM_TUNE_VERBOSE=1 timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r03/c19_bench_$arm.json 2> gpurun_out/r03/c19_bench_$arm.err
  echo "bench $arm rc=$?"
  python - <<PY
import json
try:
    d = json.loads(open('gpurun_out/r03/c19_bench_$arm.json').read().strip().splitlines()[-1])
    print('$arm', d['value'], d['ttft_p50_ms'], d['prefill_tokens_per_s'], d['config'].get('prefill_gemm_tilings'))
except Exception as e:
    print('$arm: no line', e)
PY
  grep -E "M=8192 ->" gpurun_out/r03/c19_bench_$arm.err | head -4
done
