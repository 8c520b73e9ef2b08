#!/bin/bash
# round 3, call 11: dispatch tuned for prefill size classes -- parity of every candidate, then request-stream throughput and TTFT A/B
mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_bench_launcher.py -q -m gpu -x -k "every_tuner or launcher" 2>&1 | tail -4
for t in 0 1; do
  TM_GEMM_TUNE_VERBOSE=$t timeout 900 python tools/bench_continuous.py --tune $t > gpurun_out/r03/c11_continuous_tune$t.json 2> gpurun_out/r03/c11_continuous_tune$t.err
  echo "continuous tune=$t rc=$?"; tail -c 700 gpurun_out/r03/c11_continuous_tune$t.json; echo
done
grep "tm tune.*->" gpurun_out/r03/c11_continuous_tune1.err | cut -c1-200
for t in 0 1; do
  timeout 600 python bench.py --steps 20 --warmup 5 --tune $t --no-traffic --no-cpu-baseline --no-full-run --profile-steps 0 > gpurun_out/r03/c11_bench_tune$t.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open('gpurun_out/r03/c11_bench_tune$t.json').read().strip().splitlines()[-1])
print('bench tune=$t', d['value'], d['ms_per_step'], 'ttft_p50', d['ttft_p50_ms'], 'prefill tok/s', d['prefill_tokens_per_s'])
PY
done
