#!/bin/bash
# round 3, call 24: library path warmed at engine start: TTFT without the tuner; the engine-level library tests again
mkdir -p gpurun_out/r03
timeout 600 python -m pytest tests/test_gpu_engine.py tests/test_gpu_ops.py -q -m gpu -x -k "f16_library or tuning_roundtrip or prefill_attention" > gpurun_out/r03/c24_tests.log 2>&1
echo "tests rc=$? $(tail -1 gpurun_out/r03/c24_tests.log)"
for t in 0 1; do
  timeout 300 python bench.py --steps 20 --warmup 5 --tune $t --no-cpu-baseline --no-traffic --no-full-run --profile-steps 0 > gpurun_out/r03/c24_tune$t.json 2> gpurun_out/r03/c24_tune$t.err; tail -3 gpurun_out/r03/c24_tune$t.err
  python - <<PY
import json
d = json.loads(open('gpurun_out/r03/c24_tune$t.json').read().strip().splitlines()[-1])
print('tune=$t', d['value'], d['ttft_p50_ms'], d['prefill_tokens_per_s'])
PY
done
