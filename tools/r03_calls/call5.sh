#!/bin/bash
# round 3, call 5: generalised mixed steps (fp16 KV, logits processors, chunked admissions, tp = 2 native + sampling), 2-stream A/B
mkdir -p gpurun_out/r03
timeout 1200 python -m pytest tests/test_gpu_engine.py tests/test_gpu_tp.py -q -m gpu -x > gpurun_out/r03/c5_tests.log 2>&1
echo "tests rc=$?"; tail -25 gpurun_out/r03/c5_tests.log
for two in 1 0; do
  TM_MIXED_2STREAM=$two timeout 600 python tools/bench_continuous.py > gpurun_out/r03/c5_continuous_2stream$two.json 2> gpurun_out/r03/c5_continuous_2stream$two.err
  echo "continuous 2stream=$two rc=$?"; tail -c 900 gpurun_out/r03/c5_continuous_2stream$two.json
done
