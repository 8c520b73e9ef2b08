#!/bin/bash
# round 3, call 15: flakiness screen of the final tree -- the multi-process / hand-off tests five times, the whole suite once more
mkdir -p gpurun_out/r03
for i in 1 2 3 4 5; do
  timeout 900 python -m pytest tests/test_gpu_p2p.py tests/test_gpu_tp.py tests/test_gpu_bench_launcher.py tests/test_gpu_ops.py -q -m gpu -x -k "p2p or tp2 or launcher or in_launch or residual_norm" > gpurun_out/r03/c15_pass$i.log 2>&1
  echo "pass $i rc=$? $(tail -1 gpurun_out/r03/c15_pass$i.log)"
done
timeout 1800 python -m pytest tests -q -m gpu -x > gpurun_out/r03/c15_all_gpu_tests.log 2>&1
echo "all gpu tests rc=$?"; tail -2 gpurun_out/r03/c15_all_gpu_tests.log
