#!/bin/bash
# round 3, call 13: two-shot native all-reduce -- operator tests (streams, IPC processes), the tp = 2 engine on it, launcher test
mkdir -p gpurun_out/r03
timeout 1500 python -m pytest tests/test_gpu_p2p.py tests/test_gpu_tp.py tests/test_gpu_bench_launcher.py -q -m gpu -x > gpurun_out/r03/c13_tests.log 2>&1
echo "tests rc=$?"; tail -30 gpurun_out/r03/c13_tests.log | cut -c1-400
