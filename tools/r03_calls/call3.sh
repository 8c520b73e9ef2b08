#!/bin/bash
# round 3, call 3: the new parity tests first (verbose), then the whole GPU suite
mkdir -p gpurun_out/r03
timeout 1500 python -m pytest tests/test_gpu_checkpoint.py tests/test_gpu_fullsize.py -x -q -m gpu -s -k "checkpoint or every_tuner or every_candidate or eight_full or canaries" --durations=8 > gpurun_out/r03/c3_new_tests.log 2>&1
echo "new tests rc=$?"; tail -25 gpurun_out/r03/c3_new_tests.log
timeout 1500 python -m pytest tests -q -m gpu -x --durations=10 > gpurun_out/r03/c3_all_gpu_tests.log 2>&1
echo "all gpu tests rc=$?"; tail -18 gpurun_out/r03/c3_all_gpu_tests.log
