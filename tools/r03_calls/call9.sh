#!/bin/bash
# round 3, call 9: whole GPU suite on the final code, then the round's bench lines and rocprofv3 summaries
mkdir -p gpurun_out/r03
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/r03/c9_all_gpu_tests.log 2>&1
echo "all gpu tests rc=$?"; tail -5 gpurun_out/r03/c9_all_gpu_tests.log
bash tools/r03_profiles.sh
