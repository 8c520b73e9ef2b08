#!/bin/bash
# round 3, call 21: the whole GPU suite + smoke on the final tree
mkdir -p gpurun_out/r03
timeout 1800 python -m pytest tests -q -m gpu -x > gpurun_out/r03/c21_all_gpu_tests.log 2>&1
echo "all gpu tests rc=$?"; grep -E "passed|failed" gpurun_out/r03/c21_all_gpu_tests.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
