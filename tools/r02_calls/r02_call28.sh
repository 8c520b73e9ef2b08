#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/ -q -m gpu -x > gpurun_out/full_gpu_tests.log 2>&1; echo "suite rc=$?" >> gpurun_out/full_gpu_tests.log
grep -E "passed|failed|FAILED|Error|suite rc" gpurun_out/full_gpu_tests.log | tail -8
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log
