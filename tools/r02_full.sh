#!/bin/bash
# round 2: the round-end sequence on one box -- full GPU suite, smoke(), the driver's bench command
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/ -q -m gpu -x > gpurun_out/full_gpu_tests.log 2>&1; echo "suite rc=$?" >> gpurun_out/full_gpu_tests.log
grep -E "passed|failed|FAILED|Error|suite rc" gpurun_out/full_gpu_tests.log | tail -8
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_driver_command.log 2>&1; grep '"metric"' gpurun_out/bench_driver_command.log | cut -c1-330
