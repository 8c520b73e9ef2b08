#!/bin/bash
# round 2: the round-end sequence on one box -- full GPU suite, smoke(), default bench line
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/ -q -m gpu -x > gpurun_out/full_gpu_tests.log 2>&1; echo "suite rc=$?" >> gpurun_out/full_gpu_tests.log
tail -12 gpurun_out/full_gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke.log 2>&1; tail -3 gpurun_out/smoke.log
timeout 900 python bench.py > gpurun_out/bench_default.log 2>&1; tail -2 gpurun_out/bench_default.log
