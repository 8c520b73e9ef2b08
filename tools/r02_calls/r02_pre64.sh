#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py -x -q -m gpu -k "prefill or row_block" > gpurun_out/pre64_tests.log 2>&1; echo "rc=$?" >> gpurun_out/pre64_tests.log
tail -4 gpurun_out/pre64_tests.log
{
for m in 8192 512; do
  timeout 300 python tools/bench_gemm.py --m $m --variants d4,d5 --splits 0 2>&1 | grep -v "^$" | tail -12
done
echo "-- gate_up M=8192 shape 5"; timeout 300 python tools/trace_dec32.py 4096 28672 8192 1 5 1 2>&1 | tail -2
} > gpurun_out/pre64_bench.log 2>&1
cat gpurun_out/pre64_bench.log
