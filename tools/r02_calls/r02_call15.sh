#!/bin/bash
# race fix check: the diagnostic arm that failed 5 of 6, the prefill / odd-stage tests twice, decode + prefill GEMM timings
mkdir -p gpurun_out
export TMPDIR=/tmp
{
echo "== diag default arm"; timeout 300 python tools/diag_splitk.py 6 2>&1 | grep -v "slab\|amdgpu.ids" | tail -6
for i in 1 2; do
  timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py -q -m gpu -k "prefill or row_block or odd_stage or decode_kernel_modes or fp8" 2>&1 | grep -E "AssertionError|passed|failed|FAILED" | head -8
done
echo "== decode gemm"; timeout 300 python tools/bench_gemm.py --m 64 --variants d0 2>&1 | grep -v "^$\|amdgpu.ids" | tail -5
echo "== prefill gemm"; timeout 300 python tools/bench_gemm.py --m 8192 --variants d4,d5 2>&1 | grep -v "^$\|amdgpu.ids" | tail -9
} > gpurun_out/call15.log 2>&1
cat gpurun_out/call15.log
