#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
{
timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -k "early_refill or odd_stage" 2>&1 | grep -E "AssertionError|passed|failed|FAILED|rror" | head -8
for e in 0 1; do
echo "== bench_gemm TM_D32_EARLY=$e (graph, d0)"; TM_D32_EARLY=$e timeout 300 python tools/bench_gemm.py --m 64 --variants d0 2>&1 | grep -v "^$\|amdgpu.ids" | tail -4
done
echo "== PF=4 late refill"; TM_D32_PF=4 timeout 300 python tools/bench_gemm.py --m 64 --variants d0 --only gate_up,down 2>&1 | grep -v "^$\|amdgpu.ids" | tail -2
for e in 0 1; do
echo "== trace gate_up / down EARLY=$e"; TM_D32_EARLY=$e timeout 200 python tools/trace_dec32.py 4096 28672 64 1 0 1 2>&1 | tail -2
TM_D32_EARLY=$e timeout 200 python tools/trace_dec32.py 14336 4096 64 0 0 7 2>&1 | tail -1
done
} > gpurun_out/call22.log 2>&1
cat gpurun_out/call22.log
