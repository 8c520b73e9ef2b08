#!/bin/bash
# round 4, call 1: loader / consumer decode GEMM (shape 11): parity, per-GEMM time against the library's current picks, phase traces
mkdir -p gpurun_out/r04
timeout 500 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "loader_consumer or identity_asymmetric" 2>&1 | tail -3
echo "== bench_gemm auto (current picks) + lc, PFS=3"
timeout 400 python tools/bench_gemm.py --m 64 --variants auto,lc --splits 1,2,4,7,8 2>&1 | grep -v "^$\|amdgpu.ids"
for pf in 2 4; do
  echo "== lc PFS=$pf"
  TM_LC_PFS=$pf timeout 300 python tools/bench_gemm.py --m 64 --variants lc --splits 1,4,7 --only gate_up,down 2>&1 | grep -v "^$\|amdgpu.ids"
done
echo "== traces: w1w3 shape 0 / shape 11, w2 shape 3x4 / shape 11 x7, x4"
timeout 120 python tools/trace_dec32.py 4096 28672 64 1 0 1 2>&1 | tail -2
timeout 120 python tools/trace_dec32.py 4096 28672 64 1 11 1 2>&1 | tail -2
timeout 120 python tools/trace_dec32.py 14336 4096 64 0 3 4 2>&1 | tail -2
timeout 120 python tools/trace_dec32.py 14336 4096 64 0 11 7 2>&1 | tail -2
timeout 120 python tools/trace_dec32.py 14336 4096 64 0 11 4 2>&1 | tail -2
