#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
{
for wt in 0 1 3; do
  for cfg in "4096 6144 64 0 0 4" "4096 4096 64 0 0 4" "14336 4096 64 0 0 7" "4096 28672 64 1 0 1"; do
    TM_D32_WT=$wt timeout 200 python tools/trace_boundary.py $cfg 2>&1 | tail -1
  done
done
for wt in 0 1 3; do
echo "== bench_gemm TM_D32_WT=$wt"; TM_D32_WT=$wt timeout 300 python tools/bench_gemm.py --m 64 --variants d0 2>&1 | grep -v "^$\|amdgpu.ids" | tail -4
done
} > gpurun_out/call17.log 2>&1
cat gpurun_out/call17.log
