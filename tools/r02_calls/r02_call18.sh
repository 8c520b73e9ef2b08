#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
{
for wt in 0 1 33 49 65; do
  for cfg in "4096 6144 64 0 0 4" "14336 4096 64 0 0 7"; do
    TM_D32_WT=$wt timeout 200 python tools/trace_boundary.py $cfg 2>&1 | tail -1
  done
done
echo "== fewer splits (plain stores)"
for cfg in "4096 6144 64 0 0 2" "4096 4096 64 0 0 2" "14336 4096 64 0 0 4" "14336 4096 64 0 0 2" "4096 4096 64 0 0 1"; do
    timeout 200 python tools/trace_boundary.py $cfg 2>&1 | tail -1
done
} > gpurun_out/call18.log 2>&1
cat gpurun_out/call18.log
