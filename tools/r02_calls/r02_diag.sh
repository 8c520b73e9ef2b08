#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
{
for abl in -1 0x1900 0x2900 0x3900 0x800 0; do
  echo "== TM_D32_ABL=$abl"
  if [ "$abl" = "-1" ]; then timeout 300 python tools/diag_splitk.py 6 2>&1 | grep -v "slab\|amdgpu.ids" | tail -6
  else TM_D32_ABL=$abl timeout 300 python tools/diag_splitk.py 6 2>&1 | grep -v "slab\|amdgpu.ids" | tail -6; fi
done
} > gpurun_out/diag_arms.log 2>&1
cat gpurun_out/diag_arms.log
