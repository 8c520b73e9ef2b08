#!/bin/bash
# A/B of the decode GEMM kernels (TM_GEMM_V2=0/1) on the four Llama-3-8B decode shapes: event-timed medians
cd $GRAFT_REPO_ROOT
for v in 0 1; do
  echo "== TM_GEMM_V2=$v"
  for spec in "gate_up 8,1,1" "down 8,1,8" "qkv 8,1,4" "o 8,1,4"; do
    set -- $spec
    TM_GEMM_V2=$v timeout 120 python tools/tune_gemm.py --only $1 --cfg $2 2>&1 | grep -v BEST | tail -1
  done
done
