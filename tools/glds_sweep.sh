#!/bin/bash
cd $GRAFT_REPO_ROOT
TM_GEMM_GLDS=1 timeout 200 python -m pytest tests -m gpu -x -q -k "w4a16" 2>&1 | tail -3
for spec in "gate_up 8,1,1" "down 8,1,8" "qkv 8,1,4" "o 8,1,4"; do
  set -- $spec
  echo -n "V1    "; timeout 120 python tools/tune_gemm.py --only $1 --cfg $2 2>&1 | grep -v BEST | tail -1
  echo -n "GLDS  "; TM_GEMM_GLDS=1 timeout 120 python tools/tune_gemm.py --only $1 --cfg $2 2>&1 | grep -v BEST | tail -1
done
TM_GEMM_GLDS=1 TRACE_LAUNCHES=3 timeout 120 python tools/trace_gemm.py 4096 28672 64 1 1 1 8 2>&1 | tail -6
