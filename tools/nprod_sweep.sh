#!/bin/bash
cd $GRAFT_REPO_ROOT
echo -n "V1          "; TM_GEMM_V2=0 timeout 120 python tools/tune_gemm.py --only gate_up --cfg 8,1,1 2>&1 | grep -v BEST | tail -1
for np in 1 2 4; do
  echo -n "V2 NPROD=$np  "
  TM_GEMM_V2=1 TM_GEMM_NPROD=$np timeout 120 python tools/tune_gemm.py --only gate_up --cfg 8,1,1 2>&1 | grep -v BEST | tail -1
done
TM_GEMM_NPROD=4 timeout 200 python -m pytest tests -m gpu -x -q -k "w4a16" 2>&1 | tail -2
TM_GEMM_NPROD=4 TRACE_LAUNCHES=4 timeout 120 python tools/trace_gemm.py 4096 28672 64 1 1 1 8 2>&1 | tail -6
