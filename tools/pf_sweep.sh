#!/bin/bash
# ring-depth sweep: is the decode GEMM main loop bound by HBM latency x bytes in flight?
cd $GRAFT_REPO_ROOT
for v in 0 1; do for pf in 4 8 12; do
  echo -n "V2=$v PF=$pf  "
  TM_GEMM_V2=$v TM_GEMM_PF=$pf timeout 120 python tools/tune_gemm.py --only gate_up --cfg 8,1,1 2>&1 | grep -v BEST | tail -1
done; done
