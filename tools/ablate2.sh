#!/bin/bash
# ablation combos of the symmetric decode GEMM (timing only): which subsystem bounds the main loop
cd $GRAFT_REPO_ROOT
for abl in 0 7 24 3 12 15 31; do
  echo -n "ABL=$abl  "
  TM_GEMM_V2=0 TM_GEMM_KSTAGE=4 TM_GEMM_ABL=$abl timeout 120 python tools/tune_gemm.py --only gate_up --cfg 8,1,1 2>&1 | grep -v BEST | tail -1
done
