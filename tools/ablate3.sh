#!/bin/bash
# does the activation traffic bound the decode GEMM?  half the x loads (32), half the x LDS writes (64), both (96)
cd $GRAFT_REPO_ROOT
for abl in 0 32 64 96 8; do
  echo -n "ABL=$abl  "
  TM_GEMM_KSTAGE=4 TM_GEMM_ABL=$abl timeout 120 python tools/tune_gemm.py --only gate_up --cfg 8,1,1 2>&1 | grep -v BEST | tail -1
done
