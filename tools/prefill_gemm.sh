#!/bin/bash
# prefill-shape GEMM (M = 8192): 4-wave x 4-tile vs 8-wave x 2-tile workgroups, 128-row tiles
cd $GRAFT_REPO_ROOT
for cfg in 4,4,1 8,2,1; do for shp in gate_up down qkv; do
  echo -n "cfg=$cfg  "; timeout 120 python tools/tune_gemm.py --m 8192 --iters 8 --only $shp --cfg $cfg 2>&1 | grep -v BEST | tail -1
done; done
