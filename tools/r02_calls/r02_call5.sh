#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r02e
mkdir -p $OUT
echo "== HBM-cold (distinct weights)"; timeout 200 python tools/bench_gemm.py --variants old,abl256,abl32 2>&1 | grep -v amdgpu.ids | tee $OUT/cold.log
echo "== Infinity-Cache-resident (same weight)"; timeout 200 python tools/bench_gemm.py --same --variants old,abl256,abl32 2>&1 | grep -v amdgpu.ids | tee $OUT/same.log
echo "== shape 2 (8 waves)"; timeout 200 python tools/bench_gemm.py --abl-shape 2 --variants d2,abl256,abl32 --only gate_up,o 2>&1 | grep -v amdgpu.ids | tee $OUT/shape2.log
