#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r02c
mkdir -p $OUT
for abl in 0 24 25 26 27 28 29 30 31 8 16; do
  echo "-- gate_up abl $abl"; timeout 120 python tools/trace_dec32.py 4096 28672 64 1 0 1 $abl 2>&1 | grep -v amdgpu.ids | tail -2
done > $OUT/trace_abl.log 2>&1
cat $OUT/trace_abl.log
