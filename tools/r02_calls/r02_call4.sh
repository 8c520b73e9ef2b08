#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r02d
mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "decode_kernel_modes" --maxfail=6 > $OUT/pytest_modes.log 2>&1; tail -4 $OUT/pytest_modes.log
for abl in 0 256 512 768 1024 1280 1792 263; do
  echo "-- gate_up abl $abl"; timeout 120 python tools/trace_dec32.py 4096 28672 64 1 0 1 $abl 2>&1 | grep -v amdgpu.ids | tail -2
done > $OUT/trace_modes.log 2>&1
cut -c1-200 $OUT/trace_modes.log
timeout 300 python tools/bench_gemm.py --variants d0,abl256,abl768,abl1280,abl1792 > $OUT/bench_modes.log 2>&1
cat $OUT/bench_modes.log
