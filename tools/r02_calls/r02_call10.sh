#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r02j
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "kernel_modes or row_block" --maxfail=8 > $OUT/pytest.log 2>&1; grep -E "passed|failed" $OUT/pytest.log | tail -1; grep -E "^FAILED" $OUT/pytest.log | head
for abl in 256 2304; do
  echo "-- gate_up abl $abl"; timeout 120 python tools/trace_dec32.py 4096 28672 64 1 0 1 $abl 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-220
done
timeout 300 python tools/bench_gemm.py --variants abl256,abl2304 2>&1 | grep -v amdgpu.ids | tee $OUT/bench_decode.log
for m in 8192 256; do timeout 200 python tools/bench_gemm.py --m $m --reps 8 --variants abl256,abl2304 2>&1 | grep -v amdgpu.ids; done | tee $OUT/bench_prefill.log
