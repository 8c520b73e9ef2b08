#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r02k
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py -m gpu -q -k "w4a16" --maxfail=8 > $OUT/pytest.log 2>&1; grep -E "passed|failed" $OUT/pytest.log | tail -1; grep -E "^FAILED" $OUT/pytest.log | head
for rot in 0 1; do
  echo "== rotate $rot"
  TM_D32_ROTATE=$rot timeout 120 python tools/trace_dec32.py 4096 28672 64 1 0 1 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-200
  TM_D32_ROTATE=$rot timeout 300 python tools/bench_gemm.py --variants d0 2>&1 | grep -v amdgpu.ids
  TM_D32_ROTATE=$rot timeout 200 python tools/bench_gemm.py --m 8192 --reps 6 --variants d4 --only gate_up,down 2>&1 | grep -v amdgpu.ids
done | tee $OUT/rotate.log
