#!/bin/bash
# round 4, call 7: 256 x 256 prefill tile (shape 12): parity, then time at M = 8192 / 2048 against the 128 x 512 tile (d5) and the library
timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "prefill_tiles or odd_stage" 2>&1 | tail -3
for m in 8192 2048; do
  echo "== M=$m"
  TM_GEMM_F16_LIBRARY=1 timeout 300 python tools/bench_gemm.py --m $m --variants d5,p256,lib --splits 1 --reps 10 2>&1 | grep -v "^$\|amdgpu.ids"
done
