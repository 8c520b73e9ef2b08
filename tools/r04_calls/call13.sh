#!/bin/bash
# round 4, call 13: shape 13 (256 x 256 tile fed from a per-call fp16 image): parity, time incl. the dequant pass
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py -q -m gpu -x -k "prefill_tiles or odd_stage or prefill_full_size" 2>&1 | tail -3
for m in 8192 2048; do
timeout 300 python tools/bench_gemm.py --m $m --variants d5,p256,p256f --splits 1 --reps 10 2>&1 | grep -v "^$\|amdgpu.ids"
done
