#!/bin/bash
# round 4, call 12: rotated 256 x 256 prefill loop (last k-step's MFMAs behind the barrier): parity + time + clock trace
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py -q -m gpu -x -k "prefill_tiles or odd_stage or prefill_full_size" 2>&1 | tail -3
timeout 300 python tools/bench_gemm.py --m 8192 --variants d5,p256 --splits 1 --reps 10 2>&1 | grep -v "^$\|amdgpu.ids"
timeout 200 python tools/trace_dec32.py 14336 4096 8192 0 12 1 2>&1 | tail -2 | cut -c1-200
