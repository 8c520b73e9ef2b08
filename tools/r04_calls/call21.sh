#!/bin/bash
# round 4, call 21: single-burst decode tiles (gemm_decode_burst.hip, shapes 13 .. 17): parity, phase traces and per-GEMM time
# against the current picks for w_qkv / wo, then the driver command with the tuner's choices printed
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "single_burst" 2>&1 | tail -5
echo "== traces: w_qkv current (6 x 1) / 13 x4 / 14 x4 / 15 x4 / 16 x2 / 17 x2"
for ss in "6 1" "13 4" "14 4" "15 4" "16 2" "17 2"; do
  set -- $ss; echo -n "shape $1 x $2: "; timeout 120 python tools/trace_dec32.py 4096 6144 64 0 $1 $2 2>&1 | tail -1 | cut -c1-230
done
echo "== traces: wo current (6 x 2) / 13 x4 / 14 x4 / 16 x2 / 17 x2"
for ss in "6 2" "13 4" "14 4" "16 2" "17 2"; do
  set -- $ss; echo -n "shape $1 x $2: "; timeout 120 python tools/trace_dec32.py 4096 4096 64 0 $1 $2 2>&1 | tail -1 | cut -c1-230
done
echo "== bench_gemm"
timeout 300 python tools/bench_gemm.py --only qkv,o --variants auto,d13,d14,d15 --splits 4 2>&1 | grep -v "^$\|amdgpu.ids"
timeout 300 python tools/bench_gemm.py --only qkv,o --variants d16,d17 --splits 2 2>&1 | grep -v "^$\|amdgpu.ids"
echo "== driver command"
TM_GEMM_TUNE_VERBOSE=1 timeout 600 python bench.py --steps 20 --warmup 5 2> gpurun_out/r04_call21_bench.err | cut -c1-3000
grep "tm tune.*->" gpurun_out/r04_call21_bench.err
