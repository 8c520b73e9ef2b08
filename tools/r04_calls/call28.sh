#!/bin/bash
# round 4, call 28: shape 10 = the 32-row x 64-column tile (shape 6) with stages of 8 k-blocks instead of 4 (half as many
# barriers / activation round trips, one workgroup per CU): parity, traces, per-GEMM time vs shape 6
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "row_half_tiles and 10" 2>&1 | grep -E "passed|failed|rror" | tail -3
for ss in "6 1" "10 1" "10 2"; do
  set -- $ss; echo -n "w_qkv shape $1 x $2: "; timeout 120 python tools/trace_dec32.py 4096 6144 64 0 $1 $2 2>&1 | tail -1 | cut -c1-200
done
for ss in "6 2" "10 1" "10 2"; do
  set -- $ss; echo -n "wo shape $1 x $2: "; timeout 120 python tools/trace_dec32.py 4096 4096 64 0 $1 $2 2>&1 | tail -1 | cut -c1-200
done
timeout 300 python tools/bench_gemm.py --only qkv,o,down --variants auto,d6,d10 --splits 1,2,4 2>&1 | grep -v "^$\|amdgpu.ids"
