#!/bin/bash
# round 4, call 19: the 4-wave (128 x 128 per wave) layout of the 256 x 256 prefill tile again, now with the rotated stage loop
for nw in 8 4; do
  echo "== TM_PRE256_WAVES=$nw"
  TM_PRE256_WAVES=$nw timeout 300 python tools/bench_gemm.py --m 8192 --variants p256 --splits 1 --reps 10 2>&1 | grep -v "^$\|amdgpu.ids"
done
TM_PRE256_WAVES=4 timeout 200 python tools/trace_dec32.py 14336 4096 8192 0 12 1 2>&1 | tail -2 | cut -c1-200
