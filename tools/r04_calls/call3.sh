#!/bin/bash
# round 4, call 3: loader priority (bit 32 = off) and prologue ordering (bit 64 = old order) A/B on the loader / consumer kernel
cp build/exp/libtm_mi355x.so lmdeploy_amd/lib/libtm_mi355x.so
for abl in 0 32 64 96 0 7 39 8 40; do
  echo -n "abl=$abl: "; timeout 120 python tools/trace_dec32.py 4096 28672 64 1 11 1 $abl 2>&1 | tail -1
done
echo "== w2 / qkv / wo with the new defaults"
timeout 120 python tools/trace_dec32.py 14336 4096 64 0 11 7 2>&1 | tail -1
timeout 120 python tools/trace_dec32.py 14336 4096 64 0 11 4 2>&1 | tail -1
timeout 120 python tools/trace_dec32.py 4096 6144 64 0 11 4 2>&1 | tail -1
timeout 120 python tools/trace_dec32.py 4096 4096 64 0 11 4 2>&1 | tail -1
timeout 300 python tools/bench_gemm.py --m 64 --variants auto,lc --splits 1,2,4,7 2>&1 | grep -v "^$\|amdgpu.ids"
