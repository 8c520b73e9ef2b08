#!/bin/bash
# round 4, call 2: loader / consumer kernel ablations (experiments build): which side bounds the w1w3 loop
cp build/exp/libtm_mi355x.so lmdeploy_amd/lib/libtm_mi355x.so
for abl in 0 1 2 4 8 16 7 15 24 31; do
  echo -n "abl=$abl: "; timeout 120 python tools/trace_dec32.py 4096 28672 64 1 11 1 $abl 2>&1 | tail -1
done
