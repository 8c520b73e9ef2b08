#!/bin/bash
# round 4, call 4: shader clock inside the loop of the loader / consumer kernel, full and ablated (is the step power-limited?)
cp build/exp/libtm_mi355x.so lmdeploy_amd/lib/libtm_mi355x.so
for abl in 0 1 2 4 7 8 16 24; do
  echo "abl=$abl: "; timeout 120 python tools/trace_dec32.py 4096 28672 64 1 11 1 $abl 2>&1 | tail -2
done
