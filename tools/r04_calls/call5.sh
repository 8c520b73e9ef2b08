#!/bin/bash
# round 4, call 5: activation DMA variants in the loader / consumer kernel: piece rotation per workgroup (0x80), nt (0x100), sc1 (0x200);
# memory-only (| 7) and full
cp build/exp/libtm_mi355x.so lmdeploy_amd/lib/libtm_mi355x.so
for abl in 7 0x87 0x107 0x207 7 0 0x80 0x100 0x200 0; do
  echo "abl=$abl: "; timeout 120 python tools/trace_dec32.py 4096 28672 64 1 11 1 $(($abl)) 2>&1 | tail -2
done
