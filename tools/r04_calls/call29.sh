#!/bin/bash
# round 4, call 29 (timing only, results of the hack are garbage): the int8 decode attention with its cache blocks brought in by
# LDS-DMA (buffer_load ... lds, straight into the wave's LDS image: no VGPR round trip, no ds_write) instead of global loads into
# registers -- does the cache stream run faster than 4.6 TB/s when the data does not return through the vector register path?
for v in base attn_hack attn_hack_nt base attn_hack attn_hack_nt; do
  cp build/ab/$v.so lmdeploy_amd/lib/libtm_mi355x.so
  echo "== $v"
  timeout 100 python tools/bench_attention.py --ctx 1040 --splits 1 --layers 8 --iters 40 2>&1 | grep -i "ctx=\|error\|fault\|abort" | head -3
  timeout 100 python tools/bench_attention.py --ctx 2040 --splits 1 --layers 8 --iters 40 2>&1 | grep -i "ctx=\|error\|fault\|abort" | head -3
done
