#!/bin/bash
# round 3, call 6: does the decode attention run faster when its KV bytes are Infinity-Cache resident?  (layers 1 = every launch
# re-reads the same 142 MB; layers 4 = four launches cycle over 568 MB: HBM-cold)
mkdir -p gpurun_out/r03
{
for ctx in 1040 1536 2040; do
  for layers in 1 4; do
    timeout 300 python tools/bench_attention.py --ctx $ctx --splits 1 --layers $layers --iters 24
  done
done
} > gpurun_out/r03/c6_attention_mall.txt 2>&1
grep -v amdgpu.ids gpurun_out/r03/c6_attention_mall.txt
