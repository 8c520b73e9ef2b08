#!/bin/bash
# experiment: decode attention at one workgroup per CU (LDS padded beyond 80 KB) vs two: is the launch bound by bytes in flight?
for pad in 0 16 0 16; do
  echo "pad $pad KB:"; TM_ATTN_LDS_PAD_KB=$pad timeout 200 python tools/bench_attention.py --ctx 1088 --layers 32 --splits 1 --iters 20 2>&1 | grep -v amdgpu.ids | tail -2
done
