#!/bin/bash
# round 4, call 16: why the buffer-descriptor policies of call 15 printed nothing
TM_ATTN_POL=1 timeout 60 python tools/bench_attention.py --ctx 1088 --layers 32 --splits 1 --iters 5 2>&1 | grep -v amdgpu.ids | tail -12
