#!/bin/bash
# round 4, call 15 (re-run after the pointer sign-extension fix): cache policy of the KV stream in the decode attention kernel
# (TM_ATTN_POL: 0 global nt [default], 1 buffer nt, 2 sc1, 3 sc0 sc1, 4 sc1 nt, 5 none), stand-alone at the bench's context
for pol in 0 1 2 3 4 5 0; do
  echo -n "pol=$pol: "; TM_ATTN_POL=$pol timeout 60 python tools/bench_attention.py --ctx 1088 --layers 32 --splits 1 --iters 20 2>&1 | grep "ctx=\|fault" | tail -1
done
