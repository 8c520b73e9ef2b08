#!/bin/bash
# round 4, call 8: phase / clock traces of the prefill tiles at M = 8192 (w2 and w_qkv): 128 x 512 (shape 5) vs 256 x 256 (shape 12)
for sh in 5 12; do
  echo "== shape $sh w2"; timeout 200 python tools/trace_dec32.py 14336 4096 8192 0 $sh 1 2>&1 | tail -2
  echo "== shape $sh qkv"; timeout 200 python tools/trace_dec32.py 4096 6144 8192 0 $sh 1 2>&1 | tail -2
done
