#!/bin/bash
# round 4, call 20 (experiments build, timing only): (a) activations k-block major ([kb][row][128]) for the loader / consumer
# kernel -- does the L2 -> LDS path care about the row stride?  (b) raw u4 codes as fp16 subnormals on the matrix pipe instead of
# the dequantised operand (64), plus the accumulators rescaled per group (192): what would MFMA-on-codes buy at M = 64 and at
# M = 8192 (pre64: 8 / 24)?
cp build/exp/libtm_mi355x.so lmdeploy_amd/lib/libtm_mi355x.so
for abl in -1 32 64 192 224; do
  echo "lc w1w3 abl=$abl: "; timeout 120 python tools/trace_dec32.py 4096 28672 64 1 11 1 $abl 2>&1 | tail -3 | cut -c1-260
done
for abl in -1 32; do
  echo "lc w2 x4 abl=$abl: "; timeout 120 python tools/trace_dec32.py 14336 4096 64 0 11 4 $abl 2>&1 | tail -2 | cut -c1-260
done
for abl in -1 8 24; do
  echo "pre64 w1w3 M=8192 abl=$abl: "; timeout 200 python tools/trace_dec32.py 4096 28672 8192 1 5 1 $abl 2>&1 | tail -3 | cut -c1-260
done
