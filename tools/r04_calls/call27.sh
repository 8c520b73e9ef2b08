#!/bin/bash
# round 4, call 27: upper bound of any weight-prefetch scheme: the four decode GEMMs with every launch of the graph on the SAME
# weights (cache-resident after the first replay: L2 if it survives the kernel boundary, else the Infinity Cache) vs distinct weights
timeout 300 python tools/bench_gemm.py --variants auto 2>&1 | grep -v "^$\|amdgpu.ids"
echo "== --same"
timeout 300 python tools/bench_gemm.py --variants auto --same 2>&1 | grep -v "^$\|amdgpu.ids"
