#!/bin/bash
# round 3, call 16: prefill-shape GEMM, fused W4A16 kernel vs the vendor library's fp16 GEMM
mkdir -p gpurun_out/r03
timeout 400 python tools/probes/prefill_vs_library.py --ms 1024,4096,8192 --iters 8 > gpurun_out/r03/c16_prefill_vs_library.txt 2>&1
echo rc=$?; cat gpurun_out/r03/c16_prefill_vs_library.txt | tail -20
