#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
{
echo "-- gate_up M=8192 shape 5"; timeout 300 python tools/trace_dec32.py 4096 28672 8192 1 5 1 2>&1 | tail -4
echo "-- gate_up M=8192 shape 5 no-MFMA (abl 2)"; timeout 300 python tools/trace_dec32.py 4096 28672 8192 1 5 1 2 2>&1 | tail -4
echo "-- gate_up M=512 shape 5 (short launch)"; timeout 300 python tools/trace_dec32.py 4096 28672 512 1 5 1 2>&1 | tail -4
} > gpurun_out/pre64_trace.log 2>&1
cat gpurun_out/pre64_trace.log
