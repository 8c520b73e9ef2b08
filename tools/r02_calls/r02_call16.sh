#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
{
echo "== launch floor"; timeout 120 tools/probes/launch_floor_probe
echo "== qkv trace"; timeout 200 python tools/trace_dec32.py 4096 6144 64 0 0 4 2>&1 | tail -3
} > gpurun_out/call16.log 2>&1
cat gpurun_out/call16.log
