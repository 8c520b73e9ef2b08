#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -X faulthandler -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "rectangular" > gpurun_out/call24.log 2>&1
grep -v "^$" gpurun_out/call24.log | tail -40
