#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_engine.py -x -q -m gpu -k "sampling or collective or errors" 2>&1 | grep -E "AssertionError|passed|failed|FAILED|rror|assert" | head -12 > gpurun_out/call34.log
cat gpurun_out/call34.log
