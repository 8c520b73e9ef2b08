#!/bin/bash
# launch-overhead knobs of the HIP runtime on the decode graph
mkdir -p gpurun_out
export TMPDIR=/tmp
run() { echo "== $*"; env "$@" timeout 600 python bench.py --steps 128 --warmup 16 --no-traffic --no-full-run --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*, "unit": "tokens/s"\|"ms_per_step": [0-9.]*' | head -2 | tr '\n' ' '; echo; }
{
run A=1
run HIP_FORCE_DEV_KERNARG=1
run HIP_FORCE_DEV_KERNARG=0
run GPU_MAX_HW_QUEUES=1
run A=1
} > gpurun_out/env_knobs.log 2>&1
cat gpurun_out/env_knobs.log
