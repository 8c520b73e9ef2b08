#!/bin/bash
# collectives on the engine stream vs side stream (+ weight prefetch), per-rank emulation (1-rank RCCL communicator)
mkdir -p gpurun_out
export TMPDIR=/tmp
run() { echo "== $*"; env "$@" timeout 600 python bench.py --model llama3_70b --quant-policy 4 --emulate-tp 8 --steps 128 --warmup 16 --no-traffic --no-full-run --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*, "unit": "tokens/s"\|"ms_per_step": [0-9.]*\|"allreduce": [0-9.]*' | head -3 | tr '\n' ' '; echo; }
{
run TM_COMM_STREAM=1 TM_COMM_PREFETCH=1
run TM_COMM_STREAM=1 TM_COMM_PREFETCH=0
run TM_COMM_STREAM=0
run TM_COMM_STREAM=0 TM_GRAPH_COMM=0
run TM_COMM_STREAM=1 TM_COMM_PREFETCH=1
} > gpurun_out/comm_stream.log 2>&1
cat gpurun_out/comm_stream.log
