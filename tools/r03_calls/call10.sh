#!/bin/bash
# round 3, call 10: bench.py --gpus 2 on the ONE-GPU box: launcher, RCCL bring-up failure (duplicate GPU) -> native communicator,
# rank-0 tuning table broadcast, watchdog, comm_info in the line
mkdir -p gpurun_out/r03
export GPU_MAX_HW_QUEUES=8
timeout 900 python bench.py --gpus 2 --steps 16 --warmup 4 --layers 8 --no-traffic --no-cpu-baseline --no-full-run --profile-steps 0 > gpurun_out/r03/c10_bench_gpus2.json 2> gpurun_out/r03/c10_bench_gpus2.err
echo "rc=$?"; tail -c 2500 gpurun_out/r03/c10_bench_gpus2.json; echo; grep -v amdgpu.ids gpurun_out/r03/c10_bench_gpus2.err | tail -20
