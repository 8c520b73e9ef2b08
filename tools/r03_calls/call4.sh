#!/bin/bash
# round 3, call 4: engine / communicator tests after the single-image + p2p changes, then the driver's bench command
mkdir -p gpurun_out/r03
timeout 1200 python -m pytest tests/test_gpu_engine.py tests/test_gpu_tp.py tests/test_gpu_p2p.py tests/test_gpu_checkpoint.py -q -m gpu -x > gpurun_out/r03/c4_tests.log 2>&1
echo "tests rc=$?"; tail -6 gpurun_out/r03/c4_tests.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r03/c4_bench_driver_cmd.json 2> gpurun_out/r03/c4_bench_driver_cmd.err
echo "bench rc=$?"; tail -c 3000 gpurun_out/r03/c4_bench_driver_cmd.json; tail -5 gpurun_out/r03/c4_bench_driver_cmd.err
