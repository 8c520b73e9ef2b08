set -x
mkdir -p gpurun_out/r06_logprobs
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -m gpu -k "sampling" 2>&1 | tail -15 | tee gpurun_out/r06_logprobs/pytest_ops.txt
timeout 900 python -m pytest tests/test_gpu_engine.py -q -x -m gpu -k "logprobs or sampling or pipeline" 2>&1 | tail -15 | tee gpurun_out/r06_logprobs/pytest_engine.txt
