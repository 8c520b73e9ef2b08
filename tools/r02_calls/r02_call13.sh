#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r02l
mkdir -p $OUT
timeout 120 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "fp8_quant_rows" > $OUT/pytest_q.log 2>&1; tail -15 $OUT/pytest_q.log
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "fp8_mfma_linear" --maxfail=3 > $OUT/pytest_l.log 2>&1; tail -15 $OUT/pytest_l.log
timeout 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_engine.py -m gpu -q -k "moe" --maxfail=6 > $OUT/pytest_m.log 2>&1; tail -15 $OUT/pytest_m.log
