#!/bin/bash
# round 2: native P2P all-reduce + TP-on-one-GPU tests (repeated: the protocol is timing dependent)
mkdir -p gpurun_out
export TMPDIR=/tmp
for i in 1 2 3; do
timeout 900 python -m pytest tests/test_gpu_p2p.py -q -m gpu > gpurun_out/p2p_tests_$i.log 2>&1; echo "p2p rc=$?" >> gpurun_out/p2p_tests_$i.log
tail -4 gpurun_out/p2p_tests_$i.log
done
grep -h "AssertionError" gpurun_out/p2p_tests_*.log | head
for i in 1 2; do
timeout 900 python -m pytest tests/test_gpu_tp.py -q -m gpu -rs > gpurun_out/tp_tests_$i.log 2>&1; echo "tp rc=$?" >> gpurun_out/tp_tests_$i.log
tail -4 gpurun_out/tp_tests_$i.log
done
