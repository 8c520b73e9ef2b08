#!/bin/bash
# round 4, call 36: the 128 x 512 prefill tile as a persistent workgroup per CU (TM_PRE64_PERSIST=1, gemm_prefill_persistent.hip):
# bit-identity against the plain kernel, then A/B at M = 8192 on the four Llama-3-8B shapes (one graph of 2 distinct weights each)
cd $GRAFT_REPO_ROOT
timeout 200 python -m pytest tests/test_gpu_ops.py -x -q -k "persistent_prefill" 2>&1 | tail -4
timeout 200 python tools/bench_gemm.py --m 8192 --variants d5,d5p,d5,d5p --splits 1 --reps 10 2>&1 | grep -v "^$\|amdgpu.ids"
