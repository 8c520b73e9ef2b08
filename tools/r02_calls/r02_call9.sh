#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r02i
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py tests/test_gpu_engine.py -m gpu -q -k "attention or engine_matches or other_config" --maxfail=8 > $OUT/pytest.log 2>&1; grep -E "passed|failed" $OUT/pytest.log | tail -1; grep -E "^FAILED" $OUT/pytest.log | head
for b in 4 8; do echo "== bits $b"; timeout 200 python tools/diag_attn2.py $b 2>&1 | grep -v amdgpu.ids | tail -4; done
timeout 200 python tools/bench_attention.py --ctx 1100 --splits 1 2>&1 | grep -v amdgpu.ids | tail -4
timeout 200 python tools/bench_attention.py --ctx 1100 --bits 4 --splits 1 2>&1 | grep -v amdgpu.ids | tail -4
