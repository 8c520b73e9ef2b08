#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r02h
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py -m gpu -q -k "w4a16 or ragged" --maxfail=8 > $OUT/pytest.log 2>&1; grep -E "passed|failed" $OUT/pytest.log | tail -1; grep -E "^FAILED" $OUT/pytest.log | head
for m in 8192 1024 256 128; do
  timeout 200 python tools/bench_gemm.py --m $m --reps 8 --variants oldp,d4 2>&1 | grep -v amdgpu.ids
done | tee $OUT/bench_prefill.log
timeout 600 python bench.py --steps 64 > $OUT/bench_64.json 2> $OUT/bench_64.err
tail -1 $OUT/bench_64.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('bench', d['value'], 'tok/s', d['ms_per_step'], 'ms/step ttft', d['ttft_p50_ms'], 'prefill tok/s', d['prefill_tokens_per_s'])
print('roofline', d['roofline'])
print('full', d.get('value_full_run'))
print('cpu', d.get('cpu_baseline'))"
tail -3 $OUT/bench_64.err
