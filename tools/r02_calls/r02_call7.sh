#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r02g
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q --maxfail=10 > $OUT/pytest_gpu.log 2>&1; grep -E "passed|failed" $OUT/pytest_gpu.log | tail -2; grep -E "^FAILED" $OUT/pytest_gpu.log | head
/usr/bin/time -v timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
grep -E "Elapsed|Maximum resident" $OUT/bench_default.err
tail -1 $OUT/bench_default.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('bench', d['value'], 'tok/s', d['ms_per_step'], 'ms/step ttft', d['ttft_p50_ms'])
print('roofline', d['roofline'])
print('full', d.get('value_full_run'))
print('cpu', d.get('cpu_baseline'))"
timeout 300 python tools/bench_attention.py --trace > $OUT/attn_trace.log 2>&1; tail -12 $OUT/attn_trace.log
