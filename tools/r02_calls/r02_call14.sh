#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r02m
mkdir -p $OUT
for v in 1 0; do
TM_FP8_MFMA=$v timeout 900 python bench.py --model mixtral_8x7b --steps 64 --no-cpu-baseline --no-traffic --no-full-run > $OUT/bench_mixtral_mfma$v.json 2> $OUT/bench_mixtral_mfma$v.err
tail -1 $OUT/bench_mixtral_mfma$v.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('mfma=$v', d['value'], 'tok/s', d['ms_per_step'], 'ms/step', 'step roofline', d['step_roofline']['frac'], {k:round(v,3) for k,v in d['kernel_ms_per_step'].items() if v>0.05}, d['sample_tokens'])"
tail -2 $OUT/bench_mixtral_mfma$v.err
done
