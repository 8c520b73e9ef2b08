#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
{
for arm in 1 3 1 3; do
echo "== bench TM_D32_WT=$arm"; TM_D32_WT=$arm timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-full-run --profile-steps 0 --tune 0 2>/dev/null | grep '"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
} > gpurun_out/call31.log 2>&1
cat gpurun_out/call31.log
