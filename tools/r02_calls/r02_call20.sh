#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
{
echo "== bench (driver command, rowhalf on)"; timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-full-run 2>&1 | grep '"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernel_ms_per_step'], d['roofline']['us_per_launch'], d['sample_tokens'])"
echo "== bench (driver command, rowhalf off)"; TM_D32_ROWHALF=0 timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-full-run 2>&1 | grep '"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernel_ms_per_step'], d['roofline']['us_per_launch'], d['sample_tokens'])"
echo "== gate_up shape 6"; timeout 200 python tools/trace_boundary.py 4096 28672 64 1 6 1 2>&1 | tail -1
echo "== engine tests"; timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -3
} > gpurun_out/call20.log 2>&1
cat gpurun_out/call20.log
