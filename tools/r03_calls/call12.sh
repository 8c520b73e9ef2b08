#!/bin/bash
# round 3, call 12: final state -- whole GPU suite, smoke(), the driver's bench command
mkdir -p gpurun_out/r03
timeout 1800 python -m pytest tests -q -m gpu -x > gpurun_out/r03/c12_all_gpu_tests.log 2>&1
echo "all gpu tests rc=$?"; tail -4 gpurun_out/r03/c12_all_gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r03/c12_bench_driver_cmd.json 2>/dev/null
python - <<PY
import json
d=json.loads(open('gpurun_out/r03/c12_bench_driver_cmd.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['step_roofline']['frac'], d['roofline']['frac'], d['roofline']['us_per_launch'], d['value_full_run'], d['ttft_p50_ms'], d['cpu_baseline']['value'], d['config']['gemm_tilings'])
PY
