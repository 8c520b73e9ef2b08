#!/bin/bash
# round 3, call 14: the final tree -- whole GPU suite, smoke(), driver command, one other config through the prefill-class tuner
mkdir -p gpurun_out/r03
timeout 1800 python -m pytest tests -q -m gpu -x > gpurun_out/r03/c14_all_gpu_tests.log 2>&1
echo "all gpu tests rc=$?"; tail -4 gpurun_out/r03/c14_all_gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03/c14_bench_driver_cmd.json 2>/dev/null
timeout 600 python bench.py --model internlm2_20b --batch 128 --steps 32 --no-cpu-baseline --no-full-run --no-traffic > gpurun_out/r03/c14_bench_config2.json 2>/dev/null
python - <<PY
import json
for f in ('c14_bench_driver_cmd','c14_bench_config2'):
    d=json.loads(open('gpurun_out/r03/%s.json'%f).read().strip().splitlines()[-1])
    print(f, d['value'], d['ms_per_step'], d['step_roofline']['frac'], d['ttft_p50_ms'], d['config']['gemm_tilings'])
PY
