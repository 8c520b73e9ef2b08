#!/bin/bash
# final tree of round 6: the whole GPU suite, smoke, and the two bench lines the driver's record is compared with
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06_final3
mkdir -p $OUT
cd $R
timeout 2300 python -m pytest tests/ -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3 > $OUT/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 > $OUT/smoke.txt
timeout 700 python bench.py > $OUT/bench_line_default.json 2> $OUT/bench_default.err
timeout 400 python bench.py --steps 20 --warmup 5 > $OUT/bench_line_driver_command.json 2>/dev/null
cat $OUT/pytest_gpu.txt $OUT/smoke.txt
python - <<PY
import json
for f in ('bench_line_default','bench_line_driver_command'):
    d=json.loads(open('$OUT/'+f+'.json').read().strip().splitlines()[-1])
    print(f, d['value'], d['ms_per_step'], d['step_roofline']['frac'], d.get('value_1k_out'), d.get('ttft_p50_ms'), d['roofline']['frac'], d['attention_roofline']['frac'])
PY
