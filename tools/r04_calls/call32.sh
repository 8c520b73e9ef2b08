#!/bin/bash
# round 4, call 32: the whole GPU suite on the final tree (two-phase scheduler step, measured dispatch of the general kernel / grouped
# GEMMs, pruned ablation instantiations), smoke(), and the driver's bench command
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/call32
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -8
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_line_driver_command.json 2> $O/bench_driver.err
python - <<PY
import json
d=json.loads(open('$O/bench_line_driver_command.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['step_roofline']['frac'], d.get('value_1k_out'), d['ttft_p50_ms'], d['roofline']['frac'], d['roofline']['traffic'], d['gemm_roofline']['frac'], d['config']['gemm_tilings'], d['config'].get('lm_head_tiling'), d['kernel_ms_per_step'])
PY
tail -3 $O/bench_driver.err | cut -c1-300
