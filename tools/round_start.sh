#!/bin/bash
# First GPU call of a round: regression + the design probe + a fresh default bench line, everything under gpurun_out/.
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/round_start.sh r02'
TAG=${1:-rNN}
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/start_$TAG
mkdir -p $OUT
timeout 300 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
grep -E "passed|failed|error" $OUT/pytest_gpu.log | tail -2
if [ -x tools/probes/grid_sync_probe ]; then timeout 60 tools/probes/grid_sync_probe > $OUT/grid_sync_probe.txt 2>&1; cat $OUT/grid_sync_probe.txt; fi
timeout 300 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python - <<PY
import json
d = json.loads(open("$OUT/bench_default.json").read().strip().split("\n")[-1])
print("bench", d["value"], "tok/s", d["ms_per_step"], "ms/step", "ttft", d["ttft_p50_ms"], "roofline", d["roofline"]["frac"])
print(d["kernel_ms_per_step"])
PY
