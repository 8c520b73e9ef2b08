#!/bin/bash
# decode attention split-KV count at batch 64 (finer work items dealt by the hardware dispatcher: VERDICT r05 item 5), same box, interleaved
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06_splits
mkdir -p $OUT
cd $R
for rep in 1 2; do
for s in 1 2 4; do
  timeout 400 python bench.py --decode-splits $s --no-cpu-baseline --no-traffic --profile-steps 0 > $OUT/bench_splits${s}_$rep.json 2>/dev/null
done
done
python - <<PY
import json,glob
for f in sorted(glob.glob('$OUT/bench_splits*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['value'], d['ms_per_step'], d.get('value_1k_out'), d['config']['decode_splits'])
PY
