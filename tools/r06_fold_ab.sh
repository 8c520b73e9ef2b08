#!/bin/bash
# Round 6: is the folded RMSNorm (TM_FOLD_NORM=3, a default-on departure from the reference's rounding sequence) worth more than box noise?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06_fold_ab
mkdir -p $OUT
cd $R
B="python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic"
for i in 1 2 3 4; do
  TM_FOLD_NORM=3 timeout 400 $B > $OUT/bench_fold3_$i.json 2>/dev/null
  TM_FOLD_NORM=0 timeout 400 $B > $OUT/bench_fold0_$i.json 2>/dev/null
done
python - <<PY
import json,glob,statistics as st
for arm in ('fold3','fold0'):
    v=[];k=[]
    for f in sorted(glob.glob('$OUT/bench_%s_*.json'%arm)):
        d=json.loads(open(f).read().strip().splitlines()[-1]); v.append(d['value']); k.append(d['value_1k_out'])
    print(arm, 'driver command tok/s', v, 'mean', round(st.mean(v),1), '| value_1k_out', k, 'mean', round(st.mean(k),1))
PY
