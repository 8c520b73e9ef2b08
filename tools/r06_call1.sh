#!/bin/bash
# Round 6, call 1: (a) parity of the merged split-K tiles / shape 10; (b) tuner verbose (every candidate's us per layer, merged ones included);
# (c) same-box A/B on the driver command: default dispatch vs w1w3 forced to the 256-column tile with a 2-way cross-CU k-split merged in the launch.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06_call1
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "split_k_merged or folded_norm or loader_consumer_tiles" > $OUT/pytest_merge.txt 2>&1
tail -3 $OUT/pytest_merge.txt
timeout 600 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "every_tuner_candidate and 64" > $OUT/pytest_cands.txt 2>&1
tail -3 $OUT/pytest_cands.txt
B="python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic"
TM_GEMM_TUNE_VERBOSE=1 timeout 400 $B > $OUT/bench_default.json 2> $OUT/tune_verbose.txt
grep "tm tune" $OUT/tune_verbose.txt | grep -E "M=64" > $OUT/tune_m64.txt
# forced arms: dispatch table lines K N M shape splits role
for arm in "26 2" "16 2" "17 2" "10 1" "0 1"; do
  set -- $arm
  T=/tmp/tab_$1_$2.txt
  echo "4096 28672 64 $1 $2 3" > $T
  TM_GEMM_IMPORT=$T timeout 400 $B > $OUT/bench_w1w3_$1x$2.json 2>/dev/null
done
timeout 400 $B > $OUT/bench_default_again.json 2>/dev/null
python - <<PY
import json,glob
for f in sorted(glob.glob('$OUT/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['value'], d.get('ms_per_step'), d.get('value_1k_out'), d['config']['gemm_tilings'])
    except Exception as e: print(f, 'no json', e)
PY
cat $OUT/tune_m64.txt | grep -E "w1w3|->" | head -60
echo "4096 28672 64 26 2 3" > /tmp/tab_m.txt
cd $R && TM_GEMM_IMPORT=/tmp/tab_m.txt timeout 500 python tools/fixed_cost_table.py > $OUT/fixed_cost_w1w3_merged_26x2.txt 2> $OUT/fc1.err
cd $R && timeout 500 python tools/fixed_cost_table.py > $OUT/fixed_cost_default.txt 2> $OUT/fc2.err
grep -E "w1w3|layer" $OUT/fixed_cost_w1w3_merged_26x2.txt | head; grep -E "w1w3|layer" $OUT/fixed_cost_default.txt | head
