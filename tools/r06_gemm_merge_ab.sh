#!/bin/bash
# Round 6: the 256-column decode tile with a 2-way cross-CU k-split merged inside the launch, for w1w3 -- same-box A/B
# on the driver command against the measured dispatch, the tuner's view of every candidate, and the fixed-cost table of both.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06_merge_ab
mkdir -p $OUT
cd $R
B="python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline"
TM_GEMM_TUNE_VERBOSE=1 timeout 500 $B > $OUT/bench_line_driver_command_default.json 2> $OUT/tune_verbose.txt
grep "tm tune" $OUT/tune_verbose.txt | grep -E "M=64" > $OUT/tuner_candidates_m64.txt
for arm in "26 2" "16 2" "10 1"; do
  set -- $arm
  T=/tmp/tab_$1_$2.txt
  echo "4096 28672 64 $1 $2 3" > $T
  TM_GEMM_IMPORT=$T timeout 400 $B --no-traffic > $OUT/bench_line_driver_command_w1w3_shape$1_x$2.json 2>/dev/null
done
timeout 400 $B --no-traffic > $OUT/bench_line_driver_command_default_again.json 2>/dev/null
echo "4096 28672 64 26 2 3" > /tmp/tab_m.txt
TM_GEMM_IMPORT=/tmp/tab_m.txt timeout 500 python tools/fixed_cost_table.py > $OUT/fixed_cost_w1w3_256col_2way_merged.txt 2> $OUT/fc1.err
timeout 500 python tools/fixed_cost_table.py --attn-detail > $OUT/fixed_cost_default.txt 2> $OUT/fc2.err
rm -f $OUT/tune_verbose.txt $OUT/fc1.err $OUT/fc2.err
python - <<PY
import json,glob
for f in sorted(glob.glob('$OUT/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['value'], d.get('ms_per_step'), d.get('value_1k_out'), d['config']['gemm_tilings']['w1w3'], d.get('ttft_p50_ms'))
        r=d.get('roofline',{}); print('   roofline:', r.get('kernel','')[:50], r.get('frac'), r.get('us_per_launch_group'), r.get('us_per_launch'), r.get('traffic'), '| attn', d.get('attention_roofline',{}).get('frac'), d.get('attention_roofline',{}).get('us_per_launch'))
    except Exception as e: print(f, 'no json', e)
PY
grep -E "w1w3" $OUT/tuner_candidates_m64.txt | head -30
grep -E "^w_qkv|^attn|^wo|^w1w3|^w2|layer wall" $OUT/fixed_cost_w1w3_256col_2way_merged.txt | head -8; grep -E "^w_qkv|^attn|^wo|^w1w3|^w2|layer wall" $OUT/fixed_cost_default.txt | head -8
