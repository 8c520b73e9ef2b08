#!/bin/bash
# Round 6: L2 prefetch of the decode GEMMs' weight stream one / two stages beyond the register ring (TM_D32_L2PF) -- parity + same-box A/B
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06_l2pf
mkdir -p $OUT
cd $R
TM_D32_L2PF=1 timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "w4a16_linear or folded_norm or split_k_merged or gated_silu" > $OUT/pytest_l2pf1.txt 2>&1; tail -2 $OUT/pytest_l2pf1.txt
TM_D32_L2PF=2 timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "folded_norm" > $OUT/pytest_l2pf2.txt 2>&1; tail -2 $OUT/pytest_l2pf2.txt
B="python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic"
for v in 0 1 2 0 1; do
  TM_D32_L2PF=$v TM_GEMM_TUNE_VERBOSE=1 timeout 400 $B > $OUT/bench_l2pf${v}_$RANDOM.json 2> $OUT/tune_$v.txt
done
grep "tm tune" $OUT/tune_0.txt | grep "\->" > $OUT/tuner_winners_l2pf0.txt; grep "tm tune" $OUT/tune_1.txt | grep "\->" > $OUT/tuner_winners_l2pf1.txt; grep "tm tune" $OUT/tune_2.txt | grep "\->" > $OUT/tuner_winners_l2pf2.txt
rm -f $OUT/tune_?.txt
TM_D32_L2PF=1 timeout 500 python tools/fixed_cost_table.py > $OUT/fixed_cost_l2pf1.txt 2>/dev/null
TM_D32_L2PF=0 timeout 500 python tools/fixed_cost_table.py > $OUT/fixed_cost_l2pf0.txt 2>/dev/null
python - <<PY
import json,glob
for f in sorted(glob.glob('$OUT/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['value'], d.get('ms_per_step'), d.get('value_1k_out'), d['config']['gemm_tilings'])
    except Exception as e: print(f, 'no json', e)
PY
cat $OUT/tuner_winners_l2pf0.txt $OUT/tuner_winners_l2pf1.txt $OUT/tuner_winners_l2pf2.txt | cut -c1-150
for v in 0 1; do grep -E "^w_qkv|^attn|^wo|^w1w3|^w2|layer wall" $OUT/fixed_cost_l2pf$v.txt | head -8; done
