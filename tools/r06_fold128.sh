#!/bin/bash
# Round 6: folded RMSNorm for 64 < M <= 128 -- parity + BASELINE config 3 (InternLM2-20B, batch 128) A/B, same box
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06_fold128
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "folded_norm" > $OUT/pytest_fold.txt 2>&1; tail -3 $OUT/pytest_fold.txt
timeout 900 python -m pytest tests/test_gpu_engine.py -x -q -m gpu -k "folded_norm_batch_128 or matches_oracle" > $OUT/pytest_engine.txt 2>&1; tail -3 $OUT/pytest_engine.txt
B="python $R/bench.py --model internlm2_20b --batch 128 --steps 128 --no-cpu-baseline --no-full-run --no-traffic"
TM_GEMM_TUNE_VERBOSE=1 timeout 900 $B > $OUT/bench_line_config2_internlm2_20b_b128.json 2> $OUT/tune_fold128.txt
TM_FOLD_MAX_M=64 TM_GEMM_TUNE_VERBOSE=1 timeout 900 $B > $OUT/bench_line_config2_internlm2_20b_b128_fold_max_64.json 2> $OUT/tune_fold64.txt
timeout 900 $B > $OUT/bench_line_config2_internlm2_20b_b128_again.json 2>/dev/null
grep "tm tune" $OUT/tune_fold128.txt | grep "M=128" | grep -v "M=1280" > $OUT/tuner_m128_folded.txt
grep "tm tune" $OUT/tune_fold64.txt | grep "M=128" | grep "\->" > $OUT/tuner_m128_unfolded_winners.txt
rm -f $OUT/tune_fold128.txt $OUT/tune_fold64.txt
python - <<PY
import json,glob
for f in sorted(glob.glob('$OUT/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); k=d.get('kernel_ms_per_step',{})
        print(f.split('/')[-1], d['value'], d.get('ms_per_step'), d['step_roofline']['frac'], d['config'].get('rmsnorm_fold'), d['config']['gemm_tilings'], 'res_norm', k.get('residual_norm'))
    except Exception as e: print(f, 'no json', e)
PY
grep "\->" $OUT/tuner_m128_folded.txt; cat $OUT/tuner_m128_unfolded_winners.txt
