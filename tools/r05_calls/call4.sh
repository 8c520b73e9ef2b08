#!/bin/bash
# round 5, call 4: new parity tests (threads, pipeline(tp=2) in one call, fold variants, XCD-aware split-K placement), A/B of the fold modes
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/call4
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_tp.py -x -q -k "single_call" -m gpu 2>&1 | tail -15
timeout 1200 python -m pytest tests/test_gpu_engine.py tests/test_gpu_ops.py tests/test_gpu_fullsize.py -x -q -k "two_threads or matches_oracle or folded_norm or decode_attention or w4a16" -m gpu 2>&1 | tail -8
for f in 3 1 2 0; do
  TM_FOLD_NORM=$f timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic > $O/bench_driver_fold$f.json 2> $O/bench_fold$f.err
  echo fold $f: $(grep -o '"value": [0-9.]*' $O/bench_driver_fold$f.json | head -1) $(grep -o '"value_1k_out": [0-9.]*' $O/bench_driver_fold$f.json) $(grep -o '"ms_per_step": [0-9.]*' $O/bench_driver_fold$f.json | head -1)
done
TM_D32_XCD=0 timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver_fold3_noxcd.json 2> $O/bench_fold3_noxcd.err
echo fold 3 no-xcd: $(grep -o '"value": [0-9.]*' $O/bench_driver_fold3_noxcd.json | head -1) $(grep -o '"value_1k_out": [0-9.]*' $O/bench_driver_fold3_noxcd.json) $(grep -o '"roofline_gemm_traffic": {[^}]*}' $O/bench_driver_fold3_noxcd.json)
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver_fold3_xcd.json 2> $O/bench_fold3_xcd.err
echo fold 3 xcd: $(grep -o '"value": [0-9.]*' $O/bench_driver_fold3_xcd.json | head -1) $(grep -o '"value_1k_out": [0-9.]*' $O/bench_driver_fold3_xcd.json) $(grep -o '"roofline_gemm_traffic": {[^}]*}' $O/bench_driver_fold3_xcd.json)
TM_FOLD_NORM=3 timeout 500 python tools/fixed_cost_table.py > $O/fixed_cost_fold3.txt 2> $O/fixed_cost_fold3.err
cut -c1-250 $O/fixed_cost_fold3.txt
TM_FOLD_NORM=1 timeout 500 python tools/fixed_cost_table.py > $O/fixed_cost_fold1.txt 2> $O/fixed_cost_fold1.err
cut -c1-250 $O/fixed_cost_fold1.txt
