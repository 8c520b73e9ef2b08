#!/bin/bash
# round 5, call 3: full GPU parity suite on the folded-norm tree, fixed-cost table with the attention straggler analysis, driver lines
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/call3
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -12
TM_GEMM_TUNE_VERBOSE=1 timeout 500 python tools/fixed_cost_table.py --attn-detail > $O/fixed_cost_fold1.txt 2> $O/fixed_cost_fold1.err
cut -c1-250 $O/fixed_cost_fold1.txt
grep "tm tune" $O/fixed_cost_fold1.err | grep -E "wo  |w2  |->" | cut -c1-130
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic > $O/bench_driver_fold1.json 2> $O/bench_fold1.err
cut -c1-330 $O/bench_driver_fold1.json; grep -o '"value_1k_out": [0-9.]*' $O/bench_driver_fold1.json
TM_D32_WT=3 timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic > $O/bench_driver_fold1_wt3.json 2> $O/bench_fold1_wt3.err
cut -c1-330 $O/bench_driver_fold1_wt3.json; grep -o '"value_1k_out": [0-9.]*' $O/bench_driver_fold1_wt3.json
TM_FOLD_NORM=0 timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic > $O/bench_driver_fold0.json 2> $O/bench_fold0.err
cut -c1-330 $O/bench_driver_fold0.json; grep -o '"value_1k_out": [0-9.]*' $O/bench_driver_fold0.json
