#!/bin/bash
# round 5, call 2: RMSNorm folded into the decode GEMMs -- operator + engine parity, fixed-cost table and driver line, A/B on one box
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/call2
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -k "folded_norm or w4a16_gated or w4a16_linear" -m gpu 2>&1 | tail -15
timeout 900 python -m pytest tests/test_gpu_engine.py -x -q -s -k "matches_oracle or collective_path or tuning_roundtrip or full_width or continuous_batching_matches_static" -m gpu 2>&1 | grep -v "^$" | tail -30
for f in 1 0; do
  TM_FOLD_NORM=$f timeout 500 python tools/fixed_cost_table.py > $O/fixed_cost_fold$f.txt 2> $O/fixed_cost_fold$f.err
  cut -c1-250 $O/fixed_cost_fold$f.txt
  TM_FOLD_NORM=$f timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic > $O/bench_driver_fold$f.json 2> $O/bench_fold$f.err
  cut -c1-330 $O/bench_driver_fold$f.json; grep -o '"value_1k_out": [0-9.]*' $O/bench_driver_fold$f.json
done
grep "tm tune.*->" $O/bench_fold1.err | cut -c1-200
