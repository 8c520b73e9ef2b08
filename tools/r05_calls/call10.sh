#!/bin/bash
# round 5, call 10: sums of squares as [m][tile] rows read with one 16-byte load per thread (the consumer's prologue issued eight 4-byte loads into
# a VMEM queue congested by the activation DMA: +0.4 us of "issue"): parity, fixed-cost table, driver line
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/call10
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_engine.py tests/test_gpu_fullsize.py tests/test_gpu_checkpoint.py -x -q -k "folded_norm or matches_oracle or full_width or tuning_roundtrip or checkpoint or other_config" -m gpu 2>&1 | tail -3
timeout 500 python tools/fixed_cost_table.py > $O/fixed_cost.txt 2> $O/fixed_cost.err
cut -c1-250 $O/fixed_cost.txt
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic > $O/bench_driver.json 2> $O/bench.err
echo new: $(grep -o '"value": [0-9.]*' $O/bench_driver.json | head -1) $(grep -o '"value_1k_out": [0-9.]*' $O/bench_driver.json)
TM_FOLD_NORM=0 timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic > $O/bench_driver_fold0.json 2> $O/bench0.err
echo fold0: $(grep -o '"value": [0-9.]*' $O/bench_driver_fold0.json | head -1) $(grep -o '"value_1k_out": [0-9.]*' $O/bench_driver_fold0.json)
