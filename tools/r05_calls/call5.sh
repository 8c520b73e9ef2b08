#!/bin/bash
# round 5, call 5: the split engine + ADVICE fixes + single-call tp = 2: full GPU suite; driver line; fixed-cost table of the final defaults
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/call5
mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -12
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_line_driver_command.json 2> $O/bench.err
cut -c1-400 $O/bench_line_driver_command.json; grep -o '"value_1k_out": [0-9.]*\|"roofline": {[^}]*}\|"roofline_gemm_traffic": {[^}]*}' $O/bench_line_driver_command.json
TM_FOLD_NORM=0 timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic > $O/bench_line_driver_command_fold0.json 2> $O/bench0.err
echo fold 0: $(grep -o '"value": [0-9.]*' $O/bench_line_driver_command_fold0.json | head -1) $(grep -o '"value_1k_out": [0-9.]*' $O/bench_line_driver_command_fold0.json)
timeout 500 python tools/fixed_cost_table.py --attn-detail > $O/fixed_cost_by_launch_final.txt 2> $O/fixed_cost.err
cut -c1-250 $O/fixed_cost_by_launch_final.txt
