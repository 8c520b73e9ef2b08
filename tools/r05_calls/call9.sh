#!/bin/bash
# round 5, call 9: the final tree -- full GPU parity suite, smoke(), driver-command and default bench lines
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/call9
mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -6
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_line_driver_command.json 2> $O/bench.err
cut -c1-330 $O/bench_line_driver_command.json; grep -o '"value_1k_out": [0-9.]*' $O/bench_line_driver_command.json
timeout 700 python bench.py > $O/bench_line_default.json 2> $O/bench_default.err
cut -c1-330 $O/bench_line_default.json; grep -o '"value_1k_out": [0-9.]*' $O/bench_line_default.json
