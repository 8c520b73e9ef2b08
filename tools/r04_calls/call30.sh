#!/bin/bash
# round 4, call 30: (a) parity of the two-phase scheduler step (issue / retire overlap) and of the measured dispatch of the general
# kernel / grouped GEMMs; (b) request stream A/B of TM_ASYNC_STEP on one engine; (c) kernel trace of the request stream: device busy
# vs idle inside the timed region (where do 4.9 s go?)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/call30
mkdir -p $O
cd $R
timeout 400 python -m pytest tests/test_gpu_engine.py -x -q -k "continuous or thread_serving or logits_processors or pipeline_continuous or tuning_roundtrip or moe" 2>&1 | tail -6
timeout 200 python -m pytest tests/test_gpu_tp.py -x -q -k "mixed" 2>&1 | tail -3
timeout 300 python tools/bench_continuous.py --modes 0,1,0,1 --export-table $O/table.txt > $O/cb_ab.json 2> $O/cb_ab.err
cat $O/cb_ab.json
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $O/t_cb -o trace -- python $R/tools/bench_continuous.py --modes 1 --import-table $O/table.txt > $O/cb_traced.json 2> $O/cb_traced.err
cat $O/cb_traced.json
W=$(python -c "import json;print(json.load(open('$O/cb_traced.json'))['wall_s'])")
python $R/tools/rocpd_summary.py --window-s $W $O/t_cb/trace_results.db > $O/cb_busy.txt 2>&1
rm -rf $O/t_cb
head -40 $O/cb_busy.txt
tail -3 $O/cb_ab.err $O/cb_traced.err | cut -c1-300
