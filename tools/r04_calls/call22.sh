#!/bin/bash
# round 4, call 22: one rank of a TP = 8 Llama-3-8B job (emulated): tuner picks with the single-burst tiles among the candidates,
# bench line, rocprofv3 kernel trace by grid (in-graph per-kernel time at the shard shapes)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
B="python $R/bench.py --emulate-tp 8 --no-cpu-baseline --no-traffic --no-full-run"
TM_GEMM_TUNE_VERBOSE=1 timeout 400 $B --steps 128 > $OUT/r04_call22_line.json 2> $OUT/r04_call22_tune.err
grep "tm tune" $OUT/r04_call22_tune.err | grep "M=64" | cut -c1-200
cut -c1-1800 $OUT/r04_call22_line.json
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/t8 -o trace -- $B --steps 64 --warmup 8 --profile-steps 4 > /tmp/t8.log 2>&1
python $R/tools/rocpd_summary.py --by-grid /tmp/t8/trace_results.db > $OUT/r04_call22_trace_by_grid_tp8.txt 2>&1
head -40 $OUT/r04_call22_trace_by_grid_tp8.txt
