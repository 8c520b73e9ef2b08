#!/bin/bash
# round 4, call 23: one rank of TP = 8 (Llama-3-8B and Llama-3-70B shards, emulated) with the start-up tuner ON (it was off in
# emulation): every candidate's time per layer incl. its consumer, the picks, the bench lines
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out
B="python $R/bench.py --emulate-tp 8 --no-cpu-baseline --no-traffic --no-full-run"
TM_GEMM_TUNE_VERBOSE=1 timeout 400 $B --steps 128 > $OUT/r04_call23_line_8b.json 2> $OUT/r04_call23_tune_8b.err
grep "tm tune" $OUT/r04_call23_tune_8b.err | grep "M=64" | cut -c1-200
cut -c1-1500 $OUT/r04_call23_line_8b.json
TM_GEMM_TUNE_VERBOSE=1 timeout 500 $B --model llama3_70b --quant-policy 4 --steps 128 > $OUT/r04_call23_line_70b.json 2> $OUT/r04_call23_tune_70b.err
grep "tm tune" $OUT/r04_call23_tune_70b.err | grep "M=64" | grep -e "->" | cut -c1-200
cut -c1-1500 $OUT/r04_call23_line_70b.json
