#!/bin/bash
# round 4, call 25: split-KV count of the decode attention at one rank's head count of TP = 8 (the heuristic fills 512 workgroups:
# 8 splits for the 8B shard, 4 for the 70B shard) -- sweep, tuner on, one box
R=$GRAFT_REPO_ROOT
B="python $R/bench.py --emulate-tp 8 --no-cpu-baseline --no-traffic --no-full-run --steps 128"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["kernel_ms_per_step"]["attention"], d["config"]["decode_splits"])'
for sp in 8 2 4 16 8; do
  echo -n "8B tp8 rank, splits=$sp: "; timeout 400 $B --decode-splits $sp 2>/dev/null | python -c "$P"
done
for sp in 4 2 8 4; do
  echo -n "70B tp8 rank int4 KV, splits=$sp: "; timeout 500 $B --model llama3_70b --quant-policy 4 --decode-splits $sp 2>/dev/null | python -c "$P"
done
