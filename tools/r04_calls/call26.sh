#!/bin/bash
# round 4, call 26: fused decode attention prologue with the first cache block issued BEFORE the q / k / v / RoPE loads (two
# serialised round trips instead of three): parity, then A/B on the driver-sized decode step (old / new library, alternating)
cp build/ab/new.so lmdeploy_amd/lib/libtm_mi355x.so
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "decode_attention" 2>&1 | grep -E "passed|failed|error" | tail -3
R=$GRAFT_REPO_ROOT
B="python $R/bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-traffic --no-full-run"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["kernel_ms_per_step"]["attention"], d["roofline"]["us_per_launch"], d["roofline"]["frac"])'
for v in old new old new; do
  cp build/ab/$v.so lmdeploy_amd/lib/libtm_mi355x.so
  echo -n "$v: "; timeout 400 $B 2>/dev/null | python -c "$P"
done
