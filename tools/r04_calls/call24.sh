#!/bin/bash
# round 4, call 24: split-K merge of the decode attention inside the launch (last-arriver tickets): parity (op level + engine),
# then one rank of TP = 8 (8B and 70B shards, tuner on) with and without it
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "decode_attention" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_engine.py -x -q 2>&1 | tail -4
R=$GRAFT_REPO_ROOT
B="python $R/bench.py --emulate-tp 8 --no-cpu-baseline --no-traffic --no-full-run --steps 128"
for m in 1 0 1 0; do
  echo -n "8B tp8 rank, TM_ATTN_MERGE=$m: "; TM_ATTN_MERGE=$m timeout 400 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['kernel_ms_per_step']['attention'], d['config']['decode_splits'], d['config']['gemm_tilings'])"
done
for m in 1 0; do
  echo -n "70B tp8 rank int4 KV, TM_ATTN_MERGE=$m: "; TM_ATTN_MERGE=$m timeout 500 $B --model llama3_70b --quant-policy 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['kernel_ms_per_step']['attention'], d['config']['decode_splits'])"
done
