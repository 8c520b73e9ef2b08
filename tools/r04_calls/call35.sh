#!/bin/bash
# round 4, call 35: p50 TTFT of 64 simultaneous 1024-token prompts against max_prefill_token_num (the reference's knob; 8192 = its default):
# sequences get their first token when THEIR chunk is done, so the median is quantised by the chunk size at an unchanged prefill rate
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/call35
mkdir -p $O
cd $R
for c in 8192 2048 1024; do
  timeout 150 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-full-run --max-prefill-tokens $c > $O/bench_chunk$c.json 2> $O/err$c.txt
  python -c "
import json; d=json.loads(open('$O/bench_chunk$c.json').read().strip().splitlines()[-1]); print('chunk', $c, 'ttft_p50_ms', d['ttft_p50_ms'], 'prefill_total_s', d['prefill_total_s'], 'prefill tok/s', d['prefill_tokens_per_s'], 'decode tok/s', d['value'], d['config']['prefill_gemm_tilings'])"
done
