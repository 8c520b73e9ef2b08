#!/bin/bash
# Round 6: is the decode step power-bound?  Same binary, same kernels, same launches, same bytes: real synthetic weights vs ALL-ZERO linear weights
# (TM_SYNTH_WEIGHT_SCALE=0: the dequantised operand is 0, K/V are 0 -- the low-toggle arm), interleaved on one box.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06_power
mkdir -p $OUT
cd $R
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic"
for i in 1 2; do for v in 1 0; do TM_SYNTH_WEIGHT_SCALE=$v timeout 400 $B > $OUT/bench_wscale${v}_$i.json 2>/dev/null; done; done
python - <<PY
import json,glob
for f in sorted(glob.glob('$OUT/bench_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']
    print(f.split('/')[-1], 'tok/s', d['value'], 'ms/step', d['ms_per_step'], '1k-out', d['value_1k_out'], 'ttft', d['ttft_p50_ms'], 'eager ms: qkv %.3f attn %.3f o %.3f gate_up %.3f down %.3f head %.3f' % (k['gemm_qkv'],k['attention'],k['gemm_o'],k['gemm_gate_up'],k['gemm_down'],k['lm_head']))
PY
