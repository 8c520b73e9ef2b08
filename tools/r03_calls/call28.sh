#!/bin/bash
# round 3, call 28: the other BASELINE configurations with the final tree (their prefill goes through the library path too)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/profile_r03c
mkdir -p $OUT
B="python $R/bench.py"
timeout 600 $B --model internlm2_20b --batch 128 --steps 128 --no-cpu-baseline --no-full-run > $OUT/bench_line_config2_internlm2_20b_b128.json 2> $OUT/config2.err
tail -3 $OUT/config2.err | cut -c1-200
timeout 400 $B --quant-policy 0 --steps 128 --no-cpu-baseline --no-full-run > $OUT/bench_line_config1_fp16kv.json 2>/dev/null
timeout 400 $B --quant-policy 4 --steps 256 --no-cpu-baseline --no-full-run > $OUT/bench_line_llama3_8b_int4kv.json 2>/dev/null
python - <<PY
import json,glob
for f in sorted(glob.glob('$OUT/bench_line_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['value'], d.get('ms_per_step'), d.get('step_roofline',{}).get('frac'), d.get('ttft_p50_ms'), d.get('prefill_tokens_per_s'), d['config'].get('prefill_gemm_tilings'))
    except Exception as e: print(f, 'no json', e)
PY
