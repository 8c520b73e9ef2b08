#!/bin/bash
# round 3, call 30: tuner with the 7 % bar: its tests, then the driver command and the int4 / fp16-KV lines again
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/profile_r03d
mkdir -p $OUT
(cd $R; timeout 400 python -m pytest tests/test_gpu_engine.py -q -m gpu -x -k "tuning_roundtrip or f16_library" 2>&1 | tail -1)
B="python $R/bench.py"
timeout 400 $B --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_line_driver_command.json 2>/dev/null
timeout 400 $B --quant-policy 0 --steps 128 --no-cpu-baseline --no-full-run > $OUT/bench_line_config1_fp16kv.json 2>/dev/null
timeout 400 $B --quant-policy 4 --steps 256 --no-cpu-baseline --no-full-run > $OUT/bench_line_llama3_8b_int4kv.json 2>/dev/null
python - <<PY
import json,glob
for f in sorted(glob.glob('$OUT/bench_line_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['value'], d.get('ms_per_step'), d.get('step_roofline',{}).get('frac'), d.get('ttft_p50_ms'), d['config'].get('gemm_tilings'))
    except Exception as e: print(f, 'no json', e)
PY
