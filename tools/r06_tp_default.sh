#!/bin/bash
# Round 6: the native fused all-reduce + residual + RMSNorm as the DEFAULT decode collective at tp > 1 (after its bring-up
# self-test) -- parity tests + the one-rank emulation lines, fused (default) against RCCL + the residual-norm launch (TM_COMM=rccl), same box.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r06_tp_default
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_engine.py -x -q -m gpu -k "collective_path or side_stream" > $OUT/pytest_collective.txt 2>&1; tail -3 $OUT/pytest_collective.txt
timeout 1200 python -m pytest tests/test_gpu_tp.py tests/test_gpu_p2p.py -x -q -m gpu > $OUT/pytest_tp.txt 2>&1; tail -3 $OUT/pytest_tp.txt
B="python $R/bench.py --emulate-tp 8 --steps 128 --no-cpu-baseline --no-traffic --no-full-run"
timeout 600 $B > $OUT/bench_line_llama3_8b_tp8_rank_emulation.json 2>/dev/null
TM_COMM=rccl timeout 600 $B > $OUT/bench_line_llama3_8b_tp8_rank_emulation_rccl_plus_norm_launch.json 2>/dev/null
timeout 600 $B > $OUT/bench_line_llama3_8b_tp8_rank_emulation_again.json 2>/dev/null
B2="python $R/bench.py --emulate-tp 2 --steps 128 --no-cpu-baseline --no-traffic --no-full-run"
timeout 600 $B2 > $OUT/bench_line_llama3_8b_tp2_rank_emulation.json 2>/dev/null
TM_COMM=rccl timeout 600 $B2 > $OUT/bench_line_llama3_8b_tp2_rank_emulation_rccl_plus_norm_launch.json 2>/dev/null
python - <<PY
import json,glob
for f in sorted(glob.glob('$OUT/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); k=d.get('kernel_ms_per_step',{})
        print(f.split('/')[-1], d['value'], d.get('ms_per_step'), d['config'].get('collectives'), 'res_norm', k.get('residual_norm'), 'allreduce', k.get('allreduce'), 'ttft', d.get('ttft_p50_ms'))
    except Exception as e: print(f, 'no json', e)
PY
