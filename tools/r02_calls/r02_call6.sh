#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r02f
mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_engine.py -m gpu -q -k "w4a16 or engine_matches or errors_are or moe" --maxfail=6 > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
run() { # name, env...
  name=$1; shift
  env "$@" timeout 200 python bench.py --steps 96 --no-cpu-baseline > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  tail -1 $OUT/bench_$name.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
k=d['kernel_ms_per_step']
print('$name', d['value'], 'tok/s', d['ms_per_step'], 'ms | qkv %.1f o %.1f gate_up %.1f down %.1f attn %.1f norm %.1f us/layer' % tuple(k[x]*1000/32 for x in ('gemm_qkv','gemm_o','gemm_gate_up','gemm_down','attention','residual_norm')), d['sample_tokens'])"
}
run default A=1
run old TM_GEMM_D32=0
run shape2 TM_D32_SHAPE=2
run shape3 TM_D32_SHAPE=3
run minkb8 TM_D32_MIN_KB=8
run regstage TM_D32_ABL=0
