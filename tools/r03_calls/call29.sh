#!/bin/bash
# round 3, call 29: decode A/B on ONE box: the tree before the library-path work (build/old_tree = commit f3d22cc) vs the current tree,
# heuristic tilings in both (--tune 0), alternating
mkdir -p gpurun_out/r03
R=$GRAFT_REPO_ROOT
for i in 1 2; do for t in old new; do
  if [ $t = old ]; then d=$R/build/old_tree; else d=$R; fi
  (cd $d && timeout 300 python bench.py --steps 64 --warmup 8 --tune 0 --no-cpu-baseline --no-traffic --no-full-run --profile-steps 4 > $R/gpurun_out/r03/c29_${t}_$i.json 2>/dev/null)
  python - <<PY
import json
d = json.loads(open('$R/gpurun_out/r03/c29_${t}_$i.json').read().strip().splitlines()[-1])
k = d.get('kernel_ms_per_step', {})
print('$t $i', d['value'], d['ms_per_step'], 'ttft', d['ttft_p50_ms'], {n: k[n] for n in ('gemm_qkv', 'attention', 'gemm_o', 'residual_norm', 'gemm_gate_up', 'gemm_down', 'lm_head') if n in k})
PY
done; done
