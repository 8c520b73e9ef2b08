#!/bin/bash
# round 5, call 7: (a) does hipExtAnyOrderLaunch let a dependent kernel start before its predecessor ends on gfx950 (tools/probes/any_order_probe.hip)?
# (b) eager vs graph-replayed decode steps (the any-order form would need eager launches); (c) parity of the final kernel edits
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/call7
mkdir -p $O
cd $R/tools/probes && timeout 120 hipcc --offload-arch=gfx950 -O2 -o any_order_probe any_order_probe.hip && timeout 60 ./any_order_probe | tee $O/any_order_probe.txt | cut -c1-330
cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_engine.py -x -q -k "folded_norm or matches_oracle or w4a16_linear or continuous_batching_matches_static" -m gpu 2>&1 | tail -3
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-full-run > $O/bench_graph.json 2> $O/bench_graph.err
echo graph: $(grep -o '"value": [0-9.]*' $O/bench_graph.json | head -1) $(grep -o '"ms_per_step": [0-9.]*' $O/bench_graph.json | head -1)
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-full-run --no-graph > $O/bench_eager.json 2> $O/bench_eager.err
echo eager: $(grep -o '"value": [0-9.]*' $O/bench_eager.json | head -1) $(grep -o '"ms_per_step": [0-9.]*' $O/bench_eager.json | head -1)
