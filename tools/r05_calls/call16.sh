cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "decode_attention" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_engine.py -m gpu -x -q -k "matches_oracle or full_width" 2>&1 | tail -4
for ctx in 1040 1536; do
  timeout 200 python tools/bench_attention.py --ctx $ctx --splits 1 --layers 32 --iters 40 2>&1 | grep ctx=
done
timeout 200 python tools/bench_attention.py --ctx 1040 --bits 4 --splits 1 --layers 32 --iters 40 2>&1 | grep ctx=
for i in 1 2; do
  timeout 600 python bench.py --steps 256 --warmup 32 --no-cpu-baseline --no-traffic --no-full-run 2>&1 | tail -1 | tee gpurun_out/bench_attn_prologue_$i.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
done
timeout 600 python tools/fixed_cost_table.py --attn-detail > gpurun_out/fixed_cost_attn_prologue.txt 2>&1; grep -A2 "^attn\|layer wall" gpurun_out/fixed_cost_attn_prologue.txt
