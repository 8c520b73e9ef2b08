cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "decode_attention" 2>&1 | tail -8
for pair in 0 2 0 2; do
  echo "== TM_ATTN_PAIR=$pair"
  for ctx in 1040 1536; do
    TM_ATTN_PAIR=$pair timeout 200 python tools/bench_attention.py --ctx $ctx --splits 1 --layers 32 --iters 40 2>&1 | grep ctx=
  done
done
echo "== int4"
for pair in 0 2; do
  TM_ATTN_PAIR=$pair timeout 200 python tools/bench_attention.py --ctx 1040 --bits 4 --splits 1 --layers 32 --iters 40 2>&1 | grep ctx=
done
for pair in 0 1 0 1; do
  echo "== bench TM_ATTN_PAIR=$pair"
  TM_ATTN_PAIR=$pair timeout 600 python bench.py --steps 256 --warmup 32 --no-cpu-baseline --no-traffic --no-full-run 2>&1 | tail -1 | tee gpurun_out/bench_pair$pair.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d.get('value_1k_out'))"
done
TM_ATTN_PAIR=1 timeout 600 python tools/fixed_cost_table.py --attn-detail > gpurun_out/fixed_cost_pair.txt 2>&1; tail -40 gpurun_out/fixed_cost_pair.txt
