# same-box A/B, int4 KV: library of commit eff45b9 against the current tree
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab_attn
for rep in 1 2; do
for which in old new; do
  cp gpurun_ab/libtm_$which.so lmdeploy_amd/lib/libtm_mi355x.so
  echo "== $which (rep $rep)"
  timeout 200 python tools/bench_attention.py --ctx 1040 --bits 4 --splits 1 --layers 32 --iters 40 2>&1 | grep ctx=
  timeout 200 python tools/bench_attention.py --ctx 2048 --bits 4 --splits 1 --layers 32 --iters 40 2>&1 | grep ctx=
  timeout 600 python bench.py --quant-policy 4 --steps 256 --warmup 32 --no-cpu-baseline --no-traffic --no-full-run 2>&1 | tail -1 | tee gpurun_out/ab_attn/bench_int4_${which}_$rep.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['us_per_launch'])"
done
done
