# same-box A/B: library of commit eff45b9 (one-head decode attention as of round 4 + round-5 call 12) against the current tree
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ab_attn
for rep in 1 2; do
for which in old new; do
  cp gpurun_ab/libtm_$which.so lmdeploy_amd/lib/libtm_mi355x.so
  echo "== $which (rep $rep)"
  timeout 200 python tools/bench_attention.py --ctx 1040 --splits 1 --layers 32 --iters 40 2>&1 | grep ctx=
  timeout 200 python tools/bench_attention.py --ctx 2048 --splits 1 --layers 32 --iters 40 2>&1 | grep ctx=
  timeout 600 python bench.py --steps 256 --warmup 32 --no-cpu-baseline --no-traffic --no-full-run 2>&1 | tail -1 | tee gpurun_out/ab_attn/bench_${which}_$rep.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
done
done
for which in old new; do
  cp gpurun_ab/libtm_$which.so lmdeploy_amd/lib/libtm_mi355x.so
  timeout 600 python tools/fixed_cost_table.py --attn-detail > gpurun_out/ab_attn/fixed_cost_$which.txt 2>&1; grep -A1 "^attn\|layer wall" gpurun_out/ab_attn/fixed_cost_$which.txt
done
