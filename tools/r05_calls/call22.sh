cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/step_fused
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
for rep in 1 2; do
for f in 0 1; do
  TM_STEP_FUSED=$f timeout 600 python bench.py --steps 256 --warmup 32 --no-cpu-baseline --no-traffic --no-full-run 2>&1 | tail -1 | tee gpurun_out/step_fused/bench_fused${f}_$rep.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fused $f', d['value'], d['ms_per_step'], d['kernel_ms_per_step']['sample'], d['kernel_ms_per_step']['embed'])"
done
done
for f in 0 1; do
  TM_STEP_FUSED=$f timeout 600 python bench.py --emulate-tp 8 --steps 128 --no-cpu-baseline --no-traffic --no-full-run 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('tp8-emu fused $f', d['value'], d['ms_per_step'])"
done
