cd $GRAFT_REPO_ROOT
for i in 1 2; do
timeout 600 python bench.py --steps 256 --warmup 32 --no-cpu-baseline --no-traffic --no-full-run 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'])"
done
AMD_LOG_LEVEL=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-full-run 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'])"
