cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/overlap3
timeout 400 python -m pytest tests/test_gpu_engine.py -x -q -k "side_stream or collective_path" 2>&1 | tail -4
timeout 300 python -m pytest tests/test_gpu_tp.py -x -q -k "pipeline_tp2 or dense" 2>&1 | tail -3
run() {  # tp gbps side_stream
  TM_EMULATE_AR_GBPS=$2 TM_COMM_STREAM=$3 timeout 300 python bench.py --emulate-tp $1 --steps 32 --warmup 4 --tune 0 --profile-steps 0 \
    --no-cpu-baseline --no-traffic --no-full-run 2>/dev/null | tail -1 > gpurun_out/overlap3/bench_tp$1_g$2_s$3.json
  python - <<PY
import json
d = json.load(open('gpurun_out/overlap3/bench_tp$1_g$2_s$3.json'))
print('tp$1 gbps $2 side $3: prefill_total_s', d['prefill_total_s'], 'ttft_p50_ms', d['ttft_p50_ms'], 'tok/s', d['value'])
PY
}
run 8 300 0
run 8 300 1
run 2 300 0
run 2 300 1
run 2 0 0
