cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/overlap2
timeout 400 python -m pytest tests/test_gpu_engine.py -x -q -k "side_stream or collective_path" 2>&1 | tail -15
run() {  # tp gbps side_stream microbatch
  TM_EMULATE_AR_GBPS=$2 TM_COMM_STREAM=$3 TM_PIPE_MICROBATCH=$4 timeout 300 python bench.py --emulate-tp $1 --steps 32 --warmup 4 --tune 0 --profile-steps 0 \
    --no-cpu-baseline --no-traffic --no-full-run 2>gpurun_out/overlap2/err_tp$1_g$2_s$3_mb$4.txt | tail -1 > gpurun_out/overlap2/bench_tp$1_g$2_s$3_mb$4.json
  python - <<PY
import json
d = json.load(open('gpurun_out/overlap2/bench_tp$1_g$2_s$3_mb$4.json'))
o = d['config'].get('prefill_allreduce_overlap', {})
print('tp$1 gbps $2 side $3 mb $4: prefill_total_s', d['prefill_total_s'], 'ttft_p50_ms', d['ttft_p50_ms'], 'tok/s', d['value'], {k: o.get(k) for k in ('side_stream', 'overlapped_forwards', 'microbatch_forwards', 'side_stream_allreduces')})
PY
}
run 8 0 0 1
run 8 0 1 1
run 8 150 0 1
run 8 150 1 0
run 8 150 1 1
run 2 150 0 1
run 2 150 1 0
run 2 150 1 1
