#!/bin/bash
# Round-5 evidence for the tensor-parallel prefill overlap on the one-GPU box: emulated-exchange bench lines of one rank of
# TP = 8 / TP = 2 (serial, row halves, two micro-batches; 150 and 300 GB/s), the kernel-trace overlap summary, and the GPU test of both
# schedules.  Summaries -> gpurun_out/overlap/ (copied by hand into profiles/r05_prefill_overlap*).  ~3 GPU-minutes.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/overlap
mkdir -p $OUT
cd $R
timeout 400 python -m pytest tests/test_gpu_engine.py -x -q -k "side_stream or collective_path" 2>&1 | grep -E "passed|failed|error" | tee $OUT/pytest_tail.txt
B="python $R/bench.py"
run() {  # tp gbps side_stream microbatch
  TM_EMULATE_AR_GBPS=$2 TM_COMM_STREAM=$3 TM_PIPE_MICROBATCH=$4 timeout 300 $B --emulate-tp $1 --steps 32 --warmup 4 --tune 0 --profile-steps 0 \
    --no-cpu-baseline --no-traffic --no-full-run 2>/dev/null | tail -1 > $OUT/bench_tp$1_g$2_s$3_mb$4.json
  python - <<PY
import json
d = json.load(open('$OUT/bench_tp$1_g$2_s$3_mb$4.json'))
o = d['config'].get('prefill_allreduce_overlap', {})
print('tp$1 gbps $2 side $3 mb $4: prefill_total_s', d['prefill_total_s'], 'ttft_p50_ms', d['ttft_p50_ms'], 'tok/s', d['value'],
      {k: o.get(k) for k in ('side_stream', 'overlapped_forwards', 'microbatch_forwards', 'side_stream_allreduces')})
PY
}
for tp in 8 2; do
  run $tp 0 0 1; run $tp 0 1 1
  for g in 150 300; do run $tp $g 0 1; run $tp $g 1 0; run $tp $g 1 1; done
done
cd /tmp
CMD="$B --emulate-tp 8 --steps 8 --warmup 2 --tune 0 --profile-steps 0 --no-cpu-baseline --no-traffic --no-full-run"
for s in 1 0; do
  TM_EMULATE_AR_GBPS=150 TM_COMM_STREAM=$s timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/t_$s -o trace -- $CMD > $OUT/trace_$s.log 2>&1
  python $R/tools/rocpd_summary.py --overlap exchange_standin $OUT/t_$s/trace_results.db | tee $OUT/prefill_overlap_kernel_trace_side$s.txt
  rm -rf $OUT/t_$s $OUT/trace_$s.log
done
