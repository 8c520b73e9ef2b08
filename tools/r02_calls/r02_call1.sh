#!/bin/bash
# Round 2, first GPU call: the new decode GEMM (parity first, then speed), the full-size parity tests, the whole GPU suite,
# a default bench line and a kernel trace.  Everything under gpurun_out/r02a/.
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/r02_call1.sh'
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r02a
mkdir -p $OUT
echo "== decode-kernel parity (operator tests that route M <= 64 through gemm_decode.hip)"
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "w4a16 or tp_sharded" --maxfail=10 > $OUT/pytest_gemm.log 2>&1
tail -3 $OUT/pytest_gemm.log
echo "== full-size parity"
timeout 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -q --maxfail=12 > $OUT/pytest_fullsize.log 2>&1
tail -5 $OUT/pytest_fullsize.log
echo "== GEMM micro-benchmark"
timeout 300 python tools/bench_gemm.py --variants old,d0,d0pf4,d1,d2,d3 --splits 0 --json $OUT/bench_gemm.json > $OUT/bench_gemm.log 2>&1
cat $OUT/bench_gemm.log
timeout 200 python tools/bench_gemm.py --variants abl1,abl2,abl4,abl8,abl16,abl7,abl15,abl31 --only gate_up,down --json $OUT/bench_gemm_abl.json > $OUT/bench_gemm_abl.log 2>&1
cat $OUT/bench_gemm_abl.log
timeout 200 python tools/bench_gemm.py --variants d0,d2 --splits 1,2,4,8 --only qkv,o,down --json $OUT/bench_gemm_splits.json > $OUT/bench_gemm_splits.log 2>&1
cat $OUT/bench_gemm_splits.log
echo "== whole GPU suite"
timeout 600 python -m pytest tests -m gpu -q --maxfail=10 --deselect tests/test_gpu_fullsize.py > $OUT/pytest_gpu.log 2>&1
tail -5 $OUT/pytest_gpu.log
echo "== bench"
timeout 300 python bench.py --steps 128 > $OUT/bench_128.json 2> $OUT/bench_128.err
tail -1 $OUT/bench_128.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('bench', d['value'], 'tok/s', d['ms_per_step'], 'ms/step ttft', d['ttft_p50_ms'], 'roofline', d['roofline']['frac'], 'step', d['step_roofline']['frac'])
print(d['kernel_ms_per_step'])"
echo "== kernel trace"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$OUT/trace -o trace -- python $R/bench.py --steps 64 --warmup 8 --profile-steps 4 --no-cpu-baseline > $R/$OUT/trace.log 2>&1
python $R/tools/rocpd_summary.py $R/$OUT/trace/trace_results.db > $R/$OUT/kernel_trace_stats.txt 2>&1
rm -rf $R/$OUT/trace
head -24 $R/$OUT/kernel_trace_stats.txt
