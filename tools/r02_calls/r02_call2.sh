#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r02b
mkdir -p $OUT
echo "== failed tests again"
timeout 300 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -k "linearity or ragged" --maxfail=12 > $OUT/pytest_fullsize.log 2>&1; tail -3 $OUT/pytest_fullsize.log
timeout 300 python -m pytest tests/test_gpu_engine.py -m gpu -q --maxfail=12 > $OUT/pytest_engine.log 2>&1; tail -3 $OUT/pytest_engine.log
echo "== traces"
for spec in "4096 28672 64 1 0 1" "4096 28672 64 1 0 1 31" "4096 28672 64 1 0 1 15" "4096 28672 64 1 0 1 7" "14336 4096 64 0 0 7" "4096 4096 64 0 0 4" "4096 4096 64 0 0 4 31"; do
  echo "-- $spec"; timeout 120 python tools/trace_dec32.py $spec 2>&1 | grep -v amdgpu.ids | tail -4
done > $OUT/trace_dec32.log 2>&1
cat $OUT/trace_dec32.log
echo "== ablations"
timeout 300 python tools/bench_gemm.py --variants d0,abl32,abl95,abl31,abl64,abl15,abl7,abl8,abl4,abl2,abl1 --only gate_up,o > $OUT/bench_gemm_abl.log 2>&1
cat $OUT/bench_gemm_abl.log
