#!/bin/bash
# round 5, call 8: 256 x 256 prefill tile with THREE activation buffers (x DMA two stages ahead, counted vmcnt across the barrier): parity
# under TM_PRE256_XBUF=3, then time at M = 8192 against the two-buffer form and the 128 x 512 tile, one process per arm (the switch is read once)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/call8
mkdir -p $O
cd $R
TM_PRE256_XBUF=3 timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py -x -q -k "prefill or every_tuner_candidate or odd_stage" -m gpu 2>&1 | tail -4
for xb in 2 3 2 3; do
  echo "== TM_PRE256_XBUF=$xb"
  TM_PRE256_XBUF=$xb timeout 300 python tools/bench_gemm.py --m 8192 --reps 12 --variants p256 --splits 1 2>&1 | grep "TF/s" | tee -a $O/prefill_xbuf$xb.txt
done
echo "== d5"
timeout 300 python tools/bench_gemm.py --m 8192 --reps 12 --variants d5 --splits 1 2>&1 | grep "TF/s" | tee $O/prefill_d5.txt
