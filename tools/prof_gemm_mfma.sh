#!/bin/bash
# MFMA utilisation (SQ_VALU_MFMA_BUSY_CYCLES) of the W4A16 GEMM at M in {64, 128, 1024, 8192} on the gate_up shape
# (K 4096, N 28672, gated epilogue), the library's own dispatch (P32 kernels: decode tile / 128-row tiles / 128 x 512 prefill tile),
# N(0,1) activations, weights HBM-cold (512 MB flush between launches).  Own PMC pass (no tracing domains) + a kernel-trace pass
# for the durations.  usage: tools/prof_gemm_mfma.sh [tag]
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/mfma_$TAG
mkdir -p $OUT
: > $OUT/gemm_mfma_util.txt
for M in 64 128 1024 8192; do
  timeout 120 rocprofv3 --kernel-trace -d $OUT/t -o t -- python $R/tools/run_gemm_once.py 4096 28672 $M 1 0 0 0 12 > $OUT/log.txt 2>&1
  timeout 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/p -o p -- python $R/tools/run_gemm_once.py 4096 28672 $M 1 0 0 0 12 >> $OUT/log.txt 2>&1
  echo "== M=$M  (K=4096 N=28672 gated)" >> $OUT/gemm_mfma_util.txt
  python $R/tools/mfma_util.py $OUT/t/t_results.db $OUT/p/p_results.db 4096 28672 $M >> $OUT/gemm_mfma_util.txt 2>&1
  rm -rf $OUT/t $OUT/p
done
cat $OUT/gemm_mfma_util.txt
