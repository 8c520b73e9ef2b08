#!/bin/bash
# true kernel durations (rocprofv3 --kernel-trace) of one GEMM configuration under the ablation switches
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_gemm2
mkdir -p $OUT
for abl in 0 15 31; do
  for spec in "4096 28672 64 1 1 1 8" "4096 4096 64 0 1 4 8"; do
    TM_GEMM_ABL=$abl TM_GEMM_KSTAGE=4 rocprofv3 --kernel-trace -d $OUT/t -o t -- python $R/tools/run_gemm_once.py $spec 20 > $OUT/log.txt 2>&1
    echo "ABL=$abl spec=[$spec]"; python $R/tools/rocpd_summary.py $OUT/t/t_results.db gemm_kernel | grep gemm_kernel | cut -c1-130
    python $R/tools/rocpd_summary.py $OUT/t/t_results.db splitk | grep splitk | cut -c1-130
    rm -rf $OUT/t
  done
done
