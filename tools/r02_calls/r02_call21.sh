#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/call21
mkdir -p $OUT
CMD="python $R/bench.py --steps 20 --warmup 5 --profile-steps 0 --no-cpu-baseline --no-traffic --no-full-run"
for arm in 1 0; do
  TM_D32_ROWHALF=$arm rocprofv3 --kernel-trace --stats -d $OUT/trace$arm -o trace -- $CMD > $OUT/trace$arm.log 2>&1
  python $R/tools/rocpd_summary.py $OUT/trace$arm/trace_results.db --by-grid > $OUT/by_grid_rowhalf$arm.txt 2>&1
  grep '"metric"' $OUT/trace$arm.log | cut -c1-200 >> $OUT/by_grid_rowhalf$arm.txt
  rm -rf $OUT/trace$arm
done
cd $R
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | grep -E "passed|failed|FAILED|Error" | head -5 > $OUT/tests.txt
head -16 $OUT/by_grid_rowhalf1.txt $OUT/by_grid_rowhalf0.txt; cat $OUT/tests.txt
