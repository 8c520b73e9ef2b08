#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/call26
mkdir -p $OUT
CMD="python $R/bench.py --steps 20 --warmup 5 --profile-steps 0 --no-cpu-baseline --no-traffic --no-full-run"
for arm in 1 0; do
  TM_NORM_SU8=$arm rocprofv3 --kernel-trace --stats -d $OUT/trace$arm -o trace -- $CMD > $OUT/trace$arm.log 2>&1
  python $R/tools/rocpd_summary.py $OUT/trace$arm/trace_results.db --by-grid > $OUT/by_grid_su8_$arm.txt 2>&1
  grep '"metric"' $OUT/trace$arm.log | cut -c1-200 >> $OUT/by_grid_su8_$arm.txt
  rm -rf $OUT/trace$arm
done
for arm in 1 0 1 0; do
echo "== bench TM_NORM_SU8=$arm"; TM_NORM_SU8=$arm timeout 400 python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-full-run --profile-steps 0 2>&1 | grep '"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done > $OUT/bench.txt 2>&1
grep -E "rmsnorm|dec32|decode_attention|metric" $OUT/by_grid_su8_1.txt $OUT/by_grid_su8_0.txt | cut -c1-230; cat $OUT/bench.txt
