#!/bin/bash
# round 3, call 25: PMC passes on the prefill attention kernel (separate runs, counters only), then its parity tests
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03
mkdir -p $OUT
PMC="python $R/bench.py --steps 2 --warmup 1 --tune 0 --profile-steps 0 --no-cpu-baseline --no-traffic --no-full-run"
pmc() { name=$1; shift
  timeout 400 rocprofv3 --pmc "$@" -d $OUT/p25_$name -o pmc -- $PMC > $OUT/c25_pmc_$name.log 2>&1
  python $R/tools/rocpd_summary.py $OUT/p25_$name/pmc_results.db prefill_attention > $OUT/c25_pmc_${name}_prefill_attention.txt 2>&1
  rm -rf $OUT/p25_$name
  cat $OUT/c25_pmc_${name}_prefill_attention.txt | cut -c60-200
}
pmc sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES
pmc lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA
cd $R; timeout 300 python -m pytest tests/test_gpu_ops.py -q -m gpu -x -k "prefill_attention" 2>&1 | tail -1
