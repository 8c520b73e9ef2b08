#!/bin/bash
# round 3, call 32 (last GPU seconds): derived memory-unit counters on the decode attention kernel
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03
mkdir -p $OUT
timeout 75 rocprofv3 --pmc MemUnitBusy MemUnitStalled L2CacheHit -d $OUT/p32 -o pmc -- python $R/bench.py --steps 6 --warmup 1 --profile-steps 0 --no-cpu-baseline --no-traffic --no-full-run --no-graph --tune 0 > $OUT/c32_pmc.log 2>&1
python $R/tools/rocpd_summary.py $OUT/p32/pmc_results.db decode_attention > $OUT/c32_pmc_mem_attention.txt 2>&1
python $R/tools/rocpd_summary.py $OUT/p32/pmc_results.db gemm_dec32 > $OUT/c32_pmc_mem_gemm.txt 2>&1
rm -rf $OUT/p32
tail -5 $OUT/c32_pmc_mem_attention.txt | cut -c60-180; tail -3 $OUT/c32_pmc.log | cut -c1-200
