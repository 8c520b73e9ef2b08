#!/bin/bash
# rocprofv3 passes over one GEMM configuration: kernel trace + two PMC passes (never combined with tracing domains)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_gemm
mkdir -p $OUT
ARGS="${@:-4096 28672 64 1 1 1 8 5}"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python $R/tools/run_gemm_once.py $ARGS > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/pmc1 -o pmc1 -- python $R/tools/run_gemm_once.py $ARGS > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_VMEM -d $OUT/pmc2 -o pmc2 -- python $R/tools/run_gemm_once.py $ARGS > $OUT/pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $OUT/pmc3 -o pmc3 -- python $R/tools/run_gemm_once.py $ARGS > $OUT/pmc3.log 2>&1
find $OUT -name "*.csv" | head -20
