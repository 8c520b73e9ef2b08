#!/bin/bash
for spec in "gate_up 16,1,1" "gate_up 8,1,1" "down 16,1,8" "down 16,1,4" "down 8,1,8" "qkv 16,1,4" "qkv 16,1,2" "qkv 8,1,4" "o 16,1,4" "o 16,1,2" "o 8,1,4"; do set -- $spec
TM_GEMM_KSTAGE=1 python tools/tune_gemm.py --only $1 --cfg $2 2>&1 | grep "^$1 " | sed 's/  */ /g' | cut -d' ' -f1,5-
done
