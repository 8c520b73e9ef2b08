#!/bin/bash
# round 3, call 8: non-temporal KV loads in the decode attention kernel (A/B by rebuilding one object on the box)
mkdir -p gpurun_out/r03
run() { for ctx in 1040 1536; do timeout 200 python tools/bench_attention.py --ctx $ctx --splits 1 --layers 4 --iters 40 2>&1 | grep -v amdgpu; done; }
{
echo "== plain loads"; run
touch lmdeploy_amd/csrc/attention_decode_mfma.hip
make -C lmdeploy_amd/csrc CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wall -Wno-unused-function -DTM_ATTN_NT" > /dev/null 2>&1
echo "== nontemporal loads"; run; run
touch lmdeploy_amd/csrc/attention_decode_mfma.hip
make -C lmdeploy_amd/csrc > /dev/null 2>&1
echo "== plain loads again"; run
} > gpurun_out/r03/c8_attention_nt.txt 2>&1
cat gpurun_out/r03/c8_attention_nt.txt
