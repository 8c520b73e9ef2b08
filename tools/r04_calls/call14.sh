#!/bin/bash
# round 4, call 14: decode attention: fresh phase trace at the bench's context, and the memory-pipeline counters of the launch
# (TA / TCP / TCC: what DESIGN r03 section 8 listed as "not yet looked at")
mkdir -p gpurun_out/r04
timeout 200 python tools/bench_attention.py --ctx 1088 --layers 32 --splits 1 --iters 20 --trace 2>&1 | grep -v amdgpu.ids | tail -30
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o -E "\b(TA_[A-Z_0-9]+|TCP_[A-Z_0-9]+|TCC_[A-Z_0-9]+|TD_[A-Z_0-9]+)\b" | sort -u | tr '\n' ' ' | head -c 6000 > $GRAFT_REPO_ROOT/gpurun_out/r04/c14_counters.txt
echo; echo "counters listed: $(wc -w < $GRAFT_REPO_ROOT/gpurun_out/r04/c14_counters.txt)"
