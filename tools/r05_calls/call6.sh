#!/bin/bash
# round 5, call 6: parity of the last kernel edit (slab loads of the last arriver hoisted), then the round's evidence (tools/r05_profiles.sh)
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_engine.py tests/test_gpu_fullsize.py -x -q -k "folded_norm or matches_oracle or full_width or w4a16_linear" -m gpu 2>&1 | tail -4
bash tools/r05_profiles.sh
