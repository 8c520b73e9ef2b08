#!/bin/bash
# round 4, call 39: last sanity run of the final library build (tm_gemm_export added after the full-suite run of call 32): smoke(), the
# engine-vs-oracle tests, the tuner round trip
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 120 python -m pytest tests/test_gpu_engine.py -q -k "matches_oracle and not continuous or tuning_roundtrip" 2>&1 | tail -3
