#!/bin/bash
# round 4, call 31: the scheduler / dispatch tests that call 30 did not reach (-x stopped at a test bug: the table is process-global)
cd $GRAFT_REPO_ROOT
timeout 500 python -m pytest tests/test_gpu_engine.py -q -k "continuous or thread_serving or logits_processors or pipeline_continuous or tuning_roundtrip or moe" 2>&1 | tail -12
