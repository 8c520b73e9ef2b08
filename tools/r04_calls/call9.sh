#!/bin/bash
# round 4, call 9: the whole GPU suite after removing the vendor-library path and the in-launch consumer; smoke; launcher tests (2 and 8 ranks on one GPU)
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -8
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
