#!/bin/bash
# round 4, call 18: request-stream run under rocprofv3 --kernel-trace: GPU-busy time vs wall (how much is host gap?)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace -d /tmp/cb -o trace -- python $R/tools/bench_continuous.py --tune 0 > /tmp/cb.log 2>&1
grep '"metric"' /tmp/cb.log | cut -c1-400
python - <<'PY'
import sqlite3, glob
db = sqlite3.connect(glob.glob('/tmp/cb/**/trace_results.db', recursive=True)[0])
rows = list(db.execute("select start, end, name from kernels order by start"))
# the timed session = everything after the largest idle gap following warm-up; approximate: last 60 % of launches' busy union
import itertools
busy = 0; cur_s, cur_e = rows[0][0], rows[0][1]
for s, e, _ in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = rows[-1][1] - rows[0][0]
print('kernels', len(rows), 'span_s', span / 1e9, 'busy_s', busy / 1e9)
# gaps histogram over the last 5 s
t_end = rows[-1][1]; gaps = []
prev_e = None
for s, e, n in rows:
    if e < t_end - 5.2e9: prev_e = max(prev_e or e, e); continue
    if prev_e is not None and s > prev_e: gaps.append((s - prev_e, n))
    prev_e = max(prev_e or e, e)
import collections
tot = sum(g for g, _ in gaps)
print('last 5.2 s: gap total', tot / 1e9, 's in', len(gaps), 'gaps; > 50 us:', sum(1 for g, _ in gaps if g > 50e3), 'sum', sum(g for g, _ in gaps if g > 50e3) / 1e9)
by = collections.Counter()
for g, n in gaps:
    if g > 20e3: by[n[:60]] += g
for n, g in by.most_common(8): print(f'{g / 1e6:9.1f} ms of gaps > 20 us in front of {n}')
PY
