#!/bin/bash
# round 4, call 38: idle gaps INSIDE the captured decode step and BETWEEN graph replays (rocprofv3 --kernel-trace of 512 replayed decode steps,
# tools/rocpd_summary.py --window-s 1.5 = the last 1.5 s of the trace = ~440 steps): is there a per-replay gap worth a multi-step graph?
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/call38
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $O/t -o trace -- python $R/bench.py --steps 512 --warmup 16 --no-cpu-baseline --no-traffic --no-full-run --profile-steps 0 > $O/bench.json 2> $O/err.txt
python $R/tools/rocpd_summary.py --window-s 1.5 $O/t/trace_results.db > $O/decode_gaps.txt 2>&1
python - <<PY
import sqlite3
db=sqlite3.connect('$O/t/trace_results.db')
cur=db.cursor()
tend=cur.execute("select max(end) from kernels").fetchone()[0]
rows=cur.execute("select name,start,end from kernels where start >= ? order by start",(tend-int(1.5e9),)).fetchall()
# gap in front of every launch of the step's FIRST kernel (advance_kernel) = the gap between two graph replays
g=[rows[i][1]-rows[i-1][2] for i in range(1,len(rows)) if 'advance_kernel' in rows[i][0]]
import statistics
print('replay-to-replay gaps (before advance_kernel): n', len(g), 'mean us', round(statistics.mean(g)/1e3,2), 'median', round(statistics.median(g)/1e3,2), 'max', round(max(g)/1e3,1))
others=[rows[i][1]-rows[i-1][2] for i in range(1,len(rows)) if 'advance_kernel' not in rows[i][0]]
print('in-graph gaps: n', len(others), 'mean us', round(statistics.mean(others)/1e3,3), 'median', round(statistics.median(others)/1e3,3), 'sum per step us', round(sum(others)/max(len(g),1)/1e3,1))
PY
rm -rf $O/t
head -9 $O/decode_gaps.txt
grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' $O/bench.json | head -2
