#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd sqlite database: per-kernel count / total / mean / min duration (kernel-trace) and,
if present, mean counter values per kernel.  usage: rocpd_summary.py <results.db> [kernel substring]"""
import sqlite3
import sys


def main():
    by_grid = '--by-grid' in sys.argv   # one line per (kernel, grid): separates the four decode linears of one template
    if by_grid:
        sys.argv.remove('--by-grid')
    window_s = None
    if '--window-s' in sys.argv:   # device-busy analysis of the LAST x seconds of the trace (a timed region that ends the process)
        i = sys.argv.index('--window-s')
        window_s = float(sys.argv[i + 1])
        del sys.argv[i:i + 2]
    overlap = None
    if '--overlap' in sys.argv:    # how much of the kernels named *x* ran WHILE other kernels ran (side-stream concurrency evidence)
        i = sys.argv.index('--overlap')
        overlap = sys.argv[i + 1]
        del sys.argv[i:i + 2]
    db = sqlite3.connect(sys.argv[1])
    if overlap is not None:
        cur = db.cursor()
        mine = sorted(cur.execute("select start, end from kernels where name like ?", (f'%{overlap}%',)).fetchall())
        rest = sorted(cur.execute("select start, end, name from kernels where name not like ?", (f'%{overlap}%',)).fetchall())
        # union of the other kernels' intervals
        uni = []
        for s_, e_, _ in rest:
            if uni and s_ <= uni[-1][1]:
                uni[-1][1] = max(uni[-1][1], e_)
            else:
                uni.append([s_, e_])
        import bisect
        starts = [u[0] for u in uni]
        tot = cov = 0
        for s_, e_ in mine:
            tot += e_ - s_
            j = max(0, bisect.bisect_right(starts, s_) - 1)
            while j < len(uni) and uni[j][0] < e_:
                cov += max(0, min(e_, uni[j][1]) - max(s_, uni[j][0]))
                j += 1
        by = {}
        k = 0
        for s_, e_, n_ in rest:       # which kernels ran under them (by time)
            while k < len(mine) and mine[k][1] <= s_:
                k += 1
            j = k
            while j < len(mine) and mine[j][0] < e_:
                ov = min(e_, mine[j][1]) - max(s_, mine[j][0])
                if ov > 0:
                    by[n_] = by.get(n_, 0) + ov
                j += 1
        print(f'kernels named *{overlap}*: {len(mine)} launches, {tot / 1e6:.2f} ms in total; {cov / 1e6:.2f} ms = {100.0 * cov / max(tot, 1):.1f} % of that time '
              f'at least one OTHER kernel was running on the device (union of {len(rest)} other launches)')
        print('other kernels by the time they spent under them:')
        for n_, v in sorted(by.items(), key=lambda kv: -kv[1])[:12]:
            print(f'  {n_[:90]:90s} {v / 1e6:9.2f} ms')
        return
    if window_s is not None:
        cur = db.cursor()
        t_end = cur.execute("select max(end) from kernels").fetchone()[0]
        t_lo = t_end - int(window_s * 1e9)
        iv = sorted(cur.execute("select start, end from kernels where start >= ?", (t_lo,)).fetchall())
        busy, cur_s, cur_e, gaps = 0, None, None, []
        for s_, e_ in iv:
            if cur_e is None or s_ > cur_e:
                if cur_e is not None:
                    busy += cur_e - cur_s
                    gaps.append(s_ - cur_e)
                cur_s, cur_e = s_, e_
            else:
                cur_e = max(cur_e, e_)
        if cur_e is not None:
            busy += cur_e - cur_s
        span = (iv[-1][1] - iv[0][0]) if iv else 0
        big = sorted(gaps, reverse=True)
        print(f'window: last {window_s:.3f} s of the trace: {len(iv)} kernels, span {span / 1e6:.1f} ms, device busy (union of kernel intervals) '
              f'{busy / 1e6:.1f} ms = {100.0 * busy / max(span, 1):.1f} %, idle {(span - busy) / 1e6:.1f} ms')
        for lo, hi in ((0, 2e3), (2e3, 1e4), (1e4, 1e5), (1e5, 1e6), (1e6, 1e12)):
            sel = [g for g in gaps if lo <= g < hi]
            print(f'  gaps {lo / 1e3:8.0f} .. {hi / 1e3:10.0f} us: {len(sel):7d}, {sum(sel) / 1e6:9.2f} ms')
        print('  largest gaps (us): ' + ', '.join(f'{g / 1e3:.0f}' for g in big[:12]))
        rows = list(cur.execute("select name, grid_x/workgroup_x*grid_y/workgroup_y*grid_z/workgroup_z, count(*), sum(end-start), avg(end-start) from kernels "
                                "where start >= ? group by 1, 2 order by 4 desc", (t_lo,)))
        print(f'{"kernel":60s} {"wgs":>8s} {"calls":>7s} {"total_ms":>10s} {"avg_us":>9s}')
        for r in rows[:45]:
            print(f'{r[0][:60]:60s} {int(r[1]):8d} {r[2]:7d} {r[3] / 1e6:10.3f} {r[4] / 1e3:9.2f}')
        return
    like = f'%{sys.argv[2]}%' if len(sys.argv) > 2 else '%'
    cur = db.cursor()
    if by_grid:
        rows = list(cur.execute("select name, grid_x/workgroup_x, grid_y/workgroup_y, grid_z/workgroup_z, count(*), sum(end-start), "
                                "avg(end-start), min(end-start) from kernels where name like ? group by 1, 2, 3, 4 order by 6 desc", (like,)))
        print(f'{"kernel":60s} {"grid":>14s} {"calls":>7s} {"total_ms":>10s} {"avg_us":>9s} {"min_us":>9s}')
        for r in rows[:40]:
            print(f'{r[0][:60]:60s} {int(r[1]):5d}x{int(r[2]):3d}x{int(r[3]):3d} {r[4]:7d} {r[5]/1e6:10.3f} {r[6]/1e3:9.2f} {r[7]/1e3:9.2f}')
        return
    rows = list(cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(vgpr_count), "
                            "max(lds_size), max(grid_x*1.0/workgroup_x) from kernels where name like ? group by name "
                            "order by 3 desc", (like,)))
    tot = sum(r[2] for r in rows) or 1
    print(f'{"kernel":72s} {"calls":>7s} {"total_ms":>10s} {"%":>6s} {"avg_us":>9s} {"min_us":>9s} {"vgpr":>5s} {"lds":>7s}')
    for r in rows[:40]:
        print(f'{r[0][:72]:72s} {r[1]:7d} {r[2]/1e6:10.3f} {100*r[2]/tot:6.2f} {r[3]/1e3:9.2f} {r[4]/1e3:9.2f} {r[5] or 0:5d} {r[6] or 0:7d}')
    try:
        crow = list(cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection "
                                "where kernel_name like ? group by kernel_name, counter_name order by 1, 2", (like,)))
    except sqlite3.Error:
        crow = []
    if crow:
        print('\ncounters (mean per dispatch):')
        for r in crow:
            print(f'{r[0][:72]:72s} {r[1]:28s} {r[2]:16.1f} n={r[3]}')
    # the same per (kernel, workgroups): separates the launches of one template (the four decode linears, tuner candidates) --
    # VERDICT r03 item 1a.  The counters view carries the dispatch's total grid / workgroup sizes under version-dependent names.
    try:
        cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    except sqlite3.Error:
        cols = []
    gcol = next((c for c in ('grid_size', 'grid_size_x', 'grid_x') if c in cols), None)
    wcol = next((c for c in ('workgroup_size', 'workgroup_size_x', 'workgroup_x') if c in cols), None)
    if crow and gcol and wcol:
        rows = list(cur.execute(f"select kernel_name, cast({gcol} / {wcol} as int), counter_name, avg(value), count(*) from counters_collection "
                                f"where kernel_name like ? group by 1, 2, 3 order by 1, 2, 3", (like,)))
        print(f'\ncounters by (kernel, workgroups = {gcol} / {wcol}) (mean per dispatch):')
        for r in rows:
            print(f'{r[0][:60]:60s} wgs={r[1]:<6d} {r[2]:28s} {r[3]:16.1f} n={r[4]}')
    elif crow:
        print('\n(no grid columns in counters_collection; columns: ' + ', '.join(cols) + ')')


if __name__ == '__main__':
    main()
