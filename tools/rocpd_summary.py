#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd sqlite database: per-kernel count / total / mean / min duration (kernel-trace) and,
if present, mean counter values per kernel.  usage: rocpd_summary.py <results.db> [kernel substring]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    like = f'%{sys.argv[2]}%' if len(sys.argv) > 2 else '%'
    cur = db.cursor()
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


if __name__ == '__main__':
    main()
