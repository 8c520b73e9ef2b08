#!/usr/bin/env python
"""MFMA utilisation of the GEMM kernels in a rocprofv3 run: SQ_VALU_MFMA_BUSY_CYCLES (summed over the 1024 SIMDs, PMC
database) / (1024 * kernel duration * 2.4 GHz) with the duration from the kernel-trace database -- a lower bound, the
chip clocks 2.0-2.2 GHz under this load (GRBM_GUI_ACTIVE is aggregated over several instances and only printed).
usage: mfma_util.py <trace.db> <pmc.db> K N M"""
import sqlite3
import sys


def main():
    tdb, pdb = sqlite3.connect(sys.argv[1]), sqlite3.connect(sys.argv[2])
    K, N, M = map(int, sys.argv[3:6])
    flop = 2.0 * M * K * N
    dur = {r[0]: (r[1], r[2]) for r in tdb.execute(
        "select name, count(*), avg(end-start) from kernels where name like '%gemm%' group by name")}
    ctr = {}
    for name, cname, val in pdb.execute("select kernel_name, counter_name, avg(value) from counters_collection "
                                        "where kernel_name like '%gemm%' group by kernel_name, counter_name"):
        ctr.setdefault(name, {})[cname] = val
    for name, (calls, ns) in dur.items():
        c = ctr.get(name, {})
        busy, act = c.get('SQ_VALU_MFMA_BUSY_CYCLES'), c.get('GRBM_GUI_ACTIVE')
        util = busy / (1024.0 * ns * 2.4) if busy else float('nan')
        print(f'{name[:70]:70s} calls {calls:3d}  {ns/1e3:9.1f} us  {flop/ns/1e3:8.1f} TFLOP/s ({flop/ns/1e3/2500*100:5.1f} % of 2.5 PF)  '
              f'mfma_busy {busy or 0:.3e}  gui_active {act or 0:.3e}  MFMA util {100*util:5.1f} %')


if __name__ == '__main__':
    main()
