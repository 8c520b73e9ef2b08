#!/usr/bin/env python
"""Host cost of the one-call rank group: microseconds per mirrored engine call at tp = 2 / 4 / 8 with stub engines
(a call that returns at once), over the same authenticated localhost connections the product uses -- what a continuous-batching session
pays per scheduler step ON TOP of the device time when every `step` is mirrored one by one.

  python tools/tp_group_host_cost.py [--calls 2000]            prints one JSON line per tp
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class StubEngine:
    def step(self, n=1):
        return (1, 0)

    def step_many(self, n, **kw):
        return (n, 0, 0)

    def close(self):
        pass


def child(port, key):
    import faulthandler
    faulthandler.dump_traceback_later(15, exit=True, file=open(f'/tmp/tpcost_child_{os.getpid()}.txt', 'w'))
    from multiprocessing.connection import Client
    from lmdeploy_amd.turbomind import tp_group
    conn = Client(('127.0.0.1', port), authkey=bytes.fromhex(key))
    conn.send(0)
    tp_group.WorkerLink(conn).serve(StubEngine())
    conn.close()


def measure(tp, calls, burst=8):
    from multiprocessing.connection import Listener
    from lmdeploy_amd.turbomind import tp_group
    key = os.urandom(16)
    with Listener(('127.0.0.1', 0), authkey=key) as srv:
        port = srv.address[1]
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), '--child', str(port), key.hex()], stdout=subprocess.DEVNULL,
                                  stderr=subprocess.DEVNULL, stdin=subprocess.DEVNULL) for _ in range(tp - 1)]
        conns = []
        for _ in range(tp - 1):
            c = srv.accept()
            c.recv()
            conns.append(c)
    link = tp_group.ParentLink(tp, 'unused', None, conns=conns)
    eng = tp_group.TpEngine(StubEngine(), link, 'stub')
    try:
        for _ in range(50):
            eng.step()
        t0 = time.perf_counter()
        for _ in range(calls):
            eng.step()
        us = (time.perf_counter() - t0) / calls * 1e6
        t0 = time.perf_counter()
        for _ in range(max(1, calls // burst)):
            eng.step_many(burst)
        us_burst = (time.perf_counter() - t0) / (max(1, calls // burst) * burst) * 1e6
    finally:
        link.close()
        for p in procs:
            try:
                p.wait(timeout=10)
            except Exception:       # noqa: BLE001
                p.kill()
    return us, us_burst


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == '--child':
        child(int(sys.argv[2]), sys.argv[3])
        sys.exit(0)
    import faulthandler; faulthandler.dump_traceback_later(int(os.environ.get("TM_DUMP_S", "100000")), exit=True)
    ap = argparse.ArgumentParser()
    ap.add_argument('--calls', type=int, default=2000)
    a = ap.parse_args()
    for tp in (2, 4, 8):
        us, usb = measure(tp, a.calls)
        print(json.dumps({'tp': tp, 'us_per_mirrored_step': round(us, 1), 'us_per_step_in_bursts_of_8': round(usb, 1), 'calls': a.calls,
                          'host_cpus': os.cpu_count()}), flush=True)
