#!/usr/bin/env python
"""Sweep (nt, splits) of the W4A16 GEMM for the decode shapes of a model and print GB/s per config.
Counterpart of the reference's GEMM tuner (src/turbomind/kernels/gemm/tuner/, turbomind.cc:363-487): measured
dispatch instead of a cost model.  Usage: python tools/tune_gemm.py [--m 64] [--model llama3_8b]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lmdeploy_amd import _ffi  # noqa: E402

MODELS = {
    'llama3_8b': dict(H=4096, q=32, kv=8, I=14336, V=128256),
    'internlm2_20b': dict(H=6144, q=48, kv=8, I=16384, V=92544),
    'llama3_70b_tp8': dict(H=8192, q=8, kv=1, I=3584, V=16032),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--m', type=int, default=64)
    ap.add_argument('--model', default='llama3_8b')
    ap.add_argument('--iters', type=int, default=30)
    ap.add_argument('--only', default='')
    ap.add_argument('--no-flush', action='store_true', help='leave the weights in the Infinity Cache between launches')
    ap.add_argument('--cfg', default='', help='waves,nt,splits: time one configuration only')
    args = ap.parse_args()
    tm = _ffi.load()
    mm = MODELS[args.model]
    H, D = mm['H'], 128
    shapes = {'qkv': (H, (mm['q'] + 2 * mm['kv']) * D, 0), 'o': (mm['q'] * D, H, 0), 'gate_up': (H, 2 * mm['I'], 1),
              'down': (mm['I'], H, 0)}
    M = args.m
    st = torch.cuda.current_stream().cuda_stream
    results = {}
    # a buffer larger than the 256 MiB Infinity Cache, written between timed launches, keeps weights HBM-cold
    flush = torch.empty(512 << 20, dtype=torch.uint8, device='cuda')
    for name, (K, N, gated) in shapes.items():
        if args.only and name != args.only:
            continue
        qw = torch.randint(-2**31, 2**31 - 1, (K, N // 8), dtype=torch.int32, device='cuda')
        s = (torch.rand((K // 128, N), device='cuda') * 1e-3 + 1e-3).half()
        z = torch.randint(0, 16, (K // 128, N), device='cuda').half()
        h = _ffi.C.c_void_p()
        _ffi.check(tm.tm_linear_create(_ffi.C.byref(h), K, N, 0, 128))
        _ffi.check(tm.tm_linear_prepare(h, qw.data_ptr(), s.data_ptr(), z.data_ptr(), st))
        x = torch.randn((M, K), device='cuda').half()
        y = torch.empty((M, N), device='cuda').half()
        ws = torch.empty(max(1, tm.tm_linear_workspace(h, M)), dtype=torch.uint8, device='cuda')
        wbytes = K * N / 2 + K * N / 32
        best = None
        grid_ = [(w_, n_, s_) for w_ in (4, 8, 0x108) for n_ in (1, 2, 4) for s_ in (1, 2, 4, 8, 16)]
        if args.cfg:
            grid_ = [tuple(int(v, 0) for v in args.cfg.split(','))]
        for waves, nt, splits in grid_:
            if True:
                if splits > K // 256 or (waves != 4 and nt == 4) or (waves == 4 and nt == 4 and M <= 64):
                    continue
                ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                      for _ in range(args.iters)]
                for a, b in ev:
                    if not args.no_flush:
                        flush.fill_(1)
                    a.record()
                    _ffi.check(tm.tm_linear_forward(h, x.data_ptr(), K, y.data_ptr(), N // (2 if gated else 1), M, gated,
                                                    nt, splits, waves, ws.data_ptr(), st))
                    b.record()
                torch.cuda.synchronize()
                ts = sorted(a.elapsed_time(b) for a, b in ev)
                med = ts[len(ts) // 2]
                gbs = wbytes / (med * 1e-3) / 1e9
                print(f'{name:8s} K={K:6d} N={N:6d} M={M} waves={waves} nt={nt} splits={splits:2d}  {med*1e3:8.1f} us  {gbs:7.0f} GB/s',
                      flush=True)
                if best is None or med < best[0]:
                    best = (med, nt, splits, gbs, waves)
        results[name] = dict(K=K, N=N, us=best[0] * 1e3, nt=best[1], splits=best[2], gbs=best[3], waves=best[4])
        _ffi.check(tm.tm_linear_destroy(h))
    print('BEST ' + json.dumps(results))


if __name__ == '__main__':
    main()
