#!/usr/bin/env python
"""Prefill-shape GEMM: this library's fused W4A16 kernel against the vendor library's plain fp16 GEMM (torch.matmul ->
hipBLASLt / rocBLAS) on the same (M, K, N).  The library GEMM gets fp16 weights for free here, so its figure is an upper
bound for a "dequantise a layer's weights to fp16, then call the library" prefill path (the dequantisation pass would
add K*N*2.5 bytes of HBM traffic per linear and chunk).

  python tools/probes/prefill_vs_library.py [--ms 2048,4096,8192] [--iters 10]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lmdeploy_amd import _ffi  # noqa: E402

SHAPES = {'qkv': (4096, 6144, 0), 'o': (4096, 4096, 0), 'gate_up': (4096, 28672, 1), 'down': (14336, 4096, 0)}


def timed(fn, iters):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / iters)
    return float(np.median(ts))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--ms', default='2048,4096,8192')
    ap.add_argument('--iters', type=int, default=10)
    ap.add_argument('--only', default='')
    args = ap.parse_args()
    tm = _ffi.load()
    C = _ffi.C
    g = torch.Generator(device='cuda').manual_seed(1)
    for name, (K, N, gated) in SHAPES.items():
        if args.only and name not in args.only.split(','):
            continue
        qw = torch.randint(-2**31, 2**31 - 1, (K, N // 8), generator=g, device='cuda', dtype=torch.int32)
        s = (torch.rand((K // 128, N), generator=g, device='cuda') * 0.002 + 0.001).to(torch.float16)
        z = torch.randint(4, 12, (K // 128, N), generator=g, device='cuda').to(torch.float16)
        h = C.c_void_p()
        _ffi.check(tm.tm_linear_create(C.byref(h), K, N, 0, 128))
        _ffi.check(tm.tm_linear_prepare(h, qw.data_ptr(), s.data_ptr(), z.data_ptr(), torch.cuda.current_stream().cuda_stream))
        torch.cuda.synchronize()
        del qw, s, z
        w_kn = (torch.randn((K, N), generator=g, device='cuda') * 0.02).to(torch.float16)
        w_nk = w_kn.t().contiguous()
        for M in [int(v) for v in args.ms.split(',')]:
            x = torch.randn((M, K), generator=g, device='cuda').to(torch.float16)
            y = torch.zeros((M, N // 2 if gated else N), dtype=torch.float16, device='cuda')
            ws = torch.zeros(max(1, tm.tm_linear_workspace(h, M)), dtype=torch.uint8, device='cuda')
            st = torch.cuda.current_stream().cuda_stream

            sh, sp = C.c_int(), C.c_int()
            _ffi.check(tm.tm_debug_pick_tiling(K, N, M, 0, C.byref(sh), C.byref(sp)))     # the library's own heuristic

            def ours():
                _ffi.check(tm.tm_linear_forward(h, x.data_ptr(), K, y.data_ptr(), y.shape[1], M, gated, 0, sp.value, 0x200 | sh.value,
                                                ws.data_ptr(), st))
            out = torch.empty((M, N), dtype=torch.float16, device='cuda')
            t_ours = timed(ours, args.iters)

            def ours_p256():
                _ffi.check(tm.tm_linear_forward(h, x.data_ptr(), K, y.data_ptr(), y.shape[1], M, gated, 0, 1, 0x200 | 12, ws.data_ptr(), st))
            t_lib = timed(ours_p256, args.iters) if M >= 256 and N >= 256 else float('nan')
            t_nn = timed(lambda: torch.matmul(x, w_kn, out=out), args.iters)
            t_nt = timed(lambda: torch.matmul(x, w_nk.t(), out=out), args.iters)
            fl = 2.0 * M * K * N
            print(f'{name:8s} M={M:5d} K={K:6d} N={N:6d}  fused W4A16 {t_ours:8.1f} us {fl / t_ours / 1e6:7.1f} TF/s | '
                  f'256 x 256 tile (shape 12) {t_lib:8.1f} us {fl / t_lib / 1e6:7.1f} TF/s | '
                  f'library fp16 NN {t_nn:8.1f} us {fl / t_nn / 1e6:7.1f} TF/s | NT {t_nt:8.1f} us {fl / t_nt / 1e6:7.1f} TF/s', flush=True)
            del x, y, ws, out
        tm.tm_linear_destroy(h)
        del w_kn, w_nk
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
