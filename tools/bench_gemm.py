#!/usr/bin/env python
"""Decode-GEMM micro-benchmark on the GPU: the four W4A16 linears of a Llama-3-8B layer (or --shapes K,N,gated ...) at
M = 64, every variant timed as ONE hipGraph of L launches over L DISTINCT weights (L x bytes > the 256 MB Infinity
Cache, so the weights really stream from HBM as they do in the model), replayed R times.  Prints us per launch and the
algorithmic GB/s (packed weights + scales per launch).

  python tools/bench_gemm.py [--m 64] [--reps 20] [--variants old,d0,d0pf4,d1,d2,d3,abl15,abl31,...]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lmdeploy_amd import _ffi  # noqa: E402

LLAMA3_8B = {'qkv': (4096, 6144, 0), 'o': (4096, 4096, 0), 'gate_up': (4096, 28672, 1), 'down': (14336, 4096, 0)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--m', type=int, default=64)
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--only', default='')
    ap.add_argument('--variants', default='d0,d0pf4,d1,d2,d3')
    ap.add_argument('--splits', default='0', help='comma list of split counts to try for the d* variants (0 = heuristic)')
    ap.add_argument('--json', default='')
    ap.add_argument('--same', action='store_true', help='every launch of a graph uses the SAME weight: Infinity-Cache-resident after the first replay')
    ap.add_argument('--abl-shape', type=int, default=0)
    args = ap.parse_args()
    tm = _ffi.load()
    C = _ffi.C
    M = args.m
    results = []
    for name, (K, N, gated) in LLAMA3_8B.items():
        if args.only and name not in args.only.split(','):
            continue
        wbytes = K * N // 2 + (K // 128) * N * 4
        L = max(4, int(600e6 // wbytes) + 1) if M <= 256 else 2
        g = torch.Generator(device='cuda').manual_seed(1)
        handles = []
        for _ in range(L):
            qw = torch.randint(-2**31, 2**31 - 1, (K, N // 8), generator=g, device='cuda', dtype=torch.int32)
            s = (torch.rand((K // 128, N), generator=g, device='cuda') * 0.002 + 0.001).to(torch.float16)
            z = torch.randint(4, 12, (K // 128, N), generator=g, device='cuda').to(torch.float16)
            h = C.c_void_p()
            _ffi.check(tm.tm_linear_create(C.byref(h), K, N, 0, 128))
            _ffi.check(tm.tm_linear_prepare(h, qw.data_ptr(), s.data_ptr(), z.data_ptr(), torch.cuda.current_stream().cuda_stream))
            torch.cuda.synchronize()
            handles.append(h)
            del qw, s, z
        x = torch.randn((M, K), generator=g, device='cuda').to(torch.float16)
        y = torch.zeros((M, N // 2 if gated else N), dtype=torch.float16, device='cuda')
        ws = torch.zeros(max(1, tm.tm_linear_workspace(handles[0], M)), dtype=torch.uint8, device='cuda')

        def run(variant, splits):
            env = {}
            nt, waves = 0, 0
            if variant == 'old':
                nt, waves = 1, 0x108           # round-1 default: 4 column groups x 2 k-phases, 8 waves
            elif variant == 'oldp':
                nt, waves = 2, 8               # round-1 prefill tile: 8 waves x 2 tiles, 128-row blocks
            elif variant == 'auto':
                nt, waves = 0, 0               # the library's own pick (measured table, then the heuristic)
            elif variant == 'p256':
                waves = 0x200 | 12             # 256 x 256 prefill tiles, dequant through LDS (gemm_prefill.hip)
            elif variant == 'lib':
                waves = 0x200 | 10             # dequantise + vendor fp16 GEMM (probe only)
            elif variant == 'lc':
                waves = 0x200 | 11             # loader / consumer kernel (gemm_decode_lc.hip)
            elif variant.startswith('d'):
                digits = ''.join(c for c in variant[1:3] if c.isdigit())   # d0 .. d9, d11 .. d17
                shape = int(digits)
                waves = 0x200 | shape
                if 'pf4' in variant:
                    env['TM_D32_PF'] = '4'
            elif variant.startswith('abl'):
                waves = 0x200 | (4 if M > 64 else args.abl_shape)
                env['TM_D32_ABL'] = variant[3:]
            for k, v in env.items():
                os.environ[k] = v

            def launch(h, stream):
                _ffi.check(tm.tm_linear_forward(h, x.data_ptr(), K, y.data_ptr(), y.shape[1], M, gated, nt, splits, waves,
                                                ws.data_ptr(), stream))
            for h_ in handles:                                                # warm-up (function attributes, lazy init; shape 13 builds its fp16 image on first use)
                launch(h_, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                st = torch.cuda.current_stream().cuda_stream
                for h in handles:
                    launch(handles[0] if args.same else h, st)
            for k in env:
                del os.environ[k]
            graph.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ts = []
            for _ in range(args.reps):
                e0.record()
                graph.replay()
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3 / L)
            ts = np.asarray(ts)
            return float(np.median(ts)), float(ts.min())

        for variant in args.variants.split(','):
            sp_list = [int(v) for v in args.splits.split(',')] if (variant.startswith('d') or variant in ('lc', 'p256')) else [0]
            for sp in sp_list:
                try:
                    med, mn = run(variant, sp)
                except Exception as e:   # noqa: BLE001
                    print(f'{name:8s} {variant:8s} splits={sp}: FAILED {e}', flush=True)
                    continue
                r = dict(gemm=name, K=K, N=N, M=M, variant=variant, splits=sp, us_median=round(med, 2), us_min=round(mn, 2),
                         gbps=round(wbytes / med / 1e3, 1), launches=L)
                results.append(r)
                tf = 2.0 * M * K * N / med / 1e6
                print(f"{name:8s} K={K:6d} N={N:6d} M={M:5d} {variant:8s} splits={sp:2d}  {med:8.2f} us (min {mn:8.2f})  {wbytes / med / 1e3:7.1f} GB/s  {tf:7.1f} TF/s", flush=True)
        for h in handles:
            tm.tm_linear_destroy(h)
        del x, y, ws
        torch.cuda.empty_cache()
    if args.json:
        with open(args.json, 'w') as f:
            json.dump(results, f, indent=1)


if __name__ == '__main__':
    main()
