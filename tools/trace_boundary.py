#!/usr/bin/env python
"""Kernel-boundary cost of the decode GEMM from in-kernel s_memrealtime stamps (100 MHz, one clock for the whole chip):
L launches over DISTINCT weights queued back to back on one stream, EVERY launch traced into its own buffer, so that
    gap(i) = first workgroup start of launch i+1  -  last workgroup end of launch i
is measured directly, next to the span of each launch.  Arms (env): TM_D32_WT=0/1/3 (write-through slab / output stores).
  python tools/trace_boundary.py K N M gated shape splits"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lmdeploy_amd import _ffi  # noqa: E402


def main():
    K, N, M, gated, shape, splits = [int(v, 0) for v in sys.argv[1:7]]
    tm = _ffi.load()
    st = torch.cuda.current_stream().cuda_stream
    L = 8
    hs = []
    for i in range(L):
        qw = torch.randint(-2**31, 2**31 - 1, (K, N // 8), dtype=torch.int32, device='cuda')
        s = (torch.rand((K // 128, N), device='cuda') * 1e-3 + 1e-3).half()
        z = torch.randint(0, 16, (K // 128, N), device='cuda').half()
        h = _ffi.C.c_void_p()
        _ffi.check(tm.tm_linear_create(_ffi.C.byref(h), K, N, 0, 128))
        _ffi.check(tm.tm_linear_prepare(h, qw.data_ptr(), s.data_ptr(), z.data_ptr(), st))
        torch.cuda.synchronize()
        hs.append(h)
    x = torch.randn((M, K), device='cuda').half()
    y = torch.empty((M, N), device='cuda').half()
    ws = torch.empty(max(1, tm.tm_linear_workspace(hs[0], M)), dtype=torch.uint8, device='cuda')
    flush = torch.empty(512 << 20, dtype=torch.uint8, device='cuda')
    dbg = torch.zeros((L, 8192, 8), dtype=torch.int64, device='cuda')
    gaps, spans = [], []
    for it in range(6):
        flush.fill_(it)
        dbg.zero_()
        torch.cuda.synchronize()
        for i, h in enumerate(hs):
            tm.tm_debug_set_gemm_trace(dbg[i].data_ptr())
            _ffi.check(tm.tm_linear_forward(h, x.data_ptr(), K, y.data_ptr(), N // (2 if gated else 1), M, gated, 0, splits,
                                            0x200 | shape, ws.data_ptr(), st))
        tm.tm_debug_set_gemm_trace(None)
        torch.cuda.synchronize()
        raw = dbg.cpu().numpy()
        first, last = [], []
        for i in range(L):
            r = raw[i][raw[i][:, 0] > 0]
            first.append(r[:, 0].min() / 100.0)
            last.append(r[:, 3].max() / 100.0)
        if it >= 2:   # the first passes include lazy initialisation
            gaps.append([first[i + 1] - last[i] for i in range(2, L - 1)])      # launches 0..2 may be host-launch bound
            spans.append([last[i] - first[i] for i in range(3, L)])
    gaps, spans = np.asarray(gaps), np.asarray(spans)
    print(f'K={K} N={N} M={M} splits={splits} TM_D32_WT={os.environ.get("TM_D32_WT", "0")}: span {spans.mean():.2f} us (min {spans.min():.2f}) '
          f'| gap end->next start {gaps.mean():.2f} us (min {gaps.min():.2f}, max {gaps.max():.2f}) | period {(spans.mean() + gaps.mean()):.2f} us')


if __name__ == '__main__':
    main()
