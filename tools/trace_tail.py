#!/usr/bin/env python
"""In-launch residual-norm consumer of the decode GEMM (gemm_decode.hip, dec32_norm_tail) against the two-launch sequence, from
in-kernel s_memrealtime stamps (100 MHz): L launches over DISTINCT weights back to back on one stream, every launch traced.
Per launch: loop end of the last workgroup, last arrival (ticket drawn), poll satisfied, consumer done; and the PERIOD
(first workgroup start -> first workgroup start of the next launch), which is what a decode step pays per linear.
  python tools/trace_tail.py K N M shape splits"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lmdeploy_amd import _ffi  # noqa: E402


def main():
    K, N, M, shape, splits = [int(v, 0) for v in sys.argv[1:6]]
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
    r = torch.zeros((M, N), device='cuda').half()
    nw = torch.ones(N, device='cuda').half()
    ws = torch.empty(max(1, tm.tm_linear_workspace(hs[0], M)) + M * N * 2, dtype=torch.uint8, device='cuda')
    sync = torch.zeros(4, dtype=torch.int32, device='cuda')
    flush = torch.empty(512 << 20, dtype=torch.uint8, device='cuda')
    dbg = torch.zeros((L, 8192, 8), dtype=torch.int64, device='cuda')
    for fused in (1, 0):
        acc = []
        for it in range(6):
            flush.fill_(it)
            dbg.zero_()
            torch.cuda.synchronize()
            for i, h in enumerate(hs):
                tm.tm_debug_set_gemm_trace(dbg[i].data_ptr())
                _ffi.check(tm.tm_linear_residual_norm(h, x.data_ptr(), K, y.data_ptr(), r.data_ptr(), nw.data_ptr(), 1e-5, M, shape, splits,
                                                      fused, ws.data_ptr(), sync.data_ptr(), st))
            tm.tm_debug_set_gemm_trace(None)
            torch.cuda.synchronize()
            raw = dbg.cpu().numpy().astype(np.float64) / 100.0
            if it < 2:
                continue
            for i in range(3, L - 1):
                a = raw[i][raw[i][:, 0] > 0]
                t0 = a[:, 0].min()
                nxt = raw[i + 1][raw[i + 1][:, 0] > 0][:, 0].min()
                tail = a[a[:, 5] > 0]
                row = [len(a), a[:, 2].max() - t0, a[:, 7].max() - t0]
                if fused and len(tail):
                    row += [tail[:, 5].max() - t0, tail[:, 6].min() - t0, tail[:, 6].max() - t0, tail[:, 3].max() - t0]
                else:
                    row += [np.nan, np.nan, np.nan, a[:, 3].max() - t0]
                row += [nxt - t0]
                acc.append(row)
        m = np.nanmean(np.asarray(acc), 0)
        print(f'K={K} N={N} M={M} shape {shape} splits {splits} fused={fused}: {int(m[0])} WGs | loop end {m[1]:.2f} | k-phase reduce done {m[2]:.2f} '
              f'| last ticket {m[3]:.2f} | poll ok first {m[4]:.2f} last {m[5]:.2f} | kernel end {m[6]:.2f} | PERIOD to the next launch {m[7]:.2f} us')


if __name__ == '__main__':
    main()
