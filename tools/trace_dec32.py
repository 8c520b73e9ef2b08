#!/usr/bin/env python
"""Phase breakdown of the decode GEMM (gemm_decode.hip) from in-kernel s_memrealtime stamps (100 MHz): per workgroup
start skew, prologue (until the first stage barrier), main loop, k-phase reduction, stores.  Launches run back to back
over DISTINCT weights (HBM-cold) inside one stream; the traced launch is the last one.
  python tools/trace_dec32.py K N M gated shape splits [abl]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lmdeploy_amd import _ffi  # noqa: E402


def main():
    K, N, M, gated, shape, splits = [int(v, 0) for v in sys.argv[1:7]]
    if len(sys.argv) > 7:
        os.environ['TM_D32_ABL'] = sys.argv[7]
    tm = _ffi.load()
    st = torch.cuda.current_stream().cuda_stream
    L = 6
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
    dbg = torch.zeros((8192, 8), dtype=torch.int64, device='cuda')
    rows = []
    for it in range(8):
        flush.fill_(it)
        dbg.zero_()
        torch.cuda.synchronize()
        for i, h in enumerate(hs):
            tm.tm_debug_set_gemm_trace(dbg.data_ptr() if i == L - 1 else None)
            _ffi.check(tm.tm_linear_forward(h, x.data_ptr(), K, y.data_ptr(), N // (2 if gated else 1), M, gated, 0, splits,
                                            0x200 | shape, ws.data_ptr(), st))
        tm.tm_debug_set_gemm_trace(None)
        torch.cuda.synchronize()
        raw = dbg.cpu().numpy()
        raw = raw[raw[:, 0] > 0]
        t = raw[:, [0, 1, 2, 7, 3]].astype(np.float64) / 100.0
        t0 = t[:, 0].min()
        t -= t0
        t[t < 0] = np.nan
        xcc = (raw[:, 4] >> 32) & 0xf
        rows.append(t)
        ph = [('start', t[:, 0]), ('prologue', t[:, 1] - t[:, 0]), ('loop', t[:, 2] - t[:, 1]), ('reduce', t[:, 3] - t[:, 2]),
              ('store', t[:, 4] - t[:, 3]), ('end', t[:, 4])]
        if raw[:, 5].any():      # shader-clock counter stamps around the loop (gemm_pre64_kernel)
            mhz = (raw[:, 6] - raw[:, 5]).astype(np.float64) / np.maximum((raw[:, 2] - raw[:, 1]).astype(np.float64) / 100.0, 1e-9)
            print(f'   shader clock inside the loop: mean {np.mean(mhz):.0f} MHz (min {np.min(mhz):.0f}, max {np.max(mhz):.0f})')
        print(f'launch {it}: {len(t)} WGs, span {np.nanmax(t[:, 4]):.2f} us | ' +
              ' | '.join(f'{n} {np.nanmean(a):.2f} (max {np.nanmax(a):.2f})' for n, a in ph) +
              f' | WGs per XCC {np.bincount(xcc.astype(int), minlength=8).tolist()}')


if __name__ == '__main__':
    main()
