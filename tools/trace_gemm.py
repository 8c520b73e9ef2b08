#!/usr/bin/env python
"""Phase breakdown of one GEMM launch from in-kernel s_memrealtime stamps (100 MHz): dispatch skew, prologue, main
loop, epilogue per workgroup.  python tools/trace_gemm.py K N M gated nt splits waves"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lmdeploy_amd import _ffi  # noqa: E402


def main():
    K, N, M, gated, nt, splits, waves = [int(v, 0) for v in sys.argv[1:8]]
    tm = _ffi.load()
    st = torch.cuda.current_stream().cuda_stream
    qw = torch.randint(-2**31, 2**31 - 1, (K, N // 8), dtype=torch.int32, device='cuda')
    s = (torch.rand((K // 128, N), device='cuda') * 1e-3 + 1e-3).half()
    z = torch.randint(0, 16, (K // 128, N), device='cuda').half()
    h = _ffi.C.c_void_p()
    _ffi.check(tm.tm_linear_create(_ffi.C.byref(h), K, N, 0, 128))
    _ffi.check(tm.tm_linear_prepare(h, qw.data_ptr(), s.data_ptr(), z.data_ptr(), st))
    x = torch.randn((M, K), device='cuda').half()
    y = torch.empty((M, N), device='cuda').half()
    ws = torch.empty(max(1, tm.tm_linear_workspace(h, M)), dtype=torch.uint8, device='cuda')
    flush = torch.empty(512 << 20, dtype=torch.uint8, device='cuda')
    dbg = torch.zeros((8192 * 4, 8), dtype=torch.int64, device='cuda')
    ratios = []
    for it in range(int(os.environ.get('TRACE_LAUNCHES', '24'))):
        flush.fill_(it)
        dbg.zero_()
        torch.cuda.synchronize()
        tm.tm_debug_set_gemm_trace(dbg.data_ptr())
        _ffi.check(tm.tm_linear_forward(h, x.data_ptr(), K, y.data_ptr(), N // (2 if gated else 1), M, gated, nt, splits,
                                        waves, ws.data_ptr(), st))
        tm.tm_debug_set_gemm_trace(None)
        torch.cuda.synchronize()
        raw_all = dbg.cpu().numpy()
        raw = raw_all[:8192]
        ph = raw_all[8192:]
        ph = ph[ph[:, 3] > 0].astype(np.float64)
        if len(ph):
            per = ph[:, :3] / ph[:, 3:4]
            print(f'  per-iteration cycles (mean over {len(ph)} traced waves): barrier {per[:,0].mean():.0f}  compute+compiler waits {per[:,1].mean():.0f}  '
                  f'DMA wait {per[:,2].mean():.0f}  (max barrier {per[:,0].max():.0f}, max compute {per[:,1].max():.0f})')
        ids = np.nonzero(raw[:, 0] > 0)[0]
        hw = raw[ids, 4]
        tt = raw[ids, :4].astype(np.float64) / 100.0
        tt -= tt[:, 0].min()
        loop = tt[:, 2] - tt[:, 1]
        clk = (raw[ids, 6] - raw[ids, 5]).astype(np.float64) / np.maximum(loop, 1e-3)   # shader ticks per us = MHz
        ratios.append(loop.max() / np.median(loop))
        # physical CU of every workgroup: (xcc, se, sh, cu) from XCC_ID / HW_ID
        xcc, hwid = (hw >> 32) & 0xf, hw & 0xffffffff
        cu_key = (xcc << 16) | (hwid & 0xff00)
        uniq, cnt = np.unique(cu_key, return_counts=True)
        order = np.argsort(-loop)[:5]
        print(f'launch {it}: span {tt[:, 3].max():.2f} us, loop clock {np.median(clk):.0f} MHz (min {clk.min():.0f}), loop max/median {ratios[-1]:.2f}, distinct CUs {len(uniq)} for {len(ids)} WGs '
              f'(max {cnt.max()} WGs on one CU); slowest (id xcc:hwid loop_us sharing): ' +
              '  '.join(f'{ids[i]} {xcc[i]}:{hwid[i] & 0xffff:04x} {loop[i]:.1f} x{cnt[np.searchsorted(uniq, cu_key[i])]}' for i in order))
    print('loop max/median over launches: ' + ' '.join(f'{r:.2f}' for r in ratios))
    t = dbg.cpu().numpy()[:8192, :4]
    t = t[t[:, 0] > 0].astype(np.float64) / 100.0      # us
    t0 = t[:, 0].min()
    t -= t0
    print(f'workgroups {len(t)}  kernel span {t[:, 3].max():.2f} us')
    for name, a in (('start skew', t[:, 0]), ('prologue', t[:, 1] - t[:, 0]), ('main loop', t[:, 2] - t[:, 1]),
                    ('epilogue', t[:, 3] - t[:, 2]), ('end time', t[:, 3])):
        print(f'{name:11s} min {a.min():6.2f}  mean {a.mean():6.2f}  p90 {np.percentile(a, 90):6.2f}  max {a.max():6.2f} us')


if __name__ == '__main__':
    main()
