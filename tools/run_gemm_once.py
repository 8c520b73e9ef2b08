#!/usr/bin/env python
"""Run ONE W4A16 GEMM configuration a few times (for rocprofv3 --kernel-trace / --pmc).
python tools/run_gemm_once.py K N M gated nt splits waves [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lmdeploy_amd import _ffi  # noqa: E402


def main():
    K, N, M, gated, nt, splits, waves = map(int, sys.argv[1:8])
    iters = int(sys.argv[8]) if len(sys.argv) > 8 else 5
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
    for _ in range(iters):
        flush.fill_(1)
        _ffi.check(tm.tm_linear_forward(h, x.data_ptr(), K, y.data_ptr(), N // (2 if gated else 1), M, gated, nt, splits,
                                        waves, ws.data_ptr(), st))
    torch.cuda.synchronize()


if __name__ == '__main__':
    main()
