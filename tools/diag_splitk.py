#!/usr/bin/env python
"""Diagnostic: the prefill-tile test sequence with slab-level checks (which split-K slab / rows / columns are wrong)."""
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lmdeploy_amd import _ffi  # noqa: E402
from oracle import tm_oracle as o  # noqa: E402

f16 = np.float16
tm = _ffi.load()
st = torch.cuda.current_stream().cuda_stream
cache = {}


def make(K, N):
    if (K, N) not in cache:
        w = (np.random.default_rng(K * 7 + N).standard_normal((K, N)) * (0.1 / math.sqrt(K))).astype(f16)
        q, s, z, _ = o.quantize_groupwise_u4(w, 128)
        cache[(K, N)] = (q, s, z, o.w4a16_dequant(q, s, z).astype(np.float32))
    q, s, z, wd = cache[(K, N)]
    h = _ffi.C.c_void_p()
    _ffi.check(tm.tm_linear_create(_ffi.C.byref(h), K, N, 0, 128))
    qd, sd, zd = torch.from_numpy(o.pack_u4_row(q)).cuda(), torch.from_numpy(s).cuda(), torch.from_numpy(z).cuda()
    _ffi.check(tm.tm_linear_prepare(h, qd.data_ptr(), sd.data_ptr(), zd.data_ptr(), st))
    torch.cuda.synchronize()
    return h, wd


bad = 0
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    for M in (1000, 2500):
        for K, N in ((4096, 6144), (1792, 4096), (1024, 512)):
            h, wd = make(K, N)
            x = np.random.default_rng(K + N + M + 11).standard_normal((M, K)).astype(f16)
            ref = x.astype(np.float32) @ wd
            ws = torch.zeros(max(1, tm.tm_linear_workspace(h, M)), dtype=torch.uint8, device='cuda')
            x_d = torch.from_numpy(x).cuda()
            for splits, waves in ((0, 0), (1, 0x204), (2, 0x204), (3, 0x204), (1, 0x205), (2, 0x205), (5, 0x205)):
                if splits > K // 128:
                    continue
                y = torch.zeros((M, N), dtype=torch.float16, device='cuda')
                _ffi.check(tm.tm_linear_forward(h, x_d.data_ptr(), K, y.data_ptr(), N, M, 0, 0, splits, waves, ws.data_ptr(), st))
                torch.cuda.synchronize()
                err = np.abs(y.cpu().numpy().astype(np.float32) - ref)
                ok = err <= 2e-3 + 2.0**-9 * np.abs(ref)
                if not ok.all():
                    bad += 1
                    r, c = np.nonzero(~ok)
                    print(f'rep {rep} M={M} K={K} N={N} splits={splits} waves={waves:#x}: {len(r)} wrong, rows {r.min()}..{r.max()} '
                          f'({len(np.unique(r))} distinct, row blocks {sorted(set((r // 128).tolist()))[:12]}), cols {c.min()}..{c.max()} '
                          f'({len(np.unique(c))} distinct, col groups of 256: {sorted(set((c // 256).tolist()))[:16]}) max err {err.max():.3f}')
                    if splits > 1:
                        KB = K // 128
                        per = (KB + splits - 1) // splits
                        S = 2 if waves == 0x204 else 1
                        per = (per + S - 1) // S * S
                        slabs = ws[:splits * M * N * 4].view(torch.float32).reshape(splits, M, N).cpu().numpy()
                        for sidx in range(splits):
                            k0, k1 = sidx * per * 128, min(K, (sidx + 1) * per * 128)
                            pref = x[:, k0:k1].astype(np.float32) @ wd[k0:k1]
                            e = np.abs(slabs[sidx] - pref)
                            nb = (e > 1e-2).sum()
                            if nb:
                                rr, cc = np.nonzero(e > 1e-2)
                                print(f'    slab {sidx} (k {k0}..{k1}): {nb} wrong, rows {rr.min()}..{rr.max()}, cols {cc.min()}..{cc.max()}, '
                                      f'row%128 in {sorted(set((rr % 128).tolist()))[:8]}.., sample got {slabs[sidx][rr[0], cc[0]]:.4f} want {pref[rr[0], cc[0]]:.4f}')
            _ffi.check(tm.tm_linear_destroy(h))
print('bad cases:', bad)
