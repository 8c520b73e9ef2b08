import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from lmdeploy_amd import _ffi
from oracle import tm_oracle as o
tm = _ffi.load()
C = _ffi.C
f16 = np.float16
def dev(a): return torch.from_numpy(np.ascontiguousarray(a)).cuda()
def st(): return None
rng = np.random.default_rng(0)
for (K, N, M, base, splits) in [(4096, 6144, 64, 0, 2), (4096, 6144, 64, 3, 2), (2048, 512, 64, 0, 2), (2048, 512, 33, 10, 2), (4096, 28672, 64, 10, 2)]:
    q = rng.integers(0, 16, (K, N), dtype=np.uint8)
    s = (rng.random((K // 128, N), dtype=np.float32) * 0.02 + 0.005).astype(f16)
    z = rng.integers(0, 16, (K // 128, N)).astype(f16)
    packed = o.pack_u4_row(q)
    h = C.c_void_p()
    _ffi.check(tm.tm_linear_create(C.byref(h), K, N, 0, 128))
    _ffi.check(tm.tm_linear_prepare(h, dev(packed).data_ptr(), dev(s).data_ptr(), dev(z).data_ptr(), None))
    x = rng.standard_normal((M, K)).astype(f16)
    ws = torch.zeros(max(1, tm.tm_linear_workspace(h, M)), dtype=torch.uint8, device='cuda')
    xd = dev(x)
    outs = {}
    for shape, sp in ((base, 1), (base, splits), (16 + base, splits), (16 + base, splits)):
        y = torch.zeros((M, N), dtype=torch.float16, device='cuda')
        _ffi.check(tm.tm_linear_forward(h, xd.data_ptr(), K, y.data_ptr(), N, M, 0, 0, sp, 0x200 | shape, ws.data_ptr(), None))
        torch.cuda.synchronize()
        outs.setdefault((shape, sp), []).append(y.cpu().numpy())
    a = outs[(base, splits)][0]
    for i, b in enumerate(outs[(16 + base, splits)]):
        bad = a.view(np.uint16) != b.view(np.uint16)
        rows, cols = np.nonzero(bad)
        print(f'K={K} N={N} M={M} base={base} x{splits} launch {i}: mismatches {bad.sum()} of {bad.size}; rows {sorted(set(rows.tolist()))[:10]}.. cols/128 {sorted(set((cols // 128).tolist()))[:16]}; '
              f'maxdiff {np.abs(a.astype(np.float32) - b.astype(np.float32)).max():.4f}; vs 1-split maxdiff {np.abs(outs[(base,1)][0].astype(np.float32) - b.astype(np.float32)).max():.4f}')
        if bad.sum():
            r, c = rows[0], cols[0]
            print('   first bad', r, c, 'plain', a[r, c], 'merged', b[r, c], 'c%128', c % 128, 'col set in tile', sorted(set((cols % 128).tolist()))[:40])
    tm.tm_linear_destroy(h)

# ---- second experiment: what do the bad elements hold?  slabs of the plain split-K launch (s0, s1) vs a workspace pre-filled with 1000.0
print('--- stale-or-corrupt experiment')
K, N, M, base, splits = 2048, 512, 64, 0, 2
rng = np.random.default_rng(1)
q = rng.integers(0, 16, (K, N), dtype=np.uint8)
s = (rng.random((K // 128, N), dtype=np.float32) * 0.02 + 0.005).astype(f16)
z = rng.integers(0, 16, (K // 128, N)).astype(f16)
h = C.c_void_p()
_ffi.check(tm.tm_linear_create(C.byref(h), K, N, 0, 128))
_ffi.check(tm.tm_linear_prepare(h, dev(o.pack_u4_row(q)).data_ptr(), dev(s).data_ptr(), dev(z).data_ptr(), None))
x = rng.standard_normal((M, K)).astype(f16)
nb = tm.tm_linear_workspace(h, M)
ws = torch.zeros(nb // 4, dtype=torch.float32, device='cuda')
xd = dev(x)
y = torch.zeros((M, N), dtype=torch.float16, device='cuda')
_ffi.check(tm.tm_linear_forward(h, xd.data_ptr(), K, y.data_ptr(), N, M, 0, 0, splits, 0x200 | base, ws.data_ptr(), None))
torch.cuda.synchronize()
s0 = ws[:M * N].cpu().numpy().reshape(M, N).copy()
s1 = ws[M * N:2 * M * N].cpu().numpy().reshape(M, N).copy()
good = y.cpu().numpy()
for trial in range(3):
    ws[:2 * M * N] = 1000.0
    torch.cuda.synchronize()
    y2 = torch.zeros((M, N), dtype=torch.float16, device='cuda')
    _ffi.check(tm.tm_linear_forward(h, xd.data_ptr(), K, y2.data_ptr(), N, M, 0, 0, splits, 0x200 | (16 + base), ws.data_ptr(), None))
    torch.cuda.synchronize()
    got = y2.cpu().numpy()
    after0 = ws[:M * N].cpu().numpy().reshape(M, N)
    after1 = ws[M * N:2 * M * N].cpu().numpy().reshape(M, N)
    bad = got.view(np.uint16) != good.view(np.uint16)
    r, c = np.nonzero(bad)
    print(f'trial {trial}: bad {bad.sum()}; slab0 in memory afterwards == s0: {np.array_equal(after0, s0)} ({(after0 != s0).sum()} differ), slab1: {np.array_equal(after1, s1)} ({(after1 != s1).sum()} differ)')
    for i in range(min(6, len(r))):
        rr, cc = r[i], c[i]
        print(f'   ({rr},{cc}) got {float(got[rr, cc]):.3f} want {float(good[rr, cc]):.3f} s0 {s0[rr, cc]:.3f} s1 {s1[rr, cc]:.3f} s0+1000 {s0[rr,cc]+1000:.1f} s1+1000 {s1[rr,cc]+1000:.1f} mem0 {after0[rr,cc]:.3f} mem1 {after1[rr,cc]:.3f}')
