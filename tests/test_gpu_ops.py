"""GPU parity tests: every HIP operator, called through the C-ABI, against the CPU oracle on the same seeded
inputs.  Integer / byte / index outputs must be bit exact; floating point within the stated tolerance.

Tolerances (fp16 path):
  * W4A16 GEMM: the reference's own gate for ('fp16','uint4') is max_abs <= 0.25, mean_abs <= 0.05
    (tests/turbomind/linear/fixture.py:34-42).  Ours is dequant-exact, so we hold a far tighter bound:
    max_abs <= 2e-3 + 2^-10 * |ref| (fp32 accumulation-order noise + one fp16 rounding).
  * attention: the reference's Compare thresholds rtol 1e-2 / atol 1e-4 (kernels/attention/test_utils.h:12-14),
    applied as |a-b| <= 1e-2*|b| + 2e-3 (fp16 outputs of magnitude O(1)).
  * norms / RoPE: <= 1 fp16 ulp on < 0.1 % of the elements, everything else identical.
"""
import math

import numpy as np
import pytest
import torch

from lmdeploy_amd import _ffi
from oracle import tm_oracle as o
from tests.gpu_helpers import DevCache, dev, host, rope_table, st, ulp_diff_f16

pytestmark = pytest.mark.gpu
f16 = np.float16


# ------------------------------------------------------------------------------------------------
def test_library_sees_gpu(tm, cuda):
    assert tm.tm_device_count() >= 1
    assert tm.tm_version() >= 100


@pytest.mark.parametrize('M,H', [(1, 2048), (3, 4096), (64, 4096), (17, 8192), (5, 6144)])
def test_rmsnorm(tm, cuda, M, H):
    rng = np.random.default_rng(M * 1000 + H)
    x = rng.standard_normal((M, H)).astype(f16)
    w = (1 + 0.02 * rng.standard_normal(H)).astype(f16)
    y = torch.empty((M, H), dtype=torch.float16, device='cuda')
    _ffi.check(tm.tm_rmsnorm(y.data_ptr(), dev(x).data_ptr(), dev(w).data_ptr(), 1e-5, M, H, st()))
    ref = o.rmsnorm(x, w, 1e-5)
    d = ulp_diff_f16(host(y), ref)
    assert d.max() <= 1 and (d > 0).mean() < 1e-3


@pytest.mark.parametrize('M,H,splits,bias', [(64, 4096, 0, False), (3, 2048, 0, True), (64, 4096, 4, False),
                                             (7, 8192, 3, True)])
def test_residual_rmsnorm(tm, cuda, M, H, splits, bias):
    rng = np.random.default_rng(7 + M + H + splits)
    r = rng.standard_normal((M, H)).astype(f16)
    w = (1 + 0.02 * rng.standard_normal(H)).astype(f16)
    b = (0.1 * rng.standard_normal(H)).astype(f16) if bias else None
    if splits:
        part = (0.3 * rng.standard_normal((splits, M, H))).astype(np.float32)
        acc = np.zeros((M, H), np.float32)
        for s in range(splits):          # same sequential fp32 order as the kernel
            acc = acc + part[s]
        hcur = acc.astype(f16)
    else:
        hcur = rng.standard_normal((M, H)).astype(f16)
    r_d = dev(r)
    y = torch.empty((M, H), dtype=torch.float16, device='cuda')
    _ffi.check(tm.tm_residual_rmsnorm(y.data_ptr(), r_d.data_ptr(), None if splits else dev(hcur).data_ptr(),
                                      dev(part).data_ptr() if splits else None, splits,
                                      dev(b).data_ptr() if bias else None, dev(w).data_ptr(), 1e-5, M, H, st()))
    r_ref, y_ref = o.residual_rmsnorm(r, hcur, w, 1e-5, b)
    assert np.array_equal(host(r_d).view(np.uint16), r_ref.view(np.uint16)), 'residual stream must be bit exact'
    d = ulp_diff_f16(host(y), y_ref)
    assert d.max() <= 1 and (d > 0).mean() < 1e-3


def test_rope_table_matches_oracle(tm):
    for p in (o.RopeParam(128, 10000.0), o.RopeParam(128, 500000.0, 'llama3', 8.0, 1.0, 4.0, 8192),
              o.RopeParam(128, 1e6, 'linear', 2.0)):
        tab = rope_table(tm, 300, p)
        c, s = o.rope_cos_sin(p, np.arange(300))
        assert np.array_equal(tab[..., 0].view(np.uint16), c.view(np.uint16))
        assert np.array_equal(tab[..., 1].view(np.uint16), s.view(np.uint16))


# ------------------------------------------------------------------------------------------------
# KV: RoPE + quantise + scatter must be BIT EXACT (integer / byte / index contract)
# ------------------------------------------------------------------------------------------------
def _kv_case(bits, lens_new, hist, Hq=4, Hkv=2, layers=2, seed=0):
    rng = np.random.default_rng(seed)
    B = len(lens_new)
    L = o.BlockLayout(layers, Hkv, 128, 64, bits)
    klen = [h + n for h, n in zip(hist, lens_new)]
    nblk = [(k + 63) // 64 for k in klen]
    total = sum(nblk) + 3
    perm = rng.permutation(total)
    tables, off = [], 0
    for nb in nblk:
        tables.append(perm[off:off + nb])
        off += nb
    T = sum(lens_new)
    qkv = (rng.standard_normal((T, (Hq + 2 * Hkv) * 128)) * 1.5).astype(f16)
    # edge cases: a constant row (scale == 0 -> q = 0), a row with a big outlier
    if T > 2:
        qkv[1, Hq * 128:(Hq + 1) * 128] = f16(0.75)
        qkv[2, (Hq + Hkv) * 128 + 5] = f16(300.0)
    return L, tables, total, klen, qkv


@pytest.mark.parametrize('bits', [8, 4, 16])
@pytest.mark.parametrize('lens_new,hist', [([1, 1, 1], [0, 63, 200]), ([70, 5, 129], [0, 0, 0]), ([3, 64], [61, 10])])
def test_kv_rope_store_bit_exact(tm, cuda, bits, lens_new, hist):
    Hq, Hkv, layer = 4, 2, 1
    L, tables, total, klen, qkv = _kv_case(bits, lens_new, hist, Hq, Hkv, seed=bits + len(lens_new))
    p = o.RopeParam(128, 500000.0, 'llama3', 8.0, 1.0, 4.0, 8192)
    max_pos = max(klen) + 1
    tab = rope_table(tm, max_pos, p)
    # oracle
    oc = o.PagedKVCache(L, total)
    cu = np.concatenate([[0], np.cumsum(lens_new)]).astype(np.int32)
    q_ref = np.zeros((len(qkv), Hq, 128), f16)
    for b, n in enumerate(lens_new):
        sl = slice(cu[b], cu[b + 1])
        pos = np.arange(hist[b], hist[b] + n)
        cos, sin = o.rope_cos_sin(p, pos)
        q = qkv[sl, :Hq * 128].reshape(n, Hq, 128)
        k = qkv[sl, Hq * 128:(Hq + Hkv) * 128].reshape(n, Hkv, 128)
        v = qkv[sl, (Hq + Hkv) * 128:].reshape(n, Hkv, 128)
        q_ref[sl] = o.rope_apply(q, cos, sin)
        o.process_kv(oc, tables[b], layer, k, v, cos, sin, hist[b])
    # device
    dc = DevCache(L, total, tables)
    qkv_d = dev(qkv)
    view = dc.view(layer)
    _ffi.check(tm.tm_kv_rope_store(qkv_d.data_ptr(), Hq, dev(cu).data_ptr(), dev(np.asarray(klen, np.int32)).data_ptr(),
                                   len(lens_new), len(qkv), dev(tab).data_ptr(), max_pos, view, st()))
    got = dc.download()
    assert np.array_equal(got, oc.pool), f'cache bytes differ in {np.count_nonzero(got != oc.pool)} positions'
    q_got = host(qkv_d)[:, :Hq * 128].reshape(-1, Hq, 128)
    assert np.array_equal(q_got.view(np.uint16), q_ref.view(np.uint16)), 'RoPE(q) must be bit exact'


def test_kv_roundtrip_known_answer(tm, cuda):
    """The reference's own known-answer test (kernels/attention/test_quant.cu:32-70): integer-valued data must
    survive T -> u8/u4 -> T exactly.  Here through store (no RoPE) + flatten."""
    for bits, hi in ((8, 256), (4, 16)):
        rng = np.random.default_rng(bits)
        Hq, Hkv, T = 1, 2, 100
        L = o.BlockLayout(1, Hkv, 128, 64, bits)
        x = rng.integers(0, hi, (T, (Hq + 2 * Hkv) * 128)).astype(f16)
        # force full range per row so that scale == 1 and zero == 0 exactly
        x[:, Hq * 128::128] = 0
        x[:, Hq * 128 + 1::128] = hi - 1
        dc = DevCache(L, 4, [np.array([2, 0])])
        cu = np.array([0, T], np.int32)
        klen = np.array([T], np.int32)
        _ffi.check(tm.tm_kv_rope_store(dev(x).data_ptr(), Hq, dev(cu).data_ptr(), dev(klen).data_ptr(), 1, T, None, 0,
                                       dc.view(0), st()))
        kf = torch.zeros((Hkv, 128, 128), dtype=torch.float16, device='cuda')
        vf = torch.zeros((Hkv, 128, 128), dtype=torch.float16, device='cuda')
        _ffi.check(tm.tm_flatten_kv(kf.data_ptr(), vf.data_ptr(), 0, dev(np.array([0, 128], np.int32)).data_ptr(),
                                    dev(klen).data_ptr(), 1, T, 128, dc.view(0), st()))
        k_ref = x[:, Hq * 128:(Hq + Hkv) * 128].reshape(T, Hkv, 128).transpose(1, 0, 2)
        v_ref = x[:, (Hq + Hkv) * 128:].reshape(T, Hkv, 128).transpose(1, 0, 2)
        assert np.array_equal(host(kf)[:, :T], k_ref)
        assert np.array_equal(host(vf)[:, :T], v_ref)


@pytest.mark.parametrize('bits', [8, 4, 16])
@pytest.mark.parametrize('transpose_v', [0, 1])
def test_flatten_kv(tm, cuda, bits, transpose_v):
    rng = np.random.default_rng(bits * 2 + transpose_v)
    Hkv, layers, layer = 2, 2, 1
    klen = [130, 64, 1]
    L = o.BlockLayout(layers, Hkv, 128, 64, bits)
    nblk = [(k + 63) // 64 for k in klen]
    total = sum(nblk) + 2
    perm = rng.permutation(total)
    tables, off = [], 0
    for nb in nblk:
        tables.append(perm[off:off + nb])
        off += nb
    oc = o.PagedKVCache(L, total)
    for b, n in enumerate(klen):
        k = rng.standard_normal((n, Hkv, 128)).astype(f16)
        v = rng.standard_normal((n, Hkv, 128)).astype(f16)
        o.process_kv(oc, tables[b], layer, k, v, None, None, 0)
    dc = DevCache(L, total, tables)
    dc.upload(oc)
    koff = np.concatenate([[0], np.cumsum([nb * 64 for nb in nblk])]).astype(np.int32)
    stride = int(koff[-1])
    kf = torch.zeros((Hkv, stride, 128), dtype=torch.float16, device='cuda')
    vf = torch.zeros((Hkv, 128, stride) if transpose_v else (Hkv, stride, 128), dtype=torch.float16, device='cuda')
    _ffi.check(tm.tm_flatten_kv(kf.data_ptr(), vf.data_ptr(), transpose_v, dev(koff).data_ptr(),
                                dev(np.asarray(klen, np.int32)).data_ptr(), len(klen), max(klen), stride, dc.view(layer),
                                st()))
    K, V = host(kf), host(vf)
    if transpose_v:
        V = V.transpose(0, 2, 1)
    for b, n in enumerate(klen):
        kr, vr = o.flatten_kv(oc, tables[b], layer, n)
        assert np.array_equal(K[:, koff[b]:koff[b] + n].view(np.uint16), kr.view(np.uint16))
        assert np.array_equal(V[:, koff[b]:koff[b] + n].view(np.uint16), vr.view(np.uint16))
        if transpose_v:   # padding up to the 64-token tile must be zero (prefill attention relies on it)
            assert not V[:, koff[b] + n:koff[b + 1]].any()


# ------------------------------------------------------------------------------------------------
# attention
# ------------------------------------------------------------------------------------------------
def _fill_cache(rng, L, klen, layer, spike=False):
    nblk = [(k + 63) // 64 for k in klen]
    total = sum(nblk) + 2
    perm = rng.permutation(total)
    tables, off = [], 0
    for nb in nblk:
        tables.append(perm[off:off + nb])
        off += nb
    oc = o.PagedKVCache(L, total)
    for b, n in enumerate(klen):
        k = rng.standard_normal((n, L.kv_heads, 128)).astype(f16)
        v = rng.standard_normal((n, L.kv_heads, 128)).astype(f16)
        if spike and n > 40:
            k[n // 3] *= f16(6.0)     # forces the online-softmax rescale branch at a chosen tile
        o.process_kv(oc, tables[b], layer, k, v, None, None, 0)
    return oc, tables, total


@pytest.mark.parametrize('bits', [8, 4, 16])
@pytest.mark.parametrize('Hq,Hkv,klen,splits', [
    (8, 2, [1, 64, 65, 300], 1), (8, 2, [1, 64, 65, 300], 3), (32, 8, [1000, 37], 2), (6, 1, [129, 5], 1),
    (8, 1, [257], 4), (2, 2, [31], 1), (12, 4, [513, 64], 16), (32, 8, [1000, 37, 1089, 4], 1), (16, 4, [513, 64, 129], 1),
    (32, 2, [255, 256, 257], 1),
])
def test_decode_attention(tm, cuda, bits, Hq, Hkv, klen, splits):
    rng = np.random.default_rng(bits + Hq + sum(klen) + splits)
    layer = 1
    L = o.BlockLayout(2, Hkv, 128, 64, bits)
    oc, tables, total = _fill_cache(rng, L, klen, layer, spike=True)
    B = len(klen)
    q = rng.standard_normal((B, Hq * 128)).astype(f16)
    dc = DevCache(L, total, tables)
    dc.upload(oc)
    out = torch.zeros((B, Hq * 128), dtype=torch.float16, device='cuda')
    ws = torch.zeros(max(1, tm.tm_decode_attention_workspace(B, Hq, splits)), dtype=torch.uint8, device='cuda')
    _ffi.check(tm.tm_decode_attention(out.data_ptr(), dev(q).data_ptr(), Hq * 128,
                                      dev(np.asarray(klen, np.int32)).data_ptr(), B, Hq, 0.0, splits, ws.data_ptr(),
                                      dc.view(layer), st()))
    got = host(out).reshape(B, Hq, 128)
    for b, n in enumerate(klen):
        Ks, Vs = [], []
        for hd in range(Hkv):
            kd, vd = oc.load_dequant(tables[b], layer, hd, 0, n, 'decode')
            Ks.append(kd)
            Vs.append(vd)
        ref = o.decode_attention(q[b].reshape(Hq, 128), np.stack(Ks), np.stack(Vs), None, 1).astype(np.float32)
        err = np.abs(got[b].astype(np.float32) - ref)
        assert np.all(err <= 1e-2 * np.abs(ref) + 2e-3), f'seq {b}: max err {err.max()}'
        # and against the reference's own unfused fp64 oracle (kernels/attention/reference.cu:252-367)
        ref64 = o.attention_reference_unfused(q[b].reshape(Hq, 128), np.stack(Ks), np.stack(Vs))
        assert np.abs(got[b] - ref64).max() < 5e-3


@pytest.mark.parametrize('bits', [8, 4])
def test_decode_attention_rectangular_block_table(tm, cuda, bits):
    """The engine's block table is rectangular (sequence b at b * stride): the MFMA decode kernel then keeps a sequence's
    block pointers in one register (v_readlane per block) instead of loading offset -> pointer in front of every block.
    Same bits as the ragged-table path (which the tests above pin on the oracle): plain and fused kernels, split-KV,
    contexts of more than 64 blocks (pointer window re-fetch), cache bytes written by the fused prologue."""
    rng = np.random.default_rng(bits)
    Hq, Hkv = 8, 2
    klen = [4200, 1, 64, 65, 4097, 700, 8190]
    L = o.BlockLayout(2, Hkv, 128, 64, bits)
    nblk = [(k + 63) // 64 for k in klen]
    total = sum(nblk) + 1
    perm = rng.permutation(total)
    tables, off = [], 0
    for nb in nblk:
        tables.append(perm[off:off + nb])
        off += nb
    dc = DevCache(L, total, tables)
    # every 2 bytes of the pool = a small positive fp16: any code bytes, finite (scale, zero) parameters
    pool0 = (rng.random(dc.pool.numel() // 2) * 0.01 + 0.01).astype(f16)
    B = len(klen)
    kl = dev(np.asarray(klen, np.int32))
    q = dev(rng.standard_normal((B, Hq * 128)).astype(f16))
    qkv_n = (Hq + 2 * Hkv) * 128
    qkv = dev(rng.standard_normal((B, qkv_n)).astype(f16))
    rope = dev(rope_table(tm, 8192, o.RopeParam(128, 10000.0, 'default', 1.0, 1.0, 4.0, 8192)))
    stride = max(nblk) + 3
    res = {}
    try:
        for mode in (0, stride):
            dc.set_tables(tables, stride=mode)
            _ffi.check(tm.tm_debug_set_block_stride(mode))
            for splits in (1, 3):
                ws = torch.zeros(max(1, tm.tm_decode_attention_workspace(B, Hq, splits)), dtype=torch.uint8, device='cuda')
                dc.pool.copy_(torch.from_numpy(pool0.view(np.uint8)).cuda().view_as(dc.pool))
                out = torch.zeros((B, Hq * 128), dtype=torch.float16, device='cuda')
                _ffi.check(tm.tm_decode_attention(out.data_ptr(), q.data_ptr(), Hq * 128, kl.data_ptr(), B, Hq, 0.0, splits,
                                                  ws.data_ptr(), dc.view(1), st()))
                outf = torch.zeros((B, Hq * 128), dtype=torch.float16, device='cuda')
                _ffi.check(tm.tm_decode_attention_fused(outf.data_ptr(), qkv.data_ptr(), 0, qkv_n, rope.data_ptr(), 8192, kl.data_ptr(),
                                                        B, Hq, 0.0, splits, ws.data_ptr(), dc.view(1), st()))
                res[(mode > 0, splits)] = (host(out), host(outf), dc.download())
    finally:
        tm.tm_debug_set_block_stride(0)
    for splits in (1, 3):
        a, b = res[(False, splits)], res[(True, splits)]
        assert np.isfinite(a[0].astype(np.float32)).all() and np.abs(a[0].astype(np.float32)).max() > 0
        for x, y, what in zip(a, b, ('plain', 'fused', 'cache bytes')):
            assert np.array_equal(x.view(np.uint8), y.view(np.uint8)), f'splits={splits}: {what} differ'


@pytest.mark.parametrize('Hq,Hkv,klen,splits,qkv_splits,rope', [
    (8, 2, [1, 64, 65, 300], 1, 0, True), (8, 2, [1, 64, 65, 300], 3, 2, True), (32, 8, [1000, 37, 128, 129], 2, 4, True),
    (6, 1, [129, 5], 1, 1, False), (8, 1, [257], 4, 0, True), (12, 4, [513, 64], 16, 3, True), (64, 8, [77, 192], 2, 0, True),
    (32, 8, [1000, 37, 128, 129, 1089], 1, 4, True), (64, 8, [77, 192, 1], 1, 0, True), (16, 2, [320, 2, 65], 1, 2, False),
])
@pytest.mark.parametrize('bits', [8, 4])
def test_decode_attention_fused_prologue(tm, cuda, bits, Hq, Hkv, klen, splits, qkv_splits, rope):
    """Fused decode prologue (RoPE + K/V quantise-store inside the attention kernel) == kv_rope_store followed by
    decode attention: cache bytes bit-exact against the oracle's process_kv, output bit-identical to the unfused
    device sequence (same q bits, same cache bits, same kernel arithmetic)."""
    rng = np.random.default_rng(Hq + sum(klen) + splits + qkv_splits)
    layer = 1
    L = o.BlockLayout(2, Hkv, 128, 64, bits)
    B = len(klen)
    hist = [k - 1 for k in klen]
    # history (klen-1 tokens) through the oracle; the new token goes through the device paths
    nblk = [(k + 63) // 64 for k in klen]
    total = sum(nblk) + 2
    perm = rng.permutation(total)
    tables, off = [], 0
    for nb in nblk:
        tables.append(perm[off:off + nb])
        off += nb
    oc = o.PagedKVCache(L, total)
    for b, n in enumerate(hist):
        if n:
            k = rng.standard_normal((n, Hkv, 128)).astype(f16)
            v = rng.standard_normal((n, Hkv, 128)).astype(f16)
            o.process_kv(oc, tables[b], layer, k, v, None, None, 0)
    qkv_n = (Hq + 2 * Hkv) * 128
    if qkv_splits:
        slabs = (rng.standard_normal((qkv_splits, B, qkv_n)) / np.sqrt(qkv_splits)).astype(np.float32)
        acc = np.zeros((B, qkv_n), np.float32)
        for s_ in slabs:
            acc = acc + s_                      # in-order fp32 sum, like splitk_reduce_kernel
        qkv = acc.astype(f16)
        qkv_in = dev(slabs)
    else:
        qkv = rng.standard_normal((B, qkv_n)).astype(f16)
        qkv_in = dev(qkv.copy())
    p = o.RopeParam(128, 500000.0, 'llama3', 8.0, 1.0, 4.0, 8192)
    max_pos = max(klen) + 1
    tab = rope_table(tm, max_pos, p)
    tab_d = dev(tab)
    klen_d = dev(np.asarray(klen, np.int32))
    cu = np.arange(B + 1, dtype=np.int32)

    # unfused device sequence
    dc_a = DevCache(L, total, tables)
    dc_a.upload(oc)
    qkv_a = dev(qkv.copy())
    _ffi.check(tm.tm_kv_rope_store(qkv_a.data_ptr(), Hq, dev(cu).data_ptr(), klen_d.data_ptr(), B, B,
                                   tab_d.data_ptr() if rope else None, max_pos, dc_a.view(layer), st()))
    out_a = torch.zeros((B, Hq * 128), dtype=torch.float16, device='cuda')
    ws = torch.zeros(max(1, tm.tm_decode_attention_workspace(B, Hq, splits)), dtype=torch.uint8, device='cuda')
    _ffi.check(tm.tm_decode_attention(out_a.data_ptr(), qkv_a.data_ptr(), qkv_n, klen_d.data_ptr(), B, Hq, 0.0, splits,
                                      ws.data_ptr(), dc_a.view(layer), st()))
    # fused
    dc_b = DevCache(L, total, tables)
    dc_b.upload(oc)
    out_b = torch.zeros((B, Hq * 128), dtype=torch.float16, device='cuda')
    ws_b = torch.zeros_like(ws)
    _ffi.check(tm.tm_decode_attention_fused(out_b.data_ptr(), qkv_in.data_ptr(), qkv_splits, qkv_n,
                                            tab_d.data_ptr() if rope else None, max_pos, klen_d.data_ptr(), B, Hq, 0.0,
                                            splits, ws_b.data_ptr(), dc_b.view(layer), st()))
    # oracle cache for the new token
    for b in range(B):
        cos, sin = o.rope_cos_sin(p, np.arange(hist[b], hist[b] + 1)) if rope else (None, None)
        k = qkv[b:b + 1, Hq * 128:(Hq + Hkv) * 128].reshape(1, Hkv, 128)
        v = qkv[b:b + 1, (Hq + Hkv) * 128:].reshape(1, Hkv, 128)
        o.process_kv(oc, tables[b], layer, k, v, cos, sin, hist[b])
    pool_a, pool_b = dc_a.download(), dc_b.download()
    assert np.array_equal(pool_a, oc.pool)
    assert np.array_equal(pool_b, oc.pool), f'fused cache bytes differ in {np.count_nonzero(pool_b != oc.pool)} positions'
    ga, gb = host(out_a), host(out_b)
    assert np.array_equal(ga.view(np.uint16), gb.view(np.uint16)), f'max diff {np.abs(ga.astype(np.float32) - gb.astype(np.float32)).max()}'


def test_decode_attention_block_table_permutation_invariance(tm, cuda):
    """test_attention.cu:70-141: a shuffled block table must give identical output."""
    rng = np.random.default_rng(5)
    L = o.BlockLayout(1, 2, 128, 64, 8)
    klen = [200]
    oc, tables, total = _fill_cache(rng, L, klen, 0)
    q = rng.standard_normal((1, 8 * 128)).astype(f16)
    outs = []
    for trial in range(2):
        if trial == 1:   # move the blocks elsewhere in the pool
            perm = rng.permutation(total)
            pool2 = np.zeros_like(oc.pool)
            pool2[perm] = oc.pool
            oc.pool = pool2
            tables = [perm[t] for t in tables]
        dc = DevCache(L, total, tables)
        dc.upload(oc)
        out = torch.zeros((1, 8 * 128), dtype=torch.float16, device='cuda')
        _ffi.check(tm.tm_decode_attention(out.data_ptr(), dev(q).data_ptr(), 8 * 128, dev(np.asarray(klen, np.int32)).data_ptr(),
                                          1, 8, 0.0, 1, None, dc.view(0), st()))
        outs.append(host(out).copy())
    assert np.array_equal(outs[0], outs[1])


@pytest.mark.parametrize('Hq,Hkv,qlens,hist', [(4, 2, [70, 5, 129], [0, 0, 0]), (8, 8, [33], [95]), (6, 2, [1, 64], [0, 64]),
                                               (8, 2, [200, 37], [0, 10]), (32, 8, [130], [62]), (16, 2, [64, 96], [0, 32]),
                                               (32, 8, [1024, 700], [0, 324]), (8, 4, [300], [1000])])
def test_prefill_attention(tm, cuda, Hq, Hkv, qlens, hist):
    rng = np.random.default_rng(Hq + sum(qlens))
    B = len(qlens)
    klen = [h + n for h, n in zip(hist, qlens)]
    koff = np.concatenate([[0], np.cumsum([((k + 63) // 64) * 64 for k in klen])]).astype(np.int32)
    stride = int(koff[-1])
    cu = np.concatenate([[0], np.cumsum(qlens)]).astype(np.int32)
    T = int(cu[-1])
    q = rng.standard_normal((T, Hq * 128)).astype(f16)
    K = np.zeros((Hkv, stride, 128), f16)
    Vt = np.zeros((Hkv, 128, stride), f16)
    Ks, Vs = [], []
    for b, n in enumerate(klen):
        k = rng.standard_normal((Hkv, n, 128)).astype(f16)
        v = rng.standard_normal((Hkv, n, 128)).astype(f16)
        K[:, koff[b]:koff[b] + n] = k
        K[:, koff[b] + n:koff[b + 1]] = f16(np.nan)      # garbage past the context must be masked, not multiplied
        Vt[:, :, koff[b]:koff[b] + n] = v.transpose(0, 2, 1)
        Ks.append(k)
        Vs.append(v)
    out = torch.zeros((T, Hq * 128), dtype=torch.float16, device='cuda')
    _ffi.check(tm.tm_prefill_attention(out.data_ptr(), dev(q).data_ptr(), Hq * 128, dev(K).data_ptr(), dev(Vt).data_ptr(),
                                       stride, dev(cu).data_ptr(), dev(koff).data_ptr(),
                                       dev(np.asarray(klen, np.int32)).data_ptr(), B, max(qlens), Hq, Hkv, 0.0, st()))
    got = host(out)
    for b, n in enumerate(qlens):
        ref = o.prefill_attention(q[cu[b]:cu[b + 1]].reshape(n, Hq, 128), Ks[b], Vs[b], hist[b]).reshape(n, -1)
        err = np.abs(got[cu[b]:cu[b + 1]].astype(np.float32) - ref.astype(np.float32))
        assert np.all(err <= 1e-2 * np.abs(ref.astype(np.float32)) + 2e-3), f'seq {b}: max err {err.max()}'


# ------------------------------------------------------------------------------------------------
# W4A16 linear (shapes from tests/turbomind/linear/models.yaml, scaled; batches from cases.py:26-69)
# ------------------------------------------------------------------------------------------------
_QCACHE = {}


def _make_linear(tm, rng, K, N):
    if (K, N) not in _QCACHE:       # quantising on the CPU is the slow part: once per shape
        w = (np.random.default_rng(K * 7 + N).standard_normal((K, N)) * (0.1 / math.sqrt(K))).astype(f16)
        q, s, z, _ = o.quantize_groupwise_u4(w, 128)
        _QCACHE[(K, N)] = (q, s, z, o.w4a16_dequant(q, s, z).astype(np.float32))
    q, s, z, _ = _QCACHE[(K, N)]
    h = _ffi.C.c_void_p()
    _ffi.check(tm.tm_linear_create(_ffi.C.byref(h), K, N, 0, 128))
    _ffi.check(tm.tm_linear_prepare(h, dev(o.pack_u4_row(q)).data_ptr(), dev(s).data_ptr(), dev(z).data_ptr(), st()))
    torch.cuda.synchronize()
    return h, (q, s, z)


@pytest.mark.parametrize('K,N', [(4096, 6144), (1024, 512), (256, 64), (1792, 4096), (384, 48)])
@pytest.mark.parametrize('M', [1, 3, 16, 17, 64, 65, 256])
def test_w4a16_linear(tm, cuda, K, N, M):
    rng = np.random.default_rng(K + N + M)
    h, (q, s, z) = _make_linear(tm, rng, K, N)
    x = rng.standard_normal((M, K)).astype(f16)
    ref = x.astype(np.float32) @ _QCACHE[(K, N)][3]
    ws = torch.zeros(max(1, tm.tm_linear_workspace(h, M)), dtype=torch.uint8, device='cuda')
    x_d = dev(x)
    for nt, splits, waves in ((0, 0, 0), (1, 1, 4), (2, 2, 4), (4, 4, 4), (1, 2, 8), (2, 1, 8), (2, 16, 8), (1, 1, 0x108), (2, 2, 0x108), (2, 4, 0x108)):
        y = torch.zeros((M, N), dtype=torch.float16, device='cuda')
        _ffi.check(tm.tm_linear_forward(h, x_d.data_ptr(), K, y.data_ptr(), N, M, 0, nt, splits, waves, ws.data_ptr(), st()))
        err = np.abs(host(y).astype(np.float32) - ref)
        tol = 2e-3 + 2.0**-10 * np.abs(ref)
        assert np.all(err <= tol), f'nt={nt} splits={splits} waves={waves}: max err {err.max()} (ref max {np.abs(ref).max()})'
    _ffi.check(tm.tm_linear_destroy(h))


@pytest.mark.parametrize('K,N,gated', [(4096, 6144, 0), (1792, 4096, 0), (1024, 512, 0), (4096, 1024, 1), (1024, 2048, 1)])
@pytest.mark.parametrize('M', [257, 300, 512, 1000, 2500])
def test_w4a16_linear_prefill_tiles(tm, cuda, K, N, M, gated):
    """The prefill-shaped path (M > 256): 128-row x 512-column workgroup tiles with two weight fragments per x-fragment read
    (0x205, the default from M = 257) and 128 x 256 tiles (0x204), the round-1 tile (explicit nt / waves), ragged last row
    block, column counts that do not fill the last workgroup, the automatic split-K of mid-size M (fewer than 256 workgroups otherwise) and explicit split counts, plain and gated-SiLU
    epilogues -- same oracle and tolerance as the decode shapes."""
    rng = np.random.default_rng(K + N + M + 11)
    h, (q, s, z) = _make_linear(tm, rng, K, N)
    x = rng.standard_normal((M, K)).astype(f16)
    ref = (o.w4a16_linear_gated_silu(x, q, s, z) if gated else x.astype(np.float32) @ _QCACHE[(K, N)][3]).astype(np.float32)
    ws = torch.zeros(max(1, tm.tm_linear_workspace(h, M)), dtype=torch.uint8, device='cuda')
    x_d = dev(x)
    # 0x20c = shape 12 (gemm_prefill.hip): 256 x 256 tiles, weights dequantised once per workgroup tile through LDS
    for nt, splits, waves in ((0, 0, 0), (0, 1, 0x204), (0, 2, 0x204), (0, 3, 0x204), (0, 1, 0x205), (0, 2, 0x205), (0, 5, 0x205),
                              (0, 1, 0x20c), (0, 2, 0x20c), (0, 3, 0x20c), (0, 1, 0x20d), (2, 1, 8), (2, 2, 8), (2, 4, 8), (4, 1, 4)):   # 0x20d = shape 13: the fp16 image
        if splits > K // 128:
            continue
        y = torch.zeros((M, N // 2 if gated else N), dtype=torch.float16, device='cuda')
        _ffi.check(tm.tm_linear_forward(h, x_d.data_ptr(), K, y.data_ptr(), y.shape[1], M, gated, nt, splits, waves, ws.data_ptr(), st()))
        err = np.abs(host(y).astype(np.float32) - ref)
        assert np.all(err <= 2e-3 + 2.0**-9 * np.abs(ref)), f'nt={nt} splits={splits} waves={waves}: max err {err.max()}'
    _ffi.check(tm.tm_linear_destroy(h))


@pytest.mark.parametrize('K,N,M,splits,waves', [(1792, 4096, 1000, 3, 0x204), (1792, 4096, 2500, 3, 0x204), (1536, 4096, 64, 1, 0x200),
                                                (1792, 4096, 1000, 3, 0x20c), (4096, 6144, 2500, 1, 0x20c), (1536, 4096, 300, 2, 0x20c),
                                                (1792, 4096, 1000, 1, 0x20d), (4096, 6144, 2500, 1, 0x20d), (1536, 4096, 300, 1, 0x20d),
                                                (1536, 4096, 64, 3, 0x200), (4608, 4096, 64, 1, 0x201), (1536, 2048, 32, 1, 0x203)])
def test_w4a16_odd_stage_count_is_stable(tm, cuda, K, N, M, splits, waves):
    """Regression (round 2): with an odd number of LDS stages per k slice the asynchronous x stage that the LDS-DMA fetched
    behind the LAST stage could land on top of the epilogue's reduction image -- intermittently wrong rows at the end of a
    row block.  Twelve launches of each odd-stage configuration: every one equals the oracle and the first one bit for bit."""
    rng = np.random.default_rng(K + N + M)
    h, (q, s, z) = _make_linear(tm, rng, K, N)
    x = rng.standard_normal((M, K)).astype(f16)
    ref = x.astype(np.float32) @ _QCACHE[(K, N)][3]
    ws = torch.zeros(max(1, tm.tm_linear_workspace(h, M)), dtype=torch.uint8, device='cuda')
    x_d = dev(x)
    first = None
    for rep in range(12):
        y = torch.zeros((M, N), dtype=torch.float16, device='cuda')
        _ffi.check(tm.tm_linear_forward(h, x_d.data_ptr(), K, y.data_ptr(), N, M, 0, 0, splits, waves, ws.data_ptr(), st()))
        got = host(y)
        err = np.abs(got.astype(np.float32) - ref)
        assert np.all(err <= 2e-3 + 2.0**-9 * np.abs(ref)), f'launch {rep}: max err {err.max()}'
        if first is None:
            first = got
        assert np.array_equal(got.view(np.uint16), first.view(np.uint16)), f'launch {rep} differs from launch 0'
    _ffi.check(tm.tm_linear_destroy(h))


@pytest.mark.parametrize('K,N,gated', [(4096, 6144, 0), (1792, 4096, 0), (1024, 512, 0), (4096, 1024, 1), (384, 64, 0)])
@pytest.mark.parametrize('M', [1, 32, 33, 50, 64, 100, 256])
@pytest.mark.parametrize('shape', [6, 7, 8, 9])
def test_w4a16_row_half_tiles(tm, cuda, K, N, M, gated, shape):
    """shapes 6 .. 9: the decode tilings on 32-row blocks (6: 64-column tiles over the WHOLE k range, no split-K slabs: the
    k-phases meet on chip -- the measured choice for the narrow projections) plus their split-K forms"""
    rng = np.random.default_rng(K + N + M + 6)
    h, (q, s, z) = _make_linear(tm, rng, K, N)
    x = rng.standard_normal((M, K)).astype(f16)
    ref = (o.w4a16_linear_gated_silu(x, q, s, z) if gated else x.astype(np.float32) @ _QCACHE[(K, N)][3]).astype(np.float32)
    ws = torch.zeros(max(1, tm.tm_linear_workspace(h, M)), dtype=torch.uint8, device='cuda')
    x_d = dev(x)
    for splits in (1, 2, 3):
        if splits > max(1, K // 512):
            continue
        y = torch.zeros((M, N // 2 if gated else N), dtype=torch.float16, device='cuda')
        _ffi.check(tm.tm_linear_forward(h, x_d.data_ptr(), K, y.data_ptr(), y.shape[1], M, gated, 0, splits, 0x200 | shape, ws.data_ptr(), st()))
        err = np.abs(host(y).astype(np.float32) - ref)
        assert np.all(err <= 2e-3 + 2.0**-9 * np.abs(ref)), f'shape {shape} splits={splits}: max err {err.max()}'
    _ffi.check(tm.tm_linear_destroy(h))


@pytest.mark.parametrize('K,N,gated', [(4096, 6144, 0), (1792, 4096, 0), (1024, 512, 0), (4096, 1024, 1), (384, 64, 0), (14336, 256, 0)])
@pytest.mark.parametrize('M', [1, 32, 33, 50, 64])
def test_w4a16_loader_consumer_tiles(tm, cuda, K, N, M, gated):
    """shape 11 (gemm_decode_lc.hip): 4 loader waves stage the activations, 8 consumer waves (2 column halves x 4 k-phases)
    stream the weights through a 3-stage register ring.  K slices that are not whole stages (14 and 3 k-blocks), column tiles
    that do not fill the workgroup (N = 64), both row-half instantiations, plain / gated / split-K epilogues; the same oracle
    and tolerance as the other decode tiles, and every launch bit-identical to the first."""
    rng = np.random.default_rng(K + N + M + 11)
    h, (q, s, z) = _make_linear(tm, rng, K, N)
    x = rng.standard_normal((M, K)).astype(f16)
    ref = (o.w4a16_linear_gated_silu(x, q, s, z) if gated else x.astype(np.float32) @ _QCACHE[(K, N)][3]).astype(np.float32)
    ws = torch.zeros(max(1, tm.tm_linear_workspace(h, M)), dtype=torch.uint8, device='cuda')
    x_d = dev(x)
    for splits in (1, 2, 3, 7):
        if splits > max(1, K // 512):
            continue
        first = None
        for rep in range(3):
            y = torch.zeros((M, N // 2 if gated else N), dtype=torch.float16, device='cuda')
            _ffi.check(tm.tm_linear_forward(h, x_d.data_ptr(), K, y.data_ptr(), y.shape[1], M, gated, 0, splits, 0x200 | 11, ws.data_ptr(), st()))
            got = host(y)
            err = np.abs(got.astype(np.float32) - ref)
            assert np.all(err <= 2e-3 + 2.0**-9 * np.abs(ref)), f'splits={splits} launch {rep}: max err {err.max()}'
            if first is None:
                first = got
            assert np.array_equal(got.view(np.uint16), first.view(np.uint16)), f'splits={splits}: launch {rep} differs from launch 0'
    _ffi.check(tm.tm_linear_destroy(h))


@pytest.mark.parametrize('K,N,gated', [(4096, 1024, 1), (2048, 512, 0), (1792, 4096, 0), (4096, 6144, 1)])
@pytest.mark.parametrize('M', [1, 33, 64])
def test_w4a16_split_k_merged_in_launch(tm, cuda, K, N, M, gated):
    """Round 6: shapes 16 + s -- the split-K slices of a decode tile merged INSIDE the launch by the last-arriving slice
    of each column tile, for the fp16 and the gated-SiLU epilogue (what a 256-column w1w3 tile with a cross-CU k-split needs) -- and
    shape 10, the 256-column x 2-k-phase tile on 2-k-block stages.  Oracle + tolerance of the other decode tiles; the fp16 epilogue is
    bit-identical to the same tile's slabs summed by the reduce kernel (slice order, fp32); three launches bit-identical (the arrival
    counters are left at zero by every launch)."""
    rng = np.random.default_rng(K + N + M + 16)
    h, (q, s, z) = _make_linear(tm, rng, K, N)
    x = rng.standard_normal((M, K)).astype(f16)
    ref = (o.w4a16_linear_gated_silu(x, q, s, z) if gated else x.astype(np.float32) @ _QCACHE[(K, N)][3]).astype(np.float32)
    ws = torch.zeros(max(1, tm.tm_linear_workspace(h, M)), dtype=torch.uint8, device='cuda')
    x_d = dev(x)
    cols = N // 2 if gated else N
    for base in (0, 3, 6, 10):
        for splits in (1, 2, 4):
            if splits > max(1, K // 512):
                continue
            plain = torch.zeros((M, cols), dtype=torch.float16, device='cuda')
            _ffi.check(tm.tm_linear_forward(h, x_d.data_ptr(), K, plain.data_ptr(), cols, M, gated, 0, splits, 0x200 | base, ws.data_ptr(), st()))
            err = np.abs(host(plain).astype(np.float32) - ref)
            assert np.all(err <= 2e-3 + 2.0**-9 * np.abs(ref)), f'shape {base} splits {splits}: max err {err.max()}'
            if splits == 1:
                continue
            first = None
            for rep in range(3):
                y = torch.zeros((M, cols), dtype=torch.float16, device='cuda')
                _ffi.check(tm.tm_linear_forward(h, x_d.data_ptr(), K, y.data_ptr(), cols, M, gated, 0, splits, 0x200 | (16 + base), ws.data_ptr(), st()))
                got = host(y)
                err = np.abs(got.astype(np.float32) - ref)
                assert np.all(err <= 2e-3 + 2.0**-9 * np.abs(ref)), f'merged shape {base} splits {splits} launch {rep}: max err {err.max()}'
                if first is None:
                    first = got
                assert np.array_equal(got.view(np.uint16), first.view(np.uint16)), f'merged shape {base} splits {splits}: launch {rep} differs'
            if not gated:   # same slabs, same order: the reduce kernel's bits
                assert np.array_equal(first.view(np.uint16), host(plain).view(np.uint16)), f'merged shape {base} splits {splits} != slabs + reduce'
    _ffi.check(tm.tm_linear_destroy(h))


def test_w4a16_loader_consumer_identity(tm, cuda):
    """x = rows of the identity through shape 11: the dequantised weights come back bit for bit (operand = the reference's)."""
    rng = np.random.default_rng(5)
    K, N = 512, 160
    h, (q, s, z) = _make_linear(tm, rng, K, N)
    w = o.w4a16_dequant(q, s, z)
    x = np.zeros((64, K), f16)
    rows = rng.permutation(K)[:64]
    x[np.arange(64), rows] = 1
    y = torch.zeros((64, N), dtype=torch.float16, device='cuda')
    _ffi.check(tm.tm_linear_forward(h, dev(x).data_ptr(), K, y.data_ptr(), N, 64, 0, 0, 1, 0x200 | 11, None, st()))
    assert np.array_equal(host(y).view(np.uint16), w[rows].view(np.uint16)), 'dequantised weights must be bit exact'
    _ffi.check(tm.tm_linear_destroy(h))


@pytest.mark.parametrize('K,N,M', [(4096, 1024, 64), (512, 256, 5), (1024, 2048, 130)])
def test_w4a16_gated_silu(tm, cuda, K, N, M):
    rng = np.random.default_rng(K + N + M + 1)
    h, (q, s, z) = _make_linear(tm, rng, K, N)
    x = (rng.standard_normal((M, K)) * 3).astype(f16)
    ref = o.w4a16_linear_gated_silu(x, q, s, z).astype(np.float32)
    ws = torch.zeros(max(1, tm.tm_linear_workspace(h, M)), dtype=torch.uint8, device='cuda')
    for nt, splits, waves in ((0, 0, 0), (2, 1, 4), (4, 2, 4), (2, 2, 8), (2, 1, 0x108)):
        y = torch.zeros((M, N // 2), dtype=torch.float16, device='cuda')
        _ffi.check(tm.tm_linear_forward(h, dev(x).data_ptr(), K, y.data_ptr(), N // 2, M, 1, nt, splits, waves, ws.data_ptr(), st()))
        err = np.abs(host(y).astype(np.float32) - ref)
        assert np.all(err <= 2e-3 + 2.0**-9 * np.abs(ref)), f'max err {err.max()}'
    _ffi.check(tm.tm_linear_destroy(h))


@pytest.mark.parametrize('M,H', [(128, 1024), (100, 1024), (65, 1024), (128, 6144)])
def test_w4a16_folded_norm_batch_128(tm, cuda, M, H):
    """Round 6 (BASELINE config 3 = batch 128): the folded RMSNorm for 64 < M <= 128 rows.  Producers run the 32-row-block
    tiles 6..9 (one arrival counter and one row of sums per (column tile, row block); in-launch slab merge included), consumers those tiles or
    the 128-row tile 4 (every weight unit read once).  Same claims as test_w4a16_folded_norm: residual stream bit-identical to the unfused
    device sequence, xg and the sums of squares exact on the device's own r, consumer inside the unfused bound."""
    rng = np.random.default_rng(200 + M + H)
    K1, N2 = 2048, 512
    eps = 1e-5
    hp, (qp, sp, zp) = _make_linear(tm, rng, K1, H)
    hc, (qc, sc, zc) = _make_linear(tm, rng, H, N2)
    x = (rng.standard_normal((M, K1)) * 2).astype(f16)
    resid0 = rng.standard_normal((M, H)).astype(f16)
    resid0[:, 5] *= 40.0
    g = (1.0 + 0.2 * rng.standard_normal(H)).astype(f16)
    ws = torch.zeros(int(tm.tm_linear_fold_workspace(hp, M)) + M * H * 2 + 4096, dtype=torch.uint8, device='cuda')
    wsc = torch.zeros(max(1, int(tm.tm_linear_workspace(hc, M))), dtype=torch.uint8, device='cuda')
    x_d, g_d = dev(x), dev(g)
    r_ref, n_ref = o.residual_rmsnorm(resid0, o.w4a16_linear(x, qp, sp, zp), g, eps)
    big = H > 1024
    for gated in (0, 1):
        ref_y = (o.w4a16_linear_gated_silu(n_ref, qc, sc, zc) if gated else o.w4a16_linear(n_ref, qc, sc, zc)).astype(np.float32)
        for shape, splits in (((6, 2), (7, 1)) if big else ((6, 1), (6, 2), (7, 1), (7, 2), (8, 4), (9, 1), (-1, 0))):
            r_u = dev(resid0.copy())
            n_u = torch.zeros((M, H), dtype=torch.float16, device='cuda')
            us, usp = (shape, splits) if shape >= 0 else (7, 1)
            _ffi.check(tm.tm_linear_residual_norm(hp, x_d.data_ptr(), K1, n_u.data_ptr(), r_u.data_ptr(), g_d.data_ptr(), eps, M, us, usp,
                                                  ws.data_ptr(), st()))
            r_f = dev(resid0.copy())
            xg = torch.zeros((M, H), dtype=torch.float16, device='cuda')
            ss = torch.full((H // 64, M), float('nan'), dtype=torch.float32, device='cuda')
            tiles = _ffi.C.c_int(0)
            for rep in range(2):
                r_f.copy_(torch.from_numpy(resid0).cuda())
                _ffi.check(tm.tm_linear_fold_produce(hp, x_d.data_ptr(), K1, xg.data_ptr(), r_f.data_ptr(), g_d.data_ptr(), ss.data_ptr(),
                                                     _ffi.C.byref(tiles), M, shape, splits, ws.data_ptr(), st()))
            torch.cuda.synchronize()
            r_dev = host(r_f)
            assert np.array_equal(r_dev.view(np.uint16), host(r_u).view(np.uint16)), f'residual stream differs from the unfused sequence {shape, splits}'
            want_xg = np.clip(r_dev.astype(np.float32) * g.astype(np.float32), -65504, 65504).astype(f16)
            assert np.array_equal(host(xg).view(np.uint16), want_xg.view(np.uint16)), f'xg {shape, splits}'
            ss_h = host(ss)[:tiles.value].astype(np.float64).sum(0)
            want_ss = (r_dev.astype(np.float64)**2).sum(-1)
            assert 1 <= tiles.value <= H // 64 and np.all(np.abs(ss_h - want_ss) <= 1e-5 * want_ss), f'sums of squares {shape, splits}'
            for cshape, csplits in ((4, 1), (6, 1), (7, 1), (-1, 0)):
                y = torch.zeros((M, N2 // 2 if gated else N2), dtype=torch.float16, device='cuda')
                _ffi.check(tm.tm_linear_fold_consume(hc, xg.data_ptr(), H, y.data_ptr(), y.shape[1], M, gated, ss.data_ptr(), tiles.value, H, eps,
                                                     cshape, csplits, wsc.data_ptr(), st()))
                err = np.abs(host(y).astype(np.float32) - ref_y)
                assert np.all(err <= 2e-3 + 2.0**-9 * np.abs(ref_y)), f'folded consumer {cshape, csplits} after {shape, splits}: max err {err.max()}'
    _ffi.check(tm.tm_linear_destroy(hp))
    _ffi.check(tm.tm_linear_destroy(hc))


@pytest.mark.parametrize('M,H', [(64, 1024), (33, 1024), (7, 1024), (64, 6144)])
def test_w4a16_folded_norm(tm, cuda, M, H):
    """RMSNorm folded into the two decode GEMMs around it (tm_linear_fold_produce / _consume; the engine's tp = 1 decode step):
    the producing row-parallel GEMM updates the residual stream in its epilogue -- BIT-IDENTICAL to the unfused device sequence
    (tm_linear_residual_norm: GEMM [+ fp32 slabs] -> reduce-norm kernel), for every tile and split-K count, the in-launch slab
    merge by the last-arriving slice included -- writes xg = h(f32(r) * f32(g)) (bit-exact against numpy on the device's own r)
    and the per-tile sums of squares; the consuming GEMM applies inv[m] to its fp32 accumulators.
    Tolerance of the consumer's output against the oracle's unfused sequence h(h(r * inv) * g) . W: the SAME bound as the unfused
    device sequence (2e-3 + 2^-9 |ref|) -- the folded form carries one fp16 rounding of the normalised activations instead of two
    -- and its mean error against the exact (fp64) result must not exceed the unfused device sequence's."""
    rng = np.random.default_rng(100 + M + H)
    # producer: [K1] -> H (wo / w2 role, 16 k-blocks: up to 4 slices of one stage); consumer: H -> N2 (w_qkv / w1w3 role).  H = 6144 (the
    # InternLM2-20B width): 96 column tiles -- more than the 8 sums per thread an 8-wave consumer holds in registers (its tail loop runs)
    K1, N2 = 2048, (1024 if H == 1024 else 512)
    big = H > 1024
    eps = 1e-5
    hp, (qp, sp, zp) = _make_linear(tm, rng, K1, H)
    hc, (qc, sc, zc) = _make_linear(tm, rng, H, N2)
    wp, wc = _QCACHE[(K1, H)][3], _QCACHE[(H, N2)][3]
    x = (rng.standard_normal((M, K1)) * 2).astype(f16)
    resid0 = rng.standard_normal((M, H)).astype(f16)
    resid0[:, 5] *= 40.0                                  # an outlier channel, as real residual streams have
    g = (1.0 + 0.2 * rng.standard_normal(H)).astype(f16)
    ws = torch.zeros(int(tm.tm_linear_fold_workspace(hp, M)) + M * H * 2 + 4096, dtype=torch.uint8, device='cuda')
    wsc = torch.zeros(max(1, int(tm.tm_linear_workspace(hc, M))), dtype=torch.uint8, device='cuda')
    x_d, g_d = dev(x), dev(g)
    # the oracle's unfused sequence and the exact result
    r_ref, n_ref = o.residual_rmsnorm(resid0, o.w4a16_linear(x, qp, sp, zp), g, eps)
    exact_r = r_ref.astype(np.float64)
    exact_n = exact_r / np.sqrt((exact_r**2).mean(-1, keepdims=True) + eps) * g.astype(np.float64)
    for gated in (0, 1):
        ref_y = (o.w4a16_linear_gated_silu(n_ref, qc, sc, zc) if gated else o.w4a16_linear(n_ref, qc, sc, zc)).astype(np.float32)
        acc = exact_n @ wc.astype(np.float64)
        exact_y = (acc[:, 0::2] / (1.0 + np.exp(-acc[:, 0::2])) * acc[:, 1::2]) if gated else acc
        for shape, splits in (((3, 1), (3, 4), (6, 2)) if big else ((3, 1), (3, 2), (3, 4), (6, 1), (6, 2), (0, 1), (0, 2), (2, 2), (7, 2), (8, 1), (-1, 0))):
            # unfused device sequence with the same tile: the residual stream to reproduce bit for bit
            r_u = dev(resid0.copy())
            n_u = torch.zeros((M, H), dtype=torch.float16, device='cuda')
            _ffi.check(tm.tm_linear_residual_norm(hp, x_d.data_ptr(), K1, n_u.data_ptr(), r_u.data_ptr(), g_d.data_ptr(), eps, M, shape, splits,
                                                  ws.data_ptr(), st()))
            y_u = torch.zeros((M, N2 // 2 if gated else N2), dtype=torch.float16, device='cuda')
            _ffi.check(tm.tm_linear_forward(hc, n_u.data_ptr(), H, y_u.data_ptr(), y_u.shape[1], M, gated, 0, 1, 0x200 | 3, wsc.data_ptr(), st()))
            # folded
            r_f = dev(resid0.copy())
            xg = torch.zeros((M, H), dtype=torch.float16, device='cuda')
            ss = torch.full((H // 64, M), float('nan'), dtype=torch.float32, device='cuda')
            tiles = _ffi.C.c_int(0)
            for rep in range(2):      # twice: the arrival counters must come back to zero
                r_f.copy_(torch.from_numpy(resid0).cuda())
                _ffi.check(tm.tm_linear_fold_produce(hp, x_d.data_ptr(), K1, xg.data_ptr(), r_f.data_ptr(), g_d.data_ptr(), ss.data_ptr(),
                                                     _ffi.C.byref(tiles), M, shape, splits, ws.data_ptr(), st()))
            torch.cuda.synchronize()
            r_dev = host(r_f)
            assert np.array_equal(r_dev.view(np.uint16), host(r_u).view(np.uint16)), f'residual stream differs from the unfused sequence {shape, splits}'
            # vs the oracle: the device's fp32 accumulation order may round the GEMM output h(acc) one fp16 step away from numpy's on a
            # few elements (then r differs by that step of h(acc), whatever r's own magnitude is)
            hc_ref = o.w4a16_linear(x, qp, sp, zp).astype(np.float32)
            dr = np.abs(r_dev.astype(np.float32) - r_ref.astype(np.float32))
            assert np.all(dr <= 2.0**-10 * np.abs(hc_ref) + 2.0**-10 * np.abs(r_ref.astype(np.float32)) + 1e-7) and (dr > 0).mean() < 2e-3, \
                'residual stream vs the oracle'
            want_xg = np.clip(r_dev.astype(np.float32) * g.astype(np.float32), -65504, 65504).astype(f16)
            assert np.array_equal(host(xg).view(np.uint16), want_xg.view(np.uint16)), f'xg {shape, splits}'
            ss_h = host(ss)[:tiles.value].astype(np.float64).sum(0)
            want_ss = (r_dev.astype(np.float64)**2).sum(-1)
            assert 1 <= tiles.value <= H // 64 and np.all(np.abs(ss_h - want_ss) <= 1e-5 * want_ss), f'sums of squares {shape, splits}'
            for cshape, csplits in (((3, 1), (0, 1), (6, 1)) if big else ((3, 1), (0, 1), (6, 1), (3, 2), (-1, 0))):
                y = torch.zeros((M, N2 // 2 if gated else N2), dtype=torch.float16, device='cuda')
                _ffi.check(tm.tm_linear_fold_consume(hc, xg.data_ptr(), H, y.data_ptr(), y.shape[1], M, gated, ss.data_ptr(), tiles.value, H, eps,
                                                     cshape, csplits, wsc.data_ptr(), st()))
                yf = host(y).astype(np.float32)
                err = np.abs(yf - ref_y)
                assert np.all(err <= 2e-3 + 2.0**-9 * np.abs(ref_y)), f'folded consumer {cshape, csplits} after {shape, splits}: max err {err.max()}'
                e_f = np.abs(yf - exact_y).mean()
                e_u = np.abs(host(y_u).astype(np.float32) - exact_y).mean()
                assert e_f <= 1.05 * e_u + 1e-6, f'folded form is further from the exact result than the unfused one: {e_f} vs {e_u}'
    _ffi.check(tm.tm_linear_destroy(hp))
    _ffi.check(tm.tm_linear_destroy(hc))


@pytest.mark.parametrize('tp', [2, 4])
def test_tp_sharded_ffn_sequential_shards(tm, cuda, tp):
    """tensor-parallel numerics validated on ONE GPU -- column-parallel w1w3 (gate/up interleaved, fused
    SiLU epilogue) and row-parallel w2 run shard by shard, the partial [M,H] outputs summed on the host (what the
    RCCL all-reduce does), against the unsharded FFN.  Group-wise quantisation survives the row split because K/tp
    stays a multiple of 128."""
    rng = np.random.default_rng(tp)
    H, I, M = 512, 1024, 64

    def lin(q, s, z):
        K, N = q.shape
        h = _ffi.C.c_void_p()
        _ffi.check(tm.tm_linear_create(_ffi.C.byref(h), K, N, 0, 128))
        _ffi.check(tm.tm_linear_prepare(h, dev(o.pack_u4_row(q)).data_ptr(), dev(s).data_ptr(), dev(z).data_ptr(), st()))
        torch.cuda.synchronize()
        return h

    def fwd(h, x, K, N, gated):
        ws = torch.zeros(max(1, tm.tm_linear_workspace(h, M)), dtype=torch.uint8, device='cuda')
        y = torch.zeros((M, N // 2 if gated else N), dtype=torch.float16, device='cuda')
        _ffi.check(tm.tm_linear_forward(h, x.data_ptr(), K, y.data_ptr(), y.shape[1], M, gated, 0, 0, 0, ws.data_ptr(), st()))
        torch.cuda.synchronize()
        return y

    w13 = (rng.standard_normal((H, 2 * I)) * (0.1 / math.sqrt(H))).astype(f16)
    w2 = (rng.standard_normal((I, H)) * (0.1 / math.sqrt(I))).astype(f16)
    q13, s13, z13, _ = o.quantize_groupwise_u4(w13, 128)
    q2, s2, z2, _ = o.quantize_groupwise_u4(w2, 128)
    x = dev(rng.standard_normal((M, H)).astype(f16))
    h13, h2 = lin(q13, s13, z13), lin(q2, s2, z2)
    full = host(fwd(h2, fwd(h13, x, H, 2 * I, 1), I, H, 0)).astype(np.float32)
    part = np.zeros((M, H), np.float32)
    for r in range(tp):
        c = slice(r * 2 * I // tp, (r + 1) * 2 * I // tp)       # interleaved (gate, up) pairs stay together
        k = slice(r * I // tp, (r + 1) * I // tp)
        g = k.start // 128, k.stop // 128
        a = lin(q13[:, c], s13[:, c], z13[:, c])
        b = lin(q2[k], s2[g[0]:g[1]], z2[g[0]:g[1]])
        mid = fwd(a, x, H, 2 * I // tp, 1)
        ref_mid = host(fwd(h13, x, H, 2 * I, 1))[:, k]       # split-K depth may differ between shard and full op
        dm = np.abs(host(mid).astype(np.float32) - ref_mid.astype(np.float32))
        assert np.all(dm <= 1e-4 + 2.0**-9 * np.abs(ref_mid.astype(np.float32))), 'column shard != slice of the full op'
        part += host(fwd(b, mid, I // tp, H, 0)).astype(np.float32)
        _ffi.check(tm.tm_linear_destroy(a))
        _ffi.check(tm.tm_linear_destroy(b))
    err = np.abs(part - full)
    assert np.all(err <= 2e-3 + tp * 2.0**-10 * np.abs(full)), f'max err {err.max()}'
    _ffi.check(tm.tm_linear_destroy(h13))
    _ffi.check(tm.tm_linear_destroy(h2))


@pytest.mark.parametrize('V,ld', [(1000, 1000), (4099, 4104), (128256, 128256)])
def test_sampling_matches_oracle(tm, cuda, V, ld):
    """tm_sample against the sort-based restatement of the reference pipeline (oracle.sample_filter / sample_draw):
    surviving-candidate count and the drawn token for given uniform numbers, rows with different temperature / top-k /
    top-p / min-p, logits with many exact ties (the tie order is part of the contract) and -inf (banned) entries."""
    rng = np.random.default_rng(V)
    rows = [dict(), dict(top_k=1), dict(top_k=40), dict(top_k=40, top_p=0.8), dict(top_p=0.9), dict(top_p=0.3, temperature=0.7),
            dict(min_p=0.05), dict(top_k=200, top_p=0.95, min_p=0.02, temperature=1.3), dict(top_p=0.0), dict(top_k=V + 5),
            dict(temperature=0.01), dict(top_p=0.999, temperature=2.0)]
    B = len(rows)
    logits = (rng.standard_normal((B, ld)) * 2.5).astype(f16)
    logits[:, ::3] = np.round(logits[:, ::3].astype(np.float32) * 4).astype(f16) / f16(4)       # lots of exact ties
    logits[:, 5:50:7] = f16(-np.inf)
    logits[2, :] = f16(1.5)                                                                       # a completely flat row
    if ld > V:
        logits[:, V:] = f16(100.0)                                                                # padding must be ignored
    u = rng.random(B).astype(np.float32)
    u[0], u[4] = 0.0, np.float32(1.0 - 2.0**-24)
    arr = lambda k, d, t: np.asarray([r.get(k, d) for r in rows], t)
    temp, topk, topp, minp = arr('temperature', 1.0, np.float32), arr('top_k', 0, np.int32), arr('top_p', 1.0, np.float32), arr('min_p', 0.0, np.float32)
    ws = torch.zeros(tm.tm_sample_workspace(B), dtype=torch.uint8, device='cuda')
    out = torch.full((B,), -1, dtype=torch.int32, device='cuda')
    kept = torch.zeros(B, dtype=torch.int32, device='cuda')
    for rep in range(2):      # the second call checks that the workspace was left zeroed
        _ffi.check(tm.tm_sample(out.data_ptr(), kept.data_ptr(), dev(logits).data_ptr(), B, V, ld, dev(temp).data_ptr(),
                                dev(topk).data_ptr(), dev(topp).data_ptr(), dev(minp).data_ptr(), dev(u).data_ptr(),
                                ws.data_ptr(), st()))
        got, got_kept = host(out), host(kept)
        for b, r in enumerate(rows):
            ids, p = o.sample_filter(logits[b, :V], float(temp[b]), int(topk[b]), float(topp[b]), float(minp[b]))
            assert got_kept[b] == len(ids), f'row {b} {r}: kept {got_kept[b]} vs {len(ids)}'
            assert got[b] == o.sample_draw(ids, p, float(u[b])), f'row {b} {r}'
    assert not ws.any()


@pytest.mark.parametrize('V,ld,cap', [(1000, 1000, 1024), (4099, 4104, 1024), (4099, 4104, 5), (128256, 128256, 1024), (128256, 128256, 20)])
def test_sampling_logprobs_match_oracle(tm, cuda, V, ld, cap):
    """tm_sample_logprobs = tm_sample + the kept candidates' logprobs (sampling_kernels.cu:67-90): same draw and kept count as
    tm_sample; the first min(kept, cap) candidates come out in sampling order (token ids EXACT, ties by id included), their logprobs
    = logf(renormalised probability) within 1e-5 (absolute, or relative for large magnitudes; -inf where the probability is 0), the
    drawn token's own logprob, and -- in the reference's layout (cap = 1024) -- a drawn token beyond the first 1024 candidates in
    entry 1023.  The workspace is left zeroed."""
    rng = np.random.default_rng(V + cap)
    rows = [dict(), dict(top_k=1), dict(top_k=40), dict(top_k=40, top_p=0.8), dict(top_p=0.9), dict(top_p=0.3, temperature=0.7),
            dict(min_p=0.05), dict(top_k=2000, top_p=0.999, temperature=1.3), dict(top_p=0.0), dict(top_k=V + 5),
            dict(temperature=0.01), dict(top_p=0.999, temperature=2.0), dict(top_k=1500, temperature=4.0)]
    B = len(rows)
    logits = (rng.standard_normal((B, ld)) * 2.5).astype(f16)
    logits[:, ::3] = np.round(logits[:, ::3].astype(np.float32) * 4).astype(f16) / f16(4)       # lots of exact ties
    logits[:, 5:50:7] = f16(-np.inf)
    logits[2, :] = f16(1.5)                                                                       # a completely flat row
    logits[9, 100:] = f16(0.25)                                                                   # one huge tie across the cap
    if ld > V:
        logits[:, V:] = f16(100.0)
    u = rng.random(B).astype(np.float32)
    u[0], u[4], u[9], u[12] = 0.0, np.float32(1.0 - 2.0**-24), 0.97, 0.999                        # rows 9 / 12: a draw deep in the tail
    arr = lambda k, d, t: np.asarray([r.get(k, d) for r in rows], t)
    temp, topk, topp, minp = arr('temperature', 1.0, np.float32), arr('top_k', 0, np.int32), arr('top_p', 1.0, np.float32), arr('min_p', 0.0, np.float32)
    ws = torch.zeros(tm.tm_sample_workspace(B), dtype=torch.uint8, device='cuda')
    out, kept = torch.full((B,), -1, dtype=torch.int32, device='cuda'), torch.zeros(B, dtype=torch.int32, device='cuda')
    out0, kept0 = torch.full((B,), -1, dtype=torch.int32, device='cuda'), torch.zeros(B, dtype=torch.int32, device='cuda')
    vals = torch.full((B, cap), 7.0, dtype=torch.float32, device='cuda')
    idx = torch.full((B, cap), -5, dtype=torch.int32, device='cuda')
    num = torch.full((B,), -1, dtype=torch.int32, device='cuda')
    sel = torch.full((B,), 7.0, dtype=torch.float32, device='cuda')
    dl = dev(logits)
    args = (dl.data_ptr(), B, V, ld, dev(temp).data_ptr(), dev(topk).data_ptr(), dev(topp).data_ptr(), dev(minp).data_ptr(),
            dev(u).data_ptr(), ws.data_ptr(), st())
    _ffi.check(tm.tm_sample(out0.data_ptr(), kept0.data_ptr(), *args))
    tail_draws = 0
    for rep in range(2):
        _ffi.check(tm.tm_sample_logprobs(out.data_ptr(), kept.data_ptr(), vals.data_ptr(), idx.data_ptr(), num.data_ptr(), sel.data_ptr(),
                                         cap, *args))
        assert np.array_equal(host(out), host(out0)) and np.array_equal(host(kept), host(kept0))
        g_vals, g_idx, g_num, g_sel, g_out = host(vals), host(idx), host(num), host(sel), host(out)
        for b, r in enumerate(rows):
            ids, p = o.sample_filter(logits[b, :V], float(temp[b]), int(topk[b]), float(topp[b]), float(minp[b]))
            e_ids, e_lp, e_sel = o.sample_logprobs(ids, p, int(g_out[b]), cap)
            n = len(e_ids)
            assert g_num[b] == n == min(len(ids), cap), f'row {b} {r}: num {g_num[b]} vs {n}'
            assert np.array_equal(g_idx[b, :n], e_ids), f'row {b} {r}: candidate order'
            fin = np.isfinite(e_lp)
            assert np.array_equal(np.isfinite(g_vals[b, :n]), fin) and np.all(g_vals[b, :n][~fin] == -np.inf), f'row {b} {r}'
            assert np.all(np.abs(g_vals[b, :n][fin] - e_lp[fin]) <= 1e-5 + 1e-6 * np.abs(e_lp[fin])), f'row {b} {r}'
            assert abs(g_sel[b] - e_sel) <= 1e-5 + 1e-6 * abs(e_sel) or (g_sel[b] == e_sel), f'row {b} {r}: drawn token'
            assert np.all(g_idx[b, n:] == -5) and np.all(g_vals[b, n:] == 7.0), 'entries beyond num must stay untouched'
            pos = int(np.flatnonzero(ids == g_out[b])[0])
            tail_draws += pos >= cap
    if V > 1024:
        assert tail_draws > 0, 'no row drew a token beyond the cap: the forced-last / sel path was not exercised'
    assert not ws.any()


def test_sampling_degenerate_rows_get_a_defined_token(tm, cuda):
    """Rows whose logits do not form a distribution (all NaN, a +inf maximum, all -inf) must still produce a DEFINED
    token -- the arg-max over the finite logits, lowest id on ties, token 0 when nothing is finite -- not the previous
    step's id; NaN entries inside an otherwise ordinary row carry zero probability (the row samples as if they were
    banned).  Our own contract: the reference leaves these cases undefined."""
    rng = np.random.default_rng(3)
    V = 1000
    logits = (rng.standard_normal((5, V)) * 2).astype(f16)
    logits[0, :] = f16(np.nan)
    logits[1, 17] = f16(np.inf)
    logits[1, 400] = f16(9.0)
    logits[1, 300] = f16(9.0)
    logits[2, :] = f16(-np.inf)
    logits[3, ::2] = f16(np.nan)
    logits[4, :] = f16(np.nan)
    logits[4, 777] = f16(-3.0)
    u = np.asarray([0.3, 0.6, 0.1, 0.42, 0.9], np.float32)
    temp = np.ones(5, np.float32)
    topk = np.asarray([0, 0, 5, 0, 0], np.int32)
    topp = np.asarray([1.0, 0.9, 1.0, 0.8, 1.0], np.float32)
    ws = torch.zeros(tm.tm_sample_workspace(5), dtype=torch.uint8, device='cuda')
    out = torch.full((5,), -7, dtype=torch.int32, device='cuda')
    _ffi.check(tm.tm_sample(out.data_ptr(), None, dev(logits).data_ptr(), 5, V, V, dev(temp).data_ptr(), dev(topk).data_ptr(),
                            dev(topp).data_ptr(), None, dev(u).data_ptr(), ws.data_ptr(), st()))
    got = host(out)
    assert got[0] == 0                       # nothing finite
    assert got[1] == 300                     # arg-max over the finite logits, lowest id on the tie
    assert got[2] == 0
    banned = logits[3].copy()
    banned[::2] = f16(-np.inf)               # NaN entries = zero probability
    ids, p = o.sample_filter(banned, 1.0, 0, 0.8, 0.0)
    assert got[3] == o.sample_draw(ids, p, float(u[3]))
    assert got[4] == 777                     # the only finite entry
    assert not ws.any()


@pytest.mark.parametrize('V,ld,off', [(1000, 1000, 0), (4099, 4104, 0), (128256, 128256, 0), (16032, 16032, 16032)])
def test_logits_process_matches_oracle(tm, cuda, V, ld, off):
    """tm_seen_update + tm_logits_process against oracle.logits_process, bit exact: seen masks built from a packed
    prefill (cu_q rows) plus decode-style single tokens, rows with / without penalty, bad ids, min-length bans that are
    active / expired, ids <= 0 (never banned), a vocabulary shard (off > 0: the mask covers the global vocabulary)."""
    rng = np.random.default_rng(V + off)
    vocab = off + V if off else V
    rows = [dict(p=1.3), dict(p=0.6, bad=[5, 17, 0]), dict(), dict(p=2.5, end=[off + 7, off + V - 1], k=10, ml=14),
            dict(bad=[off + 3], end=[off + 11, 0, off + 12], k=13, ml=14), dict(p=1.0001, bad=list(range(off + 100, off + 132)))]
    B = len(rows)
    words = (vocab + 31) // 32
    lens = [int(rng.integers(1, 300)) for _ in range(B)]
    prompt = [rng.integers(0, vocab, n).astype(np.int32) for n in lens]
    last = rng.integers(0, vocab, B).astype(np.int32)
    cu_q = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    seen = torch.zeros((B, words), dtype=torch.int32, device='cuda')
    _ffi.check(tm.tm_seen_update(seen.data_ptr(), words, dev(np.concatenate(prompt)).data_ptr(), dev(cu_q).data_ptr(), B,
                                 int(cu_q[-1]), vocab, st()))
    _ffi.check(tm.tm_seen_update(seen.data_ptr(), words, dev(last).data_ptr(), None, B, B, vocab, st()))
    want = np.zeros((B, words), np.uint32)
    for b in range(B):
        for t in list(prompt[b]) + [last[b]]:
            want[b, t >> 5] |= np.uint32(1) << np.uint32(t & 31)
    assert np.array_equal(host(seen).view(np.uint32), want)

    logits = (rng.standard_normal((B, ld)) * 4).astype(f16)
    logits[:, 1::97] = f16(-60000.0)                                     # penalised to -inf in fp16
    ban = np.full((B, 32), -1, np.int32)
    end = np.full((B, 9), -1, np.int32)
    for b, r in enumerate(rows):
        ban[b, :len(r.get('bad', []))] = r.get('bad', [])
        end[b, :len(r.get('end', []))] = r.get('end', [])
    arr = lambda k, d, t: np.asarray([r.get(k, d) for r in rows], t)
    rep, k_len, min_len = arr('p', 1.0, np.float32), arr('k', 5, np.int32), arr('ml', 0, np.int32)
    d_logits = dev(logits).clone()
    _ffi.check(tm.tm_logits_process(d_logits.data_ptr(), B, V, ld, off, seen.data_ptr(), words, dev(rep).data_ptr(),
                                    dev(ban).data_ptr(), dev(end).data_ptr(), dev(k_len).data_ptr(), dev(min_len).data_ptr(), st()))
    got = host(d_logits)
    for b, r in enumerate(rows):
        ref = o.logits_process(logits[b, :V], list(prompt[b]) + [last[b]], float(rep[b]), r.get('bad', []), r.get('end', []),
                               int(k_len[b]), int(min_len[b]), vocab_offset=off)
        assert np.array_equal(got[b, :V].view(np.uint16), ref.view(np.uint16)), f'row {b} {r}'
        assert np.array_equal(got[b, V:].view(np.uint16), logits[b, V:].view(np.uint16))      # padding untouched


def test_sampling_distribution(tm, cuda):
    """Statistical check of the draw: 20000 uniform numbers from the engine's Philox stream over one filtered
    distribution reproduce its probabilities (chi-square far below the rejection threshold)."""
    rng = np.random.default_rng(9)
    V, N = 512, 20000
    row = (rng.standard_normal(V) * 2).astype(f16)
    ids, p = o.sample_filter(row, 0.9, 12, 0.97, 0.0)
    logits = np.tile(row, (64, 1))
    counts = np.zeros(V, np.int64)
    ws = torch.zeros(tm.tm_sample_workspace(64), dtype=torch.uint8, device='cuda')
    out = torch.zeros(64, dtype=torch.int32, device='cuda')
    t, k, pp = dev(np.full(64, 0.9, np.float32)), dev(np.full(64, 12, np.int32)), dev(np.full(64, 0.97, np.float32))
    ld = dev(logits)
    for it in range(N // 64 + 1):
        u = np.asarray([tm.tm_philox_uniform(1234, it * 64 + j) for j in range(64)], np.float32)
        _ffi.check(tm.tm_sample(out.data_ptr(), None, ld.data_ptr(), 64, V, V, t.data_ptr(), k.data_ptr(), pp.data_ptr(), None,
                                dev(u).data_ptr(), ws.data_ptr(), st()))
        np.add.at(counts, host(out), 1)
    n = counts.sum()
    assert counts[np.setdiff1d(np.arange(V), ids)].sum() == 0          # nothing outside the filtered set
    chi2 = (((counts[ids] - n * p) ** 2) / (n * p)).sum()
    assert chi2 < 3 * len(ids) + 30, chi2                               # dof = len(ids) - 1 ~ 10; mean = dof


@pytest.mark.parametrize('K,N', [(512, 256), (4096, 1024), (384, 48)])
def test_fp8_linear(tm, cuda, K, N):
    """FP8 (e4m3, 128x128 block scales) weight-only linear (TM_WEIGHT_FP8; BASELINE config 5's weight format) against
    the oracle: dequantised weights bit exact (identity activations), products within the u4 linear's tolerance for
    decode- and prefill-sized M, gated SiLU epilogue."""
    rng = np.random.default_rng(K + N)
    w = (rng.standard_normal((K, N)) * (0.1 / math.sqrt(K))).astype(f16)
    q, sc = o.fp8_quantize_blockwise(w)
    q[::37, ::11] = 0x01                                  # subnormal codes
    q[5::53, 3::7] = 0xFE                                 # -448
    wd = o.fp8_dequant(q, sc)
    h = _ffi.C.c_void_p()
    _ffi.check(tm.tm_linear_create(_ffi.C.byref(h), K, N, 2, 128))
    _ffi.check(tm.tm_linear_prepare(h, dev(q).data_ptr(), dev(sc).data_ptr(), None, st()))
    torch.cuda.synchronize()
    ws = torch.zeros(max(1, tm.tm_linear_workspace(h, 700)), dtype=torch.uint8, device='cuda')
    x = np.zeros((64, K), f16)
    rows = rng.permutation(K)[:64]
    x[np.arange(64), rows] = 1
    y = torch.zeros((64, N), dtype=torch.float16, device='cuda')
    _ffi.check(tm.tm_linear_forward(h, dev(x).data_ptr(), K, y.data_ptr(), N, 64, 0, 0, 1, 0, ws.data_ptr(), st()))
    got, exp = host(y), wd[rows]
    normal = np.abs(exp.astype(np.float32)) >= 2.0**-14
    bad = (got.view(np.uint16) != exp.view(np.uint16)) & normal
    assert not bad.any(), f'dequantised fp8 weights must be bit exact: {int(bad.sum())} normal-range mismatches, e.g. {got[bad][:4]} vs {exp[bad][:4]}'
    # fp16 subnormal products: the matrix core may flush them (the reference's mma does not define it either)
    assert np.abs(got.astype(np.float32) - exp.astype(np.float32))[~normal].max(initial=0.0) <= 2.0**-14
    for M in (1, 17, 64, 130, 700):
        x = rng.standard_normal((M, K)).astype(f16)
        ref = x.astype(np.float32) @ wd.astype(np.float32)
        y = torch.zeros((M, N), dtype=torch.float16, device='cuda')
        _ffi.check(tm.tm_linear_forward(h, dev(x).data_ptr(), K, y.data_ptr(), N, M, 0, 0, 0, 0, ws.data_ptr(), st()))
        err = np.abs(host(y).astype(np.float32) - ref)
        assert np.all(err <= 2e-3 + 2.0**-10 * np.abs(ref)), f'M={M}: max err {err.max()}'
    if N % 32 == 0:
        x = (rng.standard_normal((33, K)) * 3).astype(f16)
        acc = x.astype(np.float32) @ wd.astype(np.float32)
        ref = o.gated_silu_epilogue(acc).astype(np.float32)
        y = torch.zeros((33, N // 2), dtype=torch.float16, device='cuda')
        _ffi.check(tm.tm_linear_forward(h, dev(x).data_ptr(), K, y.data_ptr(), N // 2, 33, 1, 0, 0, 0, ws.data_ptr(), st()))
        err = np.abs(host(y).astype(np.float32) - ref)
        assert np.all(err <= 2e-3 + 2.0**-9 * np.abs(ref)), f'gated: max err {err.max()}'
    _ffi.check(tm.tm_linear_destroy(h))


def test_fp8_quant_rows_bit_exact(tm, cuda):
    """QuantizeSymm (kernels/quantization.cu:28-125): e4m3 codes and fp32 scales per (row, 128 channels) -- integer /
    byte outputs, bit exact against the oracle, including an all-zero group (absmax clamp), a constant group, values at
    the fp16 extremes and fp16 subnormals."""
    rng = np.random.default_rng(17)
    M, K = 37, 1024
    x = (rng.standard_normal((M, K)) * rng.uniform(0.01, 30.0, (M, 1))).astype(f16)
    x[3, 128:256] = 0
    x[4, :128] = f16(0.75)
    x[5, 7] = f16(65504.0)
    x[6, 256:384] = (rng.standard_normal(128) * 1e-6).astype(f16)
    ldsx = (M + 3) // 4 * 4
    xq = torch.zeros((M, K), dtype=torch.uint8, device='cuda')
    sx = torch.zeros((K // 128, ldsx), dtype=torch.float32, device='cuda')
    _ffi.check(tm.tm_quant_fp8_rows(xq.data_ptr(), sx.data_ptr(), dev(x).data_ptr(), K, M, K, ldsx, st()))
    q_ref, s_ref = o.fp8_quant_rows(x)
    assert np.array_equal(host(sx)[:, :M].view(np.uint32), s_ref.view(np.uint32)), 'activation scales must be bit exact'
    got = host(xq)
    assert np.array_equal(got, q_ref), f'{int((got != q_ref).sum())} codes differ'


@pytest.mark.parametrize('K,N', [(512, 256), (4096, 1024), (384, 64)])
def test_fp8_mfma_linear(tm, cuda, K, N):
    """fp8 x fp8 linear on v_mfma_f32_32x32x16_fp8_fp8 (gemm_fp8.hip) against the oracle's restatement of the reference's
    fp8 GEMM path (activation quantisation per row and 128 channels + fp32 block-scaled accumulation): both sides contract
    the SAME codes, so the tolerance is the u4 linear's (accumulation order only); split-K, row blocks, gated SiLU."""
    rng = np.random.default_rng(K * 3 + N)
    w = (rng.standard_normal((K, N)) * (0.1 / math.sqrt(K))).astype(f16)
    q, sc = o.fp8_quantize_blockwise(w)
    q[::37, ::11] = 0x01
    q[5::53, 3::7] = 0xFE
    h = _ffi.C.c_void_p()
    _ffi.check(tm.tm_linear_create(_ffi.C.byref(h), K, N, 2, 128))
    _ffi.check(tm.tm_linear_prepare(h, dev(q).data_ptr(), dev(sc).data_ptr(), None, st()))
    torch.cuda.synchronize()
    for M in (1, 17, 33, 64, 130):
        x = (rng.standard_normal((M, K)) * rng.uniform(0.2, 4.0, (M, 1))).astype(f16)
        ref = o.fp8_act_linear(x, q, sc).astype(np.float32)
        ws = torch.zeros(tm.tm_linear_fp8_workspace(h, M), dtype=torch.uint8, device='cuda')
        for splits in (0, 1, 2, 3):
            if splits > max(1, K // 512):
                continue
            y = torch.zeros((M, N), dtype=torch.float16, device='cuda')
            _ffi.check(tm.tm_linear_forward_fp8(h, dev(x).data_ptr(), K, y.data_ptr(), N, M, 0, splits, ws.data_ptr(), st()))
            err = np.abs(host(y).astype(np.float32) - ref)
            assert np.all(err <= 2e-3 + 2.0**-9 * np.abs(ref)), f'M={M} splits={splits}: max err {err.max()}'
    if N % 256 == 0:   # the fused w1w3 form: (gate_j, up_j)-interleaved columns, scale row [w1 blocks | w3 blocks]
        w1, w3 = w[:, :N // 2], w[:, N // 2:]
        (q1, s1), (q3, s3) = o.fp8_quantize_blockwise(w1), o.fp8_quantize_blockwise(w3)
        q13, s13 = o.interleave_w1w3(q1, q3), np.concatenate([s1, s3], axis=1)
        hg = _ffi.C.c_void_p()
        _ffi.check(tm.tm_linear_create(_ffi.C.byref(hg), K, N, 2, 128))
        _ffi.check(tm.tm_linear_prepare_fp8_gated(hg, dev(q13).data_ptr(), dev(s13).data_ptr(), st()))
        torch.cuda.synchronize()
        x = (rng.standard_normal((40, K)) * 2).astype(f16)
        ref = o.fp8_act_linear(x, q13, s13, gated=True).astype(np.float32)
        ws = torch.zeros(tm.tm_linear_fp8_workspace(hg, 40), dtype=torch.uint8, device='cuda')
        y = torch.zeros((40, N // 2), dtype=torch.float16, device='cuda')
        _ffi.check(tm.tm_linear_forward_fp8(hg, dev(x).data_ptr(), K, y.data_ptr(), N // 2, 40, 1, 1, ws.data_ptr(), st()))
        err = np.abs(host(y).astype(np.float32) - ref)
        assert np.all(err <= 2e-3 + 2.0**-8 * np.abs(ref)), f'gated: max err {err.max()}'
        _ffi.check(tm.tm_linear_destroy(hg))
    _ffi.check(tm.tm_linear_destroy(h))


@pytest.mark.parametrize('wtype,T', [('fp8', 5), ('fp8', 64), ('fp8', 300), ('u4', 37), ('fp8wo', 5), ('fp8wo', 300)])
def test_moe_ffn(tm, cuda, monkeypatch, wtype, T):
    """MoE FFN block (router + top-2 of 8 + grouped expert GEMMs + combine; BASELINE config 5 scaled down) against the
    oracle: identical routing (expert ids), weights to 1e-5, output within the dense FFN's tolerance.  T = 300 runs
    the prefill-shaped grouped GEMM (row blocks per expert), the others the decode shape."""
    # 'fp8': e4m3 experts on the fp8 matrix cores with on-the-fly activation quantisation (the default for fp8 experts);
    # 'fp8wo': the weight-only dequant path (TM_FP8_MFMA=0)
    mfma = wtype == 'fp8'
    if wtype == 'fp8wo':
        monkeypatch.setenv('TM_FP8_MFMA', '0')
        wtype = 'fp8'
    rng = np.random.default_rng(T)
    H, I, E, k = 256, 384, 8, 2
    x = rng.standard_normal((T, H)).astype(f16)
    gate = (rng.standard_normal((H, E)) * 0.2).astype(f16)
    h = _ffi.C.c_void_p()
    experts_q = []
    _ffi.check(tm.tm_moe_create(_ffi.C.byref(h), H, I, E, k, 2 if wtype == 'fp8' else 0, 1, 1.0))
    _ffi.check(tm.tm_moe_set_gate(h, dev(gate).data_ptr(), st()))
    experts = []
    for e in range(E):
        w1 = (rng.standard_normal((H, I)) * (0.1 / math.sqrt(H))).astype(f16)
        w3 = (rng.standard_normal((H, I)) * (0.1 / math.sqrt(H))).astype(f16)
        w2 = (rng.standard_normal((I, H)) * (0.1 / math.sqrt(I))).astype(f16)
        w13 = o.interleave_w1w3(w1, w3)
        if wtype == 'fp8':
            # w1 / w3 block-quantised separately (as in a checkpoint): interleaved codes, scale row [w1 blocks | w3 blocks]
            (q1, s1), (q3, s3) = o.fp8_quantize_blockwise(w1), o.fp8_quantize_blockwise(w3)
            q13, s13 = o.interleave_w1w3(q1, q3), np.concatenate([s1, s3], axis=1)
            q2, s2 = o.fp8_quantize_blockwise(w2)
            experts.append((o.fp8_dequant(q13, s13, gated=True), o.fp8_dequant(q2, s2)))
            experts_q.append(((q13, s13), (q2, s2)))
            _ffi.check(tm.tm_moe_set_expert(h, e, dev(q13).data_ptr(), dev(s13).data_ptr(), None, dev(q2).data_ptr(),
                                            dev(s2).data_ptr(), None, st()))
        else:
            q13, s13, z13, _ = o.quantize_groupwise_u4(w13, 128)
            q2, s2, z2, _ = o.quantize_groupwise_u4(w2, 128)
            experts.append((o.w4a16_dequant(q13, s13, z13), o.w4a16_dequant(q2, s2, z2)))
            _ffi.check(tm.tm_moe_set_expert(h, e, dev(o.pack_u4_row(q13)).data_ptr(), dev(s13).data_ptr(), dev(z13).data_ptr(),
                                            dev(o.pack_u4_row(q2)).data_ptr(), dev(s2).data_ptr(), dev(z2).data_ptr(), st()))
    torch.cuda.synchronize()
    ws = torch.zeros(tm.tm_moe_workspace(h, T), dtype=torch.uint8, device='cuda')
    out = torch.zeros((T, H), dtype=torch.float16, device='cuda')
    ids_d = torch.zeros((T, k), dtype=torch.int32, device='cuda')
    w_d = torch.zeros((T, k), dtype=torch.float32, device='cuda')
    _ffi.check(tm.tm_moe_forward(h, out.data_ptr(), dev(x).data_ptr(), T, ws.data_ptr(), ids_d.data_ptr(), w_d.data_ptr(), st()))
    ref, ids, w = o.moe_ffn_fp8(x, gate, experts_q, k) if mfma else o.moe_ffn(x, gate, experts, k)
    assert np.array_equal(host(ids_d), ids), 'routing differs'
    assert np.abs(host(w_d) - w).max() <= 1e-5
    err = np.abs(host(out).astype(np.float32) - ref.astype(np.float32))
    # fp8 x fp8: the gated-SiLU rows are re-quantised to e4m3 (one code step = 2^-3 relative); an fp32-order difference that
    # flips an fp16 ulp of such a row can move a code, hence the wider relative term
    tol = (4e-3 + 2.0**-6 * np.abs(ref.astype(np.float32))) if mfma else (3e-3 + 2.0**-8 * np.abs(ref.astype(np.float32)))
    assert np.all(err <= tol), f'max err {err.max()}'
    _ffi.check(tm.tm_moe_destroy(h))


def test_w4a16_identity_asymmetric(tm, cuda):
    """Transpose-detecting check: x = I (first K rows) picks out rows of the dequantised weight exactly."""
    rng = np.random.default_rng(3)
    K, N = 256, 96
    h, (q, s, z) = _make_linear(tm, rng, K, N)
    w = o.w4a16_dequant(q, s, z)
    x = np.zeros((64, K), f16)
    rows = rng.permutation(K)[:64]
    x[np.arange(64), rows] = 1
    y = torch.zeros((64, N), dtype=torch.float16, device='cuda')
    _ffi.check(tm.tm_linear_forward(h, dev(x).data_ptr(), K, y.data_ptr(), N, 64, 0, 0, 1, 0, None, st()))
    assert np.array_equal(host(y).view(np.uint16), w[rows].view(np.uint16)), 'dequantised weights must be bit exact'
    _ffi.check(tm.tm_linear_destroy(h))


@pytest.mark.parametrize('K,N,M', [(2048, 1008 * 16, 64), (512, 160, 3)])
def test_f16_linear(tm, cuda, K, N, M):
    rng = np.random.default_rng(K + N)
    w = (rng.standard_normal((K, N)) * (0.1 / math.sqrt(K))).astype(f16)
    x = rng.standard_normal((M, K)).astype(f16)
    h = _ffi.C.c_void_p()
    _ffi.check(tm.tm_linear_create(_ffi.C.byref(h), K, N, 1, 128))
    _ffi.check(tm.tm_linear_prepare(h, dev(w).data_ptr(), None, None, st()))
    y = torch.zeros((M, N), dtype=torch.float16, device='cuda')
    _ffi.check(tm.tm_linear_forward(h, dev(x).data_ptr(), K, y.data_ptr(), N, M, 0, 0, 1, 0, None, st()))
    ref = o.gemm_f16_f32acc(x, w)
    err = np.abs(host(y).astype(np.float32) - ref)
    assert np.all(err <= 1e-3 + 2.0**-10 * np.abs(ref)), f'max err {err.max()}'   # fixture.py gate for f16: 1e-2
    _ffi.check(tm.tm_linear_destroy(h))


def test_quantize_groupwise_matches_oracle(tm, cuda):
    rng = np.random.default_rng(11)
    K, N = 512, 256
    w = (rng.standard_normal((K, N)) * 0.02).astype(f16)
    qw = torch.zeros((K, N // 8), dtype=torch.int32, device='cuda')
    s = torch.zeros((K // 128, N), dtype=torch.float16, device='cuda')
    z = torch.zeros((K // 128, N), dtype=torch.float16, device='cuda')
    d = torch.zeros((K, N), dtype=torch.float16, device='cuda')
    _ffi.check(tm.tm_quantize_groupwise(qw.data_ptr(), s.data_ptr(), z.data_ptr(), d.data_ptr(), dev(w).data_ptr(), K, N,
                                        128, st()))
    q_ref, s_ref, z_ref, d_ref = o.quantize_groupwise_u4(w, 128)
    assert np.array_equal(host(s).view(np.uint16), s_ref.view(np.uint16))
    assert np.array_equal(host(z), z_ref)
    q_got = o.unpack_u4_row(host(qw))
    # x/scale in fp32 may sit on a rounding boundary: allow a handful of +-1 codes
    assert (q_got != q_ref).mean() < 1e-4 and np.abs(q_got.astype(int) - q_ref.astype(int)).max() <= 1


# ------------------------------------------------------------------------------------------------
def test_embedding_argmax_silu(tm, cuda):
    rng = np.random.default_rng(2)
    table = rng.standard_normal((1000, 256)).astype(f16)
    ids = rng.integers(0, 1000, 37).astype(np.int32)
    out = torch.zeros((37, 256), dtype=torch.float16, device='cuda')
    _ffi.check(tm.tm_embedding(out.data_ptr(), dev(table).data_ptr(), dev(ids).data_ptr(), 37, 256, 1000, st()))
    assert np.array_equal(host(out), table[ids])

    logits = rng.standard_normal((5, 128256)).astype(f16)
    logits[2, 77] = logits[2, 9000] = f16(30.0)        # tie -> lowest index
    got = torch.zeros(5, dtype=torch.int32, device='cuda')
    _ffi.check(tm.tm_argmax(got.data_ptr(), None, dev(logits).data_ptr(), 5, 128256, 128256, st()))
    assert np.array_equal(host(got), o.greedy(logits))

    gu = (rng.standard_normal((9, 2 * 512)) * 3).astype(f16)
    y = torch.zeros((9, 512), dtype=torch.float16, device='cuda')
    _ffi.check(tm.tm_silu_mul(y.data_ptr(), dev(gu).data_ptr(), 9, 512, st()))
    d = ulp_diff_f16(host(y), o.silu_and_mul_unfused(gu))
    assert d.max() <= 1
