"""Host model of lmdeploy_amd/csrc/gemm_decode.hip: the P32 weight layout, the lane -> operand maps of
v_mfma_f32_32x32x16_f16, the swizzled x image in LDS and the swizzled reduction image of the k-phases, restated in numpy
lane by lane and checked against x @ dequant(W) of the oracle.  It pins the INDEX ALGEBRA of the kernel on the CPU (the
MFMA operand maps themselves are the ISA's: A[i = l&31][k = 8(l>>5)+e], B[k = 8(l>>5)+e][j = l&31],
D[i = (r&3) + 8(r>>2) + 4(l>>5)][j = l&31]; the GPU parity tests confirm them on hardware)."""
import numpy as np
import pytest

from oracle import tm_oracle as o

UNIT = 2176


def repack_p32(q, s, z):
    """numpy twin of repack_p32_kernel: q uint8 [K,N], s/z fp16 [K/128,N] -> uint32 [KB*ncg, 544]."""
    K, N = q.shape
    KB, ncg = K // 128, N // 32
    s_f, zs_f = o.fuse_scales_zeros(s, z)
    out = np.zeros((KB * ncg, UNIT // 4), np.uint32)
    for kb in range(KB):
        for cg in range(ncg):
            u = out[kb * ncg + cg]
            for d in range(512):
                jj, lane, p = d & 3, (d >> 2) & 63, d >> 8
                j = 4 * p + jj
                n = cg * 32 + (lane & 31)
                k0 = kb * 128 + 16 * j + 8 * (lane >> 5)
                w = 0
                for e in range(8):
                    nib = 4 + (e >> 1) if e & 1 else e >> 1
                    w |= int(q[k0 + e, n]) << (4 * nib)
                u[d] = w
            pair = np.stack([s_f[kb, cg * 32:cg * 32 + 32], zs_f[kb, cg * 32:cg * 32 + 32]], -1).astype(np.float16)
            u[512:] = pair.view(np.uint32)[:, 0]
    return out


def dequant_dword(w, s, zs):
    """8 fp16 weights of one packed dword in MFMA element order (k0..k7), the arithmetic of dequant8_p32."""
    nib = np.array([(w >> (4 * i)) & 15 for i in range(8)], np.float16)     # nibble i
    order = [0, 4, 1, 5, 2, 6, 3, 7]                                         # element e sits in nibble order[e]
    qv = nib[order]
    return o.hfma(qv, np.float16(s), np.float16(zs))


def kernel_model(x, units, K, N, M, MH, CG, WK, S, splits, gated):
    """Workgroup-by-workgroup, wave-by-wave, lane-by-lane replay of gemm_dec32_kernel."""
    KB, ncg, ROWS = K // 128, N // 32, 32 * MH
    KBB = ROWS * 256
    per = -(-KB // splits)
    per = min(-(-per // S) * S, KB)
    splits = -(-KB // per)
    lanes = np.arange(64)
    l31, half = lanes & 31, lanes >> 5
    coff = [(l31 * 256 + (((2 * j + half) ^ (l31 & 15)) << 4)) for j in range(8)]
    slabs = np.zeros((splits, M, N), np.float32)
    xb = x.view(np.uint8).reshape(M, K * 2)
    for by in range(splits):
        kb0 = by * per
        nkb = min(per, KB - kb0)
        nst = -(-nkb // S)
        for bx in range(-(-ncg // CG)):
            acc = np.zeros((CG, WK, MH, 64, 16), np.float32)
            for t in range(nst):
                # the LDS stage image as the staging threads build it
                lds = np.zeros(S * KBB, np.uint8)
                for qi in range(S * ROWS * 16):
                    kbi, row, ch = qi // (ROWS * 16), (qi >> 4) % ROWS, qi & 15
                    src = (kb0 + t * S + kbi) * 256 + ch * 16
                    chunk = xb[min(row, M - 1), src:src + 16] if src + 16 <= K * 2 else np.zeros(16, np.uint8)
                    dst = kbi * KBB + row * 256 + ((ch ^ (row & 15)) << 4)
                    lds[dst:dst + 16] = chunk
                for cgl in range(CG):
                    cg = min(bx * CG + cgl, ncg - 1)
                    for wk in range(WK):
                        for i in range(S // WK):
                            kbi = wk + i * WK
                            b = t * S + kbi
                            if b >= nkb:
                                continue                      # (the kernel multiplies by zeroed scales instead)
                            unit = units[(kb0 + b) * ncg + cg]
                            for j in range(8):
                                A = np.zeros((32, 16), np.float32)
                                for l in range(64):
                                    w = int(unit[((j >> 2) * 64 + l) * 4 + (j & 3)])
                                    pr = unit[512 + (l & 31):513 + (l & 31)].view(np.float16)
                                    A[l & 31, 8 * (l >> 5):8 * (l >> 5) + 8] = dequant_dword(w, pr[0], pr[1])
                                for h in range(MH):
                                    B = np.zeros((16, 32), np.float32)
                                    for l in range(64):
                                        a0 = kbi * KBB + h * 8192 + coff[j][l]
                                        B[8 * (l >> 5):8 * (l >> 5) + 8, l & 31] = lds[a0:a0 + 16].view(np.float16)
                                    D = A @ B
                                    for r in range(16):
                                        acc[cgl, wk, h, :, r] += D[(r & 3) + 8 * (r >> 2) + 4 * half, l31]
            # k-phase reduction image and the cooperative epilogue
            C4 = CG * 8
            red = np.zeros((WK, ROWS, C4, 4), np.float32)
            for cgl in range(CG):
                for wk in range(WK):
                    for h in range(MH):
                        for l in range(64):
                            m = 32 * h + (l & 31)
                            for g4 in range(4):
                                c4 = cgl * 8 + 2 * g4 + (l >> 5)
                                red[wk, m, c4 ^ (m & 7)] = acc[cgl, wk, h, l, 4 * g4:4 * g4 + 4]
            for e in range(ROWS * C4):
                m, c4 = e // C4, e % C4
                a = red[0, m, c4 ^ (m & 7)].copy()
                for k in range(1, WK):
                    a += red[k, m, c4 ^ (m & 7)]
                n = bx * CG * 32 + c4 * 4
                if m < M and n < N:
                    slabs[by, m, n:n + 4] = a
    y = slabs.sum(0)
    return o.gated_silu_epilogue(y) if gated else y.astype(np.float16)


@pytest.mark.parametrize('M,K,N,MH,CG,WK,S,splits,gated', [
    (64, 512, 128, 2, 4, 4, 4, 1, 0),      # the default shape: 4 column groups x 4 k-phases
    (37, 768, 96, 2, 4, 4, 4, 2, 0),       # ragged rows, a column group past N, ragged last stage, split-K
    (9, 512, 64, 1, 8, 2, 4, 1, 1),        # one 32-row half, 2 blocks per wave per stage, gated epilogue
])
def test_decode_gemm_index_algebra(M, K, N, MH, CG, WK, S, splits, gated):
    rng = np.random.default_rng(M + K + N)
    w = (rng.standard_normal((K, N)) * 0.05).astype(np.float16)
    q, s, z, _ = o.quantize_groupwise_u4(w, 128)
    x = rng.standard_normal((M, K)).astype(np.float16)
    units = repack_p32(q, s, z)
    got = kernel_model(x, units, K, N, M, MH, CG, WK, S, splits, gated).astype(np.float32)
    wd = o.w4a16_dequant(q, s, z)
    ref = (o.w4a16_linear_gated_silu(x, q, s, z) if gated else o.gemm_f16_f32acc(x, wd).astype(np.float16)).astype(np.float32)
    assert np.all(np.abs(got - ref) <= 2e-3 + 2.0**-9 * np.abs(ref)), np.abs(got - ref).max()


def test_p32_dequantises_to_the_oracle_weights():
    """every weight of the P32 image, dequantised the kernel's way, equals the oracle's w4a16_dequant bit for bit"""
    rng = np.random.default_rng(3)
    K, N = 256, 64
    w = (rng.standard_normal((K, N)) * 0.05).astype(np.float16)
    q, s, z, _ = o.quantize_groupwise_u4(w, 128)
    units = repack_p32(q, s, z)
    wd = o.w4a16_dequant(q, s, z)
    for kb in range(K // 128):
        for cg in range(N // 32):
            unit = units[kb * (N // 32) + cg]
            for j in range(8):
                for l in range(64):
                    pr = unit[512 + (l & 31):513 + (l & 31)].view(np.float16)
                    got = dequant_dword(int(unit[((j >> 2) * 64 + l) * 4 + (j & 3)]), pr[0], pr[1])
                    k0 = kb * 128 + 16 * j + 8 * (l >> 5)
                    assert np.array_equal(got.view(np.uint16), wd[k0:k0 + 8, cg * 32 + (l & 31)].view(np.uint16))
