"""CPU oracle for the TurboMind quantized decode hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import
it.  The product path (``lmdeploy_amd``) never routes through this module and
fails loudly when the HIP library is missing.

It is a numpy restatement of the reference's arithmetic (paths relative to the
reference checkout, lmdeploy v0.16.0).  ``h(x)`` below means "round to nearest
even to fp16".  fp16 add/sub/mul are evaluated in float32 and rounded once
(innocuous double rounding: 24 >= 2*11+2); fp16 fma is evaluated exactly in
float64 and rounded once.

Pinning status (see DESIGN.md "Oracle"):
  * AWQ unpack / dequant formula ......... pinned against the reference's own
    Python (`lmdeploy/pytorch/backends/default/awq_modules.py:12-46`,
    `lmdeploy/turbomind/weight_format.py:47-74,236-247`) via tests/golden.
  * RMSNorm ............................. pinned against
    `lmdeploy/pytorch/backends/default/norm.py:14-28` via tests/golden (fp32
    form; the TurboMind fp16-rounding form below is a restatement).
  * KV int8/int4 round trip ............. pinned by the reference's known-answer
    test `src/turbomind/kernels/attention/test_quant.cu:32-70` (exact round trip
    of integer-valued data with (scale,zero) = (1,0) / (1,-64)).
  * RoPE (Llama-3 inv_freq, cos/sin, application), MoE router, FP8 block
    dequant, sampling filters, logits processors: pinned against the reference's
    Python (`pytorch/backends/default/{rotary_embedding,apply_rotary_emb,moe}.py`,
    `turbomind/weight_format.py:349-384`, `pytorch/engine/logits_process.py:24-96`)
    via tests/golden/reference_python2.npz and reference_logits_process.npz.
  * KV min/max -> (scale, zero) rule, attention, u4 GEMM rounding order:
    "parity unpinned" beyond the properties the reference tests state (no
    known-answer vectors exist in the reference and its CUDA code cannot run
    here); the restatement follows the cited lines.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

f16 = np.float16
f32 = np.float32
f64 = np.float64


# --------------------------------------------------------------------------
# fp16 helpers
# --------------------------------------------------------------------------
def h(x):
    """Round to fp16 (RNE)."""
    return np.asarray(x).astype(f16)


def hadd(a, b):
    return (np.asarray(a, f16).astype(f32) + np.asarray(b, f16).astype(f32)).astype(f16)


def hsub(a, b):
    return (np.asarray(a, f16).astype(f32) - np.asarray(b, f16).astype(f32)).astype(f16)


def hmul(a, b):
    return (np.asarray(a, f16).astype(f32) * np.asarray(b, f16).astype(f32)).astype(f16)


def hfma(a, b, c):
    """fp16 fused multiply-add: single rounding of a*b+c."""
    with np.errstate(invalid='ignore', over='ignore'):
        return (np.asarray(a, f16).astype(f64) * np.asarray(b, f16).astype(f64)
                + np.asarray(c, f16).astype(f64)).astype(f16)


# --------------------------------------------------------------------------
# A1. AWQ checkpoint <-> boundary tensors
#   lmdeploy/turbomind/weight_format.py:47-74 (_unpack_awq_gemm, pack_u4_row)
# --------------------------------------------------------------------------
AWQ_ORDER = (0, 4, 1, 5, 2, 6, 3, 7)


def unpack_awq_gemm(qw: np.ndarray) -> np.ndarray:
    """int32[..., N/8] (AWQ 'gemm' nibble order) -> uint8[..., N].

    Logical column 8c+p is nibble AWQ_ORDER[p] of word c (weight_format.py:56-60).
    """
    qw = np.asarray(qw).astype(np.int64) & 0xFFFFFFFF
    nib = [((qw >> (4 * i)) & 15).astype(np.uint8) for i in range(8)]
    ys = [nib[i] for i in AWQ_ORDER]
    return np.stack(ys, axis=-1).reshape(*qw.shape[:-1], -1)


def pack_awq_gemm(q: np.ndarray) -> np.ndarray:
    """Inverse of unpack_awq_gemm (test helper to fabricate AWQ checkpoints)."""
    q = np.asarray(q, np.uint8).reshape(*q.shape[:-1], -1, 8).astype(np.int64)
    w = np.zeros(q.shape[:-1], np.int64)
    for p, nibble in enumerate(AWQ_ORDER):
        w |= q[..., p] << (4 * nibble)
    return w.astype(np.uint32).view(np.int32)


def pack_u4_row(q: np.ndarray) -> np.ndarray:
    """uint8[..., N] -> int32[..., N/8], nibble j of word c = element 8c+j.

    This is the boundary ("TM") layout handed to the engine (weight_format.py:63-74).
    """
    q = np.asarray(q, np.uint8).reshape(*q.shape[:-1], -1, 8).astype(np.int64)
    w = np.zeros(q.shape[:-1], np.int64)
    for j in range(8):
        w |= q[..., j] << (4 * j)
    return w.astype(np.uint32).view(np.int32)


def unpack_u4_row(w: np.ndarray) -> np.ndarray:
    w = np.asarray(w).view(np.uint32).astype(np.int64)
    nib = [((w >> (4 * j)) & 15).astype(np.uint8) for j in range(8)]
    return np.stack(nib, axis=-1).reshape(*w.shape[:-1], -1)


def awq_dequant_ref(q: np.ndarray, scales: np.ndarray, zeros: np.ndarray, group: int = 128) -> np.ndarray:
    """Golden formula (q - z) * s  (weight_format.py:236-247; awq_modules.py:37-46).

    q uint8[K,N], scales fp16[K/g,N], zeros fp16[K/g,N] -> fp16[K,N] computed the
    way torch does it for fp16 tensors (sub and mul each rounded to fp16).
    """
    K, N = q.shape
    qg = q.reshape(K // group, group, N).astype(f16)
    w = hmul(hsub(qg, zeros[:, None, :]), scales[:, None, :])
    return w.reshape(K, N)


# --------------------------------------------------------------------------
# Synthetic-weight quantiser == tm.QuantizeGroupwise semantics for fp16 x uint4
#   src/turbomind/kernels/quantization.cu:384-440 (IntegralQuantizer)
# --------------------------------------------------------------------------
def quantize_groupwise_u4(w: np.ndarray, group: int = 128):
    """w fp16 [K, N] (groups along K) -> (q uint8[K,N], scales fp16[K/g,N], zeros fp16[K/g,N], dequant fp16[K,N]).

    scale_ = max(max-min, 1e-5)/15 (f32); zero_ = clamp(-rint(min/scale_), 0, 15);
    q = clamp(rint(x/scale_) + zero_, 0, 15); d = h(q-zero_) * h(scale_).
    """
    K, N = w.shape
    x = np.asarray(w, f16).astype(f32).reshape(K // group, group, N)
    mn = x.min(axis=1, keepdims=True)
    mx = x.max(axis=1, keepdims=True)
    scale_ = (np.maximum(mx - mn, f32(1e-5)) / f32(15)).astype(f32)
    zero_ = np.clip(-np.rint(mn / scale_), 0, 15).astype(np.int32)
    q = np.clip(np.rint(x / scale_).astype(np.int32) + zero_, 0, 15)
    d = hmul((q - zero_).astype(f16), scale_.astype(f16))
    return (q.astype(np.uint8).reshape(K, N), scale_.astype(f16)[:, 0, :], zero_.astype(f16)[:, 0, :],
            d.reshape(K, N))


# --------------------------------------------------------------------------
# A2. W4A16 linear
#   kernels/gemm/cast.cu:134-165 (fuse (s, -z*s)), transform.h:68-74 (hfma),
#   kernels/attention/quantization.h:503-524 (u4 -> f16 exact), epilogue.h:159-176
# --------------------------------------------------------------------------
def fuse_scales_zeros(scales: np.ndarray, zeros: np.ndarray):
    """(s, z) -> (s, h(-z*s))  -- one fp16 rounding at load time (cast.cu:151-156)."""
    s = np.asarray(scales, f16)
    z = np.asarray(zeros, f16)
    return s, hmul(-z, s)


def w4a16_dequant(q: np.ndarray, scales: np.ndarray, zeros: np.ndarray, group: int = 128) -> np.ndarray:
    """w[k,n] = h(fma(h(q), s, h(-z*s)))  (transform.h:68-74)."""
    K, N = q.shape
    s, zs = fuse_scales_zeros(scales, zeros)
    qg = q.reshape(K // group, group, N).astype(f16)
    return hfma(qg, s[:, None, :], zs[:, None, :]).reshape(K, N)


def gemm_f16_f32acc(x: np.ndarray, w: np.ndarray) -> np.ndarray:
    """fp32 accumulators of x[M,K] (fp16) @ w[K,N] (fp16). Summation order is
    unspecified in the reference (tile / split-K dependent) => compare with tolerance."""
    return np.asarray(x, f16).astype(f32) @ np.asarray(w, f16).astype(f32)


def w4a16_linear(x, q, scales, zeros, group: int = 128) -> np.ndarray:
    return gemm_f16_f32acc(x, w4a16_dequant(q, scales, zeros, group)).astype(f16)


def silu_f32(g: np.ndarray) -> np.ndarray:
    g = np.asarray(g, f32)
    with np.errstate(over='ignore'):
        return (g / (f32(1) + np.exp(-g).astype(f32))).astype(f32)


def gated_silu_epilogue(acc: np.ndarray) -> np.ndarray:
    """acc fp32[M, 2I] with interleaved (gate_j, up_j) columns -> fp16[M, I]
    out = h(silu_f32(g) * u) on the fp32 accumulators (epilogue.h:159-176)."""
    g = acc[:, 0::2]
    u = acc[:, 1::2]
    return (silu_f32(g) * u.astype(f32)).astype(f16)


def w4a16_linear_gated_silu(x, q, scales, zeros, group: int = 128) -> np.ndarray:
    return gated_silu_epilogue(gemm_f16_f32acc(x, w4a16_dequant(q, scales, zeros, group)))


def silu_and_mul_unfused(gate_up_f16: np.ndarray) -> np.ndarray:
    """Unfused fallback kernels/activation.cu:27-130 on an fp16 [M,2I] tensor laid
    out [gate | up] (non-interleaved): out = h(silu_f32(f32(g)) * f32(u))."""
    M, N2 = gate_up_f16.shape
    g = gate_up_f16[:, : N2 // 2].astype(f32)
    u = gate_up_f16[:, N2 // 2:].astype(f32)
    return (silu_f32(g) * u).astype(f16)


def interleave_w1w3(w1: np.ndarray, w3: np.ndarray) -> np.ndarray:
    """[K,I],[K,I] -> [K,2I] with columns (2j, 2j+1) = (gate_j, up_j)
    (lmdeploy/turbomind/builders/ffn.py:31-34,164-165)."""
    K, I = w1.shape
    out = np.empty((K, 2 * I), w1.dtype)
    out[:, 0::2] = w1
    out[:, 1::2] = w3
    return out


# --------------------------------------------------------------------------
# A7. RMSNorm / residual RMSNorm
#   kernels/norm/rms_norm.cu:21-86,286-362 ; rms_norm_utils.cuh:6-15
# --------------------------------------------------------------------------
def rmsnorm(x: np.ndarray, w: np.ndarray, eps: float) -> np.ndarray:
    """y_i = h( h(f32(x_i) * inv) * w_i ), inv = rsqrt(sum(f32(x)^2)/H + eps)."""
    x = np.asarray(x, f16)
    xf = x.astype(f32)
    H = x.shape[-1]
    ss = (xf.astype(f64) ** 2).sum(-1, keepdims=True)      # fp32 accumulate in the kernel; order-free here
    inv = (1.0 / np.sqrt(ss / H + eps)).astype(f32)
    return hmul((xf * inv).astype(f16), np.asarray(w, f16))


def residual_rmsnorm(residual, hidden, w, eps, bias=None):
    """r <- h(r + hcur) (then h(r + bias)); y <- rmsnorm(r).  Returns (r, y).
    (rms_norm.cu:318-326: the residual stream accumulates in fp16.)"""
    r = hadd(residual, hidden)
    if bias is not None:
        r = hadd(r, bias)
    return r, rmsnorm(r, w, eps)


def rmsnorm_torch_default(x, w, eps, residual=None):
    """The PyTorchEngine 'default' backend form (backends/default/norm.py:14-28):
    everything in fp32, single final cast.  Used by the CPU baseline and by the
    golden fixture that pins the formula."""
    x = np.asarray(x)
    in_dtype = x.dtype
    if residual is not None:
        x = (x.astype(f32) + np.asarray(residual).astype(f32)).astype(in_dtype)
        residual = x
    xf = x.astype(f32)
    var = (xf * xf).mean(-1, keepdims=True, dtype=f32)
    y = (np.asarray(w).astype(f32) * (xf * (f32(1) / np.sqrt(var + f32(eps))))).astype(in_dtype)
    return y if residual is None else (y, residual)


# --------------------------------------------------------------------------
# A6. RoPE (interleaved pairs)
#   kernels/attention/rotary_embedding.h:11-52,169-181 ; models/attention_weight.cc:37-93
# --------------------------------------------------------------------------
@dataclass
class RopeParam:
    dim: int = 128
    base: float = 10000.0
    type: str = 'default'            # 'default' | 'linear' | 'llama3'
    factor: float = 1.0
    # llama3
    low_freq_factor: float = 1.0
    high_freq_factor: float = 4.0
    original_max_position_embeddings: int = 8192


def rope_inv_freq(p: RopeParam) -> np.ndarray:
    """inv_freq for even channel index i (fp32), i = 0,2,..,dim-2.

    freq = exp2f(i * scale_factor), scale_factor = -log2(base)/dim.
    exp2 is evaluated in float64 on the float32 product and rounded to float32 so that
    the engine's host-side table (same recipe in C++) is reproducible bit for bit.
    """
    i = np.arange(0, p.dim, 2, dtype=f32)
    scale_factor = f32(-math.log2(p.base) / p.dim)
    freq = np.exp2((i * scale_factor).astype(f32).astype(f64)).astype(f32)
    if p.type in ('default', 'linear'):
        inv_factor = f32(1.0 / p.factor) if p.type == 'linear' else f32(1.0)
        return (inv_factor * freq).astype(f32)
    if p.type == 'llama3':
        # attention_weight.cc:37-93: alpha = orig/(2pi)/(high-low), beta = low/(high-low)
        inv_diff = 1.0 / (p.high_freq_factor - p.low_freq_factor)
        alpha = f32(p.original_max_position_embeddings / (2 * math.pi) * inv_diff)
        beta = f32(p.low_freq_factor * inv_diff)
        inv_factor = f32(1.0 / p.factor)
        smooth = np.clip(alpha * freq - beta, f32(0), f32(1)).astype(f32)
        return ((f32(1) - smooth) * freq * inv_factor + smooth * freq).astype(f32)
    raise ValueError(p.type)


def rope_cos_sin(p: RopeParam, positions: np.ndarray):
    """(cos, sin) fp16 [len(positions), dim/2]: angle = f32(t)*inv_freq (fp32 multiply),
    sin/cos evaluated in float64 of that fp32 angle, rounded to fp32 then fp16."""
    inv = rope_inv_freq(p)
    ang = (np.asarray(positions, f32)[:, None] * inv[None, :]).astype(f32).astype(f64)
    return np.cos(ang).astype(f32).astype(f16), np.sin(ang).astype(f32).astype(f16)


def rope_apply(x: np.ndarray, cos: np.ndarray, sin: np.ndarray) -> np.ndarray:
    """x fp16 [..., T, heads, D]; cos/sin fp16 [T, D/2] -> rotated fp16.
    x'[2i] = h(h(c*x[2i]) - h(s*x[2i+1])), x'[2i+1] = h(h(c*x[2i+1]) + h(s*x[2i]))
    (rotary_embedding.h:169-181; fp16 ops, no fma)."""
    x = np.asarray(x, f16)
    D = x.shape[-1]
    rd = cos.shape[-1] * 2
    c = cos[:, None, :]
    s = sin[:, None, :]
    x0 = x[..., 0:rd:2]
    x1 = x[..., 1:rd:2]
    y0 = hsub(hmul(c, x0), hmul(s, x1))
    y1 = hadd(hmul(c, x1), hmul(s, x0))
    out = x.copy()
    out[..., 0:rd:2] = y0
    out[..., 1:rd:2] = y1
    return out


def permute_qk_for_interleaved_rope(w: np.ndarray, heads: int, D: int) -> np.ndarray:
    """Loader-side column permutation (lmdeploy/turbomind/models/utils.py:306-324): HF
    'rotate-half' channel j (first half) and j+D/2 become the adjacent pair (2j,2j+1).
    w [..., heads*D] -> same shape."""
    lead = w.shape[:-1]
    return w.reshape(*lead, heads, 2, D // 2).swapaxes(-1, -2).reshape(*lead, heads * D)


# --------------------------------------------------------------------------
# A3/A4. KV quantise / dequantise
#   kernels/attention/quantization.h:316-366,428-489 (store), 492-574,665-704 (load)
# --------------------------------------------------------------------------
KV_INT4_NIBBLE_ORDER = (0, 2, 4, 6, 1, 3, 5, 7)   # nibble i of the word holds element ORDER[i]


def kv_quant_params(x: np.ndarray, bits: int):
    """x fp16 [..., D] -> (scale fp16 [...], zero fp16 [...])."""
    x = np.asarray(x, f16)
    mn = x.min(-1)
    mx = x.max(-1)
    inv_q_max = f32(1.0) / f32((1 << bits) - 1)
    scale = ((mx.astype(f32) - mn.astype(f32)) * inv_q_max).astype(f16)
    return scale, mn


def kv_quantize_values(x: np.ndarray, scale: np.ndarray, zero: np.ndarray, bits: int) -> np.ndarray:
    """q = sat_u8(rne(h(h(x - zero) * inv))), inv = h(1/f32(scale)); b==4: min(q, 15).
    Degenerate scale==0 -> inv=inf -> 0*inf = NaN -> q = 0 (cvt.rni.sat of NaN)."""
    x = np.asarray(x, f16)
    with np.errstate(divide='ignore', invalid='ignore', over='ignore'):
        inv = (f32(1.0) / np.asarray(scale, f16).astype(f32)).astype(f16)
        y = hmul(hsub(x, np.asarray(zero, f16)[..., None]), inv[..., None]).astype(f32)
        r = np.rint(y)                       # RNE
        r = np.where(np.isnan(r), 0.0, r)
        q = np.clip(r, 0, 255).astype(np.uint8)
    if bits == 4:
        q = np.minimum(q, 15).astype(np.uint8)
    return q


def kv_pack_int4(q: np.ndarray) -> np.ndarray:
    """uint8 [..., D] (values 0..15) -> uint8 bytes [..., D/2]; per 8 values one
    little-endian 32-bit word with nibbles [q0,q2,q4,q6,q1,q3,5,q7] (quantization.h:459-471)."""
    q = np.asarray(q, np.uint8)
    g = q.reshape(*q.shape[:-1], -1, 8).astype(np.uint32)
    w = np.zeros(g.shape[:-1], np.uint32)
    for nib, el in enumerate(KV_INT4_NIBBLE_ORDER):
        w |= g[..., el] << np.uint32(4 * nib)
    return w.view(np.uint8).reshape(*q.shape[:-1], -1) if w.flags['C_CONTIGUOUS'] else \
        np.ascontiguousarray(w).view(np.uint8).reshape(*q.shape[:-1], -1)


def kv_unpack_int4(b: np.ndarray) -> np.ndarray:
    b = np.ascontiguousarray(np.asarray(b, np.uint8))
    w = b.reshape(*b.shape[:-1], -1, 4).view(np.uint32)[..., 0]
    out = np.zeros((*w.shape, 8), np.uint8)
    for nib, el in enumerate(KV_INT4_NIBBLE_ORDER):
        out[..., el] = (w >> np.uint32(4 * nib)) & 15
    return out.reshape(*b.shape[:-1], -1)


def kv_quantize(x: np.ndarray, bits: int):
    """x fp16 [..., D] -> (data bytes uint8 [..., D*bits/8], params fp16 [..., 2] = (scale, zero))."""
    scale, zero = kv_quant_params(x, bits)
    q = kv_quantize_values(x, scale, zero, bits)
    data = q if bits == 8 else kv_pack_int4(q)
    return data, np.stack([scale, zero], -1)


def kv_dequant_decode(q: np.ndarray, scale, zero) -> np.ndarray:
    """Decode-kernel form: h(fma(h(q), scale, zero)) (impl_81616.h:338-347)."""
    return hfma(np.asarray(q).astype(f16), np.asarray(scale, f16)[..., None], np.asarray(zero, f16)[..., None])


def kv_dequant_flatten(q: np.ndarray, scale, zero) -> np.ndarray:
    """Flatten (prefill) form: h(h(h(q)*scale) + zero) (quantization.h:565-573,694-703)."""
    return hadd(hmul(np.asarray(q).astype(f16), np.asarray(scale, f16)[..., None]), np.asarray(zero, f16)[..., None])


# --------------------------------------------------------------------------
# A5. Cache block byte layout  (kernels/attention/block.h:126-219)
# --------------------------------------------------------------------------
@dataclass
class BlockLayout:
    layers: int
    kv_heads: int
    head_dim: int = 128
    block_len: int = 64
    bits: int = 8            # 16 (fp16, unquantised) | 8 | 4

    @property
    def token_data_size(self):
        return self.bits * self.head_dim // 8

    @property
    def token_param_size(self):
        return 4 if self.bits < 16 else 0      # t_bits*2/8 with T=fp16

    @property
    def head_data_size(self):
        return self.block_len * self.token_data_size

    @property
    def head_param_size(self):
        return self.block_len * self.token_param_size

    @property
    def layer_size(self):
        return self.kv_heads * 2 * self.head_data_size + self.kv_heads * 2 * self.head_param_size

    @property
    def block_size(self):
        return self.layers * self.layer_size

    def layer_offset(self, layer):
        return layer * self.layer_size

    def k_data(self, head, ti):
        return head * 2 * self.head_data_size + ti * self.token_data_size

    def v_data(self, head, ti):
        return self.k_data(head, ti) + self.head_data_size

    def k_param(self, head, ti):
        return self.kv_heads * 2 * self.head_data_size + head * 2 * self.head_param_size + ti * self.token_param_size

    def v_param(self, head, ti):
        return self.k_param(head, ti) + self.head_param_size


class PagedKVCache:
    """Byte-exact model of the paged cache: a pool of blocks + per-sequence block tables
    (unified_attention_layer.cc:153-171,288-305; block.h:93-97)."""

    def __init__(self, layout: BlockLayout, num_blocks: int):
        self.layout = layout
        self.pool = np.zeros((num_blocks, layout.block_size), np.uint8)

    def store_token(self, block_table, layer, t, k, v):
        """k, v fp16 [kv_heads, D] for sequence position t (K already RoPE'd)."""
        L = self.layout
        blk = self.pool[block_table[t // L.block_len]]
        ti = t % L.block_len
        base = L.layer_offset(layer)
        for hd in range(L.kv_heads):
            for (x, doff, poff) in ((k[hd], L.k_data(hd, ti), L.k_param(hd, ti)),
                                    (v[hd], L.v_data(hd, ti), L.v_param(hd, ti))):
                if L.bits == 16:
                    blk[base + doff: base + doff + L.token_data_size] = np.asarray(x, f16).view(np.uint8)
                else:
                    data, param = kv_quantize(np.asarray(x, f16), L.bits)
                    blk[base + doff: base + doff + L.token_data_size] = data
                    blk[base + poff: base + poff + 4] = np.asarray(param, f16).view(np.uint8)

    def load_raw(self, block_table, layer, head, t0, t1):
        """-> (kq, vq, kparam, vparam) for tokens [t0,t1): integer codes (or fp16 values when bits==16)."""
        L = self.layout
        ks, vs, kp, vp = [], [], [], []
        base = L.layer_offset(layer)
        for t in range(t0, t1):
            blk = self.pool[block_table[t // L.block_len]]
            ti = t % L.block_len
            for (lst, plst, doff, poff) in ((ks, kp, L.k_data(head, ti), L.k_param(head, ti)),
                                            (vs, vp, L.v_data(head, ti), L.v_param(head, ti))):
                raw = blk[base + doff: base + doff + L.token_data_size]
                if L.bits == 16:
                    lst.append(raw.view(f16).copy())
                    plst.append(np.array([1, 0], f16))
                else:
                    lst.append(raw.copy() if L.bits == 8 else kv_unpack_int4(raw))
                    plst.append(blk[base + poff: base + poff + 4].view(f16).copy())
        return np.stack(ks), np.stack(vs), np.stack(kp), np.stack(vp)

    def load_dequant(self, block_table, layer, head, t0, t1, form='decode'):
        kq, vq, kp, vp = self.load_raw(block_table, layer, head, t0, t1)
        if self.layout.bits == 16:
            return kq, vq
        fn = kv_dequant_decode if form == 'decode' else kv_dequant_flatten
        return fn(kq, kp[:, 0], kp[:, 1]), fn(vq, vp[:, 0], vp[:, 1])


def process_kv(cache: PagedKVCache, block_table, layer, k, v, cos, sin, t_begin):
    """ProcessKV_v2 (kv_cache_utils_v2.cu:18-210): k,v fp16 [T, kv_heads, D] new tokens at
    positions t_begin.. ; K <- RoPE(K); quantise; scatter into blocks."""
    kr = rope_apply(k, cos, sin) if cos is not None else np.asarray(k, f16)
    for i in range(k.shape[0]):
        cache.store_token(block_table, layer, t_begin + i, kr[i], v[i])
    return kr


def flatten_kv(cache: PagedKVCache, block_table, layer, ctx_len):
    """flattenKV_v2 (kv_cache_utils_v2.cu:340-468): -> K,V fp16 [kv_heads, ctx, D], dequantised
    with the flatten (two-rounding) form."""
    L = cache.layout
    ks, vs = [], []
    for hd in range(L.kv_heads):
        k, v = cache.load_dequant(block_table, layer, hd, 0, ctx_len, form='flatten')
        ks.append(k)
        vs.append(v)
    return np.stack(ks), np.stack(vs)


# --------------------------------------------------------------------------
# A8. Attention
#   attention_universal.h:398-553 ; impl_81616.h:510-591 ; reduce.cu:57-226
# --------------------------------------------------------------------------
def _exp2f(x):
    with np.errstate(over='ignore', under='ignore', invalid='ignore'):
        return np.exp2(np.asarray(x, f32).astype(f64)).astype(f32)


def attention_tiles(q, K, V, scale_log2, tile=64, t_begin=0, t_end=None):
    """One split of the streaming softmax over tiles of `tile` cached tokens, newest first.

    q fp16 [G, D] (the G query heads of one kv head), K,V fp16 [ctx, D] (dequantised).
    Returns partial (O f32 [G, D], M f32 [G], L f32 [G]) with the reference's recurrences:
      S = f32 sum_d k[d]*q[d]; m' = max(m, max S); O *= exp2((m-m')c); L = L*exp2((m-m')c) + sum exp2(S c - m' c);
      P = h(exp2(S c - m' c)); O += sum f32(P) f32(v).
    """
    ctx = K.shape[0]
    t_end = ctx if t_end is None else t_end
    G, D = q.shape
    O = np.zeros((G, D), f32)
    M = np.full((G,), -np.inf, f32)
    Lsum = np.zeros((G,), f32)
    qf = np.asarray(q, f16).astype(f32)
    c = f32(scale_log2)
    # tiles are aligned to `tile` from position 0 (cache blocks), iterated high -> low
    first = (t_begin // tile) * tile
    starts = list(range(first, t_end, tile))
    for s0 in reversed(starts):
        a, b = max(s0, t_begin), min(s0 + tile, t_end)
        if a >= b:
            continue
        Kt = np.asarray(K[a:b], f16).astype(f32)
        Vt = np.asarray(V[a:b], f16).astype(f32)
        S = (qf @ Kt.T).astype(f32)                       # [G, n]
        mnew = np.maximum(M, S.max(-1))
        with np.errstate(invalid='ignore'):
            alpha = np.where(np.isinf(M) & (M < 0), f32(0), _exp2f((M - mnew) * c))
        pf = _exp2f(S * c - (mnew * c)[:, None])
        Lsum = (Lsum * alpha + pf.sum(-1, dtype=f32)).astype(f32)
        P = pf.astype(f16).astype(f32)
        O = (O * alpha[:, None] + P @ Vt).astype(f32)
        M = mnew.astype(f32)
    return O, M, Lsum


def attention_merge(parts, scale_log2):
    """Split-K merge (reduce.cu:57-226): m* = max m_i; w_i = exp2((m_i-m*)c);
    out = h(sum w_i O_i / sum w_i L_i).  Empty splits (L==0) contribute nothing."""
    c = f32(scale_log2)
    Ms = np.stack([p[1] for p in parts])          # [S, G]
    mstar = Ms.max(0)
    with np.errstate(invalid='ignore'):
        w = np.where(np.isinf(Ms) & (Ms < 0), f32(0), _exp2f((Ms - mstar[None]) * c))
    Lt = sum(w[i] * parts[i][2] for i in range(len(parts)))
    Ot = sum(w[i][:, None] * parts[i][0] for i in range(len(parts)))
    with np.errstate(invalid='ignore', divide='ignore'):
        return (Ot / Lt[:, None]).astype(f16)


def split_ranges(ctx_len: int, splits: int, tile: int = 64):
    """Tiles per split = ceil(tile_count/split_cnt) (attention_universal.h:421-427)."""
    tiles = (ctx_len + tile - 1) // tile
    per = (tiles + splits - 1) // splits
    out = []
    for s in range(splits):
        a = s * per * tile
        b = min((s + 1) * per * tile, ctx_len)
        out.append((a, max(a, b)))
    return out


def decode_attention(q, K, V, softmax_scale=None, splits=1, tile=64):
    """q fp16 [Hq, D] (already RoPE'd), K,V fp16 [Hkv, ctx, D] (dequantised, incl. the new token).
    -> out fp16 [Hq, D]."""
    Hq, D = q.shape
    Hkv, ctx, _ = K.shape
    G = Hq // Hkv
    scale = (1.0 / math.sqrt(D)) if softmax_scale is None else softmax_scale
    c = f32(scale * math.log2(math.e))
    out = np.zeros((Hq, D), f16)
    for g in range(Hkv):
        qs = q[g * G:(g + 1) * G]
        if splits == 1:
            O, M, Ls = attention_tiles(qs, K[g], V[g], c, tile)
            with np.errstate(invalid='ignore', divide='ignore'):
                out[g * G:(g + 1) * G] = (O / Ls[:, None]).astype(f16)
        else:
            parts = [attention_tiles(qs, K[g], V[g], c, tile, a, b) for (a, b) in split_ranges(ctx, splits, tile)
                     if b > a]
            out[g * G:(g + 1) * G] = attention_merge(parts, c)
    return out


def prefill_attention(q, K, V, history=0, softmax_scale=None, tile=64):
    """Causal attention for T new tokens on top of `history` cached ones.
    q fp16 [T, Hq, D] (RoPE'd), K,V fp16 [Hkv, history+T, D] (flattened round-tripped KV)."""
    T, Hq, D = q.shape
    out = np.zeros((T, Hq, D), f16)
    for i in range(T):
        ctx = history + i + 1
        out[i] = decode_attention(q[i], K[:, :ctx], V[:, :ctx], softmax_scale, 1, tile)
    return out


def attention_reference_unfused(q, K, V, softmax_scale=None):
    """The reference's own unfused test oracle (kernels/attention/reference.cu:252-367):
    QK^T in fp32 -> softmax -> PV; used to cross-check the tiled restatement."""
    Hq, D = q.shape
    Hkv = K.shape[0]
    G = Hq // Hkv
    scale = (1.0 / math.sqrt(D)) if softmax_scale is None else softmax_scale
    out = np.zeros((Hq, D), f64)
    for hq in range(Hq):
        g = hq // G
        s = (K[g].astype(f64) @ q[hq].astype(f64)) * scale
        p = np.exp(s - s.max())
        p /= p.sum()
        out[hq] = p @ V[g].astype(f64)
    return out


# --------------------------------------------------------------------------
# A9. Embedding + greedy
# --------------------------------------------------------------------------
def embedding_lookup(table: np.ndarray, ids: np.ndarray) -> np.ndarray:
    return np.asarray(table, f16)[np.asarray(ids)]


def lm_head(hidden: np.ndarray, w_out: np.ndarray) -> np.ndarray:
    """logits fp16 [B,V] = h(f32acc(hidden @ W)) (language_model.cc:285-335: fp16 dense GEMM)."""
    return gemm_f16_f32acc(hidden, w_out).astype(f16)


def greedy(logits: np.ndarray) -> np.ndarray:
    """top_k = 1 on fp32-cast logits (generation/sampling.cc:92-183). Ties: lowest index here;
    the reference does not guarantee tie order => tests avoid exact ties."""
    return np.asarray(logits).astype(f32).argmax(-1).astype(np.int32)


# --------------------------------------------------------------------------
# Whole-model restatement (layer assembly; unified_decoder.cc:163-380,
# unified_attention_layer.cc:365-441, LlamaFfnLayer.cc:28-91, language_model.cc:493-542)
# --------------------------------------------------------------------------
@dataclass
class ModelConfig:
    hidden: int
    layers: int
    q_heads: int
    kv_heads: int
    head_dim: int
    inter: int
    vocab: int
    rms_eps: float = 1e-5
    rope: RopeParam = field(default_factory=RopeParam)
    group: int = 128
    kv_bits: int = 8
    block_len: int = 64
    # BASELINE config 5: fp8 block-scaled weights, mixture-of-experts FFN (0 experts = dense)
    weight_format: str = 'u4'          # 'u4' (AWQ g128) | 'fp8' (e4m3, 128x128 block scales)
    moe_experts: int = 0
    moe_top_k: int = 0
    moe_norm_topk: bool = True
    moe_routed_scale: float = 1.0
    # fp8 experts contracted on fp8 matrix cores with on-the-fly activation quantisation (the reference's fp8 GEMM path,
    # LlamaLinear.cu:67-127) instead of the weight-only dequant path; attention / dense linears stay weight-only
    moe_fp8_act: bool = False


LLAMA3_8B = dict(hidden=4096, layers=32, q_heads=32, kv_heads=8, head_dim=128, inter=14336, vocab=128256,
                 rms_eps=1e-5)
INTERNLM2_1_8B = dict(hidden=2048, layers=24, q_heads=16, kv_heads=8, head_dim=128, inter=8192, vocab=92544,
                      rms_eps=1e-5)
INTERNLM2_20B = dict(hidden=6144, layers=48, q_heads=48, kv_heads=8, head_dim=128, inter=16384, vocab=92544,
                     rms_eps=1e-5)
LLAMA3_70B = dict(hidden=8192, layers=80, q_heads=64, kv_heads=8, head_dim=128, inter=28672, vocab=128256,
                  rms_eps=1e-5)


def make_synthetic_weights(cfg: ModelConfig, seed: int = 0, quantized: bool = True):
    """Random weights with real shapes: fp16 master N(0,1)*0.1/sqrt(K), group-128
    asymmetric u4 quantisation; norms 1+0.02 N; embeddings N(0,0.02).
    Layout = the boundary ("TM") layout: q [K,N] u4 as uint8, scales/zeros fp16 [K/g,N];
    w_qkv = [Q|K|V] fused along N (Q/K columns already in interleaved-RoPE order);
    w1w3 interleaved (gate_j, up_j)."""
    rng = np.random.default_rng(seed)
    H, D = cfg.hidden, cfg.head_dim
    nq, nkv, I = cfg.q_heads * D, cfg.kv_heads * D, cfg.inter

    def lin(K, N):
        w = (rng.standard_normal((K, N), dtype=f32) * (0.1 / math.sqrt(K))).astype(f16)
        if not quantized:
            return dict(w=w)
        if cfg.weight_format == 'fp8':
            f8, bs = fp8_quantize_blockwise(w)
            return dict(f8=f8, bs=bs)
        q, s, z, _ = quantize_groupwise_u4(w, cfg.group)
        return dict(q=q, s=s, z=z)

    def lin13(K, I2):
        """fused w1w3 [K, 2I], (gate_j, up_j) interleaved.  fp8: w1 and w3 are block-quantised separately (as in a
        checkpoint); codes interleaved, scale row = [w1 blocks | w3 blocks] ('gated' layout)"""
        if not (quantized and cfg.weight_format == 'fp8'):
            return lin(K, I2)
        w = (rng.standard_normal((K, I2), dtype=f32) * (0.1 / math.sqrt(K))).astype(f16)
        (g8, gs), (u8, us) = fp8_quantize_blockwise(w[:, 0::2]), fp8_quantize_blockwise(w[:, 1::2])
        f8 = np.empty((K, I2), np.uint8)
        f8[:, 0::2], f8[:, 1::2] = g8, u8
        return dict(f8=f8, bs=np.concatenate([gs, us], axis=1), gated=True)

    layers = []
    for _ in range(cfg.layers):
        layers.append(dict(
            attn_norm=(1 + 0.02 * rng.standard_normal(H, dtype=f32)).astype(f16),
            w_qkv=lin(H, nq + 2 * nkv),
            wo=lin(nq, H),
            ffn_norm=(1 + 0.02 * rng.standard_normal(H, dtype=f32)).astype(f16),
            **(dict(moe_gate=(0.2 * rng.standard_normal((H, cfg.moe_experts), dtype=f32)).astype(f16),
                    experts=[dict(w1w3=lin13(H, 2 * I), w2=lin(I, H)) for _ in range(cfg.moe_experts)])
               if cfg.moe_experts else dict(w1w3=lin13(H, 2 * I), w2=lin(I, H))),
        ))
    return dict(
        tok_embeddings=(0.02 * rng.standard_normal((cfg.vocab, H), dtype=f32)).astype(f16),
        layers=layers,
        norm=(1 + 0.02 * rng.standard_normal(H, dtype=f32)).astype(f16),
        output=(rng.standard_normal((H, cfg.vocab), dtype=f32) * (0.1 / math.sqrt(H))).astype(f16),
    )


def _dense_weight(W, group):
    if 'q' in W:
        return w4a16_dequant(W['q'], W['s'], W['z'], group)
    if 'f8' in W:
        return fp8_dequant(W['f8'], W['bs'], W.get('gated', False))
    return W['w']


def _linear(x, W, group, gated=False):
    if '_dense' not in W:      # dequantise once per weight (full-width parity tests re-use a layer for several forwards)
        W['_dense'] = np.asarray(_dense_weight(W, group), f16).astype(f32)
    acc = np.asarray(x, f16).astype(f32) @ W['_dense']
    return gated_silu_epilogue(acc) if gated else acc.astype(f16)


class OracleModel:
    """Static-batch Llama-style decoder that reproduces the reference's data flow:
    prefill = ProcessKV -> FlattenKV -> causal attention on round-tripped KV;
    decode = RoPE + quantise-and-store the new token, then attention over the paged cache
    with the decode (fma) dequant form."""

    def __init__(self, cfg: ModelConfig, weights, batch: int, max_ctx: int):
        self.cfg, self.w = cfg, weights
        self.layout = BlockLayout(cfg.layers, cfg.kv_heads, cfg.head_dim, cfg.block_len, cfg.kv_bits)
        nblk = (max_ctx + cfg.block_len - 1) // cfg.block_len
        self.cache = PagedKVCache(self.layout, batch * nblk)
        # deliberately non-identity block tables (the reference shuffles them in its tests)
        perm = np.random.default_rng(1234).permutation(batch * nblk)
        self.tables = [perm[b * nblk:(b + 1) * nblk] for b in range(batch)]
        self.seq_len = [0] * batch
        self.c = 1.0 / math.sqrt(cfg.head_dim)

    def _layer_io(self, x_norm, li):
        return _linear(x_norm, self.w['layers'][li]['w_qkv'], self.cfg.group)

    def forward(self, ids_per_seq, decode_splits=1):
        """ids_per_seq: list of int arrays (new tokens per sequence; len 1 => decode row).
        Returns next-token ids [B] and logits fp16 [B, V]."""
        cfg = self.cfg
        D, Hq, Hkv = cfg.head_dim, cfg.q_heads, cfg.kv_heads
        lens = [len(t) for t in ids_per_seq]
        ids = np.concatenate([np.asarray(t, np.int64) for t in ids_per_seq])
        offs = np.concatenate([[0], np.cumsum(lens)])
        resid = embedding_lookup(self.w['tok_embeddings'], ids)          # [T, H] fp16
        x = rmsnorm(resid, self.w['layers'][0]['attn_norm'], cfg.rms_eps)
        for li, Lw in enumerate(self.w['layers']):
            qkv = _linear(x, Lw['w_qkv'], cfg.group)
            attn = np.zeros((len(ids), Hq * D), f16)
            for b, n in enumerate(lens):
                if n == 0:
                    continue
                sl = slice(offs[b], offs[b + 1])
                hist = self.seq_len[b]
                pos = np.arange(hist, hist + n)
                cos, sin = rope_cos_sin(cfg.rope, pos)
                q = qkv[sl, :Hq * D].reshape(n, Hq, D)
                k = qkv[sl, Hq * D:(Hq + Hkv) * D].reshape(n, Hkv, D)
                v = qkv[sl, (Hq + Hkv) * D:].reshape(n, Hkv, D)
                q = rope_apply(q, cos, sin)
                process_kv(self.cache, self.tables[b], li, k, v, cos, sin, hist)
                if n == 1:
                    Ks, Vs = [], []
                    for hd in range(Hkv):
                        kd, vd = self.cache.load_dequant(self.tables[b], li, hd, 0, hist + 1, 'decode')
                        Ks.append(kd)
                        Vs.append(vd)
                    o = decode_attention(q[0], np.stack(Ks), np.stack(Vs), self.c, decode_splits)
                    attn[sl] = o.reshape(1, -1)
                else:
                    Kf, Vf = flatten_kv(self.cache, self.tables[b], li, hist + n)
                    attn[sl] = prefill_attention(q, Kf, Vf, hist, self.c).reshape(n, -1)
            o = _linear(attn, Lw['wo'], cfg.group)
            resid, x = residual_rmsnorm(resid, o, Lw['ffn_norm'], cfg.rms_eps)
            if cfg.moe_experts and cfg.moe_fp8_act and cfg.weight_format == 'fp8':
                exq = [((E_['w1w3']['f8'], E_['w1w3']['bs']), (E_['w2']['f8'], E_['w2']['bs'])) for E_ in Lw['experts']]
                d, _, _ = moe_ffn_fp8(x, Lw['moe_gate'], exq, cfg.moe_top_k, cfg.moe_norm_topk, cfg.moe_routed_scale)
            elif cfg.moe_experts:
                ex = [(_dense_weight(E_['w1w3'], cfg.group), _dense_weight(E_['w2'], cfg.group)) for E_ in Lw['experts']]
                d, _, _ = moe_ffn(x, Lw['moe_gate'], ex, cfg.moe_top_k, cfg.moe_norm_topk, cfg.moe_routed_scale)
            else:
                act = _linear(x, Lw['w1w3'], cfg.group, gated=True)
                d = _linear(act, Lw['w2'], cfg.group)
            nxt = self.w['layers'][li + 1]['attn_norm'] if li + 1 < cfg.layers else self.w['norm']
            resid, x = residual_rmsnorm(resid, d, nxt, cfg.rms_eps)
        last = np.array([offs[b + 1] - 1 for b in range(len(lens)) if lens[b] > 0])
        self.last_resid = resid                                           # residual stream after the last layer [T, H]
        logits = lm_head(x[last], self.w['output'])
        for b, n in enumerate(lens):
            self.seq_len[b] += n
        return greedy(logits), logits


# ------------------------------------------------------------------------------------------------
# Sampling (generation/sampling.cc:92-183; kernels/sampling_topk_kernels.cu, sampling_topp_kernels.cu:291-384,
# sampling_kernels.cu:16-94; temperature: generation/logits_processor.cc:105-112)
# ------------------------------------------------------------------------------------------------
def logits_process(logits, seen_ids, repetition_penalty=1.0, bad_ids=(), end_ids=(), k_len=0, min_len=0, vocab_offset=0):
    """Logits processors of ONE sequence, in the reference's order (LogitsProcessor::Forward,
    src/turbomind/generation/logits_processor.cc:66-115):
      1. repetition penalty (RepetitionPenaltyKernel, kernels/sampling_penalty_kernels.cu:137-175): every id that occurs
         in `seen_ids` (prompt + generated so far, each id once):  l < 0 ? l * p : l / p   in fp32;
      2. bad ids (BanBadWordsKernel single-token case, kernels/ban_bad_words.cu:51-95): l = -max; ids <= 0 skipped;
      3. min length (batchApplyMinLengthPenalty, sampling_penalty_kernels.cu:198-215): end ids (> 0) get -max while
         k_len + 1 < min_len  (k_len = sequence length of this step, min_len = prompt length + min_new_tokens).
    logits: fp16 [V] = columns vocab_offset .. vocab_offset + V of the vocabulary.  Returns fp16 [V]: the reference keeps
    the fp32 values, the MI355X path rounds them to fp16 once (its sampler consumes fp16), -max = -65504."""
    l16 = np.asarray(logits, np.float16)
    V = l16.shape[0]
    x = l16.astype(np.float32)
    p = np.float32(repetition_penalty)
    if p != 1 and p > 0:
        ids = np.unique(np.asarray(seen_ids, np.int64)) - vocab_offset
        ids = ids[(ids >= 0) & (ids < V)]
        v = x[ids]
        x[ids] = np.where(v < 0, v * p, v / p).astype(np.float32)
    with np.errstate(over='ignore'):
        out = x.astype(np.float16)
    ban = [int(t) for t in bad_ids if t > 0]
    if k_len + 1 < min_len:
        ban += [int(t) for t in end_ids if t > 0]
    for t in ban:
        c = t - vocab_offset
        if 0 <= c < V:
            out[c] = np.float16(-65504.0)
    return out


def sample_filter(logits: np.ndarray, temperature: float = 1.0, top_k: int = 0, top_p: float = 1.0, min_p: float = 0.0):
    """One row.  Returns (token ids of the surviving candidates in sampling order, their renormalised probabilities).
    Order: descending logit, ties by ascending token id (a stable descending sort, as cub's radix sort gives the
    reference).  top-k keeps the first k; softmax over them; top-p keeps the shortest prefix whose cumulative
    probability EXCEEDS top_p (sampling_topp_kernels.cu:318-338); min-p drops p < min_p * p_max (:341-369); the
    survivors are divided by their mass (:372-382)."""
    v = np.asarray(logits, np.float16).astype(np.float32)
    ids = np.arange(v.size)
    key = np.where(np.isnan(v), -np.inf, v)
    order = np.lexsort((ids, -key))
    k = v.size if top_k <= 0 else min(int(top_k), v.size)
    cand = order[:k]
    inv_t = np.float32(1.0) / np.float32(temperature if temperature > 0 else 1.0)
    z = ((v[cand] - v[cand[0]]).astype(np.float32) * inv_t).astype(np.float32)     # fp32, like the kernel
    w = np.exp(z.astype(np.float64))
    p = w / w.sum()
    kept, ksum = k, 1.0
    if top_p < 1.0:
        cum = np.cumsum(p)
        hit = np.flatnonzero(cum > top_p)
        if hit.size:
            kept, ksum = int(hit[0]) + 1, float(cum[hit[0]])
    if min_p > 0.0:
        ok = p[:kept] >= p[0] * min_p
        kept = int(ok.sum())            # a prefix, because p is non-increasing
        ksum = float(p[:kept].sum())
    return cand[:kept], p[:kept] / ksum


def sample_logprobs(ids: np.ndarray, probs: np.ndarray, selected: int, cap: int = 1024):
    """The `sampled_logprobs / sampled_indexes / sampled_nums` outputs of the reference's sampling kernel
    (kernels/sampling_kernels.cu:67-90; kMaxLogProb = 1024, utils/constant.h:7) for one row whose kept candidates are
    (ids, probs) = sample_filter(...): the first min(n, cap) candidates with logf(p); with cap == 1024 a drawn token at
    position >= 1024 replaces entry 1023 (:76-81).  Returns (token ids, float32 logprobs, logprob of `selected`)."""
    n = min(len(ids), cap)
    with np.errstate(divide='ignore'):
        lp = np.log(np.asarray(probs, np.float64).astype(np.float32)).astype(np.float32)
    out_ids, out_lp = np.asarray(ids[:n]).copy(), lp[:n].copy()
    pos = int(np.flatnonzero(np.asarray(ids) == selected)[0])
    if cap == 1024 and len(ids) > cap and pos >= cap:
        out_ids[cap - 1], out_lp[cap - 1] = selected, lp[pos]
    return out_ids, out_lp, lp[pos]


def logprobs_view(ids, lps, sel_lp, token: int, topn: int) -> dict:
    """What the Python side makes of one generated token's record (lmdeploy/turbomind/turbomind.py:472-503): the first
    min(n, topn) candidates as {token id: logprob}, the generated token added when it is not among them, -inf entries dropped."""
    res = {int(i): float(v) for i, v in zip(ids[:topn], lps[:topn])}
    if token not in res:
        res[int(token)] = float(sel_lp)
    return {k: v for k, v in res.items() if v != float('-inf')}


def sample_draw(ids: np.ndarray, probs: np.ndarray, u: float) -> int:
    """first candidate whose inclusive prefix sum exceeds u, else the last one (sampling_kernels.cu:46-63)"""
    cum = np.cumsum(probs)
    hit = np.flatnonzero(cum > u)
    return int(ids[hit[0]] if hit.size else ids[-1])


def philox_uniform(seed: int, counter: int) -> float:
    """Philox4x32-10 (Salmon et al. 2011), counter (ctr,0,0,0), key = 64-bit seed; top 24 bits of word 0 -> [0,1)."""
    m32 = 0xffffffff
    c = [counter & m32, 0, 0, 0]
    k0, k1 = seed & m32, (seed >> 32) & m32
    for _ in range(10):
        p0 = 0xD2511F53 * c[0]
        p1 = 0xCD9E8D57 * c[2]
        c = [((p1 >> 32) ^ c[1] ^ k0) & m32, p1 & m32, ((p0 >> 32) ^ c[3] ^ k1) & m32, p0 & m32]
        k0 = (k0 + 0x9E3779B9) & m32
        k1 = (k1 + 0xBB67AE85) & m32
    return float(np.float32(c[0] >> 8) * np.float32(1.0 / 16777216.0))


# ------------------------------------------------------------------------------------------------
# FP8 (e4m3fn) block-scaled weights (lmdeploy/turbomind/weight_format.py:349-393; weight-only path of
# models/linear_weight.cc:138-150: 128x128 block scales expanded to per-column K-group scales in the activation type;
# conversion kernels/attention/quantization.h:820-846 = exact e4m3 -> f16)
# ------------------------------------------------------------------------------------------------
def fp8_e4m3_to_f32(b: np.ndarray) -> np.ndarray:
    """OCP e4m3fn: 1 sign, 4 exponent (bias 7), 3 mantissa bits; no infinities, S.1111.111 = NaN, max 448."""
    b = np.asarray(b, np.uint8).astype(np.int32)
    sign = np.where(b & 0x80, -1.0, 1.0)
    e = (b >> 3) & 0xF
    m = b & 7
    v = np.where(e == 0, m / 8.0 * 2.0**-6, (1.0 + m / 8.0) * np.exp2(e.astype(np.float64) - 7))
    v = np.where((e == 15) & (m == 7), np.nan, v)
    return (sign * v).astype(np.float32)


def fp8_e4m3_from_f32(x: np.ndarray) -> np.ndarray:
    """round-to-nearest-even onto the e4m3fn grid, saturating at +-448 (test data generator)"""
    grid = fp8_e4m3_to_f32(np.arange(0, 0x7F, dtype=np.uint8)).astype(np.float64)      # 0 .. 448, ascending
    x = np.asarray(x, np.float64)
    a = np.clip(np.abs(x), 0, 448.0)
    hi = np.clip(np.searchsorted(grid, a, side='left'), 1, len(grid) - 1)
    lo = hi - 1
    dl, dh = a - grid[lo], grid[hi] - a
    pick_hi = (dh < dl) | ((dh == dl) & (hi % 2 == 0))        # ties to the even code
    code = np.where(pick_hi, hi, lo).astype(np.uint8)
    return np.where(np.signbit(x), code | 0x80, code).astype(np.uint8)


def fp8_expand_block_scales(block_scales: np.ndarray, K: int, N: int, gated: bool = False) -> np.ndarray:
    """[K/128][ceil(N/128)] fp32 -> per-column group scales fp16 [K/128][N] (BlockscaleToGroupscale,
    models/linear_weight.cc:138-150).  gated: the fused w1w3 linear with (gate_j, up_j)-interleaved columns whose scale
    row is [w1's inter/128 blocks | w3's inter/128 blocks] (w1 / w3 are block-quantised separately in a checkpoint;
    the interleave of builders/ffn.py:31-34 acts on the expanded per-column scales)."""
    bs = np.asarray(block_scales, np.float32)
    if not gated:
        return np.repeat(bs, 128, axis=1)[:, :N].astype(np.float16)
    assert N % 256 == 0 and bs.shape[1] == N // 128
    half = bs.shape[1] // 2
    out = np.empty((bs.shape[0], N), np.float32)
    out[:, 0::2] = np.repeat(bs[:, :half], 128, axis=1)
    out[:, 1::2] = np.repeat(bs[:, half:], 128, axis=1)
    return out.astype(np.float16)


def fp8_dequant(wq: np.ndarray, block_scales: np.ndarray, gated: bool = False) -> np.ndarray:
    """w[k, n] = h( f16(e4m3[k, n]) * s[k/128, n] ): exact conversion, one fp16 multiply"""
    K, N = wq.shape
    s = fp8_expand_block_scales(block_scales, K, N, gated)
    v = fp8_e4m3_to_f32(wq).astype(np.float16)              # exact
    return hmul(v, np.repeat(s, 128, axis=0)[:K])


def fp8_quantize_blockwise(w: np.ndarray):
    """fp16/fp32 [K, N] -> (e4m3 codes, fp32 128x128 block scales) with scale = amax / 448 (test data)"""
    w = np.asarray(w, np.float32)
    K, N = w.shape
    kb, nb = (K + 127) // 128, (N + 127) // 128
    scales = np.zeros((kb, nb), np.float32)
    q = np.zeros((K, N), np.uint8)
    for i in range(kb):
        for j in range(nb):
            blk = w[i * 128:(i + 1) * 128, j * 128:(j + 1) * 128]
            sc = max(float(np.abs(blk).max()) / 448.0, 1e-8)
            scales[i, j] = sc
            q[i * 128:(i + 1) * 128, j * 128:(j + 1) * 128] = fp8_e4m3_from_f32(blk / sc)
    return q, scales


# ------------------------------------------------------------------------------------------------
# Fused all-reduce + residual + RMSNorm of the native communicator (comm/cuda_ipc/fused_allreduce.cu:406-500): the ranks'
# partial tiles are summed in fp32 in RANK ORDER (identical bits on every rank), rounded to fp16 once, then A7's
# residual + RMSNorm.
# ------------------------------------------------------------------------------------------------
def p2p_allreduce_norm(partials, resid, weight, eps):
    """partials: list of fp16 [M, H] (one per rank).  Returns (new residual fp16, normed fp16)."""
    acc = np.zeros(partials[0].shape, np.float32)
    for part in partials:
        acc = acc + np.asarray(part, f16).astype(np.float32)
    return residual_rmsnorm(resid, acc.astype(f16), weight, eps)


# ------------------------------------------------------------------------------------------------
# FP8 x FP8 linear: what the reference runs for e4m3 weights on fp8 matrix cores (SM90+) and what the MI355X path runs on
# v_mfma_f32_32x32x16_fp8_fp8.  The activations are quantised on the fly, per row and per group of 128 input channels
# (QuantizeSymm, src/turbomind/kernels/quantization.cu:28-63, called from LlamaLinear::GetOperandA,
# models/llama/LlamaLinear.cu:67-93):  absmax = max(|x|) over the group, clamped to 1e-8;  scale = absmax / 448;
# q = e4m3_rne_sat(f32(x) * (448 / absmax)).  The GEMM contracts codes with codes in fp32, one k-group (128) at a time, and
# adds partial * (scale_x[m, g] * scale_w[g, n]) -- the weight scale is the fp32 128x128 block scale, NOT rounded to the
# activation type as in the weight-only path above.
# Unpinned to the last bit: the CUDA file is built with --use_fast_math (448 / absmax may differ by an ulp there).
# ------------------------------------------------------------------------------------------------
def fp8_quant_rows(x: np.ndarray, group: int = 128):
    """x fp16 [M, K] -> (codes uint8 [M, K], scales fp32 [K/group, M]) (column-major scales, quantization.cu:50-52)"""
    x = np.asarray(x, np.float16).astype(np.float32)
    M, K = x.shape
    assert K % group == 0
    g = x.reshape(M, K // group, group)
    absmax = np.maximum(np.abs(g).max(axis=2), np.float32(1e-8)).astype(np.float32)
    scale = (absmax / np.float32(448.0)).astype(np.float32)
    inv = (np.float32(448.0) / absmax).astype(np.float32)
    q = fp8_e4m3_from_f32((g * inv[:, :, None]).astype(np.float32))
    return q.reshape(M, K), np.ascontiguousarray(scale.T)


def fp8_block_scales_f32(block_scales: np.ndarray, K: int, N: int, gated: bool = False) -> np.ndarray:
    """[K/128][ceil(N/128)] fp32 -> per-column fp32 scales [K/128][N] (the gated interleave of fp8_expand_block_scales, no
    rounding)"""
    bs = np.asarray(block_scales, np.float32)
    if not gated:
        return np.repeat(bs, 128, axis=1)[:, :N]
    assert N % 256 == 0 and bs.shape[1] == N // 128
    half = bs.shape[1] // 2
    out = np.empty((bs.shape[0], N), np.float32)
    out[:, 0::2] = np.repeat(bs[:, :half], 128, axis=1)
    out[:, 1::2] = np.repeat(bs[:, half:], 128, axis=1)
    return out


def fp8_act_linear_acc(x: np.ndarray, wq: np.ndarray, block_scales: np.ndarray, gated_scales: bool = False) -> np.ndarray:
    """fp32 accumulators of the fp8 x fp8 linear: sum_g (xq_g . wq_g) * sx[g, m] * sw[g, n]"""
    K, N = wq.shape
    xq, sx = fp8_quant_rows(x)
    a = fp8_e4m3_to_f32(xq)
    w = fp8_e4m3_to_f32(wq)
    sw = fp8_block_scales_f32(block_scales, K, N, gated_scales)
    acc = np.zeros((a.shape[0], N), np.float32)
    for g in range(K // 128):
        part = a[:, g * 128:(g + 1) * 128] @ w[g * 128:(g + 1) * 128]
        acc += part * (sx[g][:, None] * sw[g][None, :])
    return acc


def fp8_act_linear(x, wq, block_scales, gated: bool = False) -> np.ndarray:
    """fp16 output (gated: the fused w1w3 linear with the gated-SiLU epilogue on the fp32 accumulators)"""
    acc = fp8_act_linear_acc(x, wq, block_scales, gated)
    return gated_silu_epilogue(acc) if gated else acc.astype(f16)


def moe_ffn_fp8(x: np.ndarray, gate: np.ndarray, experts_q: list, top_k: int, norm_topk: bool = True, routed_scale: float = 1.0):
    """moe_ffn with the fp8 x fp8 expert linears: experts_q[e] = ((w13 codes, w13 block scales), (w2 codes, w2 block scales));
    the activations of BOTH expert projections are quantised per row and 128-channel group (x once per token, the
    gated-SiLU output once per (token, expert) row)."""
    _, ids, w = moe_gate(x, gate, top_k, norm_topk, routed_scale)
    T, H = x.shape
    out = np.zeros((T, H), np.float32)
    for t in range(T):
        for j in range(top_k):
            (q13, s13), (q2, s2) = experts_q[ids[t, j]]
            act = fp8_act_linear(x[t:t + 1], q13, s13, gated=True)
            y = fp8_act_linear(act, q2, s2)
            out[t] += w[t, j] * y[0].astype(np.float32)
    return out.astype(np.float16), ids, w


# ------------------------------------------------------------------------------------------------
# Mixture of experts (models/llama/moe_ffn_layer.cc:133-325; kernels/gemm/moe_utils_v2.cu:355-690)
# ------------------------------------------------------------------------------------------------
def moe_gate(x: np.ndarray, gate: np.ndarray, top_k: int, norm_topk: bool = True, routed_scale: float = 1.0):
    """x fp16 [T,H], gate fp16 [H,E] -> (logits f32 [T,E], expert ids [T,k], weights f32 [T,k]).
    Top-k on the logits (ties: lower expert id); norm_topk: softmax over the selected experts only
    (moe_utils_v2.cu:451-481), else over all experts; times routed_scale (:534)."""
    logits = (x.astype(np.float32) @ gate.astype(np.float32)).astype(np.float32)
    T, E = logits.shape
    ids = np.zeros((T, top_k), np.int32)
    w = np.zeros((T, top_k), np.float32)
    for t in range(T):
        order = np.lexsort((np.arange(E), -logits[t]))[:top_k]
        mx = logits[t].max()
        ex = np.exp((logits[t] - mx).astype(np.float32)).astype(np.float32)
        denom = ex[order].sum(dtype=np.float32) if norm_topk else ex.sum(dtype=np.float32)
        ids[t] = order
        w[t] = ex[order] / denom * np.float32(routed_scale)
    return logits, ids, w


def moe_ffn(x: np.ndarray, gate: np.ndarray, experts: list, top_k: int, norm_topk: bool = True, routed_scale: float = 1.0):
    """experts[e] = (w13 fp16 [H, 2I] with (gate_j, up_j) interleaved, w2 fp16 [I, H]) as DEQUANTISED weights.
    out[t] = fp16( sum_j w_j * fp16( W2_e . fp16(silu(g) * u) ) ): the expert FFN is the dense FFN's arithmetic
    (fp32 accumulation, gated-SiLU epilogue, fp16 outputs), the combine accumulates in fp32 (invokeMoeCombine)."""
    _, ids, w = moe_gate(x, gate, top_k, norm_topk, routed_scale)
    T, H = x.shape
    out = np.zeros((T, H), np.float32)
    for t in range(T):
        for j in range(top_k):
            w13, w2 = experts[ids[t, j]]
            act = gated_silu_epilogue(gemm_f16_f32acc(x[t:t + 1], w13))
            y = gemm_f16_f32acc(act, w2).astype(np.float16)
            out[t] += w[t, j] * y[0].astype(np.float32)
    return out.astype(np.float16), ids, w
