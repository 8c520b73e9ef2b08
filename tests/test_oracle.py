"""CPU tests: the oracle against (a) golden vectors produced by the reference's own Python
(tests/golden/reference_python.npz, generator committed next to it), (b) the reference's known-answer tests,
(c) domain properties (round trips, split-K invariance, block-table permutation invariance)."""
import math
import os

import numpy as np
import pytest
from hypothesis import given, settings
from hypothesis import strategies as st

from oracle import tm_oracle as o

f16 = np.float16
GOLD = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'reference_python.npz'))


# ---- (a) golden vectors from the reference's Python ---------------------------------------------------------------
def test_awq_unpack_matches_reference_weight_format():
    assert np.array_equal(o.unpack_awq_gemm(GOLD['awq_qweight']), GOLD['awq_unpacked'])
    assert np.array_equal(o.unpack_awq_gemm(GOLD['awq_qzeros']), GOLD['awq_zeros_unpacked'])
    assert np.array_equal(o.pack_u4_row(GOLD['awq_unpacked']), GOLD['awq_packed_row'])
    assert np.array_equal(o.unpack_u4_row(GOLD['awq_packed_row']), GOLD['awq_unpacked'])
    assert np.array_equal(o.pack_awq_gemm(GOLD['awq_unpacked']), GOLD['awq_qweight'])


def test_awq_dequant_matches_reference_formula():
    """(q - z) * s: AWQFormat.dequant (weight_format.py:236-247) and the PyTorchEngine default backend
    (awq_modules.py:37-46) agree with each other and with the oracle bit for bit."""
    q, z, s = GOLD['awq_unpacked'], GOLD['awq_zeros_unpacked'].astype(f16), GOLD['awq_scales']
    w = o.awq_dequant_ref(q, s, z)
    assert np.array_equal(w.view(np.uint16), GOLD['awq_dequant'].view(np.uint16))
    assert np.array_equal(w.view(np.uint16), GOLD['awq_dequantize_gemm'].view(np.uint16))
    # TurboMind's kernel form fma(q, s, h(-z*s)) differs from (q-z)*s only by the rounding of z*s (<= half an ulp
    # of |z*s|) plus one final rounding: an ABSOLUTE bound (near q == z the two forms cancel differently)
    wt = o.w4a16_dequant(q, s, z).astype(np.float32)
    zs = np.repeat(np.abs(z.astype(np.float32) * s.astype(np.float32)), 128, axis=0)
    assert np.all(np.abs(wt - w.astype(np.float32)) <= 2.0**-10 * np.maximum(zs, np.abs(wt)) + 1e-8)
    y = o.gemm_f16_f32acc(GOLD['awq_x'], w)
    assert np.allclose(y, GOLD['awq_y_fp32'], rtol=1e-5, atol=1e-5)


def test_rmsnorm_matches_reference_default_backend():
    y = o.rmsnorm_torch_default(GOLD['norm_x'], GOLD['norm_w'], 1e-5)
    assert np.array_equal(y.view(np.uint16), GOLD['norm_y'].view(np.uint16))
    y2, r2 = o.rmsnorm_torch_default(GOLD['norm_x'], GOLD['norm_w'], 1e-5, GOLD['norm_res'])
    assert np.array_equal(r2.view(np.uint16), GOLD['norm_res_out'].view(np.uint16))
    assert np.array_equal(y2.view(np.uint16), GOLD['norm_y_res'].view(np.uint16))
    # the TurboMind fp16-rounding form stays within one fp16 ulp of it
    yt = o.rmsnorm(GOLD['norm_x'], GOLD['norm_w'], 1e-5)
    d = np.abs(yt.view(np.int16).astype(np.int32) - GOLD['norm_y'].view(np.int16).astype(np.int32))
    assert d.max() <= 1


def test_silu_matches_reference_default_backend():
    x = GOLD['silu_in']
    half = x.shape[1] // 2
    ours = o.silu_and_mul_unfused(x)
    # reference: silu in fp16 (torch) * up; ours: fp32 silu.  <= 2 fp16 ulp
    d = np.abs(ours.astype(np.float32) - GOLD['silu_out'].astype(np.float32))
    assert np.all(d <= 2.0**-9 * np.abs(GOLD['silu_out'].astype(np.float32)) + 1e-4)
    assert half == GOLD['silu_out'].shape[1]


# ---- (b) known-answer tests of the reference ---------------------------------------------------------------------
@pytest.mark.parametrize('bits,hi', [(8, 256), (4, 16)])
def test_kv_round_trip_known_answer(bits, hi):
    """kernels/attention/test_quant.cu:32-70: integer-valued data survives T -> u8/u4 -> T exactly with
    (scale, zero) = (1, 0)."""
    rng = np.random.default_rng(bits)
    x = rng.integers(0, hi, (4096, 128)).astype(f16)
    one, zero = np.ones(4096, f16), np.zeros(4096, f16)
    q = o.kv_quantize_values(x, one, zero, bits)
    data = q if bits == 8 else o.kv_pack_int4(q)
    back = data if bits == 8 else o.kv_unpack_int4(data)
    assert np.array_equal(o.kv_dequant_flatten(back, one, zero), x)
    assert np.array_equal(o.kv_dequant_decode(back, one, zero), x)
    # the biased u4 variant of the test: zero = -64 applied to codes offset by 64 is the same identity
    if bits == 4:
        assert np.array_equal(o.kv_dequant_flatten(back.astype(np.int32) + 64, one, np.full(4096, -64, f16)), x)


def test_kv_int4_nibble_order():
    """quantization.h:459-471: word = nibbles [q0,q2,q4,q6,q1,q3,q5,q7], LSB first."""
    q = np.arange(8, dtype=np.uint8)[None]
    b = o.kv_pack_int4(q)
    word = int(b.view(np.uint32)[0, 0])
    assert [(word >> (4 * i)) & 15 for i in range(8)] == [0, 2, 4, 6, 1, 3, 5, 7]


def test_kv_quant_rule_and_degenerate_row():
    x = np.zeros((2, 128), f16)
    x[0] = f16(0.75)                              # constant row: scale 0 -> inv inf -> NaN -> q = 0
    x[1, :] = np.linspace(-3, 5, 128).astype(f16)
    data, prm = o.kv_quantize(x, 8)
    assert prm[0, 0] == 0 and prm[0, 1] == f16(0.75) and not data[0].any()
    mn, mx = x[1].min(), x[1].max()
    assert prm[1, 1] == mn
    assert prm[1, 0] == f16((np.float32(mx) - np.float32(mn)) * (np.float32(1) / np.float32(255)))
    assert data[1].min() == 0 and data[1].max() in (254, 255)
    rec = o.kv_dequant_decode(data[1:2], prm[1:2, 0], prm[1:2, 1])
    assert np.abs(rec.astype(np.float32) - x[1].astype(np.float32)).max() <= float(prm[1, 0]) * 0.51 + 4e-3


def test_block_layout_sizes():
    """Llama-3-8B int8: 135168 B / layer / block, 4325376 B per block; int4 69632; fp16 262144."""
    assert o.BlockLayout(32, 8, 128, 64, 8).layer_size == 135168
    assert o.BlockLayout(32, 8, 128, 64, 8).block_size == 4325376
    assert o.BlockLayout(32, 8, 128, 64, 4).layer_size == 69632
    assert o.BlockLayout(32, 8, 128, 64, 16).layer_size == 262144
    L = o.BlockLayout(2, 2, 128, 64, 8)
    assert L.v_data(1, 3) == L.k_data(1, 3) + 64 * 128 and L.k_param(0, 0) == 2 * 2 * 64 * 128


# ---- (c) properties ------------------------------------------------------------------------------------------------
@settings(max_examples=25, deadline=None)
@given(st.integers(0, 2**32 - 1), st.sampled_from([8, 4]))
def test_kv_quantize_property(seed, bits):
    rng = np.random.default_rng(seed)
    x = (rng.standard_normal((3, 128)) * rng.uniform(0.01, 30)).astype(f16)
    data, prm = o.kv_quantize(x, bits)
    codes = data if bits == 8 else o.kv_unpack_int4(data)
    assert codes.max() <= (1 << bits) - 1
    rec = o.kv_dequant_decode(codes, prm[:, 0], prm[:, 1]).astype(np.float32)
    step = prm[:, 0].astype(np.float32)[:, None]
    # fp16 arithmetic in the quantiser costs a little more than half a step
    assert np.all(np.abs(rec - x.astype(np.float32)) <= 0.75 * step + 2.0**-8 * np.abs(x.astype(np.float32)) + 1e-3)


@pytest.mark.parametrize('splits', [1, 2, 3, 7])
def test_split_k_invariance_and_unfused_reference(splits):
    rng = np.random.default_rng(splits)
    Hq, Hkv, ctx = 8, 2, 333
    q = rng.standard_normal((Hq, 128)).astype(f16)
    K = rng.standard_normal((Hkv, ctx, 128)).astype(f16)
    V = rng.standard_normal((Hkv, ctx, 128)).astype(f16)
    K[0, 100] *= f16(8)            # a spike forces the online-softmax rescale path
    ref = o.attention_reference_unfused(q, K, V)
    out = o.decode_attention(q, K, V, None, splits)
    assert np.abs(out - ref).max() < 4e-3
    assert np.abs(out.astype(np.float32) - o.decode_attention(q, K, V, None, 1).astype(np.float32)).max() < 2e-3


def test_block_table_permutation_invariance():
    """test_attention.cu:70-141."""
    rng = np.random.default_rng(0)
    L = o.BlockLayout(1, 1, 128, 64, 8)
    k = rng.standard_normal((150, 1, 128)).astype(f16)
    v = rng.standard_normal((150, 1, 128)).astype(f16)
    outs = []
    for table in ([0, 1, 2], [5, 2, 7]):
        c = o.PagedKVCache(L, 8)
        o.process_kv(c, table, 0, k, v, None, None, 0)
        outs.append(c.load_dequant(table, 0, 0, 0, 150, 'decode'))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])


def test_rope_properties():
    p = o.RopeParam(128, 500000.0, 'llama3', 8.0, 1.0, 4.0, 8192)
    inv = o.rope_inv_freq(p)
    base = o.rope_inv_freq(o.RopeParam(128, 500000.0))
    assert np.allclose(inv[:8], base[:8])                    # high frequencies untouched
    assert np.allclose(inv[-4:], base[-4:] / 8.0, rtol=1e-6)  # low frequencies divided by the factor
    cos, sin = o.rope_cos_sin(p, np.array([0, 1, 77]))
    x = np.random.default_rng(1).standard_normal((3, 2, 128)).astype(f16)
    y = o.rope_apply(x, cos, sin)
    assert np.array_equal(y[0], x[0])                       # position 0 is the identity
    n0 = np.linalg.norm(x.astype(np.float32), axis=-1)
    n1 = np.linalg.norm(y.astype(np.float32), axis=-1)
    assert np.allclose(n0, n1, rtol=2e-3)                   # rotation preserves norms
    w = np.arange(2 * 8 * 128).reshape(2, 8 * 128)
    pw = o.permute_qk_for_interleaved_rope(w, 8, 128)
    assert pw[0, 0] == w[0, 0] and pw[0, 1] == w[0, 64] and pw[0, 2] == w[0, 1]


def test_quantize_groupwise_and_gated_silu():
    rng = np.random.default_rng(5)
    w = (rng.standard_normal((256, 64)) * 0.02).astype(f16)
    q, s, z, d = o.quantize_groupwise_u4(w, 128)
    assert q.max() <= 15 and z.min() >= 0 and z.max() <= 15
    assert np.abs(d.astype(np.float32) - w.astype(np.float32)).max() <= s.astype(np.float32).max() * 0.51 + 1e-4
    wd = o.w4a16_dequant(q, s, z)
    assert np.abs(wd.astype(np.float32) - d.astype(np.float32)).max() <= 2.0**-9 * np.abs(d.astype(np.float32)).max() + 1e-6
    acc = rng.standard_normal((3, 8)).astype(np.float32) * 4
    out = o.gated_silu_epilogue(acc)
    g, u = acc[:, 0::2], acc[:, 1::2]
    assert np.allclose(out.astype(np.float32), g / (1 + np.exp(-g)) * u, rtol=2e-3, atol=1e-3)
    assert np.array_equal(o.interleave_w1w3(np.ones((2, 3)), np.zeros((2, 3)))[0], [1, 0, 1, 0, 1, 0])


def test_oracle_model_is_deterministic_and_block_table_free():
    cfg = o.ModelConfig(hidden=128, layers=1, q_heads=2, kv_heads=1, head_dim=128, inter=256, vocab=300, kv_bits=8)
    w = o.make_synthetic_weights(cfg, seed=0)
    prompts = [np.arange(9) % 300, (np.arange(70) * 7) % 300]
    runs = []
    for _ in range(2):
        m = o.OracleModel(cfg, w, batch=2, max_ctx=128)
        ids, _ = m.forward(prompts)
        ids2, lg = m.forward([[int(ids[0])], [int(ids[1])]], decode_splits=2)
        runs.append((ids, ids2, lg))
    assert np.array_equal(runs[0][0], runs[1][0]) and np.array_equal(runs[0][2], runs[1][2])
    assert math.isfinite(float(np.abs(runs[0][2].astype(np.float32)).max()))


def test_philox_known_answer():
    """Philox4x32-10 known-answer vectors of the Random123 distribution (kat_vectors): counter / key all zero, all
    ones, and the digits of pi -> first output word."""
    def word0(ctr4, key2):
        m32 = 0xffffffff
        c = list(ctr4)
        k0, k1 = key2
        for _ in range(10):
            p0 = 0xD2511F53 * c[0]
            p1 = 0xCD9E8D57 * c[2]
            c = [((p1 >> 32) ^ c[1] ^ k0) & m32, p1 & m32, ((p0 >> 32) ^ c[3] ^ k1) & m32, p0 & m32]
            k0 = (k0 + 0x9E3779B9) & m32
            k1 = (k1 + 0xBB67AE85) & m32
        return c
    assert word0([0, 0, 0, 0], (0, 0)) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert word0([0xffffffff] * 4, (0xffffffff, 0xffffffff)) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert word0([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], (0xa4093822, 0x299f31d0)) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
    # the oracle's (and the library's) uniform = top 24 bits of word 0
    assert o.philox_uniform(0, 0) == float(np.float32(0x6627e8d5 >> 8) / np.float32(16777216.0))
    from lmdeploy_amd import _ffi
    lib = _ffi.load()
    for seed, ctr in ((0, 0), (1234567890123, 77), (2**64 - 1, 2**32 - 1)):
        assert lib.tm_philox_uniform(seed, ctr) == o.philox_uniform(seed, ctr)


def test_sampling_filter_semantics():
    logits = np.array([1.0, 3.0, 3.0, -2.0, 0.5, 3.0, 2.0], np.float16)
    ids, p = o.sample_filter(logits)                                   # everything, descending, ties by id
    assert ids.tolist() == [1, 2, 5, 6, 0, 4, 3] and abs(p.sum() - 1) < 1e-12
    ids, p = o.sample_filter(logits, top_k=2)
    assert ids.tolist() == [1, 2] and np.allclose(p, 0.5)
    ids, p = o.sample_filter(logits, top_k=1, temperature=0.3)
    assert ids.tolist() == [1] and p[0] == 1.0
    full = o.sample_filter(logits)[1]
    ids, p = o.sample_filter(logits, top_p=float(full[:2].sum()))      # cumulative must EXCEED top_p
    assert ids.tolist() == [1, 2, 5]
    ids, p = o.sample_filter(logits, min_p=0.2)                        # e^-1 = 0.37 stays, e^-2 = 0.135 goes
    assert ids.tolist() == [1, 2, 5, 6]
    cold = o.sample_filter(logits, temperature=0.25)[1]
    assert cold[0] > full[0]                                           # lower temperature sharpens
    assert o.sample_draw(np.array([7, 8, 9]), np.array([0.2, 0.3, 0.5]), 0.2) == 8     # prefix must exceed u
    assert o.sample_draw(np.array([7, 8, 9]), np.array([0.2, 0.3, 0.5]), 0.999999) == 9
    assert o.sample_draw(np.array([7, 8, 9]), np.array([0.2, 0.3, 0.5]), 0.0) == 7


def test_fp8_e4m3_matches_torch():
    """e4m3fn decode / encode pinned on torch.float8_e4m3fn (an independent implementation): all 256 codes, and
    round-to-nearest-even encoding of random values."""
    import torch
    codes = np.arange(256, dtype=np.uint8)
    ref = torch.from_numpy(codes).view(torch.float8_e4m3fn).float().numpy()
    got = o.fp8_e4m3_to_f32(codes)
    assert np.array_equal(np.isnan(ref), np.isnan(got)) and np.array_equal(ref[~np.isnan(ref)], got[~np.isnan(got)])
    x = (np.random.default_rng(0).standard_normal(20000) * 60).astype(np.float32)
    enc_ref = torch.from_numpy(x).to(torch.float8_e4m3fn).view(torch.uint8).numpy()
    inside = np.abs(x) <= 448
    assert np.array_equal(o.fp8_e4m3_from_f32(x)[inside], enc_ref[inside])
    w = (np.random.default_rng(1).standard_normal((256, 160)) * 0.05).astype(np.float16)
    q, sc = o.fp8_quantize_blockwise(w)
    d = o.fp8_dequant(q, sc).astype(np.float32)
    assert np.abs(d - w.astype(np.float32)).max() <= np.abs(w).max() * 2.0**-3      # 3 mantissa bits


def test_logits_process_matches_reference_python():
    """Repetition penalty / bad ids against the committed outputs of the reference's own Python
    (lmdeploy/pytorch/engine/logits_process.py:24-65, run by tests/golden/make_golden.py): the oracle equals the
    reference's fp32 result rounded to fp16 once, bit for bit; banned ids are exactly the reference's."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'reference_logits_process.npz'))
    for b in range(g['logits'].shape[0]):
        got = o.logits_process(g['logits'][b], g['ids'][b], float(g['penalty'][b]))
        assert np.array_equal(got.view(np.uint16), g['penalised_fp32'][b].astype(np.float16).view(np.uint16)), b
        bad = [int(t) for t in g['bad'][b] if t >= 0]
        got = o.logits_process(g['logits'][b], g['ids'][b], float(g['penalty'][b]), bad_ids=bad)
        assert np.array_equal(got == np.float16(-65504), g['banned_mask'][b]), b


def test_logits_process_semantics():
    """The TurboMind kernels' details (sampling_penalty_kernels.cu:137-215, ban_bad_words.cu:51-95): each seen id is
    penalised once; ids <= 0 are never banned; end ids are banned while k_len + 1 < min_len; a vocabulary shard
    (vocab_offset) gives the corresponding slice of the full result."""
    l = np.array([1.0, -2.0, 3.0, 0.5, -1.0, 4.0, 0.25, -0.5], np.float16)
    got = o.logits_process(l, [1, 2, 2, 2, 7], 2.0)
    assert got.tolist() == [1.0, -4.0, 1.5, 0.5, -1.0, 4.0, 0.25, -1.0]
    assert np.array_equal(o.logits_process(l, [1, 2], 1.0), l)
    got = o.logits_process(l, [], 1.0, bad_ids=[0, 3, -1, 100], end_ids=[0, 5], k_len=4, min_len=6)
    assert got[0] == 1.0 and got[3] == -65504 and got[5] == -65504 and got[4] == -1.0
    assert o.logits_process(l, [], 1.0, end_ids=[5], k_len=5, min_len=6)[5] == 4.0        # 5 + 1 < 6 is false
    rng = np.random.default_rng(0)
    full = (rng.standard_normal(64) * 3).astype(np.float16)
    seen = rng.integers(0, 64, 30)
    whole = o.logits_process(full, seen, 1.7, bad_ids=[9, 40], end_ids=[33], k_len=1, min_len=9)
    parts = [o.logits_process(full[c:c + 32], seen, 1.7, bad_ids=[9, 40], end_ids=[33], k_len=1, min_len=9, vocab_offset=c)
             for c in (0, 32)]
    assert np.array_equal(np.concatenate(parts).view(np.uint16), whole.view(np.uint16))


def _golden2():
    import os
    return np.load(os.path.join(os.path.dirname(__file__), 'golden', 'reference_python2.npz'))


def test_rope_matches_reference_python():
    """Llama-3 RoPE against the reference's Python (backends/default/rotary_embedding.py:51-78,141-172 and
    apply_rotary_emb.py:48-77, outputs committed by make_golden.py).
      * inv_freq: TurboMind's exp2f / smooth-clamp form (attention_weight.cc:37-93) equals the HF piecewise form to fp32
        rounding (rel 4e-6); the unscaled default base likewise;
      * cos / sin tables (fp16): within 1e-3 (the angle is an fp32 product of up to 8191 rad);
      * application: the interleaved-pair rotation on loader-permuted channels equals the reference's rotate-half
        rotation on HF-ordered channels BIT FOR BIT given the same cos / sin (same fp16 op sequence)."""
    g = _golden2()
    p = o.RopeParam(128, 500000.0, 'llama3', 8.0, 1.0, 4.0, 8192)
    inv = o.rope_inv_freq(p)
    assert np.allclose(inv, g['rope_llama3_inv_freq'], rtol=4e-6, atol=0)
    assert np.allclose(o.rope_inv_freq(o.RopeParam(128, 10000.0, 'default', 1.0, 1.0, 4.0, 8192)), g['rope_default_inv_freq'],
                       rtol=4e-6, atol=0)
    cos, sin = o.rope_cos_sin(p, g['rope_pos'])
    assert np.abs(cos.astype(np.float32) - g['rope_cos'][:, :64].astype(np.float32)).max() <= 1e-3
    assert np.abs(sin.astype(np.float32) - g['rope_sin'][:, :64].astype(np.float32)).max() <= 1e-3
    assert np.array_equal(g['rope_cos'][:, :64], g['rope_cos'][:, 64:])          # [freqs | freqs]
    c, s_ = g['rope_cos'][:, :64], g['rope_sin'][:, :64]
    for x, want, heads in ((g['rope_q'], g['rope_q_out'], 4), (g['rope_k'], g['rope_k_out'], 2)):
        T = x.shape[0]
        inter = o.permute_qk_for_interleaved_rope(x.reshape(T, heads * 128), heads, 128).reshape(T, heads, 128)
        got = o.rope_apply(inter, c, s_)
        want_inter = o.permute_qk_for_interleaved_rope(want.reshape(T, heads * 128), heads, 128).reshape(T, heads, 128)
        assert np.array_equal(got.view(np.uint16), want_inter.view(np.uint16))


def test_moe_gate_matches_reference_python():
    """Router against backends/default/moe.py:7-33 (softmax over all experts -> top-k) + Mixtral's renormalisation:
    same experts in the same order; norm_topk weights = renormalised top-k softmax, plain weights = the softmax."""
    g = _golden2()
    logits = g['moe_logits']
    x = logits.astype(np.float16)
    assert np.array_equal(x.astype(np.float32), logits)
    eye = np.eye(8, dtype=np.float16)
    lg, ids, w = o.moe_gate(x, eye, 2, norm_topk=True)
    assert np.array_equal(lg, logits) and np.array_equal(ids, g['moe_topk_ids'])
    assert np.allclose(w, g['moe_topk_renorm'], rtol=2e-6, atol=0)
    _, ids2, w2 = o.moe_gate(x, eye, 2, norm_topk=False)
    assert np.array_equal(ids2, g['moe_topk_ids']) and np.allclose(w2, g['moe_topk_softmax'], rtol=2e-6, atol=0)


def test_fp8_dequant_matches_reference_weight_format():
    """FP8Format.dequant (lmdeploy/turbomind/weight_format.py:349-384): e4m3 decode x 128x128 block scale.  The
    reference multiplies in fp32 and rounds to fp16 once; the kernel's recipe (oracle.fp8_dequant) rounds the scale to
    fp16 first and multiplies in fp16 -- two roundings, so the results agree to 2^-10 relative (one fp16 ulp), and
    exactly when the block scale is a power of two."""
    g = _golden2()
    got = o.fp8_dequant(g['fp8_codes'], g['fp8_block_scales']).astype(np.float32)
    ref = g['fp8_dequant'].astype(np.float32)
    assert np.all(np.abs(got - ref) <= np.abs(ref) * 2.0**-10 + 1e-7)
    pow2 = np.array([[2.0**-6, 2.0**-7, 2.0**-5], [2.0**-8, 2.0**-6, 2.0**-4]], np.float32)
    exact = (o.fp8_e4m3_to_f32(g['fp8_codes']) * np.repeat(np.repeat(pow2, 128, 0), 128, 1)).astype(np.float16)
    assert np.array_equal(o.fp8_dequant(g['fp8_codes'], pow2).view(np.uint16), exact.view(np.uint16))


def test_sampling_filters_match_reference_python():
    """top-k -> top-p -> min-p on descending scores against pytorch/engine/logits_process.py:68-96 (committed outputs):
    the same surviving candidates in the same order and the same renormalised probabilities."""
    g = _golden2()
    for b in range(g['samp_logits'].shape[0]):
        ids, p = o.sample_filter(g['samp_logits'][b], float(g['samp_temperature'][b]), int(g['samp_topk'][b]),
                                 float(g['samp_topp'][b]), float(g['samp_minp'][b]))
        kept = g['samp_kept'][b]
        assert ids.tolist() == g['samp_order'][b][kept].tolist(), b
        assert np.allclose(p, g['samp_probs'][b][kept], rtol=1e-4, atol=1e-7), b


def test_cpu_baseline_port_matches_reference_golden_vectors():
    """bench.py's `cpu_baseline` leg is a torch-CPU port of the reference PyTorchEngine's default-backend ops
    (oracle/cpu_baseline.py); its building blocks must reproduce the vectors generated from the reference itself."""
    import torch
    from oracle import cpu_baseline as cb
    q = torch.from_numpy(GOLD['awq_unpacked'].copy())
    s = torch.from_numpy(GOLD['awq_scales'].copy())
    z = torch.from_numpy(GOLD['awq_zeros_unpacked'].astype(f16))
    x = torch.from_numpy(GOLD['awq_x'].copy())
    y = cb._awq_linear(x, q, s, z, 128).numpy()
    ref = GOLD['awq_y_fp32'].astype(f16)                       # x @ ((q - z) * s) in fp32, cast once
    assert np.array_equal(y.view(np.uint16), ref.view(np.uint16))
    yn, _ = cb._rmsnorm(torch.from_numpy(GOLD['norm_x'].copy()), torch.from_numpy(GOLD['norm_w'].copy()), 1e-5)
    assert np.array_equal(yn.numpy().view(np.uint16), GOLD['norm_y'].view(np.uint16))
    yr, rr = cb._rmsnorm(torch.from_numpy(GOLD['norm_x'].copy()), torch.from_numpy(GOLD['norm_w'].copy()), 1e-5,
                         torch.from_numpy(GOLD['norm_res'].copy()))
    assert np.array_equal(rr.numpy().view(np.uint16), GOLD['norm_res_out'].view(np.uint16))
    assert np.array_equal(yr.numpy().view(np.uint16), GOLD['norm_y_res'].view(np.uint16))
    # and the whole baseline runs (tiny shapes, one pass): a number comes out, the sample is described
    tiny = dict(hidden=256, layers=2, q_heads=2, kv_heads=1, head_dim=128, inter=256, vocab=512)
    r = cb.run(tiny, batch=2, ctx=65, sample_layers=2, passes=2, threads=2)
    assert r['value'] > 0 and r['kind'] == 'port' and '2 of 2 decoder layers' in r['sample'] and 'warm-up' in r['sample']


def test_p2p_allreduce_norm_restatement():
    """f2: rank-ordered fp32 sum -> one fp16 rounding -> A7's residual + RMSNorm; tp = 1 degenerates to A7 itself."""
    rng = np.random.default_rng(4)
    parts = [(0.5 * rng.standard_normal((5, 256))).astype(np.float16) for _ in range(4)]
    resid = rng.standard_normal((5, 256)).astype(np.float16)
    w = (1 + 0.02 * rng.standard_normal(256)).astype(np.float16)
    r1, y1 = o.p2p_allreduce_norm(parts[:1], resid, w, 1e-5)
    r2, y2 = o.residual_rmsnorm(resid, parts[0], w, 1e-5)
    assert np.array_equal(r1.view(np.uint16), r2.view(np.uint16)) and np.array_equal(y1.view(np.uint16), y2.view(np.uint16))
    r4, y4 = o.p2p_allreduce_norm(parts, resid, w, 1e-5)
    h = ((parts[0].astype(np.float32) + parts[1].astype(np.float32)) + parts[2].astype(np.float32)) + parts[3].astype(np.float32)
    r5, y5 = o.residual_rmsnorm(resid, h.astype(np.float16), w, 1e-5)
    assert np.array_equal(r4.view(np.uint16), r5.view(np.uint16)) and np.array_equal(y4.view(np.uint16), y5.view(np.uint16))


def test_cpu_baseline_config0_runs_on_a_tiny_model():
    """bench.py --cpu-config0 (BASELINE.json configs[0]: InternLM2-1.8B fp16 on the host cores): the whole-model fp16 decode loop of
    oracle/cpu_baseline.py::run_config0 on a tiny model -- finite, positive rate, the sample string says what was timed."""
    from oracle import cpu_baseline
    tiny = dict(hidden=256, layers=2, q_heads=4, kv_heads=2, head_dim=64, inter=512, vocab=300)
    r = cpu_baseline.run_config0(tiny, batch=2, ctx=16, steps=2, threads=2, budget_s=5.0)
    assert r['value'] > 0 and np.isfinite(r['value']) and r['kind'] == 'port' and r['cores'] == 2
    assert 'whole model (2 layers' in r['sample'] and 'no extrapolation' in r['sample']


def test_logprobs_view_matches_reference_python():
    """Response.logprobs: oracle.logprobs_view and the product's lmdeploy_amd.pipeline.logprobs_of_token against the outputs of the
    reference's own _get_logprobs_impl (lmdeploy/turbomind/turbomind.py:472-503), executed on these inputs by tests/golden/make_golden.py."""
    import json
    import sys
    import lmdeploy_amd  # noqa: F401
    of_token = sys.modules['lmdeploy_amd.pipeline'].logprobs_of_token
    cases = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'reference_logprobs_view.json')))
    assert len(cases) >= 4
    for c in cases:
        for k, tok in enumerate(c['tokens']):
            t = k + c['offset']
            idx = np.asarray(c['idx'][t], np.int32)
            vals = np.asarray([float(v) for v in c['vals'][t]], np.float32)
            sel = float(vals[np.flatnonzero(idx == tok)[0]])
            want = {int(a): float(np.float32(b)) for a, b in c['expect'][k]}
            assert o.logprobs_view(idx, vals, sel, tok, c['topn']) == want
            assert of_token(vals, idx, int(c['nums'][t]), sel, tok, c['topn']) == want


def test_sample_logprobs_semantics():
    """oracle.sample_logprobs restates sampling_kernels.cu:67-90: min(n, cap) entries of logf(p) in sampling order; only in the
    reference's 1024-entry layout does a drawn token beyond the buffer replace the last entry."""
    rng = np.random.default_rng(0)
    logits = (rng.standard_normal(3000) * 2).astype(np.float16)
    ids, p = o.sample_filter(logits, 1.0, 0, 1.0, 0.0)
    assert len(ids) == 3000 and abs(p.sum() - 1) < 1e-9
    deep = int(ids[2000])
    e_ids, e_lp, e_sel = o.sample_logprobs(ids, p, deep, 1024)
    assert len(e_ids) == 1024 and np.array_equal(e_ids[:1023], ids[:1023]) and e_ids[1023] == deep
    assert e_lp[1023] == e_sel == np.float32(np.log(np.float32(p[2000])))
    e_ids, e_lp, e_sel = o.sample_logprobs(ids, p, deep, 8)
    assert np.array_equal(e_ids, ids[:8]) and e_sel == np.float32(np.log(np.float32(p[2000])))
    assert np.all(np.diff(e_lp) <= 0)
    ids1, p1 = o.sample_filter(logits, 1.0, 1, 1.0, 0.0)          # greedy = top_k 1: one candidate with logprob 0
    e_ids, e_lp, e_sel = o.sample_logprobs(ids1, p1, int(ids1[0]), 1024)
    assert list(e_ids) == [int(np.argmax(logits.astype(np.float32)))] and e_lp[0] == 0.0 and e_sel == 0.0
