"""SURVEY 8(f3) on the GPU: an HF checkpoint ON DISK -> lmdeploy_amd.pipeline(path) -> tokens / logits against the oracle.

The checkpoints are fabricated here in the on-disk formats the reference reads (AutoAWQ `qweight / qzeros / scales` in the
'gemm' nibble order, HF [out, in] fp16 tensors, rotate-half RoPE channel order, separate q / k / v and gate / up projections
for Llama, the per-kv-group fused `wqkv` for InternLM2).  The ORACLE side is assembled from the same HF tensors with oracle
functions only (oracle.tm_oracle: channel permutation, QKV fusion, gate / up interleave) -- never through
lmdeploy_amd.turbomind.checkpoint / loader -- so a mis-wired slot name, dtype, nibble order or permutation anywhere between
`checkpoint.py -> loader.py -> tm_engine_weight_copy -> tm_linear_prepare` shows up as wrong tokens.

Reference: lmdeploy/turbomind/weight_format.py:56-60 (AWQ unpack order), models/utils.py:306-373 (RoPE row permutation),
builders/attention.py:65-108 (QKV fusion), builders/ffn.py:31-65,138-243 (w1w3 interleave), models/internlm2.py:34-87.
"""
import json
import os

import numpy as np
import pytest

from oracle import tm_oracle as o

pytestmark = pytest.mark.gpu
f16 = np.float16


def _awq(tensors, name, w_kn):
    """fp16 [K, N] (input-major) -> AutoAWQ tensors under `name`; returns the boundary-layout dict the oracle consumes"""
    q, s, z, _ = o.quantize_groupwise_u4(w_kn, 128)
    tensors[name + '.qweight'] = o.pack_awq_gemm(q)
    tensors[name + '.qzeros'] = o.pack_awq_gemm(z.astype(np.uint8))
    tensors[name + '.scales'] = s
    return q, s, z


def _cat(parts):
    """fuse quantised projections along the output dim: codes, scales and zeros alike"""
    return dict(q=np.concatenate([p[0] for p in parts], -1), s=np.concatenate([p[1] for p in parts], -1),
                z=np.concatenate([p[2] for p in parts], -1))


def _perm(qsz, heads, D):
    return tuple(o.permute_qk_for_interleaved_rope(a, heads, D) for a in qsz)


def _fabricate(tmp, arch, cfg, rng):
    """Writes config.json + model.safetensors; returns the oracle's weight dict for the same model."""
    from safetensors.numpy import save_file
    H, D, Hq, Hkv, I, V = cfg.hidden, cfg.head_dim, cfg.q_heads, cfg.kv_heads, cfg.inter, cfg.vocab
    t, layers = {}, []
    rnd = lambda K, N: (rng.standard_normal((K, N)) * (0.1 / np.sqrt(K))).astype(f16)
    for li in range(cfg.layers):
        p = f'model.layers.{li}'
        an = (1 + 0.05 * rng.standard_normal(H)).astype(f16)
        fn = (1 + 0.05 * rng.standard_normal(H)).astype(f16)
        if arch == 'llama':
            qp, kp, vp = (_awq(t, f'{p}.self_attn.{n}_proj', rnd(H, h * D)) for n, h in (('q', Hq), ('k', Hkv), ('v', Hkv)))
            wo = _awq(t, f'{p}.self_attn.o_proj', rnd(Hq * D, H))
            g, u = _awq(t, f'{p}.mlp.gate_proj', rnd(H, I)), _awq(t, f'{p}.mlp.up_proj', rnd(H, I))
            w2 = _awq(t, f'{p}.mlp.down_proj', rnd(I, H))
            t[f'{p}.input_layernorm.weight'], t[f'{p}.post_attention_layernorm.weight'] = an, fn
        else:   # internlm2: wqkv fused per kv group as [q_0 .. q_{g-1}, k, v] x head_dim along the output dim
            gsz = Hq // Hkv
            qw, kw, vw = rnd(H, Hq * D), rnd(H, Hkv * D), rnd(H, Hkv * D)
            fused = np.concatenate([np.concatenate([qw[:, j * gsz * D:(j + 1) * gsz * D], kw[:, j * D:(j + 1) * D],
                                                    vw[:, j * D:(j + 1) * D]], 1) for j in range(Hkv)], 1)
            fq = _awq(t, f'{p}.attention.wqkv', fused)
            # the oracle's view of the same tensor: de-interleave the group layout by hand
            cols = np.arange((Hq + 2 * Hkv) * D).reshape(Hkv, gsz + 2, D)
            qi, ki, vi = cols[:, :gsz].reshape(-1), cols[:, gsz].reshape(-1), cols[:, gsz + 1].reshape(-1)
            qp, kp, vp = (tuple(a[:, ix] for a in fq) for ix in (qi, ki, vi))
            wo = _awq(t, f'{p}.attention.wo', rnd(Hq * D, H))
            g, u = _awq(t, f'{p}.feed_forward.w1', rnd(H, I)), _awq(t, f'{p}.feed_forward.w3', rnd(H, I))
            w2 = _awq(t, f'{p}.feed_forward.w2', rnd(I, H))
            t[f'{p}.attention_norm.weight'], t[f'{p}.ffn_norm.weight'] = an, fn
        layers.append(dict(attn_norm=an, ffn_norm=fn, w_qkv=_cat([_perm(qp, Hq, D), _perm(kp, Hkv, D), vp]),
                           wo=dict(q=wo[0], s=wo[1], z=wo[2]),
                           w1w3=dict(q=o.interleave_w1w3(g[0], u[0]), s=o.interleave_w1w3(g[1], u[1]), z=o.interleave_w1w3(g[2], u[2])),
                           w2=dict(q=w2[0], s=w2[1], z=w2[2])))
    emb = (0.05 * rng.standard_normal((V, H))).astype(f16)
    head = (rng.standard_normal((V, H)) * (0.1 / np.sqrt(H))).astype(f16)          # HF: [out = vocab, in = hidden]
    norm = (1 + 0.05 * rng.standard_normal(H)).astype(f16)
    if arch == 'llama':
        t['model.embed_tokens.weight'], t['model.norm.weight'], t['lm_head.weight'] = emb, norm, head
    else:
        t['model.tok_embeddings.weight'], t['model.norm.weight'], t['output.weight'] = emb, norm, head
    save_file(t, os.path.join(tmp, 'model.safetensors'))
    hf = {'architectures': ['LlamaForCausalLM' if arch == 'llama' else 'InternLM2ForCausalLM'], 'hidden_size': H,
          'num_hidden_layers': cfg.layers, 'num_attention_heads': Hq, 'num_key_value_heads': Hkv, 'head_dim': D,
          'intermediate_size': I, 'vocab_size': V, 'rms_norm_eps': cfg.rms_eps, 'rope_theta': cfg.rope.base, 'eos_token_id': 2,
          'max_position_embeddings': 512,
          'quantization_config': {'quant_method': 'awq', 'bits': 4, 'group_size': 128, 'zero_point': True, 'version': 'gemm'}}
    if cfg.rope.type == 'llama3':
        hf['rope_scaling'] = {'rope_type': 'llama3', 'factor': cfg.rope.factor, 'low_freq_factor': cfg.rope.low_freq_factor,
                              'high_freq_factor': cfg.rope.high_freq_factor,
                              'original_max_position_embeddings': cfg.rope.original_max_position_embeddings}
    elif cfg.rope.type == 'linear':
        hf['rope_scaling'] = {'type': 'linear', 'factor': cfg.rope.factor}
    json.dump(hf, open(os.path.join(tmp, 'config.json'), 'w'))
    return dict(tok_embeddings=emb, layers=layers, norm=norm, output=np.ascontiguousarray(head.T))


@pytest.mark.parametrize('arch,kv_bits', [('llama', 8), ('internlm2', 8), ('llama', 4)])
def test_checkpoint_on_disk_through_pipeline_matches_oracle(tmp_path, arch, kv_bits):
    """fabricated AWQ checkpoint -> pipeline(path, TurbomindEngineConfig(model_format='awq', quant_policy=8|4)) -> greedy tokens
    and first-step logits of a ragged batch against OracleModel on oracle-assembled weights (logits <= 3e-2; tokens equal
    wherever the oracle's top-2 margin exceeds the tolerance), static and continuous paths."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    from lmdeploy_amd import GenerationConfig, TurbomindEngineConfig, pipeline
    rope = (o.RopeParam(128, 500000.0, 'llama3', 8.0, 1.0, 4.0, 8192) if arch == 'llama' else o.RopeParam(128, 1e6, 'linear', 2.0))
    cfg = o.ModelConfig(hidden=256, layers=2, q_heads=4, kv_heads=2, head_dim=128, inter=512, vocab=640, kv_bits=kv_bits, rope=rope,
                        rms_eps=1e-5 if arch == 'llama' else 1e-6)
    rng = np.random.default_rng({'llama': 5, 'internlm2': 6}[arch] + kv_bits)
    w = _fabricate(str(tmp_path), arch, cfg, rng)
    prompts = [rng.integers(3, cfg.vocab, n).astype(np.int32).tolist() for n in (19, 5, 40)]
    N = 6
    pipe = pipeline(str(tmp_path), backend_config=TurbomindEngineConfig(model_format='awq', quant_policy=kv_bits, max_batch_size=3,
                                                                        session_len=128))
    assert pipe.model_cfg.quantized and pipe.model_cfg.arch == arch
    g = GenerationConfig(max_new_tokens=N, ignore_eos=True)
    got = [r.token_ids for r in pipe(prompts, g)]
    # first-step logits through the engine the pipeline built
    pipe.engine.prefill(prompts, max_new_tokens=2)
    lg0 = pipe.engine.fetch_logits().astype(np.float32)
    pipe.engine.release()
    # the scheduler path on the same engine (one more prompt than slots)
    cont = [r.token_ids for r in pipe(prompts + [prompts[1]], g)]
    pipe.close()

    om = o.OracleModel(cfg, w, batch=3, max_ctx=128)
    _, ref0 = om.forward([np.asarray(p) for p in prompts])
    ref0 = ref0.astype(np.float32)
    assert np.abs(lg0 - ref0).max() <= 3e-2, np.abs(lg0 - ref0).max()
    # teacher-force the engine's own tokens through the oracle: every token must be the oracle's arg-max unless the oracle's
    # top-2 margin is inside the logit tolerance (then either of the two)
    ref = ref0
    for s in range(N):
        for b in range(3):
            top = np.argsort(ref[b])[::-1][:2]
            margin = ref[b][top[0]] - ref[b][top[1]]
            assert got[b][s] == top[0] or (margin <= 6e-2 and got[b][s] == top[1]), (arch, b, s, got[b][s], top, margin)
        if s + 1 < N:
            _, ref = om.forward([[got[b][s]] for b in range(3)])
            ref = ref.astype(np.float32)
    assert cont[:3] == got and cont[3] == got[1]
