"""HF checkpoint -> logical TM-layout weights (the on-disk side of the boundary).

Host-side mirror of the reference loader for Llama / InternLM2 / Mixtral with AWQ (W4A16 g128), FP8 (e4m3, 128x128 block
scales, `.weight_scale_inv`) or fp16 / bf16 weights:
  * source models                    lmdeploy/turbomind/models/llama.py:45-101, internlm2.py:34-87, mixtral.py:57-106
  * FP8 normalize / dequant          lmdeploy/turbomind/weight_format.py:349-384 (HF [out, in] -> [in, out], scales alike)
  * AWQ normalize (unpack order)     lmdeploy/turbomind/weight_format.py:200-234
  * RoPE q/k channel permutation     lmdeploy/turbomind/models/utils.py:306-373 (weight, scales and zeros alike)
  * QKV fusion / w1w3 interleave     lmdeploy/turbomind/builders/attention.py:65-108, ffn.py:31-65,138-170
  * config -> engine model config    lmdeploy/turbomind/converter.py:154-259
HF linears are [out, in]; TM layout is [in, out] (= AWQ's native [K, N/8] packing for quantised tensors).
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass, field
from glob import glob

import numpy as np

from .loader import interleave_gate_up, permute_qk_for_interleaved_rope, unpack_awq_gemm


@dataclass
class RopeConfig:
    dim: int = 128
    base: float = 10000.0
    type: str = 'default'
    factor: float = 1.0
    low_freq_factor: float = 1.0
    high_freq_factor: float = 4.0
    original_max_position_embeddings: int = 8192


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
    rope: RopeConfig = field(default_factory=RopeConfig)
    group: int = 128
    arch: str = 'llama'
    quantized: bool = True
    weight_format: str = 'u4'          # 'u4' (AWQ) | 'fp8' (block 128x128) | 'f16'
    moe_experts: int = 0               # Mixtral: num_local_experts
    moe_top_k: int = 0                 # num_experts_per_tok
    moe_norm_topk: bool = True
    moe_routed_scale: float = 1.0
    eos_token_id: int | list | None = None
    max_position_embeddings: int = 8192


def read_config(model_path: str) -> ModelConfig:
    with open(os.path.join(model_path, 'config.json')) as f:
        c = json.load(f)
    arch = (c.get('architectures') or ['LlamaForCausalLM'])[0]
    kind = 'internlm2' if 'InternLM2' in arch else 'llama'
    if kind == 'llama' and not any(a in arch for a in ('Llama', 'Mistral', 'Mixtral')):
        raise NotImplementedError(f'architecture {arch}: the MI355X hot path covers Llama / InternLM2 / Mixtral decoders')
    H = c['hidden_size']
    heads = c['num_attention_heads']
    D = c.get('head_dim') or H // heads
    q = c.get('quantization_config')
    wfmt = 'f16'
    if q is not None:
        if q.get('quant_method') == 'fp8':       # converter.py:186-187; block size fixed at 128 (converter.py:82,90)
            if list(q.get('weight_block_size') or [128, 128]) != [128, 128]:
                raise NotImplementedError(f'quantization_config {q}: fp8 needs 128x128 weight blocks')
            wfmt = 'fp8'
        elif q.get('quant_method') != 'awq' or q.get('bits', 4) != 4 or q.get('group_size', 128) != 128:
            raise NotImplementedError(f'quantization_config {q}: only AWQ 4-bit group 128 or block-128 FP8 '
                                      f'(lmdeploy/turbomind/converter.py:82-92)')
        else:
            wfmt = 'u4'
    rope = RopeConfig(dim=D, base=float(c.get('rope_theta', 10000.0)))
    rs = c.get('rope_scaling')
    if rs:
        t = rs.get('rope_type', rs.get('type'))
        if t == 'llama3':
            rope = RopeConfig(D, rope.base, 'llama3', float(rs['factor']), float(rs.get('low_freq_factor', 1.0)),
                              float(rs.get('high_freq_factor', 4.0)), int(rs.get('original_max_position_embeddings', 8192)))
        elif t == 'linear':
            rope = RopeConfig(D, rope.base, 'linear', float(rs['factor']))
        elif t not in (None, 'default'):
            raise NotImplementedError(f'rope_scaling type {t}')
    # eos ids: config.json + generation_config.json (GenerationConfig.update_from_hf_gen_cfg, lmdeploy/messages.py:176-199)
    eos = c.get('eos_token_id')
    gpath = os.path.join(model_path, 'generation_config.json')
    if os.path.exists(gpath):
        with open(gpath) as f:
            ge = json.load(f).get('eos_token_id')
        if ge is not None:
            merged = list(eos if isinstance(eos, (list, tuple)) else ([] if eos is None else [eos]))
            merged += [t for t in (ge if isinstance(ge, (list, tuple)) else [ge]) if t not in merged]
            eos = merged if len(merged) > 1 else merged[0]
    c = dict(c, eos_token_id=eos)
    return ModelConfig(hidden=H, layers=c['num_hidden_layers'], q_heads=heads,
                       kv_heads=c.get('num_key_value_heads', heads), head_dim=D, inter=c['intermediate_size'],
                       vocab=c['vocab_size'], rms_eps=float(c.get('rms_norm_eps', 1e-5)), rope=rope, arch=kind,
                       quantized=q is not None, eos_token_id=c.get('eos_token_id'),
                       max_position_embeddings=int(c.get('max_position_embeddings', 8192)), weight_format=wfmt,
                       moe_experts=int(c.get('num_local_experts', 0) or 0) if 'Mixtral' in arch else 0,
                       moe_top_k=int(c.get('num_experts_per_tok', 0) or 0) if 'Mixtral' in arch else 0)


class _Tensors:
    """Lazy name -> numpy lookup over every *.safetensors shard of a checkpoint."""

    def __init__(self, model_path: str):
        from safetensors import safe_open
        self._files = []
        self._index = {}
        for fn in sorted(glob(os.path.join(model_path, '*.safetensors'))):
            f = safe_open(fn, framework='pt')      # numpy has neither bf16 nor fp8
            self._files.append(f)
            for k in f.keys():
                self._index[k] = f
        if not self._index:
            raise FileNotFoundError(f'no *.safetensors under {model_path}')

    def __contains__(self, k):
        return k in self._index

    def get(self, k) -> np.ndarray:
        """numpy view of a tensor: bf16 -> float32 (exact), fp8 -> its uint8 codes, everything else as stored"""
        import torch
        t = self._index[k].get_tensor(k)
        if t.dtype == torch.bfloat16:
            t = t.float()
        elif t.dtype in (torch.float8_e4m3fn,):
            t = t.view(torch.uint8)
        return t.numpy().copy()     # own the memory: the tensor is a view into the file mapping


def _linear(t: _Tensors, prefix: str, quantized: bool) -> dict:
    """-> {'q','s','z'} (uint8 [K,N], fp16 [K/g,N] x2), {'f8','bs'} (e4m3 codes [K,N], fp32 block scales
    [K/128, ceil(N/128)]) or {'w'} fp16 [K,N]."""
    if quantized and (prefix + '.weight_scale_inv') in t:       # FP8Format.normalize: transpose weight and scales
        w = t.get(prefix + '.weight')
        assert w.dtype == np.uint8, f'{prefix}.weight: expected float8_e4m3fn / uint8 codes, got {w.dtype}'
        return dict(f8=np.ascontiguousarray(w.T), bs=np.ascontiguousarray(t.get(prefix + '.weight_scale_inv').astype(np.float32).T))
    if quantized and (prefix + '.qweight') in t:
        return dict(q=unpack_awq_gemm(t.get(prefix + '.qweight')),
                    s=t.get(prefix + '.scales').astype(np.float16),
                    z=unpack_awq_gemm(t.get(prefix + '.qzeros')).astype(np.float16))
    return dict(w=np.ascontiguousarray(t.get(prefix + '.weight').astype(np.float16).T))


def _cat(parts: list) -> dict:
    return {k: np.concatenate([p[k] for p in parts], axis=-1) for k in parts[0]}


def _map(lin: dict, fn) -> dict:
    """apply a per-output-channel permutation; fp8 block scales cover whole heads (128 channels) and stay put"""
    return {k: (v if k == 'bs' else fn(v)) for k, v in lin.items()}


def _fuse_w1w3(w1: dict, w3: dict) -> dict:
    """(gate_j, up_j) column interleave (builders/ffn.py:31-34).  fp8: codes interleaved, the scale row becomes
    [w1 blocks | w3 blocks] -- the engine's layout for *.w1w3.scales (include/tm_mi355x.h, tm_moe_set_expert)."""
    if 'f8' in w1:
        assert w1['f8'].shape[1] % 128 == 0, 'fp8 w1/w3: intermediate size must be a multiple of the 128-column block'
        return dict(f8=interleave_gate_up(w1['f8'], w3['f8']), bs=np.concatenate([w1['bs'], w3['bs']], axis=1), gated=True)
    return {kk: interleave_gate_up(w1[kk], w3[kk]) for kk in w1}


def load_hf_weights(model_path: str, cfg: ModelConfig) -> dict:
    t = _Tensors(model_path)
    D, Hq, Hkv = cfg.head_dim, cfg.q_heads, cfg.kv_heads
    layers = []
    for i in range(cfg.layers):
        if cfg.arch == 'llama':
            p = f'model.layers.{i}'
            q = _linear(t, p + '.self_attn.q_proj', cfg.quantized)
            k = _linear(t, p + '.self_attn.k_proj', cfg.quantized)
            v = _linear(t, p + '.self_attn.v_proj', cfg.quantized)
            wo = _linear(t, p + '.self_attn.o_proj', cfg.quantized)
            moe = None
            if cfg.moe_experts:      # Mixtral (models/mixtral.py:73-106): router + per-expert w1 / w3 / w2
                m = p + '.block_sparse_moe'
                moe = dict(moe_gate=np.ascontiguousarray(t.get(m + '.gate.weight').astype(np.float16).T), experts=[])
                for x in range(cfg.moe_experts):
                    e1, e3, e2 = (_linear(t, f'{m}.experts.{x}.{n}', cfg.quantized) for n in ('w1', 'w3', 'w2'))
                    moe['experts'].append(dict(w1w3=_fuse_w1w3(e1, e3), w2=e2))
                w1 = w3 = w2 = None
            else:
                w1 = _linear(t, p + '.mlp.gate_proj', cfg.quantized)
                w3 = _linear(t, p + '.mlp.up_proj', cfg.quantized)
                w2 = _linear(t, p + '.mlp.down_proj', cfg.quantized)
            n1 = t.get(p + '.input_layernorm.weight')
            n2 = t.get(p + '.post_attention_layernorm.weight')
        else:   # internlm2: fused wqkv, per kv group [q_0..q_{g-1}, k, v] (models/internlm2.py:34-87)
            p = f'model.layers.{i}'
            moe = None
            wqkv = _linear(t, p + '.attention.wqkv', cfg.quantized)
            g = Hq // Hkv

            def split(x):
                lead = x.shape[:-1]
                x = x.reshape(*lead, Hkv, g + 2, D)
                return (x[..., :g, :].reshape(*lead, Hq * D), x[..., g, :].reshape(*lead, Hkv * D),
                        x[..., g + 1, :].reshape(*lead, Hkv * D))
            parts = {k_: split(v_) for k_, v_ in wqkv.items()}
            q = {k_: parts[k_][0] for k_ in parts}
            k = {k_: parts[k_][1] for k_ in parts}
            v = {k_: parts[k_][2] for k_ in parts}
            wo = _linear(t, p + '.attention.wo', cfg.quantized)
            w1 = _linear(t, p + '.feed_forward.w1', cfg.quantized)
            w3 = _linear(t, p + '.feed_forward.w3', cfg.quantized)
            w2 = _linear(t, p + '.feed_forward.w2', cfg.quantized)
            n1 = t.get(p + '.attention_norm.weight')
            n2 = t.get(p + '.ffn_norm.weight')
        q = _map(q, lambda a: permute_qk_for_interleaved_rope(a, Hq, D))
        k = _map(k, lambda a: permute_qk_for_interleaved_rope(a, Hkv, D))
        ffn = moe if moe is not None else dict(w1w3=_fuse_w1w3(w1, w3), w2=w2)
        layers.append(dict(attn_norm=n1.astype(np.float16), ffn_norm=n2.astype(np.float16), w_qkv=_cat([q, k, v]), wo=wo, **ffn))
    if cfg.arch == 'llama':
        emb = t.get('model.embed_tokens.weight')
        norm = t.get('model.norm.weight')
        head = t.get('lm_head.weight') if 'lm_head.weight' in t else emb
    else:
        emb = t.get('model.tok_embeddings.weight')
        norm = t.get('model.norm.weight')
        head = t.get('output.weight')
    return dict(tok_embeddings=emb.astype(np.float16), layers=layers, norm=norm.astype(np.float16),
                output=np.ascontiguousarray(head.astype(np.float16).T))
