"""Weight export: logical TM-layout tensors -> named, TP-sharded Param slots of the native engine.

Host-side mirror of the reference's loader for the hot path:
  * u4 packing `pack_u4_row` / AWQ unpack `_unpack_awq_gemm`  (lmdeploy/turbomind/weight_format.py:47-74)
  * QKV fusion [Q|K|V] per rank                                (lmdeploy/turbomind/builders/attention.py:65-112)
  * w1w3 gate/up column interleave for the fused-SiLU epilogue (lmdeploy/turbomind/builders/ffn.py:31-34,164-165)
  * TP sharding: column-parallel split of the output dim, row-parallel split of the input dim
    (lmdeploy/turbomind/builders/_base.py:102-113); KV heads are replicated when kv_heads < tp
    (builders/attention.py:30-50)
  * RoPE row permutation of HF q/k channels                     (lmdeploy/turbomind/models/utils.py:306-324)
  * slot hand-off with exact byte-size check                    (builders/_base.py:72-99)
"""
from __future__ import annotations

import numpy as np

AWQ_ORDER = (0, 4, 1, 5, 2, 6, 3, 7)


def unpack_awq_gemm(qw: np.ndarray) -> np.ndarray:
    """AWQ checkpoint int32[..., N/8] -> uint8[..., N]: logical column 8c+p is nibble AWQ_ORDER[p] of word c."""
    w = np.asarray(qw).view(np.uint32).astype(np.uint64)
    nib = [((w >> np.uint64(4 * i)) & np.uint64(15)).astype(np.uint8) for i in range(8)]
    return np.stack([nib[i] for i in AWQ_ORDER], axis=-1).reshape(*w.shape[:-1], -1)


def pack_u4_row(q: np.ndarray) -> np.ndarray:
    """uint8[..., N] -> int32[..., N/8] with nibble j of word c = element 8c+j (the engine's boundary layout)."""
    g = np.asarray(q, np.uint8).reshape(*q.shape[:-1], -1, 8).astype(np.uint64)
    w = np.zeros(g.shape[:-1], np.uint64)
    for j in range(8):
        w |= g[..., j] << np.uint64(4 * j)
    return w.astype(np.uint32).view(np.int32)


def permute_qk_for_interleaved_rope(w: np.ndarray, heads: int, head_dim: int) -> np.ndarray:
    """HF rotate-half channels (j, j+D/2) -> adjacent pairs (2j, 2j+1) along the last (output) dim."""
    lead = w.shape[:-1]
    return w.reshape(*lead, heads, 2, head_dim // 2).swapaxes(-1, -2).reshape(*lead, heads * head_dim)


def interleave_gate_up(w1: np.ndarray, w3: np.ndarray) -> np.ndarray:
    out = np.empty((*w1.shape[:-1], 2 * w1.shape[-1]), w1.dtype)
    out[..., 0::2] = w1
    out[..., 1::2] = w3
    return out


def _cols(t: np.ndarray, lo: int, hi: int) -> np.ndarray:
    return t[..., lo:hi]


def _shard_linear(lin: dict, kind: str, tp: int, rank: int, group: int, col_slices=None) -> dict:
    """lin: {'q','s','z'} (u4 as uint8 [K,N], fp16 [K/g,N]) or {'w'} fp16 [K,N].
    kind 'col': take `col_slices` (list of (lo,hi)) of the output dim; 'row': split the input dim evenly."""
    out = {}
    if 'f8' in lin:
        # e4m3 codes [K,N] + fp32 scales of 128x128 blocks [K/128, ceil(N/128)]: a shard must keep whole blocks
        f8, bs = lin['f8'], lin['bs']
        if lin.get('gated'):
            # fused w1w3, (gate_j, up_j)-interleaved codes, scale row = [w1 blocks | w3 blocks]: rank r owns the inter
            # columns [r*I/tp, (r+1)*I/tp) of both halves
            (lo, hi), = col_slices
            assert kind == 'col' and lo % 256 == 0 and hi % 256 == 0, 'fp8 w1w3 shards must keep whole 128-column blocks'
            half = bs.shape[1] // 2
            b0, b1 = lo // 256, hi // 256
            return {'f8': f8[:, lo:hi], 'bs': np.concatenate([bs[:, b0:b1], bs[:, half + b0:half + b1]], axis=1), 'gated': True}
        if kind == 'col':
            assert all(lo % 128 == 0 and (hi % 128 == 0 or hi == f8.shape[1]) for lo, hi in col_slices), \
                'fp8 column shards must be aligned to the 128-column scale blocks'
            return {'f8': np.concatenate([f8[:, lo:hi] for lo, hi in col_slices], axis=1),
                    'bs': np.concatenate([bs[:, lo // 128:(hi + 127) // 128] for lo, hi in col_slices], axis=1)}
        K = f8.shape[0]
        assert K % tp == 0 and (K // tp) % 128 == 0, 'fp8 row-parallel shards must keep whole 128-row scale blocks'
        lo, hi = rank * (K // tp), (rank + 1) * (K // tp)
        return {'f8': f8[lo:hi], 'bs': bs[lo // 128:hi // 128]}
    if kind == 'col':
        for k, t in lin.items():
            out[k] = np.concatenate([_cols(t, lo, hi) for lo, hi in col_slices], axis=-1)
    else:
        K = (lin['q'] if 'q' in lin else lin['w']).shape[0]
        assert K % tp == 0 and (K // tp) % group == 0, 'row-parallel shard must keep whole quantisation groups'
        lo, hi = rank * (K // tp), (rank + 1) * (K // tp)
        for k, t in lin.items():
            out[k] = t[lo:hi] if k in ('q', 'w') else t[lo // group:hi // group]
    return out


def _emit(slots: dict, prefix: str, lin: dict):
    if 'q' in lin:
        slots[prefix + '.qweight'] = np.ascontiguousarray(pack_u4_row(lin['q']))
        slots[prefix + '.scales'] = np.ascontiguousarray(lin['s'], dtype=np.float16)
        slots[prefix + '.zeros'] = np.ascontiguousarray(lin['z'], dtype=np.float16)
    elif 'f8' in lin:        # e4m3 codes [K,N] + fp32 128x128 block scales (lmdeploy/turbomind/weight_format.py:349-393)
        slots[prefix + '.weight'] = np.ascontiguousarray(lin['f8'], dtype=np.uint8)
        slots[prefix + '.scales'] = np.ascontiguousarray(lin['bs'], dtype=np.float32)
    else:
        slots[prefix + '.weight'] = np.ascontiguousarray(lin['w'], dtype=np.float16)


def export_weights(cfg, weights: dict, tp: int = 1, rank: int = 0) -> dict:
    """weights (unsharded, TM layout): {'tok_embeddings' [V,H], 'norm' [H], 'output' [H,V],
    'layers': [{'attn_norm','ffn_norm', 'w_qkv','wo','w1w3','w2': linear dicts}]} with w_qkv = [Q|K|V] along N and
    w1w3 already (gate_j, up_j)-interleaved.  Returns {slot name: contiguous numpy array} for this rank."""
    D = cfg.head_dim
    Hq, Hkv, I, G = cfg.q_heads, cfg.kv_heads, cfg.inter, cfg.group
    assert Hq % tp == 0 and I % tp == 0 and cfg.vocab % tp == 0
    hq_l = Hq // tp
    if Hkv >= tp:
        assert Hkv % tp == 0
        hkv_l, kv0 = Hkv // tp, rank * (Hkv // tp)
    else:                                   # replicate kv heads
        assert tp % Hkv == 0
        hkv_l, kv0 = 1, rank // (tp // Hkv)
    q0 = rank * hq_l
    nq, nkv = Hq * D, Hkv * D
    qkv_slices = [(q0 * D, (q0 + hq_l) * D), (nq + kv0 * D, nq + (kv0 + hkv_l) * D),
                  (nq + nkv + kv0 * D, nq + nkv + (kv0 + hkv_l) * D)]
    i_l = I // tp
    slots = {}
    for li, L in enumerate(weights['layers']):
        p = f'layers.{li}'
        _emit(slots, p + '.attention.w_qkv', _shard_linear(L['w_qkv'], 'col', tp, rank, G, qkv_slices))
        _emit(slots, p + '.attention.wo', _shard_linear(L['wo'], 'row', tp, rank, G))
        if 'experts' in L:     # mixture of experts: replicated router, every expert sharded like the dense FFN
            slots[p + '.moe_ffn.gate.weight'] = np.ascontiguousarray(L['moe_gate'], dtype=np.float16)
            for x, E_ in enumerate(L['experts']):
                q = f'{p}.moe_ffn.experts.{x}'
                _emit(slots, q + '.w1w3', _shard_linear(E_['w1w3'], 'col', tp, rank, G, [(2 * rank * i_l, 2 * (rank + 1) * i_l)]))
                _emit(slots, q + '.w2', _shard_linear(E_['w2'], 'row', tp, rank, G))
        else:
            _emit(slots, p + '.feed_forward.w1w3',
                  _shard_linear(L['w1w3'], 'col', tp, rank, G, [(2 * rank * i_l, 2 * (rank + 1) * i_l)]))
            _emit(slots, p + '.feed_forward.w2', _shard_linear(L['w2'], 'row', tp, rank, G))
        slots[p + '.attention_norm.weight'] = np.ascontiguousarray(L['attn_norm'], dtype=np.float16)
        slots[p + '.ffn_norm.weight'] = np.ascontiguousarray(L['ffn_norm'], dtype=np.float16)
    slots['tok_embeddings.weight'] = np.ascontiguousarray(weights['tok_embeddings'], dtype=np.float16)
    slots['norm.weight'] = np.ascontiguousarray(weights['norm'], dtype=np.float16)
    v_l = cfg.vocab // tp
    slots['output.weight'] = np.ascontiguousarray(weights['output'][:, rank * v_l:(rank + 1) * v_l], dtype=np.float16)
    return slots
