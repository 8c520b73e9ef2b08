#!/usr/bin/env python
"""Generate tests/golden/*.npz by IMPORTING the reference's own Python (read-only, /root/reference) in this
container.  /root/reference does not exist on the GPU box, so the vectors are committed and this script is only
re-run by hand:  python tests/golden/make_golden.py

What can be imported: `import lmdeploy` fails (mmengine, fire, ... missing) and
lmdeploy/turbomind/*.py import the CUDA extension `_turbomind`, so
  * leaf pure-torch modules of the PyTorchEngine default backend are imported through a namespace-package shim
    (sys.modules['lmdeploy'].__path__ = ['/root/reference/lmdeploy'], __init__.py is NOT executed):
      lmdeploy.pytorch.backends.default.norm        (DefaultRMSNormImpl)
      lmdeploy.pytorch.backends.default.awq_modules (unpack_awq, dequantize_gemm)  [needs a stub for
                                                     lmdeploy.pytorch.distributed]
      lmdeploy.pytorch.backends.default.activation  (DefaultSiluAndMulImpl)
  * lmdeploy/turbomind/weight_format.py is loaded with `_turbomind` and `.linear` stubbed (only DataType names are
    touched at import time): _unpack_awq_gemm, pack_u4_row, AWQFormat.dequant.
"""
import importlib
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = '/root/reference'
OUT = os.path.dirname(os.path.abspath(__file__))


def _shim():
    for name, path in (('lmdeploy', f'{REF}/lmdeploy'), ('lmdeploy.pytorch', f'{REF}/lmdeploy/pytorch'),
                       ('lmdeploy.pytorch.backends', f'{REF}/lmdeploy/pytorch/backends'),
                       ('lmdeploy.pytorch.backends.default', f'{REF}/lmdeploy/pytorch/backends/default'),
                       ('lmdeploy.turbomind', f'{REF}/lmdeploy/turbomind')):
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m
    dist = types.ModuleType('lmdeploy.pytorch.distributed')
    dist.all_reduce = lambda *a, **k: None
    sys.modules['lmdeploy.pytorch.distributed'] = dist
    tm = types.ModuleType('_turbomind')

    class _DT:
        def __getattr__(self, k):
            return k
    tm.DataType = _DT()
    sys.modules['_turbomind'] = tm
    lin = types.ModuleType('lmdeploy.turbomind.linear')
    lin.Linear = object
    sys.modules['lmdeploy.turbomind.linear'] = lin


def main():
    _shim()
    g = torch.Generator().manual_seed(1234)
    out = {}

    # ---- AWQ unpack / pack / dequant (weight_format.py) ------------------------------------------------------
    wf = importlib.import_module('lmdeploy.turbomind.weight_format')
    K, N, G = 256, 64, 128
    qweight = torch.randint(-2**31, 2**31 - 1, (K, N // 8), generator=g, dtype=torch.int32)
    qzeros = torch.randint(-2**31, 2**31 - 1, (K // G, N // 8), generator=g, dtype=torch.int32)
    scales = (torch.rand((K // G, N), generator=g) * 0.01 + 0.001).to(torch.float16)
    q_u8 = wf._unpack_awq_gemm(qweight)
    z_u8 = wf._unpack_awq_gemm(qzeros)
    packed = wf.pack_u4_row(q_u8)
    deq = wf.AWQFormat(block_in=G).dequant({'weight': q_u8.to(torch.float16), 'scales': scales,
                                            'zeros': z_u8.to(torch.float16)}, None)['weight']
    out.update(awq_qweight=qweight.numpy(), awq_qzeros=qzeros.numpy(), awq_scales=scales.numpy(),
               awq_unpacked=q_u8.numpy(), awq_zeros_unpacked=z_u8.numpy(), awq_packed_row=packed.numpy(),
               awq_dequant=deq.to(torch.float16).numpy())

    # ---- PyTorchEngine default backend: dequantize_gemm, RMSNorm, SiluAndMul ---------------------------------
    awq = importlib.import_module('lmdeploy.pytorch.backends.default.awq_modules')
    w_ref = awq.dequantize_gemm(qweight, qzeros, scales, 4, G)
    out['awq_dequantize_gemm'] = w_ref.to(torch.float16).numpy()
    x = torch.randn((5, K), generator=g).to(torch.float16)
    out['awq_x'] = x.numpy()
    out['awq_y_fp32'] = (x.float() @ w_ref.float()).numpy()

    norm = importlib.import_module('lmdeploy.pytorch.backends.default.norm')
    impl = norm.DefaultRMSNormImpl(512, 1e-5)
    xn = torch.randn((7, 512), generator=g).to(torch.float16)
    rn = torch.randn((7, 512), generator=g).to(torch.float16)
    wn = (1 + 0.02 * torch.randn(512, generator=g)).to(torch.float16)
    y0 = impl.forward(xn, wn)
    y1, r1 = impl.forward(xn, wn, rn)
    out.update(norm_x=xn.numpy(), norm_res=rn.numpy(), norm_w=wn.numpy(), norm_y=y0.numpy(), norm_y_res=y1.numpy(),
               norm_res_out=r1.numpy())

    act = importlib.import_module('lmdeploy.pytorch.backends.default.activation')
    gu = (torch.randn((6, 256), generator=g) * 3).to(torch.float16)
    out['silu_in'] = gu.numpy()
    out['silu_out'] = act.DefaultSiluAndMulImpl(False).forward(gu).numpy()

    np.savez_compressed(os.path.join(OUT, 'reference_python.npz'), **out)
    print('wrote', os.path.join(OUT, 'reference_python.npz'), {k: v.shape for k, v in out.items()})
    logits_process_golden()
    more_golden(wf)
    api_fields_golden()
    logprobs_view_golden()


def more_golden(wf):
    """reference_python2.npz: further leaf functions of the reference's Python.
      * Llama-3 RoPE: backends/default/rotary_embedding.py:141-172 (inv_freq) + :51-78 (cos/sin) and the rotate-half
        application backends/default/apply_rotary_emb.py:48-77 (the HF channel layout the loader permutes away);
      * MoE router: backends/default/moe.py:7-33 (softmax -> top-k) + the Mixtral renormalisation w / sum(w);
      * FP8 block-scaled dequant: lmdeploy/turbomind/weight_format.py:349-384 (FP8Format.dequant);
      * sampling filters on sorted scores: pytorch/engine/logits_process.py:68-96 (top-k, top-p, min-p)."""
    out = {}
    g = torch.Generator().manual_seed(4321)
    # ---- RoPE ---------------------------------------------------------------------------------------------------
    rot = importlib.import_module('lmdeploy.pytorch.backends.default.rotary_embedding')
    impl = rot.Llama3RotaryEmbeddingImpl(128, 500000.0, 8.0, 1.0, 4.0, 8192)
    out['rope_llama3_inv_freq'] = impl.inv_freq.float().numpy()
    pos = torch.tensor([[0, 1, 5, 63, 1023, 4097, 8191]])
    cos, sin = impl.forward(torch.zeros(1, dtype=torch.float16), pos)
    out['rope_pos'] = pos[0].numpy()
    out['rope_cos'] = cos[0].numpy()          # [T, 128] = [freqs | freqs], fp16
    out['rope_sin'] = sin[0].numpy()
    plain = rot.RotaryEmbeddingImpl(128, 10000.0, 1.0)
    out['rope_default_inv_freq'] = plain.inv_freq.float().numpy()
    ar = importlib.import_module('lmdeploy.pytorch.backends.default.apply_rotary_emb')
    q = torch.randn((7, 4, 128), generator=g).to(torch.float16)
    k = torch.randn((7, 2, 128), generator=g).to(torch.float16)
    out['rope_q'], out['rope_k'] = q.numpy().copy(), k.numpy().copy()
    qe, ke = ar.DefaultApplyRotaryEmbImpl().forward(q.clone(), k.clone(), cos[0], sin[0], inplace=False)
    out['rope_q_out'], out['rope_k_out'] = qe.numpy(), ke.numpy()
    # ---- MoE router ---------------------------------------------------------------------------------------------
    moe = importlib.import_module('lmdeploy.pytorch.backends.default.moe')
    logits = (torch.randn((33, 8), generator=g) * 2).to(torch.float16).float()      # fp16-representable router logits
    wts, ids = moe.DefaultSoftmaxTopKImpl(2).forward(logits)
    out['moe_logits'] = logits.numpy()
    out['moe_topk_ids'] = ids.numpy()
    out['moe_topk_softmax'] = wts.numpy()
    out['moe_topk_renorm'] = (wts / wts.sum(-1, keepdim=True)).numpy()
    # ---- FP8 ----------------------------------------------------------------------------------------------------
    base = types.ModuleType('lmdeploy.turbomind.builders._base')
    base._CPP_TO_TORCH = {'fp16': torch.float16}
    pkg = types.ModuleType('lmdeploy.turbomind.builders')
    pkg.__path__ = []
    sys.modules['lmdeploy.turbomind.builders'] = pkg
    sys.modules['lmdeploy.turbomind.builders._base'] = base
    codes = torch.randint(0, 256, (256, 384), generator=g, dtype=torch.int32).to(torch.uint8)
    codes[(codes & 0x7f) == 0x7f] = 0x3c                       # no NaN codes
    scales = torch.rand((2, 3), generator=g) * 0.02 + 1e-3
    deq = wf.FP8Format().dequant({'weight': codes, 'scales': scales}, 'fp16')['weight']
    out['fp8_codes'], out['fp8_block_scales'], out['fp8_dequant'] = codes.numpy(), scales.numpy(), deq.numpy()
    # ---- sampling filters ---------------------------------------------------------------------------------------
    lp = importlib.import_module('lmdeploy.pytorch.engine.logits_process')
    V = 200
    rows = []
    for _ in range(6):
        perm = torch.randperm(V, generator=g)
        rows.append((torch.linspace(-6, 6, V)[perm] * 1.0).to(torch.float16))      # distinct values: no tie order
    scores16 = torch.stack(rows)
    temperature = torch.tensor([1.0, 0.7, 1.3, 1.0, 0.5, 2.0])
    topk = torch.tensor([V, 40, 10, 5, V, 60])
    topp = torch.tensor([1.0, 0.9, 0.5, 1.0, 0.8, 0.95])
    minp = torch.tensor([0.0, 0.0, 0.0, 0.1, 0.05, 0.02])
    sc = scores16.float() / temperature[:, None]
    sc, order = sc.sort(1, descending=True)
    lp._filter_topk_sorted_(sc, topk)
    lp._filter_topp_sorted_(sc, topp)
    lp._filter_minp_sorted_(sc, minp)
    kept = torch.isfinite(sc)
    out['samp_logits'] = scores16.numpy()
    out['samp_temperature'], out['samp_topk'], out['samp_topp'], out['samp_minp'] = (temperature.numpy(), topk.numpy(),
                                                                                    topp.numpy(), minp.numpy())
    out['samp_order'] = order.numpy()
    out['samp_kept'] = kept.numpy()
    out['samp_probs'] = sc.softmax(-1).numpy()
    np.savez_compressed(os.path.join(OUT, 'reference_python2.npz'), **out)
    print('wrote', os.path.join(OUT, 'reference_python2.npz'), {k: v.shape for k, v in out.items()})


def api_fields_golden():
    """reference_api_fields.json: field names and default expressions of the user-facing dataclasses
    (lmdeploy/messages.py: GenerationConfig, TurbomindEngineConfig, Response), read with `ast` (the module itself
    cannot be imported here: pydantic / transformers glue)."""
    import ast
    import json
    tree = ast.parse(open(f'{REF}/lmdeploy/messages.py').read())
    out = {}
    for n in tree.body:
        if isinstance(n, ast.ClassDef) and n.name in ('GenerationConfig', 'TurbomindEngineConfig', 'Response'):
            fields = {}      # dataclass semantics: a re-annotated name keeps its first position, the last default wins
            for st in n.body:
                if isinstance(st, ast.AnnAssign):
                    fields[st.target.id] = ast.unparse(st.value) if st.value is not None else None
            out[n.name] = [[k, v] for k, v in fields.items()]
    json.dump(out, open(os.path.join(OUT, 'reference_api_fields.json'), 'w'), indent=1)
    print('wrote', os.path.join(OUT, 'reference_api_fields.json'), {k: len(v) for k, v in out.items()})


def logits_process_golden():
    """reference_logits_process.npz: lmdeploy/pytorch/engine/logits_process.py:24-65 (_process_bad_words_,
    _process_repetition_penalty_) -- the PyTorchEngine's restatement of the TurboMind kernels' formulas
    (kernels/sampling_penalty_kernels.cu:137-175, ban_bad_words.cu:51-95), run on fp32 scores.  Its module-level
    imports (lmdeploy.messages, pytorch.envs, SchedulerSequence, guided decoding) are stubbed; only the two leaf
    functions are called."""
    for name, path in (('lmdeploy.pytorch.engine', f'{REF}/lmdeploy/pytorch/engine'),):
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m
    for name, attrs in (('lmdeploy.messages', ['LogitsProcessor']), ('lmdeploy.pytorch.envs', []),
                        ('lmdeploy.pytorch.messages', ['SchedulerSequence']),
                        ('lmdeploy.pytorch.engine.guided_process', ['GuidedDecodingManager'])):
        m = types.ModuleType(name)
        for a_ in attrs:
            setattr(m, a_, object)
        sys.modules[name] = m
    sys.modules['lmdeploy.pytorch'].envs = sys.modules['lmdeploy.pytorch.envs']
    lp = importlib.import_module('lmdeploy.pytorch.engine.logits_process')
    g = torch.Generator().manual_seed(77)
    B, V, L = 4, 512, 40
    logits16 = (torch.randn(B, V, generator=g) * 3).to(torch.float16)
    ids = torch.randint(0, V, (B, L), generator=g)
    ids[1, 10:] = ids[1, 3]                                  # repeated ids are penalised once
    penalty = torch.tensor([1.3, 0.7, 1.0, 2.5])
    scores = logits16.float().clone()
    lp._process_repetition_penalty_(scores, ids, penalty)
    bad = torch.tensor([[5, 17, -1], [3, -1, -1], [-1, -1, -1], [100, 200, 511]])
    banned = scores.clone()
    lp._process_bad_words_(banned, bad, bad >= 0)
    np.savez_compressed(os.path.join(OUT, 'reference_logits_process.npz'), logits=logits16.numpy(), ids=ids.numpy(),
                        penalty=penalty.numpy(), penalised_fp32=scores.numpy(), bad=bad.numpy(),
                        banned_mask=torch.isinf(banned).numpy())
    print('wrote', os.path.join(OUT, 'reference_logits_process.npz'))


def logprobs_view_golden():
    """reference_logprobs_view.json: lmdeploy/turbomind/turbomind.py:472-503 (_get_logprobs_impl) -- how the Python side turns the
    engine's (logprob_vals, logprob_indexes, logprob_nums) buffers into Response.logprobs.  The module cannot be imported here
    (_turbomind, mmengine), so the one leaf function is taken out of the file with `ast` and executed as it stands; only its inputs
    and outputs are committed."""
    import ast
    import json
    tree = ast.parse(open(f'{REF}/lmdeploy/turbomind/turbomind.py').read())
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == '_get_logprobs_impl')
    ns = {'torch': torch}
    exec(compile(ast.Module([fn], []), 'turbomind.py:_get_logprobs_impl', 'exec'), ns)
    g = torch.Generator().manual_seed(5)
    T, W, V = 6, 1024, 5000
    cases = []
    for topn, offset in ((1, 0), (5, 0), (3, 2), (1024, 1)):
        idx = torch.stack([torch.randperm(V, generator=g)[:W] for _ in range(T)]).int()
        vals = -torch.rand(T, W, generator=g).sort(dim=-1).values * 9
        nums = torch.tensor([1, 3, 40, 1024, 7, 2])
        vals[2, 1] = float('-inf')                      # a zero-probability candidate inside the top-n is dropped
        n_out = T - offset
        toks = []
        for t in range(offset, T):
            n = int(nums[t])
            pick = 0 if t % 2 == 0 else n - 1           # the generated token: the best candidate / the last kept one
            toks.append(int(idx[t, pick]))
        got = ns['_get_logprobs_impl'](vals, idx, nums, toks, topn, offset)
        cases.append(dict(topn=topn, offset=offset, tokens=toks, nums=nums.tolist(),
                          idx=[idx[t, :int(nums[t])].tolist() for t in range(T)],
                          vals=[[float(v) if v != float('-inf') else '-inf' for v in vals[t, :int(nums[t])].tolist()] for t in range(T)],
                          expect=[[[int(k), float(v)] for k, v in d.items()] for d in got]))
        assert len(got) == n_out
    json.dump(cases, open(os.path.join(OUT, 'reference_logprobs_view.json'), 'w'))
    print('wrote', os.path.join(OUT, 'reference_logprobs_view.json'), len(cases), 'cases')


if __name__ == '__main__':
    main()
