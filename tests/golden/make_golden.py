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


if __name__ == '__main__':
    main()
