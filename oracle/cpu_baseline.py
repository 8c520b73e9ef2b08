"""CPU baseline for bench.py's `cpu_baseline` leg (TEST / MEASUREMENT INFRASTRUCTURE, never the product path).

kind = "port": the reference has no CPU backend (`PytorchEngineConfig.device_type` in {cuda, ascend, maca, camb},
lmdeploy/messages.py:531) and `/root/reference` does not exist on the GPU box, so this is a torch-CPU port of the
reference PyTorchEngine's pure-torch "default" op backend, restated from:
  * W4A16:   lmdeploy/pytorch/backends/default/awq_modules.py:37-46,58-81  ((q - z) * s in fp16, then matmul)
  * RMSNorm: lmdeploy/pytorch/backends/default/norm.py:14-28
  * SiLU*up: lmdeploy/pytorch/backends/default/activation.py:15-18
  * RoPE:    lmdeploy/pytorch/backends/default/apply_rotary_emb.py (rotate-half form)
  * attention: torch.nn.functional.scaled_dot_product_attention over a contiguous (dequantised int8) KV cache --
    the default backend has no attention op.
It times a BOUNDED sample of the bench workload (same model shapes, same batch, same context): `sample_layers`
decoder layers + lm_head for one decode step -- one untimed warm-up pass, then up to `passes` timed passes (fewer when
`budget_s` seconds of timed work are used up; at least one), median per-layer time -- and extrapolates to the full
layer count.  tests/test_oracle.py pins _awq_linear / _rmsnorm on the reference-generated golden vectors.
"""
from __future__ import annotations

import os
import time

import torch


def _awq_linear(x, q, s, z, group):
    # awq_modules.dequantize_gemm: (iweight - izeros) * scales, fp16 -> matmul.  CPU fp16 matmul is slow / partly
    # unsupported, so the contraction runs in fp32 (stated in the bench output).
    K, N = q.shape
    w = ((q.view(K // group, group, N).to(torch.float16) - z[:, None, :]) * s[:, None, :]).view(K, N)
    return (x.float() @ w.float()).to(torch.float16)


def _rmsnorm(x, w, eps, residual=None):
    if residual is not None:
        x = x + residual
        residual = x
    xf = x.float()
    var = xf.square().mean(-1, keepdim=True)
    y = (w.float() * (xf * torch.rsqrt(var + eps))).to(x.dtype)
    return y, residual


def _rope(x, cos, sin):
    d = x.shape[-1] // 2
    x1, x2 = x[..., :d], x[..., d:]
    return torch.cat([x1 * cos - x2 * sin, x2 * cos + x1 * sin], -1)


def physical_cores() -> int:
    """distinct (package, core) pairs of /proc/cpuinfo; SMT siblings count once.  Falls back to the logical count."""
    try:
        pairs, pkg = set(), None
        for line in open('/proc/cpuinfo'):
            if line.startswith('physical id'):
                pkg = line.split(':')[1].strip()
            elif line.startswith('core id'):
                pairs.add((pkg, line.split(':')[1].strip()))
        if pairs:
            return len(pairs)
    except OSError:
        pass
    return os.cpu_count() or 1


def run(model: dict, batch: int, ctx: int, sample_layers: int = 2, seed: int = 0, threads: int | None = None,
        passes: int = 3, budget_s: float = 25.0) -> dict:
    # one thread per PHYSICAL core: the contraction is fp32 FMA-bound, SMT siblings only add contention (round 2 ran
    # os.cpu_count() = 256 logical CPUs of a 128-core host and measured the oversubscribed rate)
    threads = threads or physical_cores()
    torch.set_num_threads(threads)
    g = torch.Generator().manual_seed(seed)
    H, D = model['hidden'], model['head_dim']
    Hq, Hkv, I, V, G = model['q_heads'], model['kv_heads'], model['inter'], model['vocab'], 128

    def lin(K, N):
        q = torch.randint(0, 16, (K, N), generator=g, dtype=torch.uint8)
        s = (torch.rand((K // G, N), generator=g) * 2e-3 + 1e-3).to(torch.float16)
        z = torch.randint(0, 16, (K // G, N), generator=g).to(torch.float16)
        return q, s, z

    layers = []
    for _ in range(sample_layers):
        layers.append(dict(qkv=lin(H, (Hq + 2 * Hkv) * D), wo=lin(Hq * D, H), w13=lin(H, 2 * I), w2=lin(I, H),
                           n1=torch.ones(H, dtype=torch.float16), n2=torch.ones(H, dtype=torch.float16),
                           # int8 KV cache + per-token (scale, zero), contiguous [B, Hkv, ctx, D]
                           kq=torch.randint(0, 256, (batch, Hkv, ctx, D), generator=g, dtype=torch.uint8),
                           vq=torch.randint(0, 256, (batch, Hkv, ctx, D), generator=g, dtype=torch.uint8),
                           kp=torch.rand((batch, Hkv, ctx, 2), generator=g).to(torch.float16),
                           vp=torch.rand((batch, Hkv, ctx, 2), generator=g).to(torch.float16)))
    w_out = (torch.randn((H, V), generator=g) * 0.01).to(torch.float16)
    x = torch.randn((batch, H), generator=g).to(torch.float16)
    resid = x.clone()
    cos = torch.ones((batch, 1, D // 2), dtype=torch.float16)
    sin = torch.zeros((batch, 1, D // 2), dtype=torch.float16)

    def layer_fwd(L, x, resid):
        qkv = _awq_linear(x, *L['qkv'], G)
        q = _rope(qkv[:, :Hq * D].view(batch, Hq, D), cos, sin)
        k = qkv[:, Hq * D:(Hq + Hkv) * D].view(batch, Hkv, 1, D)
        v = qkv[:, (Hq + Hkv) * D:].view(batch, Hkv, 1, D)
        kc = L['kq'].to(torch.float16) * L['kp'][..., :1] + L['kp'][..., 1:]
        vc = L['vq'].to(torch.float16) * L['vp'][..., :1] + L['vp'][..., 1:]
        kc = torch.cat([kc[:, :, 1:], k], 2).float()
        vc = torch.cat([vc[:, :, 1:], v], 2).float()
        rep = Hq // Hkv
        o = torch.nn.functional.scaled_dot_product_attention(
            q.view(batch, Hq, 1, D).float(), kc.repeat_interleave(rep, 1), vc.repeat_interleave(rep, 1))
        a = o.view(batch, Hq * D).to(torch.float16)
        h = _awq_linear(a, *L['wo'], G)
        x, resid = _rmsnorm(h, L['n2'], 1e-5, resid)
        gu = _awq_linear(x, *L['w13'], G)
        act = torch.nn.functional.silu(gu[:, 0::2].float()).to(torch.float16) * gu[:, 1::2]
        d = _awq_linear(act, *L['w2'], G)
        x, resid = _rmsnorm(d, L['n1'], 1e-5, resid)
        return x, resid

    with torch.no_grad():
        layer_fwd(layers[0], x, resid)                       # warm-up: allocator, thread pool, page faults
        (x.float() @ w_out.float()).argmax(-1)
        per_layer, per_head = [], []
        t_begin = time.perf_counter()
        for _ in range(max(1, passes)):
            xx, rr = x, resid
            t0 = time.perf_counter()
            for L in layers:
                xx, rr = layer_fwd(L, xx, rr)
            per_layer.append((time.perf_counter() - t0) / sample_layers)
            t0 = time.perf_counter()
            (xx.float() @ w_out.float()).argmax(-1)
            per_head.append(time.perf_counter() - t0)
            if time.perf_counter() - t_begin > budget_s:
                break
    t_layer = sorted(per_layer)[len(per_layer) // 2]
    t_head = sorted(per_head)[len(per_head) // 2]
    step_s = t_layer * model['layers'] + t_head
    return dict(value=batch / step_s, unit='tokens/s', cores=threads, logical_cpus=os.cpu_count(), kind='port',
                pinned_by='tests/test_oracle.py::test_cpu_baseline_port_matches_reference_golden_vectors (golden vectors generated by '
                          'importing the reference\'s pytorch/backends/default functions; /root/reference does not exist on the GPU box, '
                          'so the reference functions themselves cannot be the timed baseline)',
                sample=(f'{sample_layers} of {model["layers"]} decoder layers + lm_head, one decode step, batch {batch}, '
                        f'ctx {ctx}, int8 KV; 1 warm-up + {len(per_layer)} timed passes (median {t_layer:.2f} s / layer, '
                        f'{t_head:.2f} s head); fp32 contraction on CPU; extrapolated to {model["layers"]} layers'))


def run_config0(model: dict, batch: int = 1, ctx: int = 1024, steps: int = 4, seed: int = 0, threads: int | None = None,
                budget_s: float = 25.0) -> dict:
    """BASELINE.json configs[0]: InternLM2-1.8B, fp16 weights, fp16 KV, greedy decode on the host cores (the reference's own
    CPU-runnable case is its PyTorch engine's default op backend: pytorch/backends/default/{norm,activation,apply_rotary_emb}.py --
    plain torch ops; there is no quantised linear and no KV quantisation in it).  The WHOLE model runs (24 layers: no
    extrapolation): one untimed warm-up step, then up to `steps` timed decode steps of `batch` sequences at context `ctx` within
    `budget_s` seconds.  The contraction runs in fp32 over fp16-rounded weights (CPU fp16 matmul is slow / partly unsupported)."""
    threads = threads or physical_cores()
    torch.set_num_threads(threads)
    g = torch.Generator().manual_seed(seed)
    H, D = model['hidden'], model['head_dim']
    Hq, Hkv, I, V, NL = model['q_heads'], model['kv_heads'], model['inter'], model['vocab'], model['layers']

    def lin(K, N):      # fp16-rounded values kept in fp32: what `x.float() @ w.float()` would read, without a per-step conversion
        return (torch.randn((K, N), generator=g) * (0.5 / K ** 0.5)).to(torch.float16).float()

    layers = [dict(qkv=lin(H, (Hq + 2 * Hkv) * D), wo=lin(Hq * D, H), w13=lin(H, 2 * I), w2=lin(I, H),
                   n1=torch.ones(H, dtype=torch.float16), n2=torch.ones(H, dtype=torch.float16),
                   k=torch.randn((batch, Hkv, ctx, D), generator=g).to(torch.float16),
                   v=torch.randn((batch, Hkv, ctx, D), generator=g).to(torch.float16)) for _ in range(NL)]
    emb = (torch.randn((V, H), generator=g) * 0.02).to(torch.float16)
    w_out = lin(H, V)
    cos = torch.ones((batch, 1, D // 2), dtype=torch.float16)
    sin = torch.zeros((batch, 1, D // 2), dtype=torch.float16)
    rep = Hq // Hkv

    def step(ids):
        resid = emb[ids]
        x, _ = _rmsnorm(resid, layers[0]['n1'], 1e-5)
        for li, L in enumerate(layers):
            qkv = (x.float() @ L['qkv']).to(torch.float16)
            q = _rope(qkv[:, :Hq * D].view(batch, Hq, D), cos, sin)
            k = _rope(qkv[:, Hq * D:(Hq + Hkv) * D].view(batch, Hkv, D), cos, sin).view(batch, Hkv, 1, D)
            v = qkv[:, (Hq + Hkv) * D:].view(batch, Hkv, 1, D)
            L['k'][:, :, -1:], L['v'][:, :, -1:] = k, v          # the new token's K / V take the last cache position
            o = torch.nn.functional.scaled_dot_product_attention(q.view(batch, Hq, 1, D).float(), L['k'].float().repeat_interleave(rep, 1),
                                                                 L['v'].float().repeat_interleave(rep, 1))
            h = (o.view(batch, Hq * D) @ L['wo']).to(torch.float16)
            x, resid = _rmsnorm(h, L['n2'], 1e-5, resid)
            gu = (x.float() @ L['w13']).to(torch.float16)
            act = torch.nn.functional.silu(gu[:, 0::2].float()).to(torch.float16) * gu[:, 1::2]
            d = (act.float() @ L['w2']).to(torch.float16)
            x, resid = _rmsnorm(d, layers[li + 1]['n1'] if li + 1 < NL else layers[0]['n1'], 1e-5, resid)
        return (x.float() @ w_out).argmax(-1)

    with torch.no_grad():
        ids = torch.randint(0, V, (batch,), generator=g)
        ids = step(ids)                                          # warm-up
        times, t_begin = [], time.perf_counter()
        for _ in range(max(1, steps)):
            t0 = time.perf_counter()
            ids = step(ids)
            times.append(time.perf_counter() - t0)
            if time.perf_counter() - t_begin > budget_s:
                break
    t = sorted(times)[len(times) // 2]
    return dict(value=batch / t, unit='tokens/s', cores=threads, logical_cpus=os.cpu_count(), kind='port',
                sample=(f'the whole model ({NL} layers + lm_head), fp16 weights and fp16 KV, greedy decode, batch {batch}, ctx {ctx}: 1 warm-up + '
                        f'{len(times)} timed steps (median {t * 1e3:.1f} ms / step); fp32 contraction on CPU; no extrapolation'))
