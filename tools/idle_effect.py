#!/usr/bin/env python
"""How much of the driver command's number is the chip's state?  DESIGN.md 8 leaves one discrepancy open: in-kernel stamps of ONE replay
on a chip that idled just before say a decode layer is faster than the throughput line of a chip that has been busy for a second
(prefill at full matrix power, then decode).  This times the SAME 20 graph-replayed decode steps of the headline configuration
(Llama-3-8B W4A16, int8 KV, batch 64, 1024-token prompts) in three states of one engine, round-robin:
    hot    right after 200 back-to-back decode steps (steady state, what value_1k_out sees)
    idle   after the device sat idle for IDLE seconds (what a stamped single replay sees)
    burst  right after 0.5 s of dense fp16 matrix work (what the driver's --steps 20 --warmup 5 window sees behind the 0.9 s prefill)
Contexts differ by a few hundred tokens across the measurements; the attention share of that is removed with the measured marginal cost
(0.0212 us per context token and launch) so that the states are comparable.  usage: idle_effect.py [IDLE_SECONDS]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import torch  # noqa: E402,F401  (device context like the bench)
from lmdeploy_amd.turbomind.engine import Engine  # noqa: E402

IDLE = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
B, S, K = 64, 1024, 20
model = dict(bench.LLAMA3_8B)
eng = Engine.from_model_config(bench._Cfg(model), max_batch_size=B, session_len=S + 2600, quant_policy=8, max_prefill_token_num=8192, use_graph=1)
eng.init_synthetic(seed=0)
eng.start()
gen = torch.Generator().manual_seed(0)
prompts = torch.randint(0, model['vocab'], (B, S), generator=gen, dtype=torch.int32).numpy()
eng.prefill(list(prompts), max_new_tokens=2500)
eng.decode(8)
eng.sync()
ctx = S + 1 + 8
MARGINAL_MS = 0.0212e-3 * model['layers']      # per context token of the batch and step


def timed():
    global ctx
    eng.sync()
    t0 = time.perf_counter()
    eng.decode(K)
    eng.sync()
    dt = (time.perf_counter() - t0) / K * 1e3
    c = ctx + (K - 1) / 2
    ctx += K
    return dt, c


def heat_decode(n):
    global ctx
    eng.decode(n)
    ctx += n


# prefill-like matrix work without touching the session: fp16 GEMMs of the framework torch ships (plumbing, only a heater here)
_a = torch.randn(8192, 8192, dtype=torch.float16, device='cuda')
_b = torch.randn(8192, 8192, dtype=torch.float16, device='cuda')


def heat_matrix(seconds):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(8):
            torch.matmul(_a, _b)
        torch.cuda.synchronize()


rows = []
for rep in range(3):
    heat_decode(200)
    rows.append(('hot', *timed()))
    eng.sync()
    time.sleep(IDLE)
    rows.append(('idle', *timed()))
    heat_matrix(0.5)
    rows.append(('burst', *timed()))
ref_ctx = rows[0][2]
print(f'# 20 graph-replayed decode steps, ms per step; "at ctx {ref_ctx:.0f}" = minus {MARGINAL_MS * 1e3:.3f} us per context token beyond it')
for name, dt, c in rows:
    print(f'{name:6s} ctx {c:7.1f}  {dt:.4f} ms/step   at ctx {ref_ctx:.0f}: {dt - (c - ref_ctx) * MARGINAL_MS:.4f}')
for name in ('hot', 'idle', 'burst'):
    v = [dt - (c - ref_ctx) * MARGINAL_MS for n, dt, c in rows if n == name]
    print(f'{name:6s} mean {np.mean(v):.4f} ms/step  ({B / np.mean(v) * 1e3:.0f} tok/s at ctx {ref_ctx:.0f})')
eng.close()
