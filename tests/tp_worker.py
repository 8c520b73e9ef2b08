"""One TP rank of a tp-way engine, every rank a separate process on cuda:0 (not a test module; tests/test_gpu_tp.py starts tp
of these under a hard timeout).  Host-side exchange (RCCL id, IPC handles, results) goes through torch.distributed/gloo on
127.0.0.1.  argv: rank tp port moe(0/1) kv_bits.  Rank 0 prints one JSON line."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lmdeploy_amd.turbomind.engine import Engine                  # noqa: E402
from lmdeploy_amd.turbomind.loader import export_weights          # noqa: E402
from oracle import tm_oracle as o                                 # noqa: E402


def main():
    rank, tp, port, moe, kv_bits = [int(a) for a in sys.argv[1:6]]
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=tp)
    extra = dict(weight_format='fp8', moe_experts=4, moe_top_k=2, moe_fp8_act=True) if moe else {}
    cfg = o.ModelConfig(hidden=512, layers=2, q_heads=8, kv_heads=4, head_dim=128, inter=1024, vocab=1024, kv_bits=kv_bits,
                        rope=o.RopeParam(128, 500000.0, 'llama3', 8.0, 1.0, 4.0, 8192), **extra)
    w = o.make_synthetic_weights(cfg, seed=11)
    rng = np.random.default_rng(2)
    prompts = [rng.integers(0, cfg.vocab, n).astype(np.int32) for n in (41, 7, 64)]
    steps = 5
    eng = Engine.from_model_config(cfg, weight_type=2 if moe else 0, tp=tp, rank=rank, device=0, max_batch_size=4, session_len=256, quant_policy=kv_bits,
                                    cache_blocks=32, max_prefill_token_num=96, use_graph=1)
    # RCCL refuses two ranks on one device ("Duplicate GPU detected"), so this layout runs on the native communicator alone:
    # no tm_engine_comm_init -> every all-reduce (chunked prefills too) and the candidate all-gather go through comm_p2p.hip
    def gather(h):
        out = [None] * tp
        dist.all_gather_object(out, h)
        return out
    eng.comm_native_setup(gather, rows=64)
    eng.load_weights(export_weights(cfg, w, tp, rank))
    eng.start()
    eng.prefill(prompts, max_new_tokens=steps + 1)
    logits = [eng.fetch_logits()]
    resid = [np.zeros((len(prompts), cfg.hidden), np.float16)]      # prefill is chunked: only decode-step residuals compare
    for _ in range(steps):
        eng.decode(1)
        logits.append(eng.fetch_logits())
        resid.append(eng.fetch_residual(len(prompts)))
    toks = eng.fetch()
    dist.barrier()
    eng.close()
    # re-assemble the vocabulary shards on rank 0
    lg = torch.from_numpy(np.stack(logits).astype(np.float32))             # [steps+1, B, V/tp]
    parts = [torch.zeros_like(lg) for _ in range(tp)]
    dist.all_gather(parts, lg)
    rs = [torch.from_numpy(r.astype(np.float32)) for r in resid]
    same_resid = True
    for r in rs:                                                          # every rank must hold the same residual stream
        both = [torch.zeros_like(r) for _ in range(tp)]
        dist.all_gather(both, r)
        same_resid &= all(torch.equal(both[0], b) for b in both)
    tk = torch.from_numpy(toks.astype(np.int64))
    tks = [torch.zeros_like(tk) for _ in range(tp)]
    dist.all_gather(tks, tk)
    if rank == 0:
        full = torch.cat(parts, -1).numpy()
        om = o.OracleModel(cfg, w, batch=len(prompts), max_ctx=256)
        ids, ref = om.forward(prompts)
        res = {'ok': True, 'moe': moe, 'same_resid': bool(same_resid),
               'same_tokens': all(torch.equal(tks[0], t) for t in tks), 'max_logit_diff': 0.0, 'token_mismatch': 0,
               'resid_diff': 0.0}
        cur = toks[:, 0]
        for s in range(steps + 1):
            if s:
                ids, ref = om.forward([[int(t)] for t in cur])
                cur = toks[:, s]
                res['resid_diff'] = max(res['resid_diff'], float(np.abs(resid[s].astype(np.float32)
                                                                            - om.last_resid.astype(np.float32)).max()))
            d = np.abs(full[s] - ref.astype(np.float32))
            res['max_logit_diff'] = max(res['max_logit_diff'], float(d.max()))
            top2 = np.sort(ref.astype(np.float32), -1)[:, -2:]
            safe = (top2[:, 1] - top2[:, 0]) > 6e-2
            res['token_mismatch'] += int((toks[safe, s] != ids[safe]).sum())
            # the candidate all-gather must pick the arg-max of the re-assembled row
            res['token_mismatch'] += int((toks[:, s] != full[s].argmax(-1)).sum())
        print(json.dumps(res), flush=True)
    dist.destroy_process_group()


if __name__ == '__main__':
    try:
        main()
    except Exception as e:      # noqa: BLE001
        print(json.dumps({'ok': False, 'why': f'{type(e).__name__}: {e}'}), flush=True)
        raise
