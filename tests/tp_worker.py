"""One TP rank of a tp-way engine, every rank a separate process on cuda:0 (not a test module; tests/test_gpu_tp.py starts tp
of these under a hard timeout).  Host-side exchange (RCCL id, IPC handles, results) goes through torch.distributed/gloo on
127.0.0.1.  argv: rank tp port moe(0/1) kv_bits [phase2(0/1)].  Rank 0 prints one JSON line.
phase2 = 1 adds, on the same engine: stochastic sampling through the native communicator (logits shards gathered through the P2P
segments; every token = the oracle's draw from the re-assembled logits, identical on both ranks) and a continuous-batching
session whose admissions join a running batch as MIXED forwards (decode rows + prompt rows in one forward, graph-captured
fused all-reduce for the decode steps in between), teacher-forced through the unsharded oracle."""
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
    phase2 = len(sys.argv) > 6 and int(sys.argv[6]) != 0
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
    extra_res = {}
    if phase2:
        eng.release()
        # ---- stochastic sampling at tp > 1 without RCCL ------------------------------------------------------------------
        params = [(0.9, 30, 0.95, 0.0, 111), None, (1.4, 0, 0.8, 0.02, 222)]
        eng.set_sampling(params)
        eng.prefill(prompts, max_new_tokens=4)
        s_logits = [eng.fetch_logits()]
        for _ in range(3):
            eng.decode(1)
            s_logits.append(eng.fetch_logits())
        s_toks = eng.fetch().copy()
        eng.release()
        sl = torch.from_numpy(np.stack(s_logits).astype(np.float32))
        sparts = [torch.zeros_like(sl) for _ in range(tp)]
        dist.all_gather(sparts, sl)
        st_ = torch.from_numpy(s_toks.astype(np.int64))
        sts = [torch.zeros_like(st_) for _ in range(tp)]
        dist.all_gather(sts, st_)
        bad = 0
        if rank == 0:
            sfull = torch.cat(sparts, -1).numpy().astype(np.float16)
            for b, p in enumerate(params):
                for k in range(4):
                    row = sfull[k][b]
                    if p is None:
                        bad += int(s_toks[b, k] != int(np.argmax(row.astype(np.float32))))
                        continue
                    ids_, pr = o.sample_filter(row, p[0], p[1], p[2], p[3])
                    bad += int(s_toks[b, k] != o.sample_draw(ids_, pr, o.philox_uniform(p[4], len(prompts[b]) + k)))
        extra_res['sampling_mismatch'] = bad
        extra_res['sampling_same_tokens'] = all(torch.equal(sts[0], t) for t in sts)
        # ---- continuous batching with mixed steps ---------------------------------------------------------------------------
        lens2, news2 = [30, 9, 50, 20, 70, 12, 130], [5, 9, 4, 7, 3, 6, 4]
        rng2 = np.random.default_rng(9)
        pr2 = [rng2.integers(0, cfg.vocab, n).astype(np.int32) for n in lens2]
        rids = [eng.submit(p_, n_, -1) for p_, n_ in zip(pr2, news2)]
        done, nstep = {}, 0
        while len(done) < len(rids) and nstep < 300:
            eng.step()
            nstep += 1
            for i, rid in enumerate(rids):
                if i not in done:
                    stt, tk_ = eng.poll(rid)
                    if stt != 0:
                        done[i] = (stt, tk_.copy())
        extra_res['cb_finished'] = len(done) == len(rids) and all(v[0] == 7 for v in done.values())
        extra_res['cb_mixed_steps'] = eng.mixed_steps()
        flat = np.concatenate([done[i][1] for i in sorted(done)]) if done else np.zeros(0, np.int32)
        ft = torch.from_numpy(flat.astype(np.int64))
        fts = [torch.zeros_like(ft) for _ in range(tp)]
        dist.all_gather(fts, ft)
        extra_res['cb_same_tokens'] = all(torch.equal(fts[0], t) for t in fts)
        cb_bad, cb_checked = 0, 0
        if rank == 0 and extra_res['cb_finished']:
            for i in sorted(done):
                om2 = o.OracleModel(cfg, w, batch=1, max_ctx=256)
                feed = [pr2[i]]
                for k in range(news2[i]):
                    _, lg2 = om2.forward(feed)
                    row = lg2[0].astype(np.float32)
                    top2 = np.sort(row)[-2:]
                    t_ = int(done[i][1][k])
                    if top2[1] - top2[0] > 6e-2:
                        cb_bad += int(t_ != int(np.argmax(row)))
                        cb_checked += 1
                    else:
                        cb_bad += int(row[t_] < top2[1] - 4e-2)
                    feed = [[t_]]
        extra_res['cb_token_mismatch'] = cb_bad
        extra_res['cb_checked'] = cb_checked
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
               'resid_diff': 0.0, **extra_res}
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
