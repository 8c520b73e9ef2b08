"""N > 1 path on CPU: two processes (gloo, world_size 2) each take their TP shard from the product loader
(`export_weights`), run the oracle's arithmetic on the shard, exchange partial sums with all_reduce exactly where the
engine calls RCCL (after wo and w2; (value, index) all-gather for the sharded lm_head) and must reproduce the
unsharded forward.  This checks that the sharding plan of engine.hip (+ engine_forward.hip) / loader.py is correct by construction; the
GPU collective itself (ncclAllReduce on the engine stream) is exercised by the driver's multi-GPU bench."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from lmdeploy_amd.turbomind import loader
from oracle import tm_oracle as o

f16 = np.float16


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _lin(slots, prefix, x, gated=False):
    q = o.unpack_u4_row(slots[prefix + '.qweight'])
    acc = o.gemm_f16_f32acc(x, o.w4a16_dequant(q, slots[prefix + '.scales'], slots[prefix + '.zeros']))
    return o.gated_silu_epilogue(acc) if gated else acc.astype(f16)


def _allreduce_f16(x):
    """fp16 sum like ncclAllReduce(ncclHalf, ncclSum) with 2 ranks (one rounding of the two-term sum)."""
    t = torch.from_numpy(x.astype(np.float32))
    dist.all_reduce(t)
    return t.numpy().astype(f16)


def _worker(rank, world, port, cfg, w, ids, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    s = loader.export_weights(cfg, w, world, rank)
    D = cfg.head_dim
    hq, hkv = cfg.q_heads // world, max(1, cfg.kv_heads // world)
    T = len(ids)
    resid = s['tok_embeddings.weight'][ids]
    x = o.rmsnorm(resid, s['layers.0.attention_norm.weight'], cfg.rms_eps)
    for li in range(cfg.layers):
        p = f'layers.{li}'
        qkv = _lin(s, p + '.attention.w_qkv', x)
        cos, sin = o.rope_cos_sin(cfg.rope, np.arange(T))
        q = o.rope_apply(qkv[:, :hq * D].reshape(T, hq, D), cos, sin)
        k = o.rope_apply(qkv[:, hq * D:(hq + hkv) * D].reshape(T, hkv, D), cos, sin)
        v = qkv[:, (hq + hkv) * D:].reshape(T, hkv, D)
        attn = o.prefill_attention(q, k.transpose(1, 0, 2), v.transpose(1, 0, 2)).reshape(T, hq * D)   # local heads only
        h = _allreduce_f16(_lin(s, p + '.attention.wo', attn))                        # collective #1
        resid, x = o.residual_rmsnorm(resid, h, s[p + '.ffn_norm.weight'], cfg.rms_eps)
        act = _lin(s, p + '.feed_forward.w1w3', x, gated=True)
        h = _allreduce_f16(_lin(s, p + '.feed_forward.w2', act))                      # collective #2
        nxt = s[f'layers.{li + 1}.attention_norm.weight'] if li + 1 < cfg.layers else s['norm.weight']
        resid, x = o.residual_rmsnorm(resid, h, nxt, cfg.rms_eps)
    logits = o.lm_head(x[-1:], s['output.weight'])                                    # vocab shard
    v_l = cfg.vocab // world
    cand = torch.tensor([[float(logits.max()), float(int(logits.argmax()) + rank * v_l)]])
    gathered = [torch.zeros_like(cand) for _ in range(world)]
    dist.all_gather(gathered, cand)                                                   # (value, index) all-gather
    best = max(gathered, key=lambda t: (t[0, 0].item(), -t[0, 1].item()))
    ret[f'logits{rank}'] = logits
    if rank == 0:
        ret['token'] = int(best[0, 1].item())
        ret['x'] = x
        ret['resid'] = resid
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_tp2_sharded_forward_matches_unsharded():
    cfg = o.ModelConfig(hidden=256, layers=2, q_heads=4, kv_heads=2, head_dim=128, inter=512, vocab=512, kv_bits=16)
    w = o.make_synthetic_weights(cfg, seed=7)
    ids = (np.arange(12) * 37) % cfg.vocab
    # unsharded reference (fp16 KV so that the only difference is the reduction order of the row-parallel sums)
    m = o.OracleModel(cfg, w, batch=1, max_ctx=64)
    tok, logits = m.forward([ids])
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), cfg, w, ids, ret), nprocs=2, join=True)
    # element-wise: the residual stream after the last layer (every token row) and the re-assembled logits.  The sharded
    # run differs from the unsharded one by the split of each row-parallel sum into two fp32 partial sums that are rounded
    # to fp16 before the all-reduce adds them (what ncclAllReduce(ncclHalf) does) -- a few fp16 ulps per collective.
    r, rr = ret['resid'].astype(np.float32), m.last_resid.astype(np.float32)
    assert r.shape == rr.shape == (12, 256)
    assert np.all(np.abs(r - rr) <= 4e-3 + 2.0**-8 * np.abs(rr)), f'sharded residual stream differs: {np.abs(r - rr).max()}'
    lg = np.concatenate([ret['logits0'], ret['logits1']], axis=-1).astype(np.float32)
    assert lg.shape == (1, cfg.vocab)
    assert np.abs(lg[0] - logits[0].astype(np.float32)).max() <= 2e-2
    top2 = np.sort(logits[0].astype(np.float32))[-2:]
    if top2[1] - top2[0] > 4e-2:          # no near tie: the greedy token must agree
        assert ret['token'] == int(tok[0])
    assert ret['token'] == int(np.argmax(lg[0]))      # the (value, index) all-gather picks the arg-max of the full row


def _moe_worker(rank, world, port, cfg, w, x, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    s = loader.export_weights(cfg, w, world, rank)
    p = 'layers.0.moe_ffn'
    deq = lambda n: o.fp8_dequant(s[n + '.weight'], s[n + '.scales'], gated=n.endswith('w1w3'))
    experts = [(deq(f'{p}.experts.{e}.w1w3'), deq(f'{p}.experts.{e}.w2')) for e in range(cfg.moe_experts)]
    part = o.moe_ffn(x, s[p + '.gate.weight'], experts, cfg.moe_top_k, cfg.moe_norm_topk, cfg.moe_routed_scale)[0]
    full = _allreduce_f16(np.asarray(part, f16))                  # the engine's all-reduce after the MoE combine
    if rank == 0:
        ret['out'] = full
        ret['w13_shape'] = s[f'{p}.experts.0.w1w3.weight'].shape
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_tp2_moe_fp8_block_matches_unsharded():
    """BASELINE config 5's plan (Mixtral, fp8 block scales, TP = 2): replicated router, every expert's w1w3 column-
    sharded (interleaved codes, [w1 blocks | w3 blocks] scale rows) and w2 row-sharded by the product loader; each rank
    runs the MoE block on its shard, the partial outputs are summed where the engine calls RCCL -- must reproduce the
    unsharded block (routing is identical on every rank because the router is replicated)."""
    cfg = o.ModelConfig(hidden=256, layers=1, q_heads=4, kv_heads=2, head_dim=128, inter=256, vocab=64, weight_format='fp8',
                        moe_experts=4, moe_top_k=2)
    w = o.make_synthetic_weights(cfg, seed=9)
    x = (np.random.default_rng(3).standard_normal((9, 256)) * 0.5).astype(f16)
    L = w['layers'][0]
    experts = [(o.fp8_dequant(E['w1w3']['f8'], E['w1w3']['bs'], True), o.fp8_dequant(E['w2']['f8'], E['w2']['bs']))
               for E in L['experts']]
    ref = np.asarray(o.moe_ffn(x, L['moe_gate'], experts, 2)[0], np.float32)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_moe_worker, args=(2, _free_port(), cfg, w, x, ret), nprocs=2, join=True)
    assert tuple(ret['w13_shape']) == (256, 256)                    # [hidden, 2 * inter / tp]
    got = ret['out'].astype(np.float32)
    assert np.all(np.abs(got - ref) <= 1e-2 * np.abs(ref) + 2e-3), np.abs(got - ref).max()


def _sampling_worker(rank, world, port, logits, params, ctx, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    B, V = logits.shape
    vl = V // world
    shard = torch.from_numpy(np.ascontiguousarray(logits[:, rank * vl:(rank + 1) * vl]).view(np.uint8))      # this rank's lm_head output (bytes)
    gathered = [torch.empty_like(shard) for _ in range(world)]
    dist.all_gather(gathered, shard)                                   # engine: ncclAllGather(ncclHalf) -> [tp][B][vl]
    stacked = np.stack([g.numpy().view(f16).reshape(B, vl) for g in gathered])        # [tp][B][vl]
    full = np.ascontiguousarray(stacked.transpose(1, 0, 2)).reshape(B, V)      # gather_vocab_kernel: [B][tp * vl]
    toks = []
    for b, p in enumerate(params):
        ids, pr = o.sample_filter(full[b], p[0], p[1], p[2], p[3])
        toks.append(o.sample_draw(ids, pr, o.philox_uniform(p[4], ctx[b])))
    ret[rank] = (toks, full)
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_tp2_sampling_gathers_logits_and_draws_identically():
    """Stochastic sampling at tp = 2 (engine_forward.hip head(): all-gather of the vocabulary shards, gather_vocab_kernel, the same
    Philox number on every rank): both ranks rebuild exactly the unsharded logits row and draw exactly the token the
    unsharded oracle draws -- no exchange after the gather."""
    rng = np.random.default_rng(17)
    B, V = 3, 512
    logits = rng.standard_normal((B, V)).astype(f16)
    params = [(0.9, 30, 0.95, 0.0, 111), (1.0, 1, 1.0, 0.0, 5), (1.4, 0, 0.8, 0.02, 222)]
    ctx = [13, 40, 8]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_sampling_worker, args=(2, _free_port(), logits, params, ctx, ret), nprocs=2, join=True)
    want = []
    for b, p in enumerate(params):
        ids, pr = o.sample_filter(logits[b], p[0], p[1], p[2], p[3])
        want.append(o.sample_draw(ids, pr, o.philox_uniform(p[4], ctx[b])))
    for rank in (0, 1):
        toks, full = ret[rank]
        assert np.array_equal(full.view(np.uint16), logits.view(np.uint16))
        assert toks == want
    assert want[1] == int(np.argmax(logits[1].astype(np.float32)))      # top_k = 1 row = greedy


def _table_worker(rank, world, port, tmpdir, ret):
    """bench.py's start-up protocol for tp > 1: rank 0 holds the measured dispatch tables (here: imported lines stand in for the
    tuner), exports them as text, the text is broadcast, every other rank imports it -- all ranks must then pick the same tilings"""
    import ctypes as C

    from lmdeploy_amd import _ffi
    from lmdeploy_amd.turbomind.engine import Engine
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    lib = _ffi.load()
    table = None
    if rank == 0:
        src = os.path.join(tmpdir, 'measured.txt')
        open(src, 'w').write('5120 768 64 3 4 1\n640 5120 64 6 1 2\n5120 3584 64 2 2 3\n1792 5120 64 6 1 4\n5120 768 8192 12 1 1\n'
                             'G 17 5 5120 16032 64 1 2 4 1\nG 18 2 640 5120 64 2 4 8 1\nG 34 0 5120 3584 64 64 0 0 0\n')
        assert lib.tm_gemm_import(src.encode()) == 0
        out = os.path.join(tmpdir, 'export.txt')
        Engine.export_gemm_table(out)
        table = open(out).read()
    box = [table]
    dist.broadcast_object_list(box, src=0)
    if rank != 0:
        path = os.path.join(tmpdir, f'import{rank}.txt')
        open(path, 'w').write(box[0])
        assert lib.tm_gemm_import(path.encode()) == 0
    picks = []
    for role, (K, N) in enumerate(((5120, 768), (640, 5120), (5120, 3584), (1792, 5120)), start=1):
        picks.append(Engine.pick_tiling(K, N, 64, role=role))
    picks.append(Engine.pick_tiling(5120, 768, 8192, role=1))
    picks.append(Engine.pick_general(1, 5, 5120, 16032, 64))
    picks.append(Engine.pick_general(2, 2, 640, 5120, 64))
    r = C.c_int(-1)
    _ffi.check(lib.tm_debug_grouped_tile(2, 5120, 3584, 64, C.byref(r)))
    picks.append(r.value)
    ret[rank] = picks
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_tp2_measured_dispatch_tables_travel_from_rank0(tmp_path):
    """tp > 1: rank 0's measured GEMM dispatch (P32 lines keyed by role + the `G` lines of the general kernel / grouped GEMMs) is
    exported as text, broadcast and imported by the other ranks (bench.py does exactly this), so that every rank runs identical
    tilings -- a rank-local winner could differ by measurement noise and desynchronise the ranks' step times."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_table_worker, args=(2, _free_port(), str(tmp_path), ret), nprocs=2, join=True)
    assert ret[0] == ret[1]
    assert ret[0][:5] == [(3, 4), (6, 1), (2, 2), (6, 1), (12, 1)]
    assert ret[0][5] == (1, 2, 4, 1) and ret[0][6] == (2, 4, 8, 1) and ret[0][7] == 64


def _mb_worker(rank, world, port, cfg, w, seqs, split, ret):
    """One rank of a tp = 2 prefill forward over several sequences, twice: `serial` = every collective over all rows where the engine
    calls RCCL on its own stream; `piped` = the engine's two-micro-batch choreography (engine_forward.hip:
    forward_layers_two_microbatches) with ASYNCHRONOUS all-reduces issued in the engine's order -- attention block A, AR(wo A) in
    flight under attention block B, AR(wo B) under FFN A, AR(w2 A) under FFN B, AR(w2 B) under the next attention block A -- and
    waited for exactly where the engine stream waits for the side stream's event."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    s = loader.export_weights(cfg, w, world, rank)
    D = cfg.head_dim
    hq, hkv = cfg.q_heads // world, max(1, cfg.kv_heads // world)
    cu = np.concatenate([[0], np.cumsum([len(t) for t in seqs])])
    ids = np.concatenate(seqs)

    def attn_block(x, s0, s1):
        """rows of sequences [s0, s1): w_qkv, RoPE, causal attention per sequence (local heads), wo -> partial sums"""
        r0, r1 = cu[s0], cu[s1]
        qkv = _lin(s, p + '.attention.w_qkv', x[r0:r1])
        out = np.zeros((r1 - r0, hq * D), f16)
        for b in range(s0, s1):
            a, e = cu[b] - r0, cu[b + 1] - r0
            T = e - a
            cos, sin = o.rope_cos_sin(cfg.rope, np.arange(T))
            q = o.rope_apply(qkv[a:e, :hq * D].reshape(T, hq, D), cos, sin)
            k = o.rope_apply(qkv[a:e, hq * D:(hq + hkv) * D].reshape(T, hkv, D), cos, sin)
            v = qkv[a:e, (hq + hkv) * D:].reshape(T, hkv, D)
            out[a:e] = o.prefill_attention(q, k.transpose(1, 0, 2), v.transpose(1, 0, 2)).reshape(T, hq * D)
        return _lin(s, p + '.attention.wo', out)

    class Async:
        """fp16 partial rows -> all-reduce in flight (the side stream); .wait() = the engine stream's wait for its event"""
        def __init__(self, x):
            self.t = torch.from_numpy(x.astype(np.float32))
            self.h = dist.all_reduce(self.t, async_op=True)

        def wait(self):
            self.h.wait()
            return self.t.numpy().astype(f16)

    out = {}
    for mode in ('serial', 'piped'):
        resid = s['tok_embeddings.weight'][ids].copy()
        x = o.rmsnorm(resid, s['layers.0.attention_norm.weight'], cfg.rms_eps)
        parts = [(0, len(seqs))] if mode == 'serial' else [(0, split), (split, len(seqs))]
        rows = [slice(cu[a], cu[b]) for a, b in parts]
        pend = [None] * len(parts)
        for li in range(cfg.layers):
            p = f'layers.{li}'
            for i, (a, b) in enumerate(parts):
                if li > 0:      # this part's w2 sum of the previous layer
                    resid[rows[i]], x[rows[i]] = o.residual_rmsnorm(resid[rows[i]], pend[i].wait(), s[p + '.attention_norm.weight'], cfg.rms_eps)
                pend[i] = Async(attn_block(x, a, b))                                  # collective #1 of the part, in flight
            for i in range(len(parts)):
                resid[rows[i]], x[rows[i]] = o.residual_rmsnorm(resid[rows[i]], pend[i].wait(), s[p + '.ffn_norm.weight'], cfg.rms_eps)
                act = _lin(s, p + '.feed_forward.w1w3', x[rows[i]], gated=True)
                pend[i] = Async(_lin(s, p + '.feed_forward.w2', act))                 # collective #2 of the part, in flight
        for i in range(len(parts)):
            resid[rows[i]], x[rows[i]] = o.residual_rmsnorm(resid[rows[i]], pend[i].wait(), s['norm.weight'], cfg.rms_eps)
        out[mode] = (resid.copy(), x.copy())
    if rank == 0:
        ret['serial'], ret['piped'] = out['serial'], out['piped']
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_tp2_prefill_two_microbatches_match_serial_collectives():
    """The overlap schedule of tensor-parallel prefill on CPU, world size 2 over gloo: the forward is split where the
    engine's own policy splits it (tm_prefill_split), the two micro-batches leapfrog through the layers with their all-reduces in
    flight under the other one's blocks, both ranks issue the collectives in the same order (no deadlock), and the residual stream
    and the final normed rows equal the serial schedule's BIT FOR BIT -- every operator between two attention blocks is row-wise and
    attention never crosses a sequence, so which rows share a collective cannot change a row.  Also against the unsharded oracle."""
    import ctypes as C
    from lmdeploy_amd import _ffi
    cfg = o.ModelConfig(hidden=256, layers=2, q_heads=4, kv_heads=2, head_dim=128, inter=512, vocab=512, kv_bits=16)
    w = o.make_synthetic_weights(cfg, seed=11)
    rng = np.random.default_rng(5)
    seqs = [rng.integers(0, cfg.vocab, n) for n in (7, 5, 9)]
    cu = np.concatenate([[0], np.cumsum([len(t) for t in seqs])]).astype(np.int32)
    sa, ra = C.c_int(), C.c_int()
    _ffi.check(_ffi.load().tm_prefill_split(cu.ctypes.data, len(seqs), 8, C.byref(sa), C.byref(ra)))
    assert (sa.value, ra.value) == (2, 12)                  # 12 | 9 rows
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_mb_worker, args=(2, _free_port(), cfg, w, seqs, sa.value, ret), nprocs=2, join=True)
    for a, b, what in zip(ret['serial'], ret['piped'], ('residual stream', 'final normed rows')):
        assert a.shape == (21, 256) and np.array_equal(a.view(np.uint16), b.view(np.uint16)), what
    m = o.OracleModel(cfg, w, batch=3, max_ctx=64)
    m.forward(seqs)
    r, rr = ret['piped'][0].astype(np.float32), m.last_resid.astype(np.float32)
    assert r.shape == rr.shape
    assert np.all(np.abs(r - rr) <= 4e-3 + 2.0**-8 * np.abs(rr)), f'sharded residual stream differs: {np.abs(r - rr).max()}'


def _fused_worker(rank, world, port, parts, resid, weight, ret):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        r, outs = resid.copy(), []
        for call in range(len(parts)):
            mine = parts[call][rank]
            # the row-flag one-shot exchange as data flow: every rank makes its partial rows visible to all (all-gather = the pushes into the
            # segments + the per-row flags), then sums ALL ranks' rows in rank order in fp32, rounds once, residual + RMSNorm locally
            got = [torch.zeros(mine.shape, dtype=torch.float16) for _ in range(world)]
            dist.all_gather(got, torch.from_numpy(mine))
            r_f, y_f = o.p2p_allreduce_norm([g.numpy() for g in got], r, weight, 1e-5)
            # the RCCL arrangement it replaces: ncclAllReduce(fp16 sum) + the residual-norm launch
            y_a = _allreduce_f16(mine)
            r_a, y_n = o.residual_rmsnorm(r, y_a, weight, 1e-5)
            outs.append((np.array_equal(r_f.view(np.uint16), r_a.view(np.uint16)), np.array_equal(y_f.view(np.uint16), y_n.view(np.uint16)),
                         r_f.copy(), y_f.copy()))
            r = r_f
        ret[rank] = outs
    finally:
        dist.destroy_process_group()


def test_tp2_fused_allreduce_norm_equals_allreduce_then_norm():
    """Round 6: at tp > 1 the decode-sized exchanges run as ONE fused all-reduce + residual + RMSNorm launch by default (comm_p2p.hip's
    row-flag kernel; reference AllreduceResidualBiasRMSnorm, fused_allreduce.cu:406-500 at unified_decoder.cc:278-285,328-335) instead of
    ncclAllReduce + the residual-norm launch.  World size 2 over gloo, three chained exchanges: the fused form's data flow (every rank sees
    every rank's rows, rank-ordered fp32 sum, one fp16 rounding) gives the bits of all-reduce-then-norm, and both ranks hold identical
    residual streams and normed rows -- the schedule change is invisible to the numbers."""
    rng = np.random.default_rng(3)
    M, H, world = 5, 256, 2
    parts = [[(rng.standard_normal((M, H)) * 2).astype(f16) for _ in range(world)] for _ in range(3)]
    resid = rng.standard_normal((M, H)).astype(f16)
    weight = (1 + 0.05 * rng.standard_normal(H)).astype(f16)
    ret = mp.Manager().dict()
    mp.spawn(_fused_worker, args=(world, _free_port(), parts, resid, weight, ret), nprocs=world, join=True)
    for call in range(3):
        for rank in range(world):
            same_r, same_y, _, _ = ret[rank][call]
            assert same_r and same_y, (call, rank)
        assert np.array_equal(ret[0][call][2], ret[1][call][2]) and np.array_equal(ret[0][call][3], ret[1][call][3])
