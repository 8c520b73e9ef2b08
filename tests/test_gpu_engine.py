"""End-to-end parity: the C++ engine (prefill + hipGraph decode) against the oracle's whole-model restatement on a
small synthetic Llama-shaped model, same weights, same prompts.

Tolerance (SURVEY 8c "end-to-end tokens"): fp16 logits max_abs <= 0.25 * logit scale is the reference-style gate;
we require max |logit diff| <= 3e-2 (logits are O(1)) at every compared step and greedy-token agreement wherever the
oracle's top-2 margin exceeds that tolerance (no exact ties by construction)."""
import numpy as np
import pytest

from lmdeploy_amd import _ffi
from lmdeploy_amd.turbomind.engine import Engine
from lmdeploy_amd.turbomind.loader import export_weights
from oracle import tm_oracle as o

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('fold', [3, 1, 2, 0])
@pytest.mark.parametrize('kv_bits,use_graph', [(8, 1), (4, 0), (16, 1)])
def test_engine_matches_oracle(cuda, monkeypatch, kv_bits, use_graph, fold):
    """prefill + 6 decode steps of a ragged batch against the oracle model: logits of every step, greedy tokens (eager and
    graph-replayed).  fold = 3 (TM_FOLD_NORM=3; the default is 0 since round 6; bit 0: wo -> w1w3, bit 1: w2 -> next w_qkv): the decode steps run the RMSNorms folded into the
    GEMMs (5 launches per layer) --
    the oracle stays the reference's unfused sequence and the bound stays the unfused engine's (3e-2 on O(1) logits; measured
    max differences are printed for both arms)."""
    monkeypatch.setenv('TM_FOLD_NORM', str(fold))
    cfg = o.ModelConfig(hidden=256, layers=2, q_heads=4, kv_heads=2, head_dim=128, inter=512, vocab=1024,
                        kv_bits=kv_bits, rope=o.RopeParam(128, 500000.0, 'llama3', 8.0, 1.0, 4.0, 8192))
    w = o.make_synthetic_weights(cfg, seed=3)
    rng = np.random.default_rng(0)
    prompts = [rng.integers(0, cfg.vocab, n).astype(np.int32) for n in (70, 5, 64)]
    steps = 6

    eng = Engine.from_model_config(cfg, max_batch_size=4, session_len=256, quant_policy=0 if kv_bits == 16 else kv_bits,
                                    max_prefill_token_num=96, use_graph=use_graph)
    eng.load_weights(export_weights(cfg, w))
    eng.start()
    eng.prefill(prompts, max_new_tokens=steps + 1)
    logits = [eng.fetch_logits()]
    for _ in range(steps):
        eng.decode(1)
        logits.append(eng.fetch_logits())
    toks = eng.fetch()          # [B, steps+1]
    eng.close()

    om = o.OracleModel(cfg, w, batch=len(prompts), max_ctx=256)
    ids, lg = om.forward(prompts)
    ref_logits, ref_toks = [lg], [ids]
    cur = toks[:, 0]            # teacher-force the ENGINE's tokens so that one flipped near-tie cannot cascade
    for s in range(steps):
        ids, lg = om.forward([[int(t)] for t in cur])
        ref_logits.append(lg)
        ref_toks.append(ids)
        cur = toks[:, s + 1]
    worst = 0.0
    for s in range(steps + 1):
        d = np.abs(logits[s].astype(np.float32) - ref_logits[s].astype(np.float32))
        worst = max(worst, float(d.max()))
        assert d.max() <= 3e-2, f'step {s}: max logit diff {d.max()}'
        top2 = np.sort(ref_logits[s].astype(np.float32), -1)[:, -2:]
        safe = (top2[:, 1] - top2[:, 0]) > 6e-2
        assert np.array_equal(toks[safe, s], ref_toks[s][safe]), f'step {s}: greedy tokens differ'
    print(f'[engine vs oracle] kv_bits {kv_bits} graph {use_graph} folded norm {fold}: max logit diff over {steps + 1} steps {worst:.5f}')


@pytest.mark.parametrize('batch,fold_max', [(100, 128), (128, 128), (100, 64)])
def test_engine_folded_norm_batch_128_matches_oracle(cuda, monkeypatch, batch, fold_max):
    """Round 6 (BASELINE config 3 = batch 128): decode batches of 65 .. 128 rows run the folded layer too (TM_FOLD_MAX_M; default 64 = the
    round-5 limit: reduce-norm launches above it) -- producers on the 32-row-block tiles, consumers on those or the 128-row tile.  Prefill + 3
    decode steps (graph replay) against the oracle model, the bound of test_engine_matches_oracle."""
    monkeypatch.setenv('TM_FOLD_NORM', '3')
    monkeypatch.setenv('TM_FOLD_MAX_M', str(fold_max))
    cfg = o.ModelConfig(hidden=256, layers=2, q_heads=4, kv_heads=2, head_dim=128, inter=512, vocab=1024,
                        kv_bits=8, rope=o.RopeParam(128, 500000.0, 'llama3', 8.0, 1.0, 4.0, 8192))
    w = o.make_synthetic_weights(cfg, seed=4)
    rng = np.random.default_rng(batch)
    prompts = [rng.integers(0, cfg.vocab, int(n)).astype(np.int32) for n in rng.integers(3, 20, batch)]
    steps = 3
    eng = Engine.from_model_config(cfg, max_batch_size=batch, session_len=64, quant_policy=8, max_prefill_token_num=512, use_graph=1)
    eng.load_weights(export_weights(cfg, w))
    eng.start()
    eng.prefill(prompts, max_new_tokens=steps + 1)
    logits = [eng.fetch_logits()]
    for _ in range(steps):
        eng.decode(1)
        logits.append(eng.fetch_logits())
    toks = eng.fetch()
    eng.close()
    om = o.OracleModel(cfg, w, batch=len(prompts), max_ctx=64)
    ids, lg = om.forward(prompts)
    ref_logits = [lg]
    cur = toks[:, 0]
    for s in range(steps):
        ids, lg = om.forward([[int(t)] for t in cur])
        ref_logits.append(lg)
        cur = toks[:, s + 1]
    worst = 0.0
    for s in range(steps + 1):
        d = np.abs(logits[s].astype(np.float32) - ref_logits[s].astype(np.float32))
        worst = max(worst, float(d.max()))
        assert d.max() <= 3e-2, f'step {s}: max logit diff {d.max()}'
    print(f'[engine vs oracle] batch {batch} TM_FOLD_MAX_M {fold_max}: max logit diff over {steps + 1} steps {worst:.5f}')


@pytest.mark.parametrize('graph_comm,side_stream,native', [(0, 0, 0), (1, 0, 0), (1, 1, 0), (0, 1, 0), (1, 1, 1), (0, 1, 1)])
def test_engine_collective_path_single_rank(cuda, monkeypatch, graph_comm, side_stream, native):
    """The tp > 1 data path (RCCL all-reduce after wo / w2, vocabulary-sharded lm_head + candidate all-gather), driven
    on one GPU through a 1-rank communicator (TM_FORCE_COMM=1): must reproduce the collective-free engine exactly
    (a 1-rank sum is the identity; the non-deferred split-K reduce rounds the same fp32 sums).  graph_comm=1 also
    captures the RCCL calls into the decode hipGraph; side_stream = TM_COMM_STREAM (prefill-sized forwards only: these prompts
    stay below the split size, so both values must behave identically here; the overlapped form has its own test below).
    native=1 (round 6): the DEFAULT tp > 1 arrangement -- the native communicator on top of RCCL after its bring-up self-test, so every
    decode-sized wo / w2 exchange is ONE fused all-reduce + residual + RMSNorm launch (fused_allreduce.cu:406-500) -- same tokens, same logits."""
    cfg = o.ModelConfig(hidden=256, layers=2, q_heads=4, kv_heads=2, head_dim=128, inter=512, vocab=1024,
                        kv_bits=8, rope=o.RopeParam(128, 500000.0, 'llama3', 8.0, 1.0, 4.0, 8192))
    w = o.make_synthetic_weights(cfg, seed=5)
    rng = np.random.default_rng(1)
    prompts = [rng.integers(0, cfg.vocab, n).astype(np.int32) for n in (33, 64, 7)]

    def run(force):
        if force:
            monkeypatch.setenv('TM_FORCE_COMM', '1')
            monkeypatch.setenv('TM_GRAPH_COMM', str(graph_comm))
            monkeypatch.setenv('TM_COMM_STREAM', str(side_stream))
        else:
            monkeypatch.delenv('TM_FORCE_COMM', raising=False)
            monkeypatch.setenv('TM_FOLD_NORM', '0')     # the collective-free engine with the SAME arithmetic: reduce-norm launches, not the folded norm
        eng = Engine.from_model_config(cfg, max_batch_size=4, session_len=128, quant_policy=8, use_graph=1)
        if force:
            eng.comm_init(Engine.comm_unique_id())
            if native:
                eng.comm_native_setup(lambda h: [h], rows=4)
                assert eng.comm_native_selftest(), 'bring-up self-test of the native communicator (one rank)'
                assert eng.comm_info()['backend'] == 'native-p2p (decode) + rccl (large forwards)'
        eng.load_weights(export_weights(cfg, w))
        eng.start()
        eng.prefill(prompts, max_new_tokens=6)
        eng.decode(5)
        toks, lg = eng.fetch(), eng.fetch_logits()
        if force and native:    # the fall-back: dropped -> RCCL + the residual-norm launch again, same numbers
            eng.release()
            eng.comm_native_drop()
            assert eng.comm_info()['backend'] == 'rccl'
            eng.prefill(prompts, max_new_tokens=6)
            eng.decode(5)
            assert np.array_equal(eng.fetch(), toks)
        eng.close()
        return toks, lg

    t0, l0 = run(False)
    t1, l1 = run(True)
    assert np.array_equal(t0, t1)
    assert np.abs(l0.astype(np.float32) - l1.astype(np.float32)).max() <= 2e-3


@pytest.mark.parametrize('min_rows,chunk,lens,want_fw,want_mb', [
    (256, 1024, (300, 77, 190, 33), 1, 1),    # one forward, boundary 300 | 300: two micro-batches
    (64, 448, (300, 77, 190, 33), 2, 2),      # two forwards (448 + 152 rows, the second starts inside a sequence: history), uneven parts
    (64, 1024, (560, 20), 1, 0),              # lopsided boundary: the row-half schedule inside every layer
    (256, 1024, (600,), 1, 0),                # one sequence: row halves
])
def test_engine_prefill_allreduce_on_side_stream(cuda, monkeypatch, min_rows, chunk, lens, want_fw, want_mb):
    """Tensor-parallel prefill forwards run their RCCL all-reduces on the side stream under the other half's kernels (TM_COMM_STREAM,
    default on): at a sequence boundary near half the rows the forward becomes two micro-batches that leapfrog through ALL layers
    (engine_forward.hip: forward_layers_two_microbatches), otherwise the row-wise part of every layer (wo .. next w_qkv) runs as two
    row halves (forward_tail_two_halves).  Driven on one GPU through a 1-rank communicator (TM_FORCE_COMM=1, a 1-rank sum is the
    identity): the event choreography, the row / sequence offsets of every buffer, the shifted block table and the rebased cu_q of the
    second micro-batch must reproduce (a) the same engine with every collective on the engine stream (TM_COMM_STREAM=0), (b) the
    collective-free engine -- tokens equal, logits to the rounding of a different GEMM tiling per launch size -- and (c) the oracle.
    The decode steps behind the prefill run on the engine stream as before."""
    cfg = o.ModelConfig(hidden=256, layers=3, q_heads=4, kv_heads=2, head_dim=128, inter=512, vocab=1024,
                        kv_bits=8, rope=o.RopeParam(128, 500000.0, 'llama3', 8.0, 1.0, 4.0, 8192))
    w = o.make_synthetic_weights(cfg, seed=5)
    rng = np.random.default_rng(2)
    prompts = [rng.integers(0, cfg.vocab, n).astype(np.int32) for n in lens]

    def run(force, side):
        monkeypatch.setenv('TM_PIPE_MIN_ROWS', str(min_rows))
        monkeypatch.setenv('TM_COMM_STREAM', str(side))
        if force:
            monkeypatch.setenv('TM_FORCE_COMM', '1')
        else:
            monkeypatch.delenv('TM_FORCE_COMM', raising=False)
            monkeypatch.setenv('TM_FOLD_NORM', '0')
        eng = Engine.from_model_config(cfg, max_batch_size=4, session_len=1024, quant_policy=8, max_prefill_token_num=chunk, use_graph=1)
        if force:
            eng.comm_init(Engine.comm_unique_id())
        eng.load_weights(export_weights(cfg, w))
        eng.start()
        eng.prefill(prompts, max_new_tokens=5)
        lg0 = eng.fetch_logits().copy()
        eng.decode(4)
        out = eng.fetch().copy(), lg0, eng.fetch_logits().copy(), eng.comm_info()
        eng.close()
        return out

    base = run(False, 1)
    serial = run(True, 0)
    piped = run(True, 1)
    assert base[3]['side_stream'] is False and base[3]['overlapped_forwards'] == 0
    assert serial[3]['side_stream'] is False and serial[3]['side_stream_allreduces'] == 0
    info = piped[3]
    assert info['side_stream'] is True and (info['overlapped_forwards'], info['microbatch_forwards']) == (want_fw, want_mb), info
    assert info['side_stream_allreduces'] == 4 * cfg.layers * want_fw, info
    for other in (base, serial):
        assert np.array_equal(piped[0], other[0])
        for a, b in ((piped[1], other[1]), (piped[2], other[2])):
            assert np.abs(a.astype(np.float32) - b.astype(np.float32)).max() <= 2e-3
    om = o.OracleModel(cfg, w, batch=len(prompts), max_ctx=1024)
    _, ref = om.forward([p.tolist() for p in prompts])
    assert np.abs(piped[1].astype(np.float32) - ref.astype(np.float32)).max() <= 3e-2


@pytest.mark.parametrize('B,pf_class', [(4, 0), (80, 0), (4, 512)])
def test_engine_gemm_tuning_roundtrip(cuda, tmp_path, B, pf_class):
    """Measured GEMM dispatch (the reference's TM_GEMM_TUNE / EXPORT / IMPORT): the tuner times the candidate tilings of the
    four decode linears on the engine's own weights, the table can be exported and imported, and whatever it picked the
    engine still reproduces the oracle (every candidate is a parity-tested tiling of the same arithmetic).  pf_class = 512:
    additionally the prefill size class of forwards with 257 .. 512 tokens is tuned (table key M = 512) and the prompts are long
    enough that the prefill forward falls into it."""
    cfg = o.ModelConfig(hidden=512, layers=3, q_heads=4, kv_heads=2, head_dim=128, inter=1024, vocab=1024, kv_bits=8,
                        rope=o.RopeParam(128, 500000.0, 'llama3', 8.0, 1.0, 4.0, 8192))
    w = o.make_synthetic_weights(cfg, seed=5)
    rng = np.random.default_rng(9)
    prompts = [rng.integers(0, cfg.vocab, n).astype(np.int32) for n in ((40, 7, 65, 12) if not pf_class else (100, 90, 110, 85))]
    path = str(tmp_path / 'gemm_dispatch.txt')
    eng = Engine.from_model_config(cfg, max_batch_size=B, session_len=128, quant_policy=8, max_prefill_token_num=pf_class or 128)
    eng.load_weights(export_weights(cfg, w))
    eng.start()
    eng.tune_gemm(B, path)     # B = 80: the 64 < M <= 256 decode path (128-row tiles vs the 32-row-block shapes)
    lines = open(path).read().splitlines()
    rows = [tuple(int(v) for v in ln.split()) for ln in lines if not ln.startswith('G') and int(ln.split()[2]) == B]
    assert len(rows) == 4 and all(r[4] >= 1 for r in rows), rows        # w_qkv, wo, w1w3, w2 at M = B
    # the fp16 lm_head goes through the general kernel: its measured tiling travels as a `G 17 5 K N M nt splits waves kphases` line
    head = [ln.split() for ln in lines if ln.startswith('G 17 5 ') and ln.split()[5] == str(B)]    # (the table is process-global)
    assert len(head) == 1 and head[0][3:5] == ['512', '1024'] and head[0][6] in ('1', '2'), head
    assert {(r[0], r[1]) for r in rows} == {(512, 1024), (512, 512), (512, 2048), (1024, 512)}
    assert {(r[0], r[1], r[5]) for r in rows} == {(512, 1024, 1), (512, 512, 2), (512, 2048, 3), (1024, 512, 4)}    # keyed by role
    if pf_class:               # one 385-token prefill forward: size class 512
        eng.tune_gemm(pf_class, path)
        rows = [tuple(int(v) for v in ln.split()) for ln in open(path).read().splitlines()
                if not ln.startswith('G') and int(ln.split()[2]) == pf_class]
        assert len(rows) == 4 and all(r[3] >= 4 for r in rows), rows     # 128-row tiles or 32-row-block shapes
    prompts = (prompts * ((B + 3) // 4))[:B]
    eng.prefill(prompts, max_new_tokens=5)
    eng.decode(4)
    toks = eng.fetch()
    lg = eng.fetch_logits()
    eng.close()
    _ffi.check(_ffi.load().tm_gemm_import(path.encode()))
    om = o.OracleModel(cfg, w, batch=4, max_ctx=128)      # the batch repeats four prompts: the oracle runs them once
    _, ref = om.forward(prompts[:4])
    for s in range(4):
        _, ref = om.forward([[int(t)] for t in toks[:4, s]])
    assert all(np.array_equal(toks[b], toks[b % 4]) for b in range(B))
    lg = lg[:4]
    err = np.abs(lg.astype(np.float32) - ref.astype(np.float32))
    assert err.max() <= 4e-2, err.max()


@pytest.mark.parametrize('async_step', [1, 0])
@pytest.mark.parametrize('slots', [1, 3])
def test_continuous_batching_matches_static(cuda, monkeypatch, slots, async_step):
    """async_step = 1 (TM_ASYNC_STEP=1): the two-phase schedule / forward overlap (reference: turbomind.cc:171) -- a pure decode
    step is issued before the previous one is retired, a sequence that ended rides one more step as a dead row; the test insists
    that steps overlapped.  async_step = 0 (default): every step is retired by the call that issued it.  Same token streams.
    SURVEY 8f-1: the engine scheduler (submit / step / poll / cancel).  9 requests through 1 or 3 batch slots,
    different prompt lengths and generation lengths, one with an EOS stop, one cancelled mid-flight, one chunked prompt
    (longer than max_prefill_token_num); slots and blocks are reused.
    slots = 1: every request runs exactly the kernels of the static single-prompt run -> tokens must be identical.
    slots = 3: prompts share prefill iterations, so GEMM shapes (split-K association) differ from the single-prompt
    run in the last fp32 bits; tokens must agree up to the first position where the static run's top-2 logit margin is
    below 5e-2 (greedy decoding is only defined up to such near ties)."""
    cfg = o.ModelConfig(hidden=256, layers=2, q_heads=4, kv_heads=2, head_dim=128, inter=512, vocab=1024,
                        kv_bits=8, rope=o.RopeParam(128, 500000.0, 'llama3', 8.0, 1.0, 4.0, 8192))
    w = o.make_synthetic_weights(cfg, seed=11)
    rng = np.random.default_rng(4)
    lens = [70, 5, 64, 33, 150, 9, 1, 40, 17]
    news = [6, 12, 3, 9, 5, 20, 8, 1, 10]
    prompts = [rng.integers(0, cfg.vocab, n).astype(np.int32) for n in lens]
    weights = export_weights(cfg, w)

    monkeypatch.setenv('TM_ASYNC_STEP', str(async_step))
    # reference: every prompt alone through the static path (tokens + top-2 margin of every step)
    eng = Engine.from_model_config(cfg, max_batch_size=slots, session_len=256, quant_policy=8, max_prefill_token_num=96)
    eng.load_weights(weights)
    eng.start()
    ref, margin = [], []
    for p, n in zip(prompts, news):
        eng.prefill([p], max_new_tokens=n)
        mg = []
        for k in range(n):
            top2 = np.sort(eng.fetch_logits()[0].astype(np.float32))[-2:]
            mg.append(float(top2[1] - top2[0]))
            if k + 1 < n:
                eng.decode(1)
        ref.append(eng.fetch()[0].copy())
        margin.append(mg)
        eng.release()

    def check(i, toks, upto):
        """toks must reproduce ref[i][:upto] (exactly for one slot; up to the first near tie otherwise)"""
        assert len(toks) == upto, f'request {i}: {len(toks)} tokens, expected {upto}'
        for k in range(upto):
            if toks[k] != ref[i][k]:
                assert slots > 1 and margin[i][k] < 5e-2, f'request {i} token {k}: {toks} vs {ref[i]} (margin {margin[i][k]})'
                return

    # continuous batching on the same engine object
    eos_req, eos_tok = 5, int(ref[5][7])                      # request 5 stops at its 8th token
    first = int(np.flatnonzero(ref[5] == eos_tok)[0])        # (or earlier, if that id occurs before)
    ids = [eng.submit(p, n, eos_tok if i == eos_req else -1) for i, (p, n) in enumerate(zip(prompts, news))]
    cancel_req, cancelled_at = 1, None
    done, steps, max_active = {}, 0, 0
    while len(done) < len(ids):
        na, nw = eng.step()
        max_active = max(max_active, na)
        steps += 1
        assert steps < 400, 'scheduler does not make progress'
        for i, rid in enumerate(ids):
            if i in done:
                continue
            st, toks = eng.poll(rid)
            if i == cancel_req and st == 0 and len(toks) >= 4 and cancelled_at is None:
                eng.cancel(rid)
                cancelled_at = len(toks)
                st, toks = eng.poll(rid)
            if st != 0:
                done[i] = (st, toks.copy())
    assert max_active <= slots
    assert (eng.overlapped_steps() > 0) if async_step else (eng.overlapped_steps() == 0), eng.overlapped_steps()
    for i, (st, toks) in done.items():
        if i == cancel_req:
            assert st == 8
            check(i, toks, cancelled_at)
        elif i == eos_req and slots == 1:
            assert st == 7
            check(i, toks, first + 1)
        elif i == eos_req:
            assert st == 7 and (toks[-1] == eos_tok or len(toks) == news[i])
        else:
            assert st == 7
            check(i, toks, news[i])
    # the session can be left and the static path used again
    eng.release()
    eng.prefill([prompts[0]], max_new_tokens=3)
    eng.decode(2)
    assert np.array_equal(eng.fetch()[0], ref[0][:3])
    eng.close()


@pytest.mark.parametrize('mixed', [1, 0])
@pytest.mark.parametrize('kv_bits,banned', [(8, 0), (4, 0), (16, 0), (8, 1)])
def test_continuous_batching_matches_oracle(cuda, monkeypatch, kv_bits, banned, mixed):
    """mixed = 1 (default): an admission that arrives while other requests decode runs its LAST prefill forward together with
    the decode rows of every slot (the reference's unified batch, unified_attention_layer.cc:310-311) -- int8 / int4 KV through
    the fused decode prologue, fp16 KV through kv_rope_store, chunked and multi-iteration admissions included, with logits
    processors on (banned = 1: every request carries 24 banned token ids, so the seen-mask / processor kernels run in the
    merged forward) -- and the test insists that this happened; mixed = 0: prefill forwards and decode steps alternate.
    The scheduler path against the ORACLE (not against the engine's own static path): 8 requests of different prompt and
    generation lengths through 3 batch slots (admissions join a running batch, slots and KV blocks are reused, one prompt
    is chunked); every request's token stream is replayed through the oracle model alone (batch 1, teacher-forced with
    the engine's tokens; banned ids masked in the oracle's row) and every engine token must be the arg-max of the oracle's
    logits wherever the oracle's top-2 margin exceeds 1.5e-2 -- the synthetic model's logits have sigma = 0.1, the engine's
    measured logit error is a few 1e-3 -- and within 1e-2 of the oracle's best logit everywhere else (near ties)."""
    cfg = o.ModelConfig(hidden=256, layers=2, q_heads=4, kv_heads=2, head_dim=128, inter=512, vocab=1024,
                        kv_bits=kv_bits, rope=o.RopeParam(128, 500000.0, 'llama3', 8.0, 1.0, 4.0, 8192))
    w = o.make_synthetic_weights(cfg, seed=13)
    rng = np.random.default_rng(6)
    lens = [70, 5, 64, 33, 150, 9, 1, 40]
    news = [6, 12, 3, 9, 5, 14, 8, 2]
    prompts = [rng.integers(0, cfg.vocab, n).astype(np.int32) for n in lens]
    bans = [sorted(rng.choice(np.arange(1, cfg.vocab), 24, replace=False).tolist()) for _ in lens] if banned else [None] * len(lens)
    monkeypatch.setenv('TM_MIXED_STEP', str(mixed))
    eng = Engine.from_model_config(cfg, max_batch_size=3, session_len=256, quant_policy=0 if kv_bits == 16 else kv_bits,
                                   max_prefill_token_num=96)
    eng.load_weights(export_weights(cfg, w))
    eng.start()
    ids = [eng.submit(p, n, -1, None, dict(bad_ids=b) if b else None) for p, n, b in zip(prompts, news, bans)]
    done, steps = {}, 0
    while len(done) < len(ids):
        eng.step()
        steps += 1
        assert steps < 400, 'scheduler does not make progress'
        for i, rid in enumerate(ids):
            if i not in done:
                st, toks = eng.poll(rid)
                if st != 0:
                    done[i] = (st, toks.copy())
    n_mixed = eng.mixed_steps()
    eng.close()
    # 5 admissions join a running batch (the first three start together); the 150-token prompt is chunked (96 + 54): all merge
    assert (n_mixed >= 4) if mixed else (n_mixed == 0), f'{n_mixed} mixed steps'
    checked = 0
    for i, (st, toks) in done.items():
        assert st == 7 and len(toks) == news[i], f'request {i}: status {st}, {len(toks)} tokens'
        om = o.OracleModel(cfg, w, batch=1, max_ctx=256)
        feed = [prompts[i]]
        for k in range(news[i]):
            _, lg = om.forward(feed)
            row = lg[0].astype(np.float32)
            if bans[i]:
                assert int(toks[k]) not in bans[i], f'request {i} token {k}: banned id {toks[k]} generated'
                row[bans[i]] = -np.inf
            top2 = np.sort(row)[-2:]
            if top2[1] - top2[0] > 1.5e-2:
                assert int(toks[k]) == int(np.argmax(row)), f'request {i} token {k}: engine {toks[k]} oracle {np.argmax(row)}'
                checked += 1
            else:   # near tie: the engine's token must score within the tolerance of the oracle's best
                assert row[int(toks[k])] >= top2[1] - 1e-2
            feed = [[int(toks[k])]]
    assert checked >= sum(news) // 3


def test_engine_thread_serving(cuda):
    """The engine thread (tm_engine_serve_start; reference: Engine::Impl::InternalThreadEntry + the signal thread's
    callback contract, engine.cc:770-870, turbomind.py:808-812): 9 requests submitted concurrently from 3 Python
    threads into 3 slots while the loop runs; every thread streams its requests with tm_engine_wait; one request is
    cancelled mid-flight.  Tokens must match the static single-prompt run up to the first greedy near tie; the callback
    sees every token exactly once, in order, and nothing after a non-zero status; step / prefill are refused
    (TM_CONFLICT) while the thread runs; after serve_stop the caller-driven loop works again."""
    import threading
    from lmdeploy_amd._ffi import TmError
    cfg = o.ModelConfig(hidden=256, layers=2, q_heads=4, kv_heads=2, head_dim=128, inter=512, vocab=1024,
                        kv_bits=8, rope=o.RopeParam(128, 500000.0, 'llama3', 8.0, 1.0, 4.0, 8192))
    w = o.make_synthetic_weights(cfg, seed=11)
    rng = np.random.default_rng(4)
    lens = [70, 5, 64, 33, 150, 9, 1, 40, 17]
    news = [6, 12, 3, 9, 5, 20, 8, 1, 10]
    prompts = [rng.integers(0, cfg.vocab, n).astype(np.int32) for n in lens]
    eng = Engine.from_model_config(cfg, max_batch_size=3, session_len=256, quant_policy=8, max_prefill_token_num=96)
    eng.load_weights(export_weights(cfg, w))
    eng.start()
    ref, margin = [], []
    for p, n in zip(prompts, news):
        eng.prefill([p], max_new_tokens=n)
        mg = []
        for k in range(n):
            top2 = np.sort(eng.fetch_logits()[0].astype(np.float32))[-2:]
            mg.append(float(top2[1] - top2[0]))
            if k + 1 < n:
                eng.decode(1)
        ref.append(eng.fetch()[0].copy())
        margin.append(mg)
        eng.release()

    events, cancel = {}, {}                       # req id -> [(status, n_tokens)] as seen by the callback
    def on_update(rid, st, n):
        events.setdefault(rid, []).append((st, n))
        if cancel.get('rid') == rid and n >= 4 and st == 0 and 'done' not in cancel:
            cancel['done'] = n
            eng.cancel(rid)                       # API calls are allowed from the callback (engine thread, lock not held)
    eng.serve_start(on_update)
    with pytest.raises(TmError) as ei:
        eng.step()
    assert ei.value.status == 2
    with pytest.raises(TmError) as ei:
        eng.prefill([prompts[0]], max_new_tokens=2)
    assert ei.value.status == 2

    cancel_req = 5
    results, rids, errors = {}, {}, []
    def client(mine):
        try:
            for i in mine:
                rids[i] = eng.submit(prompts[i], news[i])
                if i == cancel_req:
                    cancel['rid'] = rids[i]
            for i in mine:
                have, st = 0, 0
                while st == 0:
                    st, n = eng.wait(rids[i], have, timeout_ms=20000)
                    assert n > have or st != 0, 'wait timed out'
                    have = n
                results[i] = eng.poll(rids[i])
        except Exception as ex:                  # surfaced by the main thread
            errors.append(ex)
    threads = [threading.Thread(target=client, args=(list(range(k, 9, 3)),)) for k in range(3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(120)
    assert not errors, errors
    assert all(not t.is_alive() for t in threads)
    eng.serve_stop()

    for i in range(9):
        st, toks = results[i]
        if i == cancel_req:
            assert st == 8 and 4 <= len(toks) < news[i]
            upto = len(toks)
        else:
            assert st == 7 and len(toks) == news[i]
            upto = news[i]
        for k in range(upto):
            if toks[k] != ref[i][k]:
                assert margin[i][k] < 5e-2, f'request {i} token {k}: {toks} vs {ref[i]} (margin {margin[i][k]})'
                break
        ev = events[rids[i]]
        ns = [n for _, n in ev]
        assert ns == list(range(1, len(ns) + 1)), f'callback token counts of request {i}: {ns}'
        assert all(s_ == 0 for s_, _ in ev[:-1])
        if i != cancel_req:
            assert ev[-1] == (7, news[i])
        else:
            assert ev[-1][0] == 0 and len(ev) <= len(toks)     # cancel is reported to the caller, not via callback

    # the caller-driven loop continues on the same session after the thread stopped
    rid = eng.submit(prompts[1], 4)
    while eng.poll(rid)[0] == 0:
        eng.step()
    assert len(eng.poll(rid)[1]) == 4
    eng.release()
    eng.close()


def test_engine_logits_processors(cuda):
    """Repetition penalty / bad ids / min_new_tokens + stop ids inside the engine (static and continuous batching).
    Exact: the processed logits of the first generated token equal oracle.logits_process(raw logits of the plain run,
    prompt, ...) bit for bit, and every token is the arg-max of the processed logits the engine reports for its step;
    default parameters reproduce the plain run; the continuous-batching path reproduces the static one.
    Properties: banned ids never appear; with a huge penalty no token of the prompt or the output repeats; stop ids
    cannot appear before min_new_tokens and end the request afterwards."""
    cfg = o.ModelConfig(hidden=256, layers=2, q_heads=4, kv_heads=2, head_dim=128, inter=512, vocab=1024,
                        kv_bits=8, rope=o.RopeParam(128, 500000.0, 'llama3', 8.0, 1.0, 4.0, 8192))
    w = o.make_synthetic_weights(cfg, seed=31)
    rng = np.random.default_rng(6)
    prompts = [rng.integers(1, cfg.vocab, n).astype(np.int32) for n in (12, 70, 33)]
    N = 12
    eng = Engine.from_model_config(cfg, max_batch_size=3, session_len=256, quant_policy=8, max_prefill_token_num=64)
    eng.load_weights(export_weights(cfg, w))
    eng.start()

    def run_static(params):
        """tokens [3][N] and the (processed) logits of every step"""
        eng.set_logits_params(params)
        eng.prefill(prompts, max_new_tokens=N)
        logits = [eng.fetch_logits().copy()]
        for _ in range(N - 1):
            eng.decode(1)
            logits.append(eng.fetch_logits().copy())
        toks = eng.fetch().copy()
        eng.release()
        for b in range(3):
            for k in range(N):
                row = logits[k][b].astype(np.float32)
                assert row[toks[b, k]] == row.max(), (b, k)
        return toks, logits

    base, base_logits = run_static(None)
    same, _ = run_static([dict()] * 3)                                    # processors on, every parameter neutral
    assert np.array_equal(same, base)

    # first-token exactness + properties
    bad = [[int(base[b, 0]), int(base[b, 1])] for b in range(3)]
    params = [dict(repetition_penalty=1.7, bad_ids=bad[0]), dict(repetition_penalty=0.8, bad_ids=bad[1], min_new_tokens=5,
              stop_ids=[int(base[1, 2])]), dict(repetition_penalty=1e4, bad_ids=bad[2])]
    toks, logits = run_static(params)
    for b, pr in enumerate(params):
        end = pr.get('stop_ids', [])
        ref = o.logits_process(base_logits[0][b], prompts[b], pr['repetition_penalty'], pr['bad_ids'], end,
                               k_len=len(prompts[b]), min_len=len(prompts[b]) + pr.get('min_new_tokens', 0) if end else 0)
        assert np.array_equal(logits[0][b].view(np.uint16), ref.view(np.uint16)), b
        assert not set(bad[b]) & set(toks[b].tolist())
    assert int(base[1, 2]) not in toks[1, :4].tolist()                    # banned for the first 4 of min_new_tokens = 5
    seq2 = prompts[2].tolist() + toks[2].tolist()
    assert len(set(toks[2].tolist())) == N and not set(toks[2].tolist()) & set(prompts[2].tolist()), seq2

    # continuous batching = static (one request at a time through one slot keeps the GEMM shapes identical)
    eng1 = Engine.from_model_config(cfg, max_batch_size=1, session_len=256, quant_policy=8, max_prefill_token_num=64)
    eng1.load_weights(export_weights(cfg, w))
    eng1.start()
    for b, pr in enumerate(params):
        eng1.set_logits_params([pr])
        eng1.prefill([prompts[b]], max_new_tokens=N)
        eng1.decode(N - 1)
        want = eng1.fetch()[0].copy()
        eng1.release()
        rid = eng1.submit(prompts[b], N, -1, None, pr)
        while eng1.poll(rid)[0] == 0:
            eng1.step()
        got = eng1.poll(rid)[1]
        assert np.array_equal(got, want[:len(got)]), b                    # the scheduler may stop early on a stop id
        assert len(got) == N or int(got[-1]) in pr.get('stop_ids', []), b
        eng1.release()
    # stop id + min_new_tokens in the scheduler: the plain run's 1st token as stop id ends the request at once ...
    stop = int(base[0, 0])
    rid = eng1.submit(prompts[0], N, -1, None, dict(stop_ids=[stop]))
    while eng1.poll(rid)[0] == 0:
        eng1.step()
    assert eng1.poll(rid)[1].tolist() == [stop]
    # ... unless min_new_tokens bans it: at least 4 tokens, none of the first 3 is the stop id
    rid = eng1.submit(prompts[0], N, stop, None, dict(min_new_tokens=4))
    while eng1.poll(rid)[0] == 0:
        eng1.step()
    out = eng1.poll(rid)[1].tolist()
    assert len(out) >= 4 and stop not in out[:3] and (out[-1] == stop or len(out) == N)
    eng1.release()
    eng1.close()
    eng.close()


def test_pipeline_continuous_generation(cuda):
    """Pipeline: more prompts than batch slots go through the engine scheduler; results keep the prompt order."""
    from lmdeploy_amd import GenerationConfig, TurbomindEngineConfig, pipeline
    pipe = pipeline('synthetic:tiny', backend_config=TurbomindEngineConfig(max_batch_size=2, session_len=128, quant_policy=8))
    rng = np.random.default_rng(0)
    prompts = [rng.integers(0, 512, n).astype(np.int32).tolist() for n in (10, 3, 25, 7, 16)]
    g = GenerationConfig(max_new_tokens=6, ignore_eos=True)
    out = pipe(prompts, g)
    assert [r.index for r in out] == list(range(5)) and all(len(r.token_ids) == 6 for r in out)
    one = [pipe([p], g)[0].token_ids for p in prompts]       # static path, one at a time
    assert [r.token_ids for r in out] == one
    streamed = sorted(pipe.stream_infer(prompts, g, stream_response=False), key=lambda r: r.index)
    assert [r.token_ids for r in streamed] == one
    # token streaming (the reference's default): deltas accumulate to the same outputs, lengths grow monotonically,
    # exactly one final Response per request
    acc, last_len, finals = {i: [] for i in range(5)}, {}, []
    for r in pipe.stream_infer(prompts, g):
        acc[r.index] += r.token_ids
        assert r.generate_token_len == len(acc[r.index]) and (r.generate_token_len > last_len.get(r.index, 0) or r.finish_reason)
        last_len[r.index] = r.generate_token_len
        if r.finish_reason is not None:
            finals.append(r.index)
    assert [acc[i] for i in range(5)] == one and sorted(finals) == list(range(5))
    # stop ids inside the engine: the first request's 3rd token as stop id cuts every output at its first occurrence
    stop = one[0][2]
    gs = GenerationConfig(max_new_tokens=6, ignore_eos=True, stop_token_ids=[stop])
    for r, full in zip(sorted(pipe.stream_infer(prompts, gs, stream_response=False), key=lambda r: r.index), one):
        want = full[:full.index(stop)] if stop in full else full
        assert r.token_ids == want and r.finish_reason == ('stop' if stop in full else 'length')
    pipe.close()


def test_engine_sampling_on_the_collective_path(cuda, monkeypatch):
    """Stochastic sampling at tp > 1: the vocabulary shards of the logits are all-gathered (RCCL) into full rows and every
    rank samples from them.  Driven on one GPU through a 1-rank communicator (TM_FORCE_COMM=1): same seeds -> the tokens of
    the collective-free engine, static and continuous."""
    cfg = o.ModelConfig(hidden=256, layers=2, q_heads=4, kv_heads=2, head_dim=128, inter=512, vocab=1024,
                        kv_bits=8, rope=o.RopeParam(128, 500000.0, 'llama3', 8.0, 1.0, 4.0, 8192))
    w = o.make_synthetic_weights(cfg, seed=23)
    rng = np.random.default_rng(3)
    prompts = [rng.integers(0, cfg.vocab, n).astype(np.int32) for n in (12, 40, 7)]
    params = [(0.9, 30, 0.95, 0.0, 111), None, (1.4, 0, 0.8, 0.02, 222)]

    def run(force):
        if force:
            monkeypatch.setenv('TM_FORCE_COMM', '1')
        else:
            monkeypatch.delenv('TM_FORCE_COMM', raising=False)
        eng = Engine.from_model_config(cfg, max_batch_size=3, session_len=128, quant_policy=8)
        if force:
            eng.comm_init(Engine.comm_unique_id())
        eng.load_weights(export_weights(cfg, w))
        eng.start()
        eng.set_sampling(params)
        eng.prefill(prompts, max_new_tokens=6)
        eng.decode(5)
        toks = eng.fetch().copy()
        eng.release()
        ids = [eng.submit(p_, 6, -1, sp) for p_, sp in zip(prompts, params)]
        while eng.step() != (0, 0):
            pass
        cont = [eng.poll(r)[1].copy() for r in ids]
        eng.close()
        return toks, cont

    t0, c0 = run(False)
    t1, c1 = run(True)
    assert np.array_equal(t0, t1)
    assert all(np.array_equal(a, b) for a, b in zip(c0, c1))
    assert not np.array_equal(t0[0], t0[1])


def test_engine_sampling_static_and_continuous(cuda):
    """Stochastic sampling inside the engine: every generated token must be the oracle's draw (sample_filter +
    sample_draw with the Philox number of (seed, context length)) from the engine's own logits of that step; the same
    seeds give the same tokens through the static path, the continuous-batching path and a second run."""
    cfg = o.ModelConfig(hidden=256, layers=2, q_heads=4, kv_heads=2, head_dim=128, inter=512, vocab=1024,
                        kv_bits=8, rope=o.RopeParam(128, 500000.0, 'llama3', 8.0, 1.0, 4.0, 8192))
    w = o.make_synthetic_weights(cfg, seed=21)
    rng = np.random.default_rng(2)
    prompts = [rng.integers(0, cfg.vocab, n).astype(np.int32) for n in (12, 40, 7)]
    params = [(0.9, 30, 0.95, 0.0, 111), None, (1.4, 0, 0.8, 0.02, 222)]          # row 1 stays greedy
    steps = 6
    eng = Engine.from_model_config(cfg, max_batch_size=3, session_len=128, quant_policy=8)
    eng.load_weights(export_weights(cfg, w))
    eng.start()

    def run_static():
        eng.set_sampling(params)
        eng.prefill(prompts, max_new_tokens=steps)
        logits = [eng.fetch_logits()]
        for _ in range(steps - 1):
            eng.decode(1)
            logits.append(eng.fetch_logits())
        toks = eng.fetch().copy()
        eng.release()
        return toks, logits

    toks, logits = run_static()
    for b, p in enumerate(params):
        for s_ in range(steps):
            row = logits[s_][b]
            if p is None:
                assert toks[b, s_] == int(np.argmax(row.astype(np.float32)))
                continue
            ids, pr = o.sample_filter(row, p[0], p[1], p[2], p[3])
            u = o.philox_uniform(p[4], len(prompts[b]) + s_)
            assert toks[b, s_] == o.sample_draw(ids, pr, u), f'seq {b} step {s_}'
    toks2, _ = run_static()
    assert np.array_equal(toks, toks2)
    assert not np.array_equal(toks[0], toks[1])
    # continuous batching, one request per slot run: same seeds -> same tokens
    ids = [eng.submit(p_, steps, -1, sp) for p_, sp in zip(prompts, params)]
    while eng.step() != (0, 0):
        pass
    for b, rid in enumerate(ids):
        st, t = eng.poll(rid)
        assert st == 7 and np.array_equal(t, toks[b]), f'request {b}: {t} vs {toks[b]}'
    eng.release()
    # greedy again after a sampling session
    eng.prefill(prompts, max_new_tokens=2)
    eng.decode(1)
    g = eng.fetch()
    eng.close()
    assert g[1, 0] == toks[1, 0]


@pytest.mark.parametrize('async_step', [0, 1])
def test_engine_logprobs_static_batch(cuda, monkeypatch, async_step):
    """GenerationConfig.logprobs inside the engine (tm_engine_set_logprobs): for every generated token of a mixed batch -- two
    stochastic sequences and a greedy one -- the record of (sequence, step) must be what the oracle makes of the engine's own logits
    of that step (sample_filter -> sample_logprobs): candidate ids exact, logprobs within 1e-5, the drawn token's logprob; the tokens
    themselves must not change when logprobs are switched on; eager first step, captured graph afterwards; a second batch with another
    count re-captures; the scheduler path (tm_engine_request_logprobs) records the same per request; the pipeline hands the reference's
    dictionaries out, streamed or not."""
    monkeypatch.setenv('TM_ASYNC_STEP', str(async_step))      # the scheduler part: steps retired by the call that issued them / one call later
    cfg = o.ModelConfig(hidden=256, layers=2, q_heads=4, kv_heads=2, head_dim=128, inter=512, vocab=1024,
                        kv_bits=8, rope=o.RopeParam(128, 500000.0, 'llama3', 8.0, 1.0, 4.0, 8192))
    w = o.make_synthetic_weights(cfg, seed=21)
    rng = np.random.default_rng(2)
    prompts = [rng.integers(0, cfg.vocab, n).astype(np.int32) for n in (12, 70, 7)]
    params = [(0.9, 30, 0.95, 0.0, 111), None, (1.4, 0, 0.8, 0.02, 222)]          # row 1 stays greedy
    steps = 6
    eng = Engine.from_model_config(cfg, max_batch_size=3, session_len=128, quant_policy=8, max_prefill_token_num=64)   # two prefill iterations
    eng.load_weights(export_weights(cfg, w))
    eng.start()

    def run(n_lp):
        eng.set_sampling(params)
        eng.set_logprobs(n_lp)
        eng.prefill(prompts, max_new_tokens=steps + 2)
        logits = [eng.fetch_logits()]
        for _ in range(steps - 1):
            eng.decode(1)
            logits.append(eng.fetch_logits())
        toks = eng.fetch().copy()
        rec = eng.fetch_logprobs() if n_lp else None
        eng.release()
        return toks, logits, rec

    plain, _, _ = run(0)
    for n_lp in (4, 1024, 2):
        toks, logits, (vals, idx, num, sel) = run(n_lp)
        assert np.array_equal(toks, plain), 'switching logprobs on changed the tokens'
        assert vals.shape == (3, steps + 2, n_lp) and np.all(num[:, steps:] == 0)
        for b, p in enumerate(params):
            for s_ in range(steps):
                row = logits[s_][b]
                ids, pr = o.sample_filter(row, *(p[:4] if p else (1.0, 1, 1.0, 0.0)))
                e_ids, e_lp, e_sel = o.sample_logprobs(ids, pr, int(toks[b, s_]), n_lp)
                n = len(e_ids)
                assert num[b, s_] == n, f'n={n_lp} seq {b} step {s_}: {num[b, s_]} vs {n}'
                assert np.array_equal(idx[b, s_, :n], e_ids), f'n={n_lp} seq {b} step {s_}'
                fin = np.isfinite(e_lp)
                assert np.array_equal(np.isfinite(vals[b, s_, :n]), fin)
                assert np.all(np.abs(vals[b, s_, :n][fin] - e_lp[fin]) <= 1e-5 + 1e-6 * np.abs(e_lp[fin]))
                assert abs(sel[b, s_] - e_sel) <= 1e-5 + 1e-6 * abs(e_sel)
                if p is None:
                    assert n == 1 and vals[b, s_, 0] == 0.0 and idx[b, s_, 0] == toks[b, s_]
    with pytest.raises(_ffi.TmError):
        eng.set_logprobs(1025)
    # continuous batching: the same requests, each asking for its own count; the per-request records equal the static batch's (same
    # kernels on the same rows); a request without logprobs records nothing; a late request joins a running session
    toks, _, (vals, idx, num, sel) = run(4)
    want_n = [4, 0, 2]
    ids = [eng.submit(p_, steps, -1, sp, None, n_) for p_, sp, n_ in zip(prompts, params, want_n)]
    eng.step()
    late = eng.submit(prompts[0], steps, -1, params[0], None, 3)
    while eng.step() != (0, 0):
        pass
    for b, rid in enumerate(ids + [late]):
        ref_b = b if b < 3 else 0
        n_ = want_n[b] if b < 3 else 3
        st_, t = eng.poll(rid)
        v, ix, nm, sl = eng.poll_logprobs(rid)
        assert st_ == 7 and np.array_equal(t, toks[ref_b, :steps]), f'request {b}'
        if n_ == 0:
            assert v.shape[0] == 0
            continue
        assert v.shape == (steps, n_) and np.array_equal(nm, np.minimum(num[ref_b, :steps], n_)), f'request {b}'
        for s_ in range(steps):
            k = nm[s_]
            assert np.array_equal(ix[s_, :k], idx[ref_b, s_, :k]) and np.all(ix[s_, k:] == -1), f'request {b} step {s_}'
            assert np.allclose(v[s_, :k], vals[ref_b, s_, :k], rtol=0, atol=2e-3), f'request {b} step {s_}'
        assert np.allclose(sl, sel[ref_b, :steps], rtol=0, atol=2e-3)
    with pytest.raises(_ffi.TmError):
        _ffi.check(eng._lib.tm_engine_request_logprobs(eng._h, ids[0], 2))      # not queued any more
    eng.release()
    eng.close()
    # the pipeline surface: greedy -> every token's dictionary is {token: 0.0}; sampling -> the token is in its dictionary, at most n + 1 entries
    import lmdeploy_amd
    pipe = lmdeploy_amd.pipeline('synthetic:tiny', backend_config=lmdeploy_amd.TurbomindEngineConfig(quant_policy=8, session_len=128,
                                                                                                      max_batch_size=2))
    ps = [list(range(5, 40)), list(range(100, 103)), list(range(7, 30))]
    res = pipe(ps, lmdeploy_amd.GenerationConfig(max_new_tokens=5, ignore_eos=True, logprobs=3))
    assert all(r.logprobs == [{t: 0.0} for t in r.token_ids] and r.generate_token_len == 5 for r in res)
    res = pipe(ps, lmdeploy_amd.GenerationConfig(max_new_tokens=5, ignore_eos=True, logprobs=3, do_sample=True, top_k=50, temperature=1.5,
                                                 random_seed=5))
    for r in res:
        assert len(r.logprobs) == 5
        for t, d in zip(r.token_ids, r.logprobs):
            assert t in d and 3 <= len(d) <= 4 and all(v <= 0.0 for v in d.values())
    # streaming through the scheduler: the deltas carry their tokens' dictionaries, together they equal the non-streamed answer
    g5 = lmdeploy_amd.GenerationConfig(max_new_tokens=5, ignore_eos=True, logprobs=3, do_sample=True, top_k=50, temperature=1.5, random_seed=5)
    acc = {}
    for r in pipe.stream_infer(ps, g5):
        assert len(r.logprobs) == len(r.token_ids)
        acc.setdefault(r.index, []).extend(r.logprobs)
    sched = pipe(ps, g5)
    for i, r in enumerate(sched):
        assert len(acc[i]) == 5 and [sorted(d) for d in acc[i]] == [sorted(d) for d in r.logprobs]
    # output_logits='generation': row s of Response.logits is the row token s was drawn from -- greedy: its arg-max; and the tokens are those
    # of the plain run (reading the logits back step by step must not change anything)
    plain = pipe(ps, lmdeploy_amd.GenerationConfig(max_new_tokens=5, ignore_eos=True))
    withl = pipe(ps, lmdeploy_amd.GenerationConfig(max_new_tokens=5, ignore_eos=True, output_logits='generation'))
    for a, b_ in zip(plain, withl):
        assert a.token_ids == b_.token_ids and b_.logits.shape == (5, 1024) and np.isfinite(b_.logits).all()
        assert b_.logits.argmax(-1).tolist() == b_.token_ids
    pipe.close()


@pytest.mark.parametrize('q_heads,kv_heads,kv_bits,hidden', [(6, 1, 8, 384), (8, 1, 4, 512), (12, 2, 16, 256)])
def test_engine_other_config_shapes(cuda, q_heads, kv_heads, kv_bits, hidden):
    """BASELINE.json configs 2-4 scaled down: GQA group 6 with int8 KV (InternLM2-20B: 48 q / 8 kv heads), one kv head
    per rank with group 8 and int4 KV (Llama-3-70B at TP = 8), group 6 with fp16 KV; hidden != q_heads * head_dim as in
    a TP shard.  Same bar as test_engine_matches_oracle."""
    cfg = o.ModelConfig(hidden=hidden, layers=2, q_heads=q_heads, kv_heads=kv_heads, head_dim=128, inter=384, vocab=512,
                        kv_bits=kv_bits, rope=o.RopeParam(128, 1000000.0, 'default', 1.0, 1.0, 4.0, 8192))
    w = o.make_synthetic_weights(cfg, seed=q_heads)
    rng = np.random.default_rng(q_heads)
    prompts = [rng.integers(0, cfg.vocab, n).astype(np.int32) for n in (66, 3)]
    steps = 4
    eng = Engine.from_model_config(cfg, max_batch_size=2, session_len=128, quant_policy=0 if kv_bits == 16 else kv_bits)
    eng.load_weights(export_weights(cfg, w))
    eng.start()
    eng.prefill(prompts, max_new_tokens=steps + 1)
    logits = [eng.fetch_logits()]
    for _ in range(steps):
        eng.decode(1)
        logits.append(eng.fetch_logits())
    toks = eng.fetch()
    eng.close()
    om = o.OracleModel(cfg, w, batch=len(prompts), max_ctx=128)
    ids, lg = om.forward(prompts)
    ref = [lg]
    for s_ in range(steps):
        ids, lg = om.forward([[int(t)] for t in toks[:, s_]])
        ref.append(lg)
    for s_ in range(steps + 1):
        d = np.abs(logits[s_].astype(np.float32) - ref[s_].astype(np.float32))
        assert d.max() <= 3e-2, f'step {s_}: max logit diff {d.max()}'


@pytest.mark.parametrize('fmt', ['fp8', 'u4'])
def test_engine_moe_matches_oracle(cuda, fmt):
    """BASELINE config 5 scaled down: mixture-of-experts decoder (router, top-2 of 4 experts, grouped expert GEMMs,
    combine) with fp8 block-scaled (or AWQ) weights in every linear, int8 KV; prefill (grouped GEMM with row blocks) +
    decode against the oracle model.  Same bar as test_engine_matches_oracle."""
    cfg = o.ModelConfig(hidden=256, layers=2, q_heads=4, kv_heads=2, head_dim=128, inter=256, vocab=512, kv_bits=8,
                        rope=o.RopeParam(128, 1000000.0, 'default', 1.0, 1.0, 4.0, 8192), weight_format=fmt,
                        moe_experts=4, moe_top_k=2, moe_fp8_act=True)   # fp8 experts run on the fp8 matrix cores
    w = o.make_synthetic_weights(cfg, seed=5)
    rng = np.random.default_rng(8)
    prompts = [rng.integers(0, cfg.vocab, n).astype(np.int32) for n in (70, 9, 33)]
    steps = 4
    eng = Engine.from_model_config(cfg, weight_type=2 if fmt == 'fp8' else 0, max_batch_size=3, session_len=128, quant_policy=8)
    eng.load_weights(export_weights(cfg, w))
    eng.start()
    eng.prefill(prompts, max_new_tokens=steps + 1)
    logits = [eng.fetch_logits()]
    for _ in range(steps):
        eng.decode(1)
        logits.append(eng.fetch_logits())
    toks = eng.fetch()
    eng.close()
    om = o.OracleModel(cfg, w, batch=len(prompts), max_ctx=128)
    ids, lg = om.forward(prompts)
    ref = [lg]
    for s_ in range(steps):
        ids, lg = om.forward([[int(t)] for t in toks[:, s_]])
        ref.append(lg)
    for s_ in range(steps + 1):
        d = np.abs(logits[s_].astype(np.float32) - ref[s_].astype(np.float32))
        assert d.max() <= 3e-2, f'step {s_}: max logit diff {d.max()}'


@pytest.mark.parametrize('fmt', ['fp8', 'u4'])
def test_engine_moe_measured_dispatch(cuda, tmp_path, fmt):
    """Measured dispatch for what is not a dense u4 linear: the start-up tuner also times the e4m3
    weight-only attention linears and the fp16 lm_head (general kernel: tiles per wave x split-K) and the row-tile height of the
    grouped expert GEMMs, over the model's own layers; the winners travel as `G` lines of the table file.  Whatever the tuner or
    an imported table selects is the same arithmetic: (a) after tuning at the decode batch and at a 64-token forward the engine
    reproduces the oracle; (b) every row-tile height the grouped kernels have, forced through an imported line, reproduces it too
    on a 64-token prefill forward (the size the entry is keyed by) followed by decode steps."""
    cfg = o.ModelConfig(hidden=256, layers=2, q_heads=4, kv_heads=2, head_dim=128, inter=256, vocab=512, kv_bits=8,
                        rope=o.RopeParam(128, 1000000.0, 'default', 1.0, 1.0, 4.0, 8192), weight_format=fmt,
                        moe_experts=4, moe_top_k=2, moe_fp8_act=True)
    w = o.make_synthetic_weights(cfg, seed=5)
    weights = export_weights(cfg, w)
    rng = np.random.default_rng(8)
    prompts = [rng.integers(0, cfg.vocab, n).astype(np.int32) for n in (40, 9, 15)]      # one 64-token prefill forward

    def run(table_line=None, tune=False):
        eng = Engine.from_model_config(cfg, weight_type=2 if fmt == 'fp8' else 0, max_batch_size=3, session_len=128, quant_policy=8)
        eng.load_weights(weights)
        eng.start()
        path = str(tmp_path / 'table.txt')
        if tune:
            eng.tune_gemm(3, path)
            eng.tune_gemm(64, path)
        elif table_line:
            open(path, 'w').write(table_line + '\n')
            eng.import_gemm_table(path)
        eng.prefill(prompts, max_new_tokens=3)
        lg = [eng.fetch_logits()]
        for _ in range(2):
            eng.decode(1)
            lg.append(eng.fetch_logits())
        toks = eng.fetch()
        eng.close()
        return toks, lg, (open(path).read().splitlines() if tune else [])

    def check(toks, lg, what):
        om2 = o.OracleModel(cfg, w, batch=3, max_ctx=128)
        _, ref = om2.forward(prompts)
        for s_ in range(3):
            d = np.abs(lg[s_].astype(np.float32) - ref.astype(np.float32))
            assert d.max() <= 3e-2, f'{what}: step {s_}: max logit diff {d.max()}'
            if s_ < 2:
                _, ref = om2.forward([[int(t)] for t in toks[:, s_]])

    toks, lg, lines = run(tune=True)
    check(toks, lg, 'tuned')
    g = [ln.split() for ln in lines if ln.startswith('G')]
    assert any(x[1] == '17' and x[2] == '5' and x[5] == '3' for x in g), lines           # the lm_head at the decode batch
    if fmt == 'fp8':     # the dense e4m3 linears (w_qkv, wo: weight-only through the general kernel) at both sizes
        assert {(x[2], x[5]) for x in g if x[1] == '18'} == {('1', '3'), ('2', '3'), ('1', '64'), ('2', '64')}, lines
    kind = 34 if fmt == 'fp8' else 32
    for K, N in ((256, 512), (256, 256)):                                                  # experts' w1w3 and w2
        for rows in ((32, 64) if fmt == 'fp8' else (16, 32, 64)):
            t2, l2, _ = run(table_line=f'G {kind} 0 {K} {N} 64 {rows} 0 0 0')
            check(t2, l2, f'experts K={K} N={N} forced {rows}-row tiles')


def test_engine_errors_are_status_codes(cuda):
    cfg = o.ModelConfig(hidden=256, layers=1, q_heads=2, kv_heads=1, head_dim=128, inter=256, vocab=512)
    eng = Engine.from_model_config(cfg, max_batch_size=2, session_len=128, quant_policy=8)
    with pytest.raises(_ffi.TmError) as ei:
        eng.start()                      # weights not processed
    assert ei.value.status == 1
    eng.init_synthetic(0)
    eng.start()
    with pytest.raises(_ffi.TmError) as ei:
        eng.prefill([np.arange(100, dtype=np.int32)], max_new_tokens=100)   # 200 > session_len
    assert ei.value.status == 6          # TM_TOO_LONG == Request::kTooLong
    eng.prefill([np.arange(10, dtype=np.int32)], max_new_tokens=4)
    eng.decode(3)
    assert eng.fetch().shape == (1, 4)
    with pytest.raises(_ffi.TmError):
        eng.decode(1)                    # past max_new_tokens
    eng.close()


def test_pipeline_surface_on_gpu(cuda):
    """lmdeploy.pipeline()-style call path end to end on synthetic weights (no checkpoints on disk)."""
    import lmdeploy_amd
    pipe = lmdeploy_amd.pipeline('synthetic:tiny', backend_config=lmdeploy_amd.TurbomindEngineConfig(
        quant_policy=8, session_len=256, max_batch_size=2))
    prompts = [list(range(5, 40)), list(range(100, 103)), list(range(7, 90))]      # 3 prompts, batches of 2
    res = pipe(prompts, lmdeploy_amd.GenerationConfig(max_new_tokens=5, ignore_eos=True))
    assert [r.index for r in res] == [0, 1, 2]
    assert all(r.generate_token_len == 5 and r.finish_reason == 'length' for r in res)
    assert [r.input_token_len for r in res] == [35, 3, 83]
    again = pipe(prompts[0], lmdeploy_amd.GenerationConfig(max_new_tokens=5, ignore_eos=True))
    assert again.token_ids == res[0].token_ids            # batch composition must not change a sequence's tokens
    too_long = pipe([list(range(300))], lmdeploy_amd.GenerationConfig(max_new_tokens=5))
    assert too_long[0].finish_reason == 'error' and too_long[0].error_code == 'INPUT_LENGTH_ERROR'
    pipe.close()


def test_two_engines_from_two_threads_of_one_process(cuda):
    """SURVEY 8(b) threading row: the per-rank set-up calls are callable concurrently (the reference drives create_context /
    process_weight / create_engine from one thread per GPU of ONE process, lmdeploy/turbomind/turbomind.py:187-217).  Two engines with
    different weights are created, loaded, started, tuned, driven (prefill + graph-captured decode) and destroyed from two threads at
    the same time, twice over; each must produce exactly the tokens it produces alone (errors / last-error text are per thread,
    the dispatch tables are shared behind a mutex)."""
    import threading
    cfg = o.ModelConfig(hidden=256, layers=2, q_heads=4, kv_heads=2, head_dim=128, inter=512, vocab=1024, kv_bits=8,
                        rope=o.RopeParam(128, 500000.0, 'llama3', 8.0, 1.0, 4.0, 8192))
    ws = [export_weights(cfg, o.make_synthetic_weights(cfg, seed=s)) for s in (31, 32)]
    rng = np.random.default_rng(4)
    prompts = [[rng.integers(0, cfg.vocab, n).astype(np.int32) for n in lens] for lens in ((33, 7), (12, 64, 5))]

    def run(i, out, barrier=None):
        try:
            if barrier:
                barrier.wait()
            eng = Engine.from_model_config(cfg, max_batch_size=4, session_len=128, quant_policy=8, use_graph=1)
            eng.load_weights(ws[i])
            eng.start()
            if barrier:
                barrier.wait()
            eng.tune_gemm(4)
            eng.prefill(prompts[i], max_new_tokens=8)
            eng.decode(7)
            out[i] = eng.fetch().copy()
            if barrier:
                barrier.wait()
            eng.close()
        except Exception as e:      # noqa: BLE001 -- reported by the main thread
            out[i] = e

    alone = [None, None]
    for i in range(2):
        run(i, alone)
        assert not isinstance(alone[i], Exception), alone[i]
    for _ in range(2):
        together = [None, None]
        bar = threading.Barrier(2)
        ts = [threading.Thread(target=run, args=(i, together, bar)) for i in range(2)]
        for t in ts:
            t.start()
        for t in ts:
            t.join(timeout=300)
            assert not t.is_alive(), 'an engine thread hangs'
        for i in range(2):
            assert not isinstance(together[i], Exception), together[i]
            assert np.array_equal(together[i], alone[i]), f'engine {i} driven beside another engine differs from its solo run'
