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


@pytest.mark.parametrize('kv_bits,use_graph', [(8, 1), (4, 0), (16, 1)])
def test_engine_matches_oracle(cuda, kv_bits, use_graph):
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
    for s in range(steps + 1):
        d = np.abs(logits[s].astype(np.float32) - ref_logits[s].astype(np.float32))
        assert d.max() <= 3e-2, f'step {s}: max logit diff {d.max()}'
        top2 = np.sort(ref_logits[s].astype(np.float32), -1)[:, -2:]
        safe = (top2[:, 1] - top2[:, 0]) > 6e-2
        assert np.array_equal(toks[safe, s], ref_toks[s][safe]), f'step {s}: greedy tokens differ'


@pytest.mark.parametrize('graph_comm', [0, 1])
def test_engine_collective_path_single_rank(cuda, monkeypatch, graph_comm):
    """The tp > 1 data path (RCCL all-reduce after wo / w2, vocabulary-sharded lm_head + candidate all-gather), driven
    on one GPU through a 1-rank communicator (TM_FORCE_COMM=1): must reproduce the collective-free engine exactly
    (a 1-rank sum is the identity; the non-deferred split-K reduce rounds the same fp32 sums).  graph_comm=1 also
    captures the RCCL calls into the decode hipGraph."""
    cfg = o.ModelConfig(hidden=256, layers=2, q_heads=4, kv_heads=2, head_dim=128, inter=512, vocab=1024,
                        kv_bits=8, rope=o.RopeParam(128, 500000.0, 'llama3', 8.0, 1.0, 4.0, 8192))
    w = o.make_synthetic_weights(cfg, seed=5)
    rng = np.random.default_rng(1)
    prompts = [rng.integers(0, cfg.vocab, n).astype(np.int32) for n in (33, 64, 7)]

    def run(force):
        if force:
            monkeypatch.setenv('TM_FORCE_COMM', '1')
            monkeypatch.setenv('TM_GRAPH_COMM', str(graph_comm))
        else:
            monkeypatch.delenv('TM_FORCE_COMM', raising=False)
        eng = Engine.from_model_config(cfg, max_batch_size=4, session_len=128, quant_policy=8, use_graph=1)
        if force:
            eng.comm_init(Engine.comm_unique_id())
        eng.load_weights(export_weights(cfg, w))
        eng.start()
        eng.prefill(prompts, max_new_tokens=6)
        eng.decode(5)
        toks, lg = eng.fetch(), eng.fetch_logits()
        eng.close()
        return toks, lg

    t0, l0 = run(False)
    t1, l1 = run(True)
    assert np.array_equal(t0, t1)
    assert np.abs(l0.astype(np.float32) - l1.astype(np.float32)).max() <= 2e-3


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
