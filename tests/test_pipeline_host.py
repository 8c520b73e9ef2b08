"""Host logic of the `pipeline()` surface on the CPU: the Pipeline class driven through a FAKE engine (same method
surface as lmdeploy_amd.turbomind.engine.Engine, deterministic tokens, the scheduler's slot / status behaviour), so that
batching, result ordering, stop handling, streaming deltas and error mapping are covered without a GPU.  The real
engine behind the same calls is covered by tests/test_gpu_engine.py."""
import sys

import numpy as np
import pytest

import lmdeploy_amd  # noqa: F401  (loads lmdeploy_amd.pipeline into sys.modules)
from lmdeploy_amd import GenerationConfig, TurbomindEngineConfig, _ffi

P = sys.modules['lmdeploy_amd.pipeline']      # the module (the package attribute `pipeline` is the factory function)

VOCAB = 1024


def _tok(prompt, k):
    """token k of the continuation of `prompt`: a pure function of the prompt, like greedy decoding"""
    return int((int(np.sum(prompt)) * 31 + 7 * k + 3) % VOCAB)


class FakeEngine:
    instances = []

    def __init__(self, max_batch_size, session_len):
        self.B, self.session_len = max_batch_size, session_len
        self.reqs, self.next_id, self.static = {}, 1, None
        self.sampling, self.logits, self.released, self.logprobs = None, None, 0, 0
        FakeEngine.instances.append(self)

    @classmethod
    def from_model_config(cls, cfg, weight_type=0, **kw):
        return cls(kw['max_batch_size'], kw['session_len'])

    # ---- life cycle -------------------------------------------------------------------------
    def init_synthetic(self, seed=0): pass
    def start(self): pass
    def close(self): pass

    # ---- static batch -----------------------------------------------------------------------
    def set_sampling(self, p): self.sampling = p
    def set_logits_params(self, p): self.logits = p
    def set_logprobs(self, n): self.logprobs = n

    def fetch_logprobs(self):
        """records like Engine.fetch_logprobs: candidate k of step s = (token + k) % VOCAB with logprob -k - 0.5 s (so the generated token is
        the best candidate), except every third step, whose generated token is NOT among the first candidates (its logprob is in sel)"""
        n, B, S = self.logprobs, len(self.static), self.max_new
        vals, idx = np.zeros((B, S, n), np.float32), np.zeros((B, S, n), np.int32)
        num, sel = np.zeros((B, S), np.int32), np.zeros((B, S), np.float32)
        for b, p in enumerate(self.static):
            for s_ in range(self.done):
                t = _tok(p, s_)
                off = 1 if s_ % 3 == 2 else 0
                idx[b, s_] = [(t + k + off) % VOCAB for k in range(n)]
                vals[b, s_] = [-k - 0.5 * s_ for k in range(n)]
                num[b, s_] = max(1, n - (s_ % 2))
                sel[b, s_] = -9.0 if off else vals[b, s_, 0]
        return vals, idx, num, sel

    def prefill(self, prompts, max_new_tokens):
        assert len(prompts) <= self.B
        if any(len(p) + max_new_tokens > self.session_len for p in prompts):
            raise _ffi.TmError(6, 'too long')
        self.static = [list(map(int, p)) for p in prompts]
        self.max_new, self.done = max_new_tokens, 1

    def decode(self, n): self.done += n

    def fetch(self):
        return np.asarray([[_tok(p, k) for k in range(self.done)] for p in self.static], np.int32)

    def fetch_logits(self):
        """[batch, VOCAB] fp16 of the last step: one-hot-ish rows whose arg-max is the token that step produced"""
        out = np.full((len(self.static), VOCAB), -1.0, np.float16)
        for b, p in enumerate(self.static):
            out[b, _tok(p, self.done - 1)] = np.float16(self.done)
        return out

    def release(self):
        self.static, self.reqs, self.released = None, {}, self.released + 1
        self.logprobs = 0          # like tm_engine_release

    # ---- scheduler --------------------------------------------------------------------------
    def submit(self, prompt, max_new, eos_id=-1, sampling=None, logits=None, logprobs=0):
        if len(prompt) + max_new > self.session_len:
            raise _ffi.TmError(6, 'prompt + max_new_tokens exceeds session_len')
        stops = {eos_id} | set((logits or {}).get('stop_ids', []) if isinstance(logits, dict) else [])
        rid, self.next_id = self.next_id, self.next_id + 1
        self.reqs[rid] = dict(prompt=list(map(int, prompt)), max_new=max_new, stops=stops - {-1}, out=[], status=0, slot=None,
                              logits=logits, logprobs=logprobs)
        return rid

    def poll_logprobs(self, rid):
        """the same fake records as fetch_logprobs, per request"""
        r = self.reqs[rid]
        n, T = r['logprobs'], len(r['out'])
        vals, idx = np.zeros((T, n), np.float32), np.zeros((T, n), np.int32)
        num, sel = np.zeros(T, np.int32), np.zeros(T, np.float32)
        for s_, t in enumerate(r['out']):
            off = 1 if s_ % 3 == 2 else 0
            idx[s_] = [(t + k + off) % VOCAB for k in range(n)]
            vals[s_] = [-k - 0.5 * s_ for k in range(n)]
            num[s_] = max(1, n - (s_ % 2))
            sel[s_] = -9.0 if off else vals[s_, 0]
        return vals, idx, num, sel

    def step(self):
        running = [r for r in self.reqs.values() if r['status'] == 0 and r['slot'] is not None]
        for r in self.reqs.values():                       # arrival-order admission into free slots
            if r['status'] == 0 and r['slot'] is None and len(running) < self.B:
                r['slot'] = True
                running.append(r)
        for r in running:
            t = _tok(r['prompt'], len(r['out']))
            r['out'].append(t)
            if t in r['stops'] or len(r['out']) >= r['max_new']:
                r['status'], r['slot'] = 7, None
        act = sum(1 for r in self.reqs.values() if r['status'] == 0 and r['slot'] is not None)
        wait = sum(1 for r in self.reqs.values() if r['status'] == 0 and r['slot'] is None)
        return act, wait

    def poll(self, rid):
        r = self.reqs[rid]
        return r['status'], np.asarray(r['out'], np.int32)

    def cancel(self, rid):
        r = self.reqs[rid]
        if r['status'] == 0:
            r['status'], r['slot'] = 8, None


@pytest.fixture
def pipe(monkeypatch):
    monkeypatch.setattr(P, 'Engine', FakeEngine)
    FakeEngine.instances.clear()
    p = P.Pipeline('synthetic:tiny', backend_config=TurbomindEngineConfig(max_batch_size=2, session_len=64, quant_policy=8))
    yield p
    p.close()


def _expect(prompt, n, stops=()):
    out = []
    for k in range(n):
        t = _tok(prompt, k)
        if t in stops:
            return out, 'stop'
        out.append(t)
    return out, 'length'


def test_static_and_scheduled_batches_agree_and_keep_order(pipe):
    rng = np.random.default_rng(0)
    prompts = [rng.integers(0, VOCAB, n).tolist() for n in (5, 9, 3, 12, 7)]
    g = GenerationConfig(max_new_tokens=6, ignore_eos=True)
    two = pipe(prompts[:2], g)                                   # <= max_batch_size: one static batch
    assert [r.token_ids for r in two] == [_expect(p, 6)[0] for p in prompts[:2]]
    out = pipe(prompts, g)                                       # more prompts than slots: the engine scheduler
    assert [r.index for r in out] == list(range(5))
    assert [r.token_ids for r in out] == [_expect(p, 6)[0] for p in prompts]
    assert all(r.finish_reason == 'length' and r.generate_token_len == 6 and r.input_token_len == len(p)
               for r, p in zip(out, prompts))
    single = pipe(prompts[0], g)                                 # a single token-id prompt returns a single Response
    assert single.token_ids == _expect(prompts[0], 6)[0]
    assert FakeEngine.instances[0].released >= 3                 # every call hands its blocks back


def test_stop_ids_static_scheduled_and_beyond_the_engine_limit(pipe):
    rng = np.random.default_rng(1)
    prompts = [rng.integers(0, VOCAB, n).tolist() for n in (4, 6, 8)]
    stop = _tok(prompts[0], 2)
    g = GenerationConfig(max_new_tokens=8, ignore_eos=True, stop_token_ids=[stop])
    for res in (pipe(prompts[:2], g), pipe(prompts, g)):
        for r, p in zip(res, prompts):
            want, reason = _expect(p, 8, {stop})
            assert (r.token_ids, r.finish_reason) == (want, reason)
            assert stop not in r.token_ids                           # the stop token is never part of the output
    # more stop ids than the engine takes (1 + 8): the pipeline cuts on the host and cancels the request
    many = [stop] + [VOCAB + i for i in range(12)]
    res = pipe(prompts, GenerationConfig(max_new_tokens=8, ignore_eos=True, stop_token_ids=many))
    assert [r.token_ids for r in res] == [_expect(p, 8, {stop})[0] for p in prompts]
    eng = FakeEngine.instances[0]
    assert eng.logits is None or eng.logits == [None]                # static path: no processors were requested


def test_streaming_deltas_and_final_responses(pipe):
    rng = np.random.default_rng(2)
    prompts = [rng.integers(0, VOCAB, n).tolist() for n in (3, 5, 7, 9)]
    g = GenerationConfig(max_new_tokens=5, ignore_eos=True)
    acc, finals, steps_seen = {i: [] for i in range(4)}, [], 0
    for r in pipe.stream_infer(prompts, g):
        steps_seen += 1
        acc[r.index] += r.token_ids
        assert r.generate_token_len == len(acc[r.index])
        if r.finish_reason is not None:
            finals.append(r.index)
    assert [acc[i] for i in range(4)] == [_expect(p, 5)[0] for p in prompts]
    assert sorted(finals) == [0, 1, 2, 3] and steps_seen == 4 * 5         # one Response per request per step
    assert finals[:2] == [0, 1]                                          # two slots: the first two finish first
    whole = sorted(pipe.stream_infer(prompts, g, stream_response=False), key=lambda r: r.index)
    assert [r.token_ids for r in whole] == [acc[i] for i in range(4)]


def test_errors_become_responses_and_processors_reach_the_engine(pipe):
    g = GenerationConfig(max_new_tokens=8, ignore_eos=True, repetition_penalty=1.2, min_new_tokens=2, bad_token_ids=[5])
    ok, too_long, tight = list(range(10)), list(range(64)), list(range(60))   # 64 >= session_len 64; 60 + 8 > 64: room for 4
    res = pipe([ok, too_long, ok], g)
    assert [r.index for r in res] == [0, 1, 2]
    assert res[1].finish_reason == 'error' and res[1].error_code == 'INPUT_LENGTH_ERROR' and res[1].token_ids == []
    assert res[0].token_ids == res[2].token_ids == _expect(ok, 8)[0]
    # the static-batch path (<= max_batch_size prompts) answers the offender alone as well: the others still run
    res2 = pipe([too_long, ok], g)
    assert [r.index for r in res2] == [0, 1]
    assert res2[0].finish_reason == 'error' and res2[0].error_code == 'INPUT_LENGTH_ERROR' and res2[0].token_ids == []
    assert res2[1].finish_reason == 'length' and res2[1].token_ids == _expect(ok, 8)[0]
    # a prompt that fits but lies within max_new_tokens of the session end is NOT refused (the reference rejects only
    # input_len >= session_len, async_engine.py:561-565): it generates until the session is full and ends with 'length'
    for res3 in (pipe([tight, ok], g), pipe([ok, tight, ok], g)):
        r = [x for x in res3 if x.input_token_len == 60][0]
        assert r.finish_reason == 'length' and r.token_ids == _expect(tight, 4)[0]
        assert all(x.token_ids == _expect(ok, 8)[0] for x in res3 if x.input_token_len == 10)
    lp = g.logits_params([])
    # greedy (do_sample=False): the penalty is reset to 1.0 like the reference does (async_engine.py:424-430) ...
    assert lp == dict(repetition_penalty=pytest.approx(1.0), min_new_tokens=2, bad_ids=[5], stop_ids=[])
    assert GenerationConfig(repetition_penalty=1.2).logits_params([]) is None
    # ... and reaches the engine only for sampling requests
    assert GenerationConfig(do_sample=True, repetition_penalty=1.2).logits_params([7])['repetition_penalty'] == pytest.approx(1.2)
    # the scheduler path hands the processors of the GenerationConfig to the engine with every request ...
    eng = FakeEngine.instances[0]
    seen = []
    real_submit = eng.submit
    eng.submit = lambda *a, **k: (seen.append(a[4] if len(a) > 4 else k.get('logits')), real_submit(*a, **k))[1]
    pipe([ok, ok, ok], g)
    assert seen == [lp, lp, lp]
    pipe([ok], g)                                                           # ... the static path through set_logits_params
    assert eng.logits == [lp]
    with pytest.raises(NotImplementedError):
        P.Pipeline('synthetic:tiny', backend_config=TurbomindEngineConfig(), chat_template_config=object())


def test_logprobs_reach_the_responses(pipe):
    """GenerationConfig.logprobs: the request runs as a static batch (also with more prompts than slots), the engine is armed with
    set_logprobs(n), and every generated token gets the reference's dictionary (turbomind.py:472-503): the first min(num, n) candidates,
    plus the generated token itself when it is not among them; the scheduler path (more prompts than slots, per-request configs,
    streaming deltas) hands out the same dictionaries from the per-request records."""
    prompts = [[1, 2, 3], [9, 9], [4, 5, 6, 7]]
    res = pipe(prompts, GenerationConfig(max_new_tokens=6, logprobs=3, ignore_eos=True))
    eng = FakeEngine.instances[-1]
    assert [r.index for r in res] == [0, 1, 2]
    for r, p in zip(res, prompts):
        assert len(r.logprobs) == r.generate_token_len == 6
        for s_, (tok, d) in enumerate(zip(r.token_ids, r.logprobs)):
            n = max(1, 3 - (s_ % 2))
            if s_ % 3 == 2:
                assert d == {**{(tok + k + 1) % VOCAB: -k - 0.5 * s_ for k in range(n)}, tok: -9.0}
            else:
                assert d == {(tok + k) % VOCAB: -k - 0.5 * s_ for k in range(n)}
    plain = pipe(prompts, GenerationConfig(max_new_tokens=6, ignore_eos=True))
    assert eng.logprobs == 0 and all(r.logprobs is None for r in plain)
    assert [r.token_ids for r in plain] == [r.token_ids for r in res]

    def want(tok, s_, n_req):
        n = max(1, n_req - (s_ % 2))
        d = {(tok + k + (1 if s_ % 3 == 2 else 0)) % VOCAB: -k - 0.5 * s_ for k in range(n)}
        if s_ % 3 == 2:
            d[tok] = -9.0
        return d
    # the scheduler path (more prompts than slots -> continuous batching; per-request configs; streaming deltas)
    many = prompts + [[8, 8, 8]]
    sched = pipe(many, GenerationConfig(max_new_tokens=5, logprobs=2, ignore_eos=True))
    for r in sched:
        assert r.logprobs == [want(t, s_, 2) for s_, t in enumerate(r.token_ids)] and len(r.token_ids) == 5
    per = pipe(prompts, [GenerationConfig(max_new_tokens=4, logprobs=2, ignore_eos=True), GenerationConfig(max_new_tokens=4, ignore_eos=True),
                         GenerationConfig(max_new_tokens=4, logprobs=3, ignore_eos=True)])
    assert per[1].logprobs is None
    assert per[0].logprobs == [want(t, s_, 2) for s_, t in enumerate(per[0].token_ids)]
    assert per[2].logprobs == [want(t, s_, 3) for s_, t in enumerate(per[2].token_ids)]
    got = {}
    for r in pipe.stream_infer(prompts, GenerationConfig(max_new_tokens=4, logprobs=2, ignore_eos=True)):
        assert len(r.logprobs) == len(r.token_ids)
        got.setdefault(r.index, []).extend(zip(r.token_ids, r.logprobs))
    for i in range(3):
        assert [d for _, d in got[i]] == [want(t, s_, 2) for s_, (t, _) in enumerate(got[i])] and len(got[i]) == 4
    with pytest.warns(UserWarning):
        assert GenerationConfig(logprobs=5000).logprobs == 1024


def test_output_logits_of_the_generated_tokens(pipe):
    """GenerationConfig.output_logits='generation': Response.logits[s] = the logits token s was drawn from (one decode step per engine call,
    read back behind every step); more prompts than slots still run as static chunks; streaming / per-request configs / 'all' refuse."""
    prompts = [[1, 2, 3], [9, 9], [4, 5, 6, 7]]
    res = pipe(prompts, GenerationConfig(max_new_tokens=5, output_logits='generation', ignore_eos=True))
    for r in res:
        assert r.logits.shape == (5, VOCAB) and r.logits.dtype == np.float32
        assert r.logits.argmax(-1).tolist() == r.token_ids and r.logits.max(-1).tolist() == [1.0, 2.0, 3.0, 4.0, 5.0]
    assert all(r.logits is None for r in pipe(prompts[:2], GenerationConfig(max_new_tokens=3, ignore_eos=True)))
    with pytest.raises(NotImplementedError):
        list(pipe.stream_infer(prompts, GenerationConfig(max_new_tokens=3, output_logits='generation')))
    with pytest.raises(NotImplementedError):
        pipe(prompts[:2], GenerationConfig(max_new_tokens=3, output_logits='generation', bad_token_ids=[5]))
    with pytest.raises(NotImplementedError):
        GenerationConfig(output_logits='all')


def test_one_generation_config_per_prompt(pipe):
    """infer(prompts, [gen_config, ...]) (lmdeploy/pipeline.py:97-143): every request runs with its own limits / stops."""
    rng = np.random.default_rng(5)
    prompts = [rng.integers(0, VOCAB, n).tolist() for n in (4, 6, 8)]
    stop = _tok(prompts[1], 1)
    gs = [GenerationConfig(max_new_tokens=3, ignore_eos=True), GenerationConfig(max_new_tokens=9, ignore_eos=True, stop_token_ids=[stop]),
          GenerationConfig(max_new_tokens=5, ignore_eos=True)]
    res = pipe(prompts, gs)
    assert [r.index for r in res] == [0, 1, 2]
    assert res[0].token_ids == _expect(prompts[0], 3)[0] and res[2].token_ids == _expect(prompts[2], 5)[0]
    assert (res[1].token_ids, res[1].finish_reason) == _expect(prompts[1], 9, {stop})
    with pytest.raises(ValueError):
        pipe(prompts, gs[:2])


# ---- pipeline(tp = N) as ONE call: the rank group's host protocol (lmdeploy_amd/turbomind/tp_group.py) ---------------------------------
class _StubRank:
    """the engine surface the group drives, recording every call; rank 1's RCCL bring-up can be made to fail"""

    def __init__(self, rank, rccl_fails=False):
        self.rank, self.rccl_fails, self.calls, self.handles = rank, rccl_fails, [], None

    def comm_init(self, uid):
        self.calls.append(('comm_init', len(uid)))
        if self.rccl_fails:
            raise _ffi.TmError(5, 'ncclCommInitRank: Duplicate GPU detected')

    def comm_drop_rccl(self):
        self.calls.append(('comm_drop_rccl',))

    def comm_native_setup(self, all_gather, rows):
        self.handles = all_gather(b'handle-of-rank-%d' % self.rank)
        self.calls.append(('comm_native_setup', rows))

    def comm_native_selftest(self):
        self.calls.append(('comm_native_selftest',))
        return not getattr(self, 'selftest_fails', False)

    def comm_native_drop(self):
        self.calls.append(('comm_native_drop',))

    def prefill(self, prompts, max_new_tokens):
        self.calls.append(('prefill', [list(map(int, p)) for p in prompts], max_new_tokens))

    def decode(self, n):
        self.calls.append(('decode', n))
        if n == 13 and self.rank == 1:
            raise _ffi.TmError(6, 'rank 1 says no')

    def submit(self, prompt, max_new, eos=-1, sampling=None, logits=None):
        self.calls.append(('submit', list(map(int, prompt)), max_new, eos, sampling, logits))
        return 40 + len(self.calls)

    def step(self):
        self.calls.append(('step',))
        return (1, 0)

    def step_many(self, n):
        self.calls.append(('step_many', n))
        return (n, 1, 0)

    def set_sampling(self, params):
        self.calls.append(('set_sampling', params))
        if params == 'bad-on-rank-0' and self.rank == 0:
            raise ValueError('rank 0 cannot convert this')      # a Python-side failure of rank 0's OWN call

    def serve_start(self, on_update=None):
        self.calls.append(('serve_start',))

    def stats(self):
        return dict(rank=self.rank)

    def release(self):
        self.calls.append(('release',))

    def poll(self, rid):
        return 0, np.asarray([self.rank], np.int32)

    def close(self):
        self.calls.append(('close',))


@pytest.mark.parametrize('rccl_fails_on,want_native', [(None, False), (1, False), (2, False), (None, True), (2, True), ('selftest', False),
                                                       ('rccl-only', False)])
def test_tp_group_protocol_with_stub_engines(rccl_fails_on, want_native, monkeypatch):
    """three ranks: rank 0 = ParentLink + TpEngine in this thread, ranks 1 / 2 = WorkerLink.serve on threads over real duplex pipes.
    Every rank must take the SAME communicator branch (RCCL only when every rank's init succeeded, else all drop it and exchange
    the native communicator's handles in rank order), see the same mirrored calls in the same order, and a worker's error must
    surface in the caller with its status code after every rank answered.  want_native = TurbomindEngineConfig.communicator 'native' /
    'cuda-ipc': the native communicator comes up ON TOP of a healthy RCCL (decode-sized exchanges native, prefill-sized RCCL); rank 0's
    choice reaches the workers with the unique id -- a worker never decides for itself."""
    import multiprocessing as mp
    import threading
    from lmdeploy_amd.turbomind import tp_group
    tp = 3
    pipes = [mp.Pipe(duplex=True) for _ in range(tp - 1)]
    # round 6: with RCCL up on every rank the DEFAULT is RCCL (large forwards) + the native fused decode collective, taken when the
    # native communicator's bring-up self-test passes on EVERY rank; 'selftest': rank 1's fails -> all ranks drop it together;
    # 'rccl-only': TM_COMM=rccl keeps RCCL alone
    selftest_case, rccl_only = rccl_fails_on == 'selftest', rccl_fails_on == 'rccl-only'
    if selftest_case or rccl_only:
        rccl_fails_on = None
    if rccl_only:
        monkeypatch.setenv('TM_COMM', 'rccl')
    else:
        monkeypatch.delenv('TM_COMM', raising=False)
    ranks = [_StubRank(r, rccl_fails=(r == rccl_fails_on)) for r in range(tp)]
    if selftest_case:
        ranks[1].selftest_fails = True
    backends = [None] * tp

    def worker(r):
        link = tp_group.WorkerLink(pipes[r - 1][1])
        backends[r] = link.setup_comm(ranks[r], True, 0, not want_native)    # the worker's own arguments are ignored: rank 0's travel
        pipes[r - 1][1].send(('ready', r))
        link.serve(ranks[r])

    threads = [threading.Thread(target=worker, args=(r,), daemon=True) for r in range(1, tp)]
    for t in threads:
        t.start()
    link = tp_group.ParentLink(tp, 'unused', None, conns=[p[0] for p in pipes])
    backends[0] = link.setup_comm(ranks[0], True, 64, want_native)
    link.wait_ready()
    eng = tp_group.TpEngine(ranks[0], link, backends[0])
    want = ('native-p2p' if rccl_fails_on is not None else 'rccl' if rccl_only else
            'rccl (native communicator failed its bring-up self-test)' if selftest_case else 'native-p2p (decode) + rccl (large forwards)')
    assert backends == [want] * tp
    for r in ranks:
        if rccl_only:
            assert [c[0] for c in r.calls] == ['comm_init']
        elif selftest_case:             # one rank's self-test failed: EVERY rank drops the native communicator
            assert [c[0] for c in r.calls] == ['comm_init', 'comm_native_setup', 'comm_native_selftest', 'comm_native_drop']
        elif rccl_fails_on is None:     # both communicators: RCCL kept, the native one's handles exchanged in rank order, self-test run
            assert [c[0] for c in r.calls] == ['comm_init', 'comm_native_setup', 'comm_native_selftest'] and ('comm_native_setup', 64) in r.calls
            assert r.handles == [b'handle-of-rank-%d' % q for q in range(tp)]
        else:
            assert r.handles == [b'handle-of-rank-%d' % q for q in range(tp)], 'handles must arrive in rank order on every rank'
            assert ('comm_native_setup', 64) in r.calls, 'rows of the parent reach every rank'
            assert (('comm_drop_rccl',) in r.calls) == (r.rank != rccl_fails_on), 'exactly the ranks whose RCCL came up drop it'
    n0 = [len(r.calls) for r in ranks]
    eng.prefill([np.asarray([1, 2, 3], np.int32), [4, 5]], max_new_tokens=7)
    eng.decode(6)
    rid = eng.submit(np.asarray([9, 8], np.int32), 5, 2, (0.8, 40, 0.9, 0.0, 123), dict(repetition_penalty=1.1, min_new_tokens=0, bad_ids=[3], stop_ids=[]))
    assert eng.step() == (1, 0)
    st, toks = eng.poll(rid)                     # a read: rank 0 alone
    assert toks.tolist() == [0]
    with pytest.raises(_ffi.TmError) as ei:      # rank 1 refuses; the others ran the call; the caller sees rank 1's status
        eng.decode(13)
    assert ei.value.status == 6 and 'rank 1' in str(ei.value)
    # a NON-TmError failure of rank 0's own call (argument conversion) must still drain the workers' replies -- the next
    # mirrored call has to read ITS replies, not the failed call's
    with pytest.raises(ValueError):
        eng.set_sampling('bad-on-rank-0')
    assert eng.step_many(8) == (8, 1, 0), 'the burst call is mirrored like a single step and returns rank 0\'s result'
    assert eng.step() == (1, 0)
    # whatever is neither mirrored nor a rank-0 read would run a forward / change state on rank 0 alone: refused loudly
    with pytest.raises(NotImplementedError):
        eng.serve_start()
    assert eng.stats() == dict(rank=0)          # a read: rank 0 alone, no message to the workers
    eng.release()
    mirrored = [r.calls[n:] for r, n in zip(ranks, n0)]
    assert mirrored[0] == mirrored[1] == mirrored[2], 'every rank must see the same calls in the same order'
    assert [c[0] for c in mirrored[0]] == ['prefill', 'decode', 'submit', 'step', 'decode', 'set_sampling', 'step_many', 'step', 'release']
    eng.close()
    for t in threads:
        t.join(timeout=10)
        assert not t.is_alive(), 'workers must leave their serve loop when the group closes'


def test_tp_group_host_cost_of_a_mirrored_step_and_of_a_burst():
    """what a mirrored scheduler step costs on the host (real worker processes over the product's authenticated
    localhost connections, stub engines), and that `step_many` -- the call the pipeline mirrors at tp > 1 -- amortises it: one round
    trip per burst of 8 steps.  Numbers of this container: profiles/r06_tp_group_host_cost.txt (55 us .. 1.3 ms per step at tp = 2,
    2.6 .. 4.8 ms at tp = 8 on 8 shared CPUs -- far above the 50 us bar, hence the burst)."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location('tp_group_host_cost', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                                     'tools', 'tp_group_host_cost.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    us, us_burst = mod.measure(2, 160)
    assert us > 0 and us_burst > 0
    assert us_burst < 0.6 * us, (us, us_burst)       # ~1/8 when the host is quiet; the bound leaves room for a noisy one
