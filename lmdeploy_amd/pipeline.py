"""`lmdeploy.pipeline()` surface over the MI355X engine.

Reference: lmdeploy/api.py:15-82 (pipeline), lmdeploy/pipeline.py:33-183,317-327 (Pipeline.infer / __call__ /
stream_infer / close).  The reference routes through AsyncEngine + its continuous-batching scheduler; here up to
`max_batch_size` prompts run as one static batch (chunked prefill + decode, the configuration BASELINE.json's metric is
quoted on), more prompts -- and every stream_infer call -- go through the engine's scheduler (tm_engine_submit / step /
poll: requests leave the batch when they stop, waiting ones take their slots).
"""
from __future__ import annotations

import os
from typing import Sequence

import numpy as np

from . import _ffi
from .messages import STATUS_TO_RESPONSE, GenerationConfig, Response, ResponseType, TurbomindEngineConfig
from .turbomind import checkpoint
from .turbomind.engine import Engine
from .turbomind.loader import export_weights

# synthetic model shapes (no checkpoints on disk in CI): "synthetic:<name>"
SYNTHETIC = {
    'llama3_8b': dict(hidden=4096, layers=32, q_heads=32, kv_heads=8, head_dim=128, inter=14336, vocab=128256,
                      rope=checkpoint.RopeConfig(128, 500000.0, 'llama3', 8.0, 1.0, 4.0, 8192)),
    'internlm2_1_8b': dict(hidden=2048, layers=24, q_heads=16, kv_heads=8, head_dim=128, inter=8192, vocab=92544,
                           rope=checkpoint.RopeConfig(128, 1000000.0)),
    'internlm2_20b': dict(hidden=6144, layers=48, q_heads=48, kv_heads=8, head_dim=128, inter=16384, vocab=92544,
                          rope=checkpoint.RopeConfig(128, 1000000.0)),
    'llama3_70b': dict(hidden=8192, layers=80, q_heads=64, kv_heads=8, head_dim=128, inter=28672, vocab=128256,
                       rope=checkpoint.RopeConfig(128, 500000.0, 'llama3', 8.0, 1.0, 4.0, 8192)),
    'tiny': dict(hidden=256, layers=2, q_heads=4, kv_heads=2, head_dim=128, inter=512, vocab=1024,
                 rope=checkpoint.RopeConfig(128, 10000.0)),
}


def logprobs_of_token(vals, idx, num: int, sel: float, token: int, topn: int) -> dict:
    """One generated token's entry of Response.logprobs, the reference's rule (lmdeploy/turbomind/turbomind.py:472-503): the first
    min(num, topn) kept candidates as {token id: logprob}, the generated token itself added when it is not among them, -inf dropped."""
    n = min(num, topn)
    res = {int(i): float(v) for i, v in zip(idx[:n], vals[:n])}
    if token not in res:
        res[int(token)] = sel
    return {k: v for k, v in res.items() if v != float('-inf')}


class Pipeline:

    def __init__(self, model_path: str, backend_config: TurbomindEngineConfig | None = None, rank: int = 0,
                 comm_unique_id: bytes | None = None, _tp_link=None, **kwargs):
        """tp > 1, the reference's way (lmdeploy/turbomind/turbomind.py:187-217): ONE call -- this process becomes rank 0 and starts
        the other tp - 1 ranks of the node itself (one process per GPU, turbomind/tp_group.py), broadcasts every request to them and
        returns rank 0's responses.  Expert path, unchanged: an externally launched process per GPU (torch.distributed.run, ...)
        passes its `rank` and the broadcast RCCL id (`comm_unique_id`); `_tp_link` is the worker side of the one-call form."""
        if kwargs.get('speculative_config') is not None or kwargs.get('chat_template_config') is not None:
            raise NotImplementedError('speculative decoding / chat templates are outside the MI355X hot path')
        cfg = backend_config or TurbomindEngineConfig()
        if not isinstance(cfg, TurbomindEngineConfig):
            raise NotImplementedError('only TurbomindEngineConfig is supported (the PyTorch engine is out of scope)')
        self.backend_config = cfg
        self.tokenizer = None
        synthetic = model_path.startswith('synthetic:')
        if synthetic:
            self.model_cfg = checkpoint.ModelConfig(**SYNTHETIC[model_path.split(':', 1)[1]],
                                                    quantized=cfg.model_format != 'hf',
                                                    weight_format={'hf': 'f16', 'fp8': 'fp8'}.get(cfg.model_format, 'u4'))
        else:
            self.model_cfg = checkpoint.read_config(model_path)
            have = self.model_cfg.weight_format if self.model_cfg.quantized else 'f16'
            if cfg.model_format in ('awq', 'fp8') and {'awq': 'u4', 'fp8': 'fp8'}[cfg.model_format] != have:
                raise ValueError(f'model_format="{cfg.model_format}" but the checkpoint\'s quantization_config says {have} '
                                 f'(lmdeploy/turbomind/converter.py:174-176)')
        session_len = cfg.session_len or self.model_cfg.max_position_embeddings
        self.session_len = int(session_len)
        devices = cfg.devices or list(range(cfg.tp))
        per_dev = max([sum(1 for r in range(cfg.tp) if devices[r % len(devices)] == d) for d in set(devices)] or [1])
        if per_dev > 1:     # ranks share a device (bring-up on a smaller box): their persistent two-shot all-reduce grids must fit TOGETHER
            os.environ.setdefault('TM_P2P_2SHOT_GRID', str(max(8, 256 // per_dev // 2)))     # (read by this rank's engine and inherited by the workers)
        self.engine = Engine.from_model_config(
            self.model_cfg, weight_type={'u4': 0, 'f16': 1, 'fp8': 2}[self.model_cfg.weight_format if self.model_cfg.quantized
                                                                        else 'f16'], tp=cfg.tp, rank=rank,
            device=devices[rank % len(devices)], max_batch_size=cfg.max_batch_size or 64, session_len=session_len,
            quant_policy=int(cfg.quant_policy), cache_max_entry_count=cfg.cache_max_entry_count,
            max_prefill_token_num=cfg.max_prefill_token_num or 8192)
        self._group = None
        if cfg.tp > 1:
            rows = cfg.max_batch_size or 64
            # ranks that share a device (a 1-GPU box driving tp = 2 for bring-up): RCCL refuses duplicate devices -> native P2P communicator
            want_rccl = len(set(devices[r % len(devices)] for r in range(cfg.tp))) == cfg.tp
            want_native = cfg.communicator in ('native', 'cuda-ipc')    # the reference's names for its in-house communicator
            if comm_unique_id is not None:              # expert path: the caller launched one process per GPU and broadcast the id
                if want_native:
                    raise ValueError(f'communicator={cfg.communicator!r} needs the one-call form (the ranks exchange IPC handles through '
                                     f'the rank group); externally launched ranks run on RCCL, or call Engine.comm_native_setup themselves')
                self.engine.comm_init(comm_unique_id)
            elif _tp_link is not None:                  # a worker of the one-call form
                _tp_link.setup_comm(self.engine, want_rccl, rows, want_native)
            else:                                       # the one-call form: this process is rank 0 and owns the other ranks
                if rank != 0:
                    raise ValueError('tp > 1 with rank != 0 needs comm_unique_id (externally launched ranks)')
                from .turbomind import tp_group
                self._group = tp_group.ParentLink(cfg.tp, model_path, cfg)
                try:
                    self._comm_backend = self._group.setup_comm(self.engine, want_rccl, rows, want_native)
                except Exception:
                    self._group.close()
                    raise
        try:
            if synthetic:
                self.engine.init_synthetic(seed=0)
            else:
                w = checkpoint.load_hf_weights(model_path, self.model_cfg)
                self.engine.load_weights(export_weights(self.model_cfg, w, cfg.tp, rank))
                if os.path.exists(os.path.join(model_path, 'tokenizer.json')) or \
                        os.path.exists(os.path.join(model_path, 'tokenizer.model')):
                    from transformers import AutoTokenizer
                    self.tokenizer = AutoTokenizer.from_pretrained(model_path)
            self.engine.start()
            self.max_batch_size = cfg.max_batch_size or 64
            if self._group is not None:
                from .turbomind import tp_group
                self._group.wait_ready()
                self.engine = tp_group.TpEngine(self.engine, self._group, self._comm_backend)
        except BaseException:      # weights / tokenizer / start / a worker that failed to come up: the workers (and their GPU memory) must not
            if self._group is not None:     # outlive the failed constructor
                self._group.close()
            raise

    # ---- reference-compatible entry points -----------------------------------------------------------
    def __call__(self, prompts, gen_config: GenerationConfig | None = None, **kwargs):
        return self.infer(prompts, gen_config, **kwargs)

    def infer(self, prompts, gen_config: GenerationConfig | None = None, **kwargs) -> list[Response] | Response:
        single = isinstance(prompts, str) or (len(prompts) > 0 and isinstance(prompts[0], (int, np.integer)))
        batch = [prompts] if single else list(prompts)
        out = []
        for res in self._generate(batch, gen_config or GenerationConfig()):
            out.append(res)
        return out[0] if single else out

    def stream_infer(self, prompts, gen_config: GenerationConfig | None = None, stream_response: bool = True, **kwargs):
        """Streaming generation through the engine scheduler (lmdeploy/pipeline.py:145-178).  stream_response=True
        (the reference's default): after every scheduler step one Response per request that produced tokens, carrying the
        NEW token ids (`token_ids`), the total generated so far (`generate_token_len`) and `finish_reason` None until the
        request ends ('stop' | 'length'); the stop token itself is never part of the output.  stream_response=False: one
        final Response per request, in completion order."""
        yield from self.generate_continuous(list(prompts), gen_config or GenerationConfig(), stream=stream_response)

    def close(self):
        self.engine.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- static batcher ----------------------------------------------------------------------------------
    def _encode(self, p) -> np.ndarray:
        if isinstance(p, str):
            if self.tokenizer is None:
                raise ValueError('string prompts need a tokenizer in the model directory; pass token ids instead')
            return np.asarray(self.tokenizer.encode(p), np.int32)
        return np.asarray(p, np.int32)

    def _stop_ids(self, g: GenerationConfig) -> set:
        ids = set(g.stop_token_ids or [])
        if not g.ignore_eos and self.model_cfg.eos_token_id is not None:
            e = self.model_cfg.eos_token_id
            ids |= set(e if isinstance(e, (list, tuple)) else [e])
        return ids

    def _resolve_words(self, g: GenerationConfig):
        """stop_words / bad_words need the tokenizer (lmdeploy/serve/core/async_engine.py converts them per request)"""
        if g.stop_words or g.bad_words:
            if self.tokenizer is None:
                raise ValueError('stop_words / bad_words need a tokenizer in the model directory; pass stop_token_ids / '
                                 'bad_token_ids instead')
            g.convert_stop_bad_words_to_ids(self.tokenizer)
        if g.bad_token_ids and len(g.bad_token_ids) > _ffi.MAX_BAD_IDS:
            raise ValueError(f'at most {_ffi.MAX_BAD_IDS} bad token ids per request')

    def _generate(self, prompts: Sequence, g):
        if isinstance(g, (list, tuple)):            # one GenerationConfig per prompt (lmdeploy/pipeline.py:97-143)
            yield from sorted(self.generate_continuous(prompts, g), key=lambda r: r.index)
            return
        self._resolve_words(g)
        if g.output_logits:                         # per-step logits are read back from the static batch (Engine.fetch_logits)
            yield from self._generate_static(prompts, g)
            return
        if len(prompts) > self.max_batch_size:      # more work than batch slots: let the engine schedule it
            yield from sorted(self.generate_continuous(prompts, g), key=lambda r: r.index)
            return
        yield from self._generate_static(prompts, g)

    def generate_continuous(self, prompts: Sequence, g: GenerationConfig | None = None, stream: bool = False):
        """Continuous batching (engine scheduler: tm_engine_submit / step / poll): any number of prompts, each request
        leaves the batch when it stops and the next waiting one takes its slot.  Yields Responses in completion order
        (stream=False) or incremental Responses after every scheduler step (stream=True, see stream_infer)."""
        ids = [self._encode(p) for p in prompts]
        if isinstance(g, (list, tuple)):            # per-request generation configs: every request carries its own
            if len(g) != len(ids):
                raise ValueError(f'{len(g)} generation configs for {len(ids)} prompts')
            gs = [gi or GenerationConfig() for gi in g]
        else:
            gs = [g or GenerationConfig()] * len(ids)
        if any(gi.output_logits for gi in gs):
            raise NotImplementedError('GenerationConfig.output_logits with streaming or per-request generation configs: the per-step logits are read '
                                      'back from the static batch path (one GenerationConfig for all prompts, infer / __call__)')
        stops = []
        pending, out_of_engine, sent = {}, [], {}
        for i, p in enumerate(ids):
            gi = gs[i]
            self._resolve_words(gi)
            stop = self._stop_ids(gi)
            stops.append(stop)
            # stop ids live inside the engine (eos id + up to 8 more); beyond that the loop below cuts on the host
            stop_l = sorted(stop)
            in_engine = stop_l if len(stop_l) <= 1 + _ffi.MAX_STOP_IDS else []
            eos = in_engine[0] if in_engine else -1
            lp = gi.logits_params(in_engine[1:])
            if lp is None and len(in_engine) > 1:
                lp = dict(stop_ids=in_engine[1:])
            try:
                # the reference clamps per request: generate until the session is full, then 'length' (async_engine.py:561-565)
                room = self.session_len - len(p)
                if len(p) < 1 or room < 1:
                    raise _ffi.TmError(6, 'empty prompt' if len(p) < 1 else
                                       f'prompt ({len(p)}) leaves no room for a new token in session_len ({self.session_len})')
                pending[self.engine.submit(p, min(gi.max_new_tokens, room), eos, gi.sampling_params(i), lp,
                                           **({'logprobs': gi.logprobs} if gi.logprobs else {}))] = i
            except _ffi.TmError as e:
                rt = STATUS_TO_RESPONSE.get(e.status, ResponseType.INTERNAL_ENGINE_ERROR)
                out_of_engine.append(Response('', 0, len(p), 'error', [], index=i, error_code=rt.name, error_message=str(e)))
        yield from out_of_engine
        # scheduler steps per host iteration: one native call runs the burst (requests finish / are admitted INSIDE it, step by step); the
        # host polls once per burst.  At tp > 1 every engine call is mirrored to the other ranks over localhost connections -- 0.05 ms
        # (tp = 2) to several ms (tp = 8) per call on a busy host (tools/tp_group_host_cost.py) against a 1.5 .. 3 ms device step -- so
        # a burst is one round trip, not one per step (the reference's per-rank engine threads exchange admissions, not steps:
        # src/turbomind/engine/engine.cc:770-870).  Streaming keeps single steps at tp = 1 (a Response per token) and short bursts above.
        tp = self.backend_config.tp
        burst = int(os.environ.get('TM_STREAM_BURST', '4' if tp > 1 else '1')) if stream else int(os.environ.get('TM_STEP_BURST', '8'))
        step_many = getattr(self.engine, 'step_many', None) if burst > 1 else None
        try:
            while pending:
                if step_many is not None:
                    step_many(burst)
                else:
                    self.engine.step()
                for rid, i in list(pending.items()):
                    st, toks = self.engine.poll(rid)
                    toks = toks.tolist()
                    g, stop = gs[i], stops[i]
                    cut = next((k for k, t in enumerate(toks) if t in stop), None)
                    if cut is not None and st == 0:      # a stop id the engine does not know about
                        self.engine.cancel(rid)
                        st = 8
                    out = toks if cut is None else toks[:cut]
                    if st == 0 and not stream:
                        continue
                    reason = None if st == 0 else ('stop' if cut is not None else 'length')
                    if st != 0:
                        del pending[rid]
                    def lps(first, toks_):    # Response.logprobs of tokens first .. first + len(toks_) of this request
                        if not g.logprobs or not toks_:
                            return None
                        v, ix, nm, sl = self.engine.poll_logprobs(rid)
                        return [logprobs_of_token(v[first + k], ix[first + k], int(nm[first + k]), float(sl[first + k]), t, g.logprobs)
                                for k, t in enumerate(toks_)]
                    if stream:          # the delta since the last Response of this request
                        first = sent.get(rid, 0)
                        new = out[first:]
                        sent[rid] = len(out)
                        if not new and st == 0:
                            continue
                        text = self.tokenizer.decode(new, skip_special_tokens=g.skip_special_tokens) if self.tokenizer else ''
                        yield Response(text, len(out), len(ids[i]), reason, new, logprobs=lps(first, new), index=i)
                        continue
                    text = self.tokenizer.decode(out, skip_special_tokens=g.skip_special_tokens) if self.tokenizer else ''
                    yield Response(text, len(out), len(ids[i]), reason, out, logprobs=lps(0, out), index=i)
        finally:
            self.engine.release()

    def _generate_static(self, prompts: Sequence, g: GenerationConfig):
        ids = [self._encode(p) for p in prompts]
        stop = self._stop_ids(g)
        # per-request admission on the host as the reference does it (lmdeploy/serve/async_engine.py:561-565): only a prompt
        # that leaves no room for a single new token (len >= session_len) is refused with INPUT_LENGTH_ERROR; otherwise the
        # request generates until its session is full and finishes with 'length' (max_new_tokens clamped per request).
        # Requests that need the clamp run in a chunk of their own: a static chunk shares one max_new_tokens.
        limit = self.session_len
        chunks, plain = [], []
        for i, p in enumerate(ids):
            if len(p) < 1 or len(p) >= limit:
                yield Response('', 0, len(p), 'error', [], index=i, error_code=ResponseType.INPUT_LENGTH_ERROR.name,
                               error_message='empty prompt' if len(p) < 1 else
                               f'prompt ({len(p)}) leaves no room for a new token in session_len ({limit})')
            elif len(p) + g.max_new_tokens > limit:
                chunks.append(([i], limit - len(p)))
            else:
                plain.append(i)
        chunks += [(plain[b0:b0 + self.max_batch_size], g.max_new_tokens) for b0 in range(0, len(plain), self.max_batch_size)]
        for idx, max_new in sorted(chunks, key=lambda c: c[0][0]):
            chunk = [ids[i] for i in idx]
            try:
                self.engine.set_sampling([g.sampling_params(i) for i in idx] if g.sampling_params() else None)
                lp = g.logits_params(sorted(stop)[:_ffi.MAX_STOP_IDS])
                self.engine.set_logits_params([lp] * len(chunk) if lp else None)
                self.engine.set_logprobs(g.logprobs or 0)
                if g.output_logits and (lp or self.backend_config.tp > 1):
                    # the engine's processors rewrite the logits in place and a tensor-parallel rank holds a vocabulary shard: what could be read
                    # back would not be the reference's output (the raw lm_head row, src/turbomind/models/language_model.cc OutputLogits)
                    raise NotImplementedError('output_logits together with repetition_penalty / min_new_tokens / bad_token_ids, or with tp > 1')
                self.engine.prefill(chunk, max_new_tokens=max_new)
                step_logits = [self.engine.fetch_logits().copy()] if g.output_logits else None   # [batch, vocab] fp16 behind every step
                done = 1
                while done < max_new:
                    n = 1 if step_logits is not None else min(32, max_new - done)
                    self.engine.decode(n)
                    done += n
                    if step_logits is not None:
                        step_logits.append(self.engine.fetch_logits().copy())
                    if stop:
                        toks = self.engine.fetch()
                        if all(any(int(t) in stop for t in row) for row in toks):
                            break
                toks = self.engine.fetch()
                records = self.engine.fetch_logprobs() if g.logprobs else None
            except _ffi.TmError as e:
                self.engine.release()
                rt = STATUS_TO_RESPONSE.get(e.status, ResponseType.INTERNAL_ENGINE_ERROR)
                for i, p in zip(idx, chunk):
                    yield Response('', 0, len(p), 'error', [], index=i, error_code=rt.name, error_message=str(e))
                continue
            self.engine.release()
            for b, (i, p, row) in enumerate(zip(idx, chunk, toks)):
                out, reason = [], 'length'
                for tkn in row.tolist():
                    if tkn in stop:
                        reason = 'stop'
                        break
                    out.append(tkn)
                text = self.tokenizer.decode(out, skip_special_tokens=g.skip_special_tokens) if self.tokenizer else ''
                lps = None
                if records is not None:
                    lps = [logprobs_of_token(records[0][b, s], records[1][b, s], int(records[2][b, s]), float(records[3][b, s]), tkn,
                                             g.logprobs) for s, tkn in enumerate(out)]
                lg = None
                if step_logits is not None:     # [generated tokens, vocab]: row s = the logits token s was drawn from (turbomind.py:444-451, 'generation')
                    lg = np.stack([step_logits[s_][b] for s_ in range(len(out))]).astype(np.float32) if out else np.zeros((0, 0), np.float32)
                yield Response(text, len(out), len(p), reason, out, logprobs=lps, logits=lg, index=i)
