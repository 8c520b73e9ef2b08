"""Python adapter over the native engine (C-ABI `tm_engine_*`).

Host-side mirror of `lmdeploy/turbomind/turbomind.py:108-850` (class TurboMind / TurboMindInstance) for the
hot path: config -> native EngineConfig, weight slot hand-off, static-batch prefill / decode, token fetch.
Everything numeric happens in libtm_mi355x.so; a failing native call raises `_ffi.TmError` carrying the
reference's request status code.
"""
from __future__ import annotations

import ctypes as C
from typing import Sequence

import numpy as np

from .. import _ffi

_ROPE_TYPES = {'default': 0, 'linear': 1, 'llama3': 2}


def make_model_config(cfg, weight_type: int = 0) -> _ffi.ModelConfig:
    """cfg: any object with hidden, layers, q_heads, kv_heads, head_dim, inter, vocab, rms_eps, rope(.dim,.base,.type,
    .factor,.low_freq_factor,.high_freq_factor,.original_max_position_embeddings), group."""
    r = cfg.rope
    return _ffi.ModelConfig(cfg.hidden, cfg.layers, cfg.q_heads, cfg.kv_heads, cfg.head_dim, cfg.inter, cfg.vocab,
                            cfg.rms_eps, r.base, _ROPE_TYPES[r.type], r.factor, r.low_freq_factor, r.high_freq_factor,
                            r.original_max_position_embeddings, cfg.group, weight_type,
                            int(getattr(cfg, 'moe_experts', 0) or 0), int(getattr(cfg, 'moe_top_k', 0) or 0),
                            int(bool(getattr(cfg, 'moe_norm_topk', True))), float(getattr(cfg, 'moe_routed_scale', 1.0)))


class Engine:

    def __init__(self, model_cfg: _ffi.ModelConfig, *, tp: int = 1, rank: int = 0, device: int = 0,
                 max_batch_size: int = 64, session_len: int = 2048, quant_policy: int = 8,
                 cache_max_entry_count: float = 0.8, cache_blocks: int = 0, max_prefill_token_num: int = 8192,
                 decode_splits: int = 0, use_graph: int = 1):
        self._lib = _ffi.load()
        self.cfg = _ffi.EngineConfig(model_cfg, tp, rank, device, max_batch_size, session_len, quant_policy, 64,
                                     cache_max_entry_count, cache_blocks, max_prefill_token_num, decode_splits,
                                     use_graph)
        self._h = C.c_void_p()
        _ffi.check(self._lib.tm_engine_create(C.byref(self._h), C.byref(self.cfg)))
        self.batch = 0
        self.max_new = 0

    @classmethod
    def from_model_config(cls, cfg, weight_type: int = 0, **kw) -> 'Engine':
        return cls(make_model_config(cfg, weight_type), **kw)

    # ---- weights ---------------------------------------------------------------------------------
    def load_weights(self, slots: dict):
        """slots: {name: numpy array} as produced by loader.export_weights (already sharded for this rank)."""
        for name, arr in slots.items():
            arr = np.ascontiguousarray(arr)
            expect = self._lib.tm_engine_weight_bytes(self._h, name.encode())
            if expect != arr.nbytes:      # same assertion as builders/_base.py:72-99 (dst.byte_size == shard.nbytes)
                raise ValueError(f'{name}: {arr.nbytes} bytes given, slot holds {expect}')
            _ffi.check(self._lib.tm_engine_weight_copy(self._h, name.encode(), arr.ctypes.data, arr.nbytes))
        _ffi.check(self._lib.tm_engine_process_weights(self._h))

    def init_synthetic(self, seed: int = 0):
        _ffi.check(self._lib.tm_engine_init_synthetic(self._h, seed))
        _ffi.check(self._lib.tm_engine_process_weights(self._h))

    def comm_init(self, unique_id: bytes):
        buf = C.create_string_buffer(unique_id, 128)
        _ffi.check(self._lib.tm_engine_comm_init(self._h, buf))

    def comm_drop_rccl(self):
        _ffi.check(self._lib.tm_engine_comm_drop_rccl(self._h))

    def comm_native_setup(self, all_gather, rows: int = 256):
        """TM_COMM=native: switch the row-parallel all-reduces of forwards with <= `rows` tokens to the fused P2P kernel.
        `all_gather(bytes) -> list[bytes]` (rank order) is the caller's host-side exchange, e.g. torch.distributed's
        all_gather_object.  Call after comm_init, before start."""
        buf = C.create_string_buffer(64)
        err = None
        try:
            _ffi.check(self._lib.tm_engine_comm_native_export(self._h, rows, buf))
        except _ffi.TmError as e:       # the gather below must still happen on this rank, or the others wait in it forever
            err = e
        handles = all_gather(None if err else buf.raw)
        if err is not None or any(h is None for h in handles):
            raise err or _ffi.TmError(5, 'native communicator: a peer rank could not export its segment')
        blob = b''.join(handles)
        _ffi.check(self._lib.tm_engine_comm_native_import(self._h, C.create_string_buffer(blob, len(blob)), len(handles)))

    def comm_native_selftest(self) -> bool:
        """collective (every rank, right after comm_native_setup): two fused all-reduce launches over a known pattern; False = fall back
        (comm_native_drop on every rank).  The default tp > 1 arrangement takes the native fused decode collective only when this
        passed everywhere -- the hop between devices is what no single-GPU test covers."""
        ok = C.c_int(0)
        _ffi.check(self._lib.tm_engine_comm_native_selftest(self._h, C.byref(ok)))
        return bool(ok.value)

    def comm_native_drop(self):
        _ffi.check(self._lib.tm_engine_comm_native_drop(self._h))

    @staticmethod
    def comm_unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        _ffi.check(_ffi.load().tm_comm_unique_id(buf))
        return buf.raw

    def start(self):
        _ffi.check(self._lib.tm_engine_start(self._h))

    def tune_gemm(self, M: int = 0, export_path: str = ''):
        """Measured GEMM dispatch for the decode batch (the reference's TM_GEMM_TUNE warm-up): times every candidate tiling
        of the decode linears at M rows (default: max_batch_size) on this engine's weights and keeps the winners; optional
        text export (`tm_gemm_import` / TM_GEMM_IMPORT loads it in a later process)."""
        M = M or int(self.cfg.max_batch_size)
        _ffi.check(self._lib.tm_engine_tune_gemm(self._h, int(M), export_path.encode() if export_path else None))

    def import_gemm_table(self, path: str):
        """load a `K N M shape splits` dispatch table written by tune_gemm (the reference's TM_GEMM_IMPORT)"""
        _ffi.check(self._lib.tm_gemm_import(path.encode()))

    @staticmethod
    def export_gemm_table(path: str):
        """write this process's measured dispatch tables (the reference's TM_GEMM_EXPORT) -- P32 lines and `G` lines"""
        _ffi.check(_ffi.load().tm_gemm_export(path.encode()))

    @staticmethod
    def pick_general(weight_type: int, role: int, K: int, N: int, M: int):
        """(tiles per wave, split-K, waves, k-phases) of the general kernel for a dense linear that the P32 kernels do not serve
        (fp16 lm_head: weight_type 1, role 5; e4m3 weight-only: weight_type 2) -- measured entry first, then the heuristic"""
        v = (C.c_int * 4)()
        _ffi.check(_ffi.load().tm_debug_pick_general(int(weight_type), int(role), int(K), int(N), int(M), v))
        return tuple(v)

    @staticmethod
    def pick_tiling(K: int, N: int, M: int, use_table: bool = True, role: int = 0):
        """(shape, split-K) the decode GEMM dispatch runs a K x N W4A16 linear with at M rows (measured table first); role: 0 any,
        1 w_qkv, 2 wo, 3 w1w3, 4 w2 -- the measured table is keyed (role, K, N, M)"""
        sh, sp = C.c_int(), C.c_int()
        _ffi.check(_ffi.load().tm_debug_pick_tiling(K, N, M, (1 if use_table else 0) | (role << 8), C.byref(sh), C.byref(sp)))
        return sh.value, sp.value

    # ---- static batch ----------------------------------------------------------------------------
    def set_sampling(self, params):
        """Static batch: per-sequence (temperature, top_k, top_p, min_p, seed) tuples for the NEXT prefill; None = greedy."""
        if not params:
            _ffi.check(self._lib.tm_engine_set_sampling(self._h, None, 0))
            return
        arr = (_ffi.Sampling * len(params))(*[_ffi.Sampling(*(p if p is not None else (1.0, 1, 1.0, 0.0, 0))) for p in params])
        _ffi.check(self._lib.tm_engine_set_sampling(self._h, arr, len(params)))

    def set_logits_params(self, params):
        """Static batch: per-sequence dicts (repetition_penalty, min_new_tokens, bad_ids, stop_ids) or _ffi.LogitsParam for
        the NEXT prefill; None / empty = no logits processors."""
        if not params:
            _ffi.check(self._lib.tm_engine_set_logits_params(self._h, None, 0))
            return
        items = [p if isinstance(p, _ffi.LogitsParam) else _ffi.LogitsParam.make(**(p or {})) for p in params]
        arr = (_ffi.LogitsParam * len(items))(*items)
        _ffi.check(self._lib.tm_engine_set_logits_params(self._h, arr, len(items)))

    def set_logprobs(self, n: int):
        """Static batch: the NEXT prefill records the first n kept candidates (and the drawn token's own logprob) of every
        generated token (GenerationConfig.logprobs; 0 / None = off)."""
        _ffi.check(self._lib.tm_engine_set_logprobs(self._h, int(n or 0)))
        self._logprobs_n = int(n or 0)

    def fetch_logprobs(self):
        """(vals [batch, max_new, n] float32, idx [batch, max_new, n] int32, num [batch, max_new] int32, sel [batch, max_new]
        float32): the records of tm_engine_set_logprobs; columns beyond the generated steps have num = 0."""
        n = getattr(self, '_logprobs_n', 0)
        if n < 1:
            raise ValueError('no logprobs were requested (set_logprobs before prefill)')
        vals = np.zeros((self.batch, self.max_new, n), np.float32)
        idx = np.zeros((self.batch, self.max_new, n), np.int32)
        num = np.zeros((self.batch, self.max_new), np.int32)
        sel = np.zeros((self.batch, self.max_new), np.float32)
        _ffi.check(self._lib.tm_engine_fetch_logprobs(self._h, vals.ctypes.data, idx.ctypes.data, num.ctypes.data, sel.ctypes.data))
        return vals, idx, num, sel

    def prefill(self, prompts: Sequence[Sequence[int]], max_new_tokens: int):
        lens = np.asarray([len(p) for p in prompts], np.int32)
        ids = np.concatenate([np.asarray(p, np.int32) for p in prompts]).astype(np.int32)
        _ffi.check(self._lib.tm_engine_prefill(self._h, ids.ctypes.data, lens.ctypes.data, len(prompts),
                                               max_new_tokens))
        self.batch, self.max_new = len(prompts), max_new_tokens

    def decode(self, steps: int = 1):
        _ffi.check(self._lib.tm_engine_decode(self._h, steps))

    PROF_CATEGORIES = ('embed', 'gemm_qkv', 'kv_store', 'attention', 'gemm_o', 'residual_norm', 'gemm_gate_up',
                       'gemm_down', 'lm_head', 'sample', 'allreduce')

    def profile_decode(self, steps: int = 1) -> dict:
        """Eager decode steps with HIP events around every kernel category -> {category: (ms per step, launches)}."""
        ms = (C.c_float * len(self.PROF_CATEGORIES))()
        n = (C.c_int * len(self.PROF_CATEGORIES))()
        _ffi.check(self._lib.tm_engine_profile_decode(self._h, steps, ms, n))
        return {k: (ms[i], n[i]) for i, k in enumerate(self.PROF_CATEGORIES)}

    def prefill_times_ms(self) -> np.ndarray:
        out = np.zeros(self.batch, np.float32)
        _ffi.check(self._lib.tm_engine_prefill_times(self._h, out.ctypes.data))
        return out

    def sync(self):
        _ffi.check(self._lib.tm_engine_sync(self._h))

    def fetch(self) -> np.ndarray:
        """tokens generated so far, [batch, steps].  After a communicator give-up (terminal) the native call hands over
        the columns known to be valid together with its error status: they travel on the raised TmError as `.partial_tokens`."""
        out = np.zeros((self.batch, self.max_new), np.int32)
        n = C.c_int(0)
        rc = self._lib.tm_engine_fetch(self._h, out.ctypes.data, C.byref(n))
        if rc != 0:
            err = _ffi.TmError(rc, _ffi.last_error())
            err.partial_tokens = out[:, :n.value]
            raise err
        return out[:, :n.value]

    def fetch_logits(self) -> np.ndarray:
        v_local = self.cfg.model.vocab // self.cfg.tp
        out = np.zeros((self.batch, v_local), np.float16)
        _ffi.check(self._lib.tm_engine_fetch_logits(self._h, out.ctypes.data))
        return out

    def fetch_residual(self, rows: int) -> np.ndarray:
        """parity tests: the residual stream of the last forward, fp16 [rows][hidden]"""
        out = np.zeros((rows, self.cfg.model.hidden), np.float16)
        _ffi.check(self._lib.tm_engine_debug_read(self._h, 0, rows, 0, out.ctypes.data, out.nbytes))
        return out

    def fetch_kv_block(self, seq: int, block: int) -> np.ndarray:
        """parity tests: the bytes of KV block `block` of static-batch sequence `seq` (all layers)"""
        out = np.zeros(self.stats()['kv_bytes_per_token'] * 64, np.uint8)
        _ffi.check(self._lib.tm_engine_debug_read(self._h, 1, seq, block, out.ctypes.data, out.nbytes))
        return out

    def mixed_steps(self) -> int:
        """continuous batching: scheduler steps whose decode rows shared ONE forward with an admission's prefill"""
        out = np.zeros(1, np.int64)
        _ffi.check(self._lib.tm_engine_debug_read(self._h, 2, 0, 0, out.ctypes.data, 8))
        return int(out[0])

    def overlapped_steps(self) -> int:
        """continuous batching: decode steps issued while the previous one was still unretired (the two-phase schedule /
        forward overlap of the reference, turbomind.cc:171; TM_ASYNC_STEP=1 switches it on)"""
        out = np.zeros(1, np.int64)
        _ffi.check(self._lib.tm_engine_debug_read(self._h, 3, 0, 0, out.ctypes.data, 8))
        return int(out[0])

    # ---- continuous batching (tm_engine_submit / step / poll / cancel) -----------------------------
    def submit(self, prompt: Sequence[int], max_new_tokens: int, eos_id: int = -1, sampling=None, logits=None, logprobs: int = 0) -> int:
        """Queue one request; returns its id.  eos_id < 0 = ignore_eos.  sampling = (temperature, top_k, top_p, min_p,
        seed) or None (greedy); logits = dict(repetition_penalty, min_new_tokens, bad_ids, stop_ids) or None; logprobs = n > 0:
        every generated token carries its first n kept candidates (poll_logprobs).  Raises TmError with the reference's status
        code (TM_TOO_LONG, TM_OOM, TM_INVALID) when the request can never run."""
        ids = np.ascontiguousarray(np.asarray(prompt, np.int32))
        rid = C.c_int64(0)
        sp = C.byref(_ffi.Sampling(*sampling)) if sampling is not None else None
        if logits is not None and not isinstance(logits, _ffi.LogitsParam):
            logits = _ffi.LogitsParam.make(**logits)
        lp = C.byref(logits) if logits is not None else None
        _ffi.check(self._lib.tm_engine_submit_gen(self._h, ids.ctypes.data, int(ids.size), int(max_new_tokens), int(eos_id), sp,
                                                  lp, C.byref(rid)))
        if logprobs:
            _ffi.check(self._lib.tm_engine_request_logprobs(self._h, rid.value, int(logprobs)))
        return rid.value

    def poll_logprobs(self, req_id: int):
        """(vals [tokens, n] float32, idx [tokens, n] int32, num [tokens] int32, sel [tokens] float32) recorded so far for a request
        submitted with logprobs = n (entries beyond num[t]: 0 / -1)."""
        nt, n = C.c_int(0), C.c_int(0)
        _ffi.check(self._lib.tm_engine_poll_logprobs(self._h, req_id, None, None, None, None, 0, C.byref(nt), C.byref(n)))
        t, w = nt.value, n.value     # the engine thread may append between the two calls: copy what the first call saw
        vals, idx = np.zeros((max(t, 1), max(w, 1)), np.float32), np.zeros((max(t, 1), max(w, 1)), np.int32)
        num, sel = np.zeros(max(t, 1), np.int32), np.zeros(max(t, 1), np.float32)
        if t and w:
            _ffi.check(self._lib.tm_engine_poll_logprobs(self._h, req_id, vals.ctypes.data, idx.ctypes.data, num.ctypes.data, sel.ctypes.data,
                                                         t, C.byref(nt), C.byref(n)))
        return vals[:t, :w], idx[:t, :w], num[:t], sel[:t]

    def step(self):
        """One scheduler iteration: admit + prefill waiting requests, then one decode step for everything running.
        Returns (n_active, n_waiting) after the step."""
        na, nw = C.c_int(0), C.c_int(0)
        _ffi.check(self._lib.tm_engine_step(self._h, C.byref(na), C.byref(nw)))
        return na.value, nw.value

    def step_many(self, max_steps: int):
        """Up to max_steps scheduler iterations in one native call (stops when nothing runs and nothing waits).  Returns
        (steps_done, n_active, n_waiting).  The call a tensor-parallel rank group mirrors instead of single steps (tp_group.py)."""
        nd, na, nw = C.c_int(0), C.c_int(0), C.c_int(0)
        _ffi.check(self._lib.tm_engine_step_many(self._h, int(max_steps), C.byref(nd), C.byref(na), C.byref(nw)))
        return nd.value, na.value, nw.value

    def poll(self, req_id: int, cap: int = 0):
        """(status, tokens generated so far) -- status 0 = waiting / running, 7 = finished, 8 = cancelled."""
        st, n = C.c_int(0), C.c_int(0)
        _ffi.check(self._lib.tm_engine_poll(self._h, req_id, C.byref(st), None, 0, C.byref(n)))
        cap = n.value  # the engine thread may append between the two calls: copy what the first call saw
        out = np.zeros(max(cap, 1), np.int32)
        _ffi.check(self._lib.tm_engine_poll(self._h, req_id, C.byref(st), out.ctypes.data, cap, C.byref(n)))
        return st.value, out[:min(cap, n.value)]

    def cancel(self, req_id: int):
        _ffi.check(self._lib.tm_engine_cancel(self._h, req_id))

    def forget(self, req_id: int):
        """Drop the record of a finished request (long-lived serving sessions; release() drops everything)."""
        _ffi.check(self._lib.tm_engine_forget(self._h, req_id))

    # -- engine thread (the reference's InternalThreadEntry + signal thread): submit / poll / cancel / wait from any thread
    def serve_start(self, on_update=None):
        """Start the engine-owned scheduler thread.  on_update(req_id, status, n_tokens) runs ON THAT THREAD once per
        step for every request that produced a token or finished (ctypes takes the GIL for the call)."""
        if on_update is not None:
            proto = C.CFUNCTYPE(None, C.c_void_p, C.c_int64, C.c_int, C.c_int)
            self._cb_keepalive = proto(lambda _u, rid, st, n: on_update(int(rid), int(st), int(n)))
            cb = C.cast(self._cb_keepalive, C.c_void_p)
        else:
            self._cb_keepalive, cb = None, None
        _ffi.check(self._lib.tm_engine_serve_start(self._h, cb, None))

    def serve_stop(self):
        _ffi.check(self._lib.tm_engine_serve_stop(self._h))

    def wait(self, req_id: int, have_tokens: int = 0, timeout_ms: int = -1):
        """Block (GIL released) until the request has more than have_tokens tokens or a non-zero status.
        Returns (status, n_tokens)."""
        st, n = C.c_int(0), C.c_int(0)
        _ffi.check(self._lib.tm_engine_wait(self._h, req_id, int(have_tokens), int(timeout_ms), C.byref(st), C.byref(n)))
        return st.value, n.value

    def release(self):
        _ffi.check(self._lib.tm_engine_release(self._h))
        self.batch = 0

    @property
    def stream(self) -> int:
        return self._lib.tm_engine_stream(self._h)

    def stats(self) -> dict:
        wb, kv, nb, sp = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int()
        _ffi.check(self._lib.tm_engine_stats(self._h, C.byref(wb), C.byref(kv), C.byref(nb), C.byref(sp)))
        return dict(weight_bytes=wb.value, kv_bytes_per_token=kv.value, num_blocks=nb.value, decode_splits=sp.value)

    def comm_info(self) -> dict:
        """which communicator the tensor-parallel data path runs on, its own rank count, graph replay or eager"""
        b, n, g = C.c_int(), C.c_int(), C.c_int()
        _ffi.check(self._lib.tm_engine_comm_info(self._h, C.byref(b), C.byref(n), C.byref(g)))
        name = {0: 'none', 1: 'rccl', 2: 'native-p2p', 3: 'native-p2p (decode) + rccl (large forwards)'}[b.value]
        ss, fw, mbf, ar = C.c_int(), C.c_int64(), C.c_int64(), C.c_int64()
        _ffi.check(self._lib.tm_engine_comm_overlap_info(self._h, C.byref(ss), C.byref(fw), C.byref(mbf), C.byref(ar)))
        # side_stream: prefill-sized forwards run their all-reduces on a side stream under the other row half's GEMMs
        return dict(backend=name, ranks=n.value, hipgraph=bool(g.value), side_stream=bool(ss.value), overlapped_forwards=fw.value,
                    microbatch_forwards=mbf.value, side_stream_allreduces=ar.value)

    def close(self):
        if self._h:
            self._lib.tm_engine_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
