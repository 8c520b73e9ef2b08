"""The part of `lmdeploy/messages.py` the hot path honours, with the reference's names and validation.

Reference: QuantPolicy (lmdeploy/messages.py:20-27), GenerationConfig (:35-205), TurbomindEngineConfig (:208-367),
ResponseType / Response (:553-602).  Fields that the MI355X engine does not implement raise NotImplementedError when
set to a non-default value -- loudly, instead of being silently ignored.
"""
from __future__ import annotations

import enum
from dataclasses import dataclass, field
from typing import Any, Literal


class QuantPolicy(enum.IntEnum):
    """Quantization policy constants for KV cache (TurboMind accepts NONE / INT4 / INT8 only)."""
    NONE = 0
    INT4 = 4
    INT8 = 8
    FP8 = 16
    FP8_E5M2 = 17
    TURBO_QUANT = 42


class ResponseType(enum.Enum):
    SUCCESS = enum.auto()
    FINISH = enum.auto()
    ENGINE_STOP_ERROR = enum.auto()
    SESSION_REPEAT = enum.auto()
    SESSION_NOT_EXIST = enum.auto()
    HANDLER_NOT_EXIST = enum.auto()
    INPUT_LENGTH_ERROR = enum.auto()
    INTERNAL_ENGINE_ERROR = enum.auto()
    CANCEL = enum.auto()
    PREFIX_CACHE_CONFLICT = enum.auto()
    NO_QUEUE = enum.auto()
    NOT_SUPPORTED = enum.auto()
    OUT_OF_MEMORY = enum.auto()


MAX_LOGPROBS = 1024     # lmdeploy/turbomind/turbomind.py:45 = kMaxLogProb (src/turbomind/utils/constant.h:7)

# native status code (src/turbomind/engine/request.h:120-131) -> ResponseType (lmdeploy/turbomind/turbomind.py:587-599)
STATUS_TO_RESPONSE = {
    0: ResponseType.SUCCESS, 1: ResponseType.INTERNAL_ENGINE_ERROR, 2: ResponseType.PREFIX_CACHE_CONFLICT,
    5: ResponseType.INTERNAL_ENGINE_ERROR, 6: ResponseType.INPUT_LENGTH_ERROR, 7: ResponseType.FINISH,
    8: ResponseType.CANCEL, 10: ResponseType.NO_QUEUE, 11: ResponseType.OUT_OF_MEMORY,
}


@dataclass
class GenerationConfig:
    """Same fields and defaults as the reference (lmdeploy/messages.py:35-205).  do_sample=False is greedy (the
    reference forces top_k=1 then: lmdeploy/serve/core/async_engine.py:424-428); do_sample=True samples with
    temperature / top_k / top_p / min_p / random_seed on the GPU (csrc/sampling.hip)."""
    n: int = 1
    max_new_tokens: int = 512
    do_sample: bool = False
    top_p: float = 1.0
    top_k: int = 50
    min_p: float = 0.0
    temperature: float = 0.8
    repetition_penalty: float = 1.0
    ignore_eos: bool = False
    random_seed: int = None
    stop_words: list[str] = None
    bad_words: list[str] = None
    stop_token_ids: list[int] = None
    bad_token_ids: list[int] = None
    min_new_tokens: int = None
    skip_special_tokens: bool = True
    spaces_between_special_tokens: bool = True
    logprobs: int = None
    response_format: dict | None = None
    logits_processors: list | None = None
    output_logits: Literal['all', 'generation'] = None
    output_last_hidden_state: Literal['all', 'generation'] = None
    include_stop_str_in_output: bool = False
    # fields of the reference that belong to subsystems outside the hot path (PPL scoring, prefix-cache control,
    # PD-disaggregation migration, expert-routing dumps, n-gram repetition filter): accepted at their defaults only
    return_ppl: bool = False
    with_cache: bool = False
    preserve_cache: bool = False
    migration_request: Any = None
    return_routed_experts: bool = False
    repetition_ngram_size: int = 0
    repetition_ngram_threshold: int = 0

    def __post_init__(self):
        assert isinstance(self.n, int) and self.n > 0, 'n is not a positive integer'
        assert self.top_p >= 0 and self.top_p <= 1
        assert self.top_k >= 0, 'top_k can not be a negative integer'
        assert self.temperature >= 0 and self.temperature <= 2
        assert 0 <= self.min_p <= 1
        assert self.repetition_penalty > 0, 'repetition_penalty must be > 0'
        unsupported = {'n': 1, 'response_format': None, 'logits_processors': None,
                       'output_last_hidden_state': None, 'return_ppl': False, 'with_cache': False,
                       'preserve_cache': False, 'migration_request': None, 'return_routed_experts': False,
                       'repetition_ngram_size': 0, 'repetition_ngram_threshold': 0}
        if self.do_sample and self.temperature == 0:
            raise ValueError('temperature must be > 0 when do_sample=True')
        for k, default in unsupported.items():
            if getattr(self, k) != default:
                raise NotImplementedError(f'GenerationConfig.{k}={getattr(self, k)!r}: the MI355X hot path implements '
                                          f'greedy decoding, temperature / top-k / top-p / min-p sampling, repetition '
                                          f'penalty, min_new_tokens, bad_token_ids, stop_token_ids, logprobs and output_logits="generation" only')
        if self.output_logits not in (None, 'generation'):
            raise NotImplementedError(f'GenerationConfig.output_logits={self.output_logits!r}: only "generation" (the logits every generated token was '
                                      f'drawn from); the prompt positions\' logits are not computed (the lm_head runs on the last token of a prompt only)')
        if self.logprobs is not None:
            assert isinstance(self.logprobs, int) and self.logprobs >= 0, 'logprobs must be a non-negative integer'
            if self.logprobs > MAX_LOGPROBS:      # the reference clamps with a warning (lmdeploy/turbomind/turbomind.py:840-845)
                import warnings
                warnings.warn(f'logprobs should be in range [1, {MAX_LOGPROBS}]: update logprobs={MAX_LOGPROBS}')
                self.logprobs = MAX_LOGPROBS


    def convert_stop_bad_words_to_ids(self, tokenizer):
        """stop_words / bad_words -> stop_token_ids / bad_token_ids, the reference's rule (lmdeploy/messages.py:154-174,
        lmdeploy/tokenizer.py:152-187,539-547): a word must encode to exactly ONE token (multi-token words cannot be
        used and are skipped with a warning); it then stands for every vocabulary entry whose token string contains it
        (' ' matches the sentencepiece '\u2581'); more than 5 candidates are narrowed to the ids that decode to exactly
        the word, at most 5.  `tokenizer`: a HF tokenizer (encode / get_vocab / decode)."""
        import warnings

        def ids_of(words, what):
            if words is None:
                return []
            assert isinstance(words, list) and all(isinstance(w_, str) for w_ in words), f'{what} must be a list of str'
            vocab = tokenizer.get_vocab()
            out = []
            for word in words:
                enc = tokenizer.encode(word, add_special_tokens=False)
                if len(enc) != 1:
                    warnings.warn(f'{what}: {word!r} encodes to {len(enc)} tokens; only single-token words can be used')
                    continue
                key = '\u2581' if word == ' ' else word
                idx = sorted(i for tok, i in vocab.items() if key in tok)
                if len(idx) > 5:
                    idx = [i for i in idx if tokenizer.decode([i]) == word][:5]
                out += idx or enc
            return out

        stop = ids_of(self.stop_words, 'stop_words') + list(self.stop_token_ids or [])
        bad = ids_of(self.bad_words, 'bad_words') + list(self.bad_token_ids or [])
        self.stop_token_ids = sorted(set(stop)) or None
        self.bad_token_ids = sorted(set(bad)) or None
        self.stop_words = self.bad_words = None          # consumed

    def logits_params(self, stop_ids=()):
        """dict for the engine's logits processors (tm_logits_param), or None when every processor is off.
        stop_ids: the ids that end the sequence besides the engine-side eos id (banned until min_new_tokens)."""
        # greedy requests run with repetition_penalty 1.0 whatever the field says -- the reference resets it together
        # with top_k / temperature (lmdeploy/serve/core/async_engine.py:424-430); bad ids and min_new_tokens stay on
        rep = float(self.repetition_penalty) if self.do_sample else 1.0
        if rep == 1.0 and not self.bad_token_ids and not self.min_new_tokens:
            return None
        return dict(repetition_penalty=rep, min_new_tokens=int(self.min_new_tokens or 0),
                    bad_ids=list(self.bad_token_ids or ()), stop_ids=list(stop_ids))

    def sampling_params(self, index: int = 0):
        """(temperature, top_k, top_p, min_p, seed) for the engine, or None for greedy.  Every sequence of a call gets
        its own stream: seed = random_seed (a fresh one if None) + index."""
        if not self.do_sample or self.top_k == 1:
            return None
        if self.random_seed is None:
            import random
            self.random_seed = random.getrandbits(62)
        return (float(self.temperature), int(self.top_k), float(self.top_p), float(self.min_p),
                (int(self.random_seed) + index) & (2**64 - 1))


@dataclass
class TurbomindEngineConfig:
    """TurboMind Engine config -- same field names and defaults as the reference (lmdeploy/messages.py:302-343)."""
    dtype: str = 'auto'
    model_format: str | None = None
    tp: int = 1
    dp: int = 1
    cp: int = 1
    ep: int = 1
    device_num: int = None
    attn_tp_size: int = None
    attn_cp_size: int = None
    attn_dp_size: int = None
    mlp_tp_size: int = None
    mlp_dp_size: int = None
    outer_dp_size: int = None
    nnodes: int = 1
    node_rank: int = 0
    dist_init_addr: str | None = None
    devices: list[int] | None = None
    session_len: int | None = None
    max_batch_size: int = None
    cache_max_entry_count: float = 0.8
    cache_chunk_size: int = -1
    cache_block_seq_len: int = 64
    enable_prefix_caching: bool = False
    cache_checkpoint_interval: int = 4096
    cache_prompt: str = 'auto'
    cache_prompt_boundary_skip: int = 1
    cache_generation: str = 'auto'
    quant_policy: int = 0
    rope_scaling_factor: float = 0.0
    use_logn_attn: bool = False
    download_dir: str | None = None
    revision: str | None = None
    max_prefill_token_num: int = 8192
    num_tokens_per_iter: int = 0
    max_prefill_iters: int = 1
    async_: int = 1
    empty_init: bool = False
    language_model_only: bool = False
    communicator: str = 'nccl'
    hf_overrides: dict[str, Any] | None = None
    enable_metrics: bool = True

    def __post_init__(self):
        """Reference validation (lmdeploy/messages.py:345-367) ..."""
        assert self.dtype in ['auto', 'float16', 'bfloat16']
        assert self.tp >= 1, 'tp must be a positive integer'
        assert self.ep >= 1, 'ep must be a positive integer'
        assert self.cache_max_entry_count > 0, 'invalid cache_max_entry_count'
        try:
            self.quant_policy = QuantPolicy(self.quant_policy)
        except ValueError as e:
            raise ValueError(f'invalid quant_policy: {self.quant_policy}') from e
        assert self.quant_policy not in (QuantPolicy.FP8, QuantPolicy.FP8_E5M2), \
            'invalid quant_policy for TurboMind, FP8 quantization is not supported'
        assert self.rope_scaling_factor >= 0, 'invalid rope_scaling_factor'
        assert self.max_prefill_token_num >= 0, 'invalid max_prefill_token_num'
        assert self.num_tokens_per_iter >= 0, 'invalid num_tokens_per_iter'
        assert self.async_ in (0, 1), 'async_ must be 0 (disabled) or 1 (enabled)'
        # ... plus: everything outside the hot path must be asked for explicitly and fails loudly
        if self.quant_policy == QuantPolicy.TURBO_QUANT:
            raise NotImplementedError('quant_policy=TURBO_QUANT')
        if self.dtype == 'bfloat16':
            raise NotImplementedError('dtype=bfloat16: the AWQ path is fp16 (lmdeploy/turbomind/converter.py:40-48)')
        if self.model_format not in (None, 'awq', 'hf', 'fp8'):
            raise NotImplementedError(f'model_format={self.model_format!r}: only "awq" (W4A16 g128), "fp8" (e4m3, 128x128 '
                                      f'block scales) and "hf" (fp16)')
        if self.cache_block_seq_len != 64:
            raise NotImplementedError('cache_block_seq_len must be 64')
        unsupported = {'dp': 1, 'cp': 1, 'ep': 1, 'attn_tp_size': None, 'attn_cp_size': None, 'attn_dp_size': None,
                       'mlp_tp_size': None, 'mlp_dp_size': None, 'outer_dp_size': None, 'nnodes': 1, 'node_rank': 0,
                       'dist_init_addr': None, 'enable_prefix_caching': False, 'rope_scaling_factor': 0.0,
                       'use_logn_attn': False, 'num_tokens_per_iter': 0, 'max_prefill_iters': 1,
                       'language_model_only': False, 'hf_overrides': None, 'cache_chunk_size': -1}
        for k, default in unsupported.items():
            if getattr(self, k) != default:
                raise NotImplementedError(f'TurbomindEngineConfig.{k}={getattr(self, k)!r} is outside the MI355X hot path')
        # the reference's values (src/turbomind/comm/device_comm.cc:14-30): 'nccl' = RCCL on ROCm; 'native' / 'cuda-ipc' = its in-house
        # communicator -> here the native P2P communicator (csrc/comm_p2p.hip) for the decode-sized exchanges, RCCL kept for prefill
        if self.communicator not in ('nccl', 'native', 'cuda-ipc'):
            raise ValueError(f'communicator={self.communicator!r}: "nccl" (RCCL on ROCm), "native" or "cuda-ipc" (the native P2P communicator)')


@dataclass
class Response:
    text: str
    generate_token_len: int
    input_token_len: int
    finish_reason: Literal['stop', 'length', 'error', 'abort'] | None = None
    token_ids: list[int] = field(default_factory=list)
    logprobs: list | None = None
    logits: Any = None
    last_hidden_state: Any = None
    index: int = 0
    routed_experts: Any = None
    cached_tokens: int = 0
    error_code: str | None = None
    error_message: str | None = None
