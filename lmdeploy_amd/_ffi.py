"""ctypes binding of libtm_mi355x.so (the C-ABI declared in include/tm_mi355x.h).

This is the only place where Python touches the native library.  There is NO fallback: if the
shared object is missing or a call fails, an exception is raised (the product path never routes
through a CPU implementation).
"""
from __future__ import annotations

import ctypes as C
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int64, c_size_t, c_uint64, c_void_p

_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'lib', 'libtm_mi355x.so')


class TmError(RuntimeError):

    def __init__(self, status: int, message: str):
        super().__init__(f'tm_mi355x status {status}: {message}')
        self.status = status


class Sampling(C.Structure):
    """struct tm_sampling"""
    _fields_ = [('temperature', c_float), ('top_k', c_int), ('top_p', c_float), ('min_p', c_float), ('seed', c_uint64)]


MAX_BAD_IDS, MAX_STOP_IDS = 32, 8


class LogitsParam(C.Structure):
    """struct tm_logits_param"""
    _fields_ = [('repetition_penalty', c_float), ('min_new_tokens', c_int), ('n_bad_ids', c_int),
                ('bad_ids', c_int * MAX_BAD_IDS), ('n_stop_ids', c_int), ('stop_ids', c_int * MAX_STOP_IDS)]

    @classmethod
    def make(cls, repetition_penalty=1.0, min_new_tokens=0, bad_ids=(), stop_ids=()):
        bad_ids, stop_ids = list(bad_ids or ()), list(stop_ids or ())
        if len(bad_ids) > MAX_BAD_IDS or len(stop_ids) > MAX_STOP_IDS:
            raise ValueError(f'at most {MAX_BAD_IDS} bad ids and {MAX_STOP_IDS} stop ids per sequence')
        p = cls(float(repetition_penalty), int(min_new_tokens or 0), len(bad_ids), (c_int * MAX_BAD_IDS)(*bad_ids),
                len(stop_ids), (c_int * MAX_STOP_IDS)(*stop_ids))
        return p


class KvCache(C.Structure):
    """struct tm_kv_cache"""
    _fields_ = [('block_ptrs', c_void_p), ('cu_block_nums', c_void_p), ('layer_offset', c_int64),
                ('kv_heads', c_int), ('head_dim', c_int), ('block_len', c_int), ('bits', c_int)]


class ModelConfig(C.Structure):
    """struct tm_model_config"""
    _fields_ = [('hidden', c_int), ('layers', c_int), ('q_heads', c_int), ('kv_heads', c_int), ('head_dim', c_int),
                ('inter', c_int), ('vocab', c_int), ('rms_eps', c_float), ('rope_base', c_float),
                ('rope_type', c_int), ('rope_factor', c_float), ('rope_low_freq_factor', c_float),
                ('rope_high_freq_factor', c_float), ('rope_original_max_position', c_int), ('group_size', c_int),
                ('weight_type', c_int), ('moe_experts', c_int), ('moe_top_k', c_int), ('moe_norm_topk', c_int),
                ('moe_routed_scale', c_float)]


class EngineConfig(C.Structure):
    """struct tm_engine_config"""
    _fields_ = [('model', ModelConfig), ('tp', c_int), ('rank', c_int), ('device', c_int),
                ('max_batch_size', c_int), ('session_len', c_int), ('quant_policy', c_int),
                ('cache_block_seq_len', c_int), ('cache_max_entry_count', c_float), ('cache_blocks', c_int),
                ('max_prefill_token_num', c_int), ('decode_splits', c_int), ('use_graph', c_int)]


# name -> (restype, argtypes); every int-returning function is a tm_status
_SIGNATURES = {
    'tm_version': (c_int, []),
    'tm_last_error': (c_char_p, []),
    'tm_device_count': (c_int, []),
    'tm_kv_layer_size': (c_int64, [c_int, c_int, c_int, c_int]),
    'tm_rmsnorm': (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_int, c_int, c_void_p]),
    'tm_residual_rmsnorm': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_float,
                                    c_int, c_int, c_void_p]),
    'tm_rope_table': (c_int, [c_void_p, c_int, c_int, c_float, c_int, c_float, c_float, c_float, c_int]),
    'tm_kv_rope_store': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int,
                                 POINTER(KvCache), c_void_p]),
    'tm_flatten_kv': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, POINTER(KvCache),
                              c_void_p]),
    'tm_decode_attention_workspace': (c_size_t, [c_int, c_int, c_int]),
    'tm_decode_attention': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_float, c_int, c_void_p,
                                    POINTER(KvCache), c_void_p]),
    'tm_decode_attention_fused': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_int,
                                          c_float, c_int, c_void_p, POINTER(KvCache), c_void_p]),
    'tm_prefill_attention': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                     c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    'tm_embedding': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    'tm_argmax': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    'tm_silu_mul': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    'tm_sample_workspace': (c_size_t, [c_int]),
    'tm_moe_create': (c_int, [POINTER(c_void_p), c_int, c_int, c_int, c_int, c_int, c_int, c_float]),
    'tm_moe_set_gate': (c_int, [c_void_p, c_void_p, c_void_p]),
    'tm_moe_set_expert': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'tm_moe_workspace': (c_size_t, [c_void_p, c_int]),
    'tm_moe_forward': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    'tm_moe_destroy': (c_int, [c_void_p]),
    'tm_sample': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                          c_void_p, c_void_p]),
    'tm_sample_logprobs': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'tm_philox_uniform': (C.c_float, [C.c_uint64, C.c_uint32]),
    'tm_linear_create': (c_int, [POINTER(c_void_p), c_int, c_int, c_int, c_int]),
    'tm_linear_prepare': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'tm_linear_workspace': (c_size_t, [c_void_p, c_int]),
    'tm_linear_forward': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                  c_void_p, c_void_p]),
    'tm_linear_dequant_f16': (c_int, [c_void_p, c_void_p, c_void_p]),
    'tm_linear_residual_norm': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_float, c_int, c_int, c_int,
                                        c_void_p, c_void_p]),
    'tm_linear_fold_workspace': (c_size_t, [c_void_p, c_int]),
    'tm_linear_fold_produce': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, POINTER(c_int), c_int, c_int,
                                       c_int, c_void_p, c_void_p]),
    'tm_linear_fold_consume': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_float, c_int,
                                       c_int, c_void_p, c_void_p]),
    'tm_linear_destroy': (c_int, [c_void_p]),
    'tm_linear_prepare_fp8_gated': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    'tm_linear_fp8_workspace': (c_size_t, [c_void_p, c_int]),
    'tm_quant_fp8_rows': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    'tm_linear_forward_fp8': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    'tm_quantize_groupwise': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                      c_void_p]),
    'tm_debug_set_gemm_trace': (c_int, [c_void_p]),
    'tm_debug_trace_arena': (c_int, [c_void_p, c_int64]),
    'tm_debug_trace_records': (c_int64, [c_char_p, c_int64]),
    'tm_engine_tune_gemm': (c_int, [c_void_p, c_int, c_char_p]),
    'tm_gemm_import': (c_int, [c_char_p]),
    'tm_gemm_export': (c_int, [c_char_p]),
    'tm_debug_set_block_stride': (c_int, [c_int]),
    'tm_debug_pick_tiling': (c_int, [c_int, c_int, c_int, c_int, POINTER(c_int), POINTER(c_int)]),
    'tm_debug_pick_general': (c_int, [c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'tm_debug_grouped_tile': (c_int, [c_int, c_int, c_int, c_int, POINTER(c_int)]),
    'tm_engine_comm_drop_rccl': (c_int, [c_void_p]),
    'tm_engine_comm_info': (c_int, [c_void_p, POINTER(c_int), POINTER(c_int), POINTER(c_int)]),
    'tm_prefill_split': (c_int, [c_void_p, c_int, c_int, POINTER(c_int), POINTER(c_int)]),
    'tm_engine_comm_overlap_info': (c_int, [c_void_p, POINTER(c_int), POINTER(c_int64), POINTER(c_int64), POINTER(c_int64)]),
    'tm_debug_tiling_candidates': (c_int, [c_int, c_int, c_int, c_void_p, c_void_p, c_int, POINTER(c_int)]),
    'tm_engine_comm_native_export': (c_int, [c_void_p, c_int, c_void_p]),
    'tm_engine_comm_native_import': (c_int, [c_void_p, c_void_p, c_int]),
    'tm_engine_comm_native_selftest': (c_int, [c_void_p, POINTER(c_int)]),
    'tm_engine_comm_native_drop': (c_int, [c_void_p]),
    'tm_p2p_segment_create': (c_int, [c_size_t, POINTER(c_void_p), c_void_p]),
    'tm_p2p_segment_open': (c_int, [c_void_p, POINTER(c_void_p)]),
    'tm_p2p_segment_close': (c_int, [c_void_p, c_int]),
    'tm_p2p_segment_bytes': (c_size_t, [c_int, c_int]),
    'tm_p2p_allreduce_norm': (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int,
                               c_int, c_void_p]),
    'tm_p2p_segment_bytes2': (c_size_t, [c_int, c_int, c_int]),
    'tm_p2p_segment_bytes_rows': (c_size_t, [c_int, c_int, c_int]),
    'tm_p2p_allreduce_norm_rows': (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_float,
                                           c_int, c_int, c_void_p]),
    'tm_p2p_allreduce_norm_2shot': (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_float,
                                            c_int, c_int, c_void_p]),
    'tm_p2p_allgather': (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    'tm_engine_create': (c_int, [POINTER(c_void_p), POINTER(EngineConfig)]),
    'tm_engine_destroy': (c_int, [c_void_p]),
    'tm_comm_unique_id': (c_int, [c_void_p]),
    'tm_engine_comm_init': (c_int, [c_void_p, c_void_p]),
    'tm_engine_weight_bytes': (c_int64, [c_void_p, c_char_p]),
    'tm_engine_weight_copy': (c_int, [c_void_p, c_char_p, c_void_p, c_int64]),
    'tm_engine_init_synthetic': (c_int, [c_void_p, c_uint64]),
    'tm_engine_process_weights': (c_int, [c_void_p]),
    'tm_engine_start': (c_int, [c_void_p]),
    'tm_engine_prefill': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int]),
    'tm_engine_decode': (c_int, [c_void_p, c_int]),
    'tm_engine_sync': (c_int, [c_void_p]),
    'tm_engine_prefill_times': (c_int, [c_void_p, c_void_p]),
    'tm_engine_profile_decode': (c_int, [c_void_p, c_int, c_void_p, c_void_p]),
    'tm_engine_fetch': (c_int, [c_void_p, c_void_p, POINTER(c_int)]),
    'tm_engine_fetch_logits': (c_int, [c_void_p, c_void_p]),
    'tm_engine_debug_read': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int64]),
    'tm_engine_release': (c_int, [c_void_p]),
    'tm_engine_set_sampling': (c_int, [c_void_p, c_void_p, c_int]),
    'tm_engine_set_logits_params': (c_int, [c_void_p, c_void_p, c_int]),
    'tm_engine_set_logprobs': (c_int, [c_void_p, c_int]),
    'tm_engine_fetch_logprobs': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'tm_engine_request_logprobs': (c_int, [c_void_p, c_int64, c_int]),
    'tm_engine_poll_logprobs': (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int, POINTER(c_int), POINTER(c_int)]),
    'tm_engine_submit_gen': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, POINTER(c_int64)]),
    'tm_seen_update': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    'tm_logits_process': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_void_p, c_void_p]),
    'tm_engine_submit_ex': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, POINTER(c_int64)]),
    'tm_engine_submit': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, POINTER(c_int64)]),
    'tm_engine_step': (c_int, [c_void_p, POINTER(c_int), POINTER(c_int)]),
    'tm_engine_step_many': (c_int, [c_void_p, c_int, POINTER(c_int), POINTER(c_int), POINTER(c_int)]),
    'tm_engine_poll': (c_int, [c_void_p, c_int64, POINTER(c_int), c_void_p, c_int, POINTER(c_int)]),
    'tm_engine_cancel': (c_int, [c_void_p, c_int64]),
    'tm_engine_forget': (c_int, [c_void_p, c_int64]),
    'tm_sched_forget': (c_int, [c_void_p, c_int64]),
    'tm_engine_serve_start': (c_int, [c_void_p, c_void_p, c_void_p]),
    'tm_engine_serve_stop': (c_int, [c_void_p]),
    'tm_engine_wait': (c_int, [c_void_p, c_int64, c_int, c_int, POINTER(c_int), POINTER(c_int)]),
    'tm_sched_abort_all': (c_int, [c_void_p, c_int]),
    'tm_sched_create': (c_int, [POINTER(c_void_p), c_int, c_int, c_int]),
    'tm_sched_destroy': (c_int, [c_void_p]),
    'tm_sched_submit': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, POINTER(c_int64)]),
    'tm_sched_admit': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, POINTER(c_int)]),
    'tm_sched_on_token': (c_int, [c_void_p, c_int, c_int, POINTER(c_int)]),
    'tm_sched_admit_ready': (c_int, [c_void_p, POINTER(c_int)]),
    'tm_sched_cancel': (c_int, [c_void_p, c_int64, POINTER(c_int)]),
    'tm_sched_query': (c_int, [c_void_p, c_int64, POINTER(c_int), POINTER(c_int), POINTER(c_int), POINTER(c_int)]),
    'tm_sched_counts': (c_int, [c_void_p, POINTER(c_int), POINTER(c_int), POINTER(c_int)]),
    'tm_engine_stream': (c_void_p, [c_void_p]),
    'tm_engine_stats': (c_int, [c_void_p, POINTER(c_int64), POINTER(c_int64), POINTER(c_int64), POINTER(c_int)]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None


def lib_path() -> str:
    return _LIB_PATH


def load():
    """Load the native library (raises if it has not been built: no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise ImportError(f'{_LIB_PATH} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                          f'or `make -C lmdeploy_amd/csrc`. There is no CPU fallback.')
    # Load order: PyTorch-ROCm bundles its own libamdhip64 / libhsa-runtime64; when this library (linked against
    # /opt/rocm) is loaded FIRST and torch later, the process ends up with mismatched runtimes and aborts at exit
    # ("double free or corruption").  torch first is the order every GPU test and bench run uses.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(_LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error() -> str:
    msg = load().tm_last_error()
    return msg.decode() if msg else ''


def check(status: int):
    if status != 0:
        raise TmError(status, last_error())


def ptr(t) -> int:
    """Device (or host) address of a torch tensor / numpy array, None -> NULL."""
    if t is None:
        return None
    if hasattr(t, 'data_ptr'):
        return t.data_ptr()
    return t.ctypes.data


def stream_ptr(stream=None):
    if stream is None:
        import torch
        return torch.cuda.current_stream().cuda_stream
    return stream
