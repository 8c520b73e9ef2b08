"""Shared helpers for the GPU parity tests: device buffers via torch (plumbing only) + paged-cache builders
that mirror oracle.tm_oracle.PagedKVCache byte for byte."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from lmdeploy_amd import _ffi
from oracle import tm_oracle as o


_KEEP = []   # tensors whose raw pointers were handed to the C-ABI must outlive the (asynchronous) call


def dev(a: np.ndarray) -> torch.Tensor:
    """numpy -> device tensor with the same bytes (fp16 stays fp16, ints stay ints).  The tensor is kept alive
    until release_all() (called after every test): `dev(x).data_ptr()` would otherwise free the buffer before the
    kernel that reads it has even been launched."""
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    _KEEP.append(t)
    return t


def release_all():
    torch.cuda.synchronize()
    _KEEP.clear()


def host(t: torch.Tensor) -> np.ndarray:
    return t.detach().cpu().numpy()


def st():
    return torch.cuda.current_stream().cuda_stream


class DevCache:
    """A device copy of an oracle PagedKVCache pool plus block tables for `tables` (list of index arrays)."""

    def __init__(self, layout: o.BlockLayout, num_blocks: int, tables):
        self.layout = layout
        self.pool = torch.zeros((num_blocks, layout.block_size), dtype=torch.uint8, device='cuda')
        self.set_tables(tables)

    def set_tables(self, tables, stride: int = 0):
        """stride > 0: rectangular table, every sequence owns `stride` entries (unused ones are null pointers)"""
        self.tables = [np.asarray(t, np.int64) for t in tables]
        base = self.pool.data_ptr()
        rows = [base + t * self.layout.block_size for t in self.tables]
        if stride > 0:
            rows = [np.concatenate([r, np.zeros(stride - len(r), np.int64)]) for r in rows]
        ptrs = np.concatenate(rows).astype(np.uint64)
        cu = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int32)
        self.block_ptrs = torch.from_numpy(ptrs.view(np.int64)).cuda()
        self.cu_block_nums = torch.from_numpy(cu).cuda()

    def view(self, layer: int) -> _ffi.KvCache:
        L = self.layout
        return _ffi.KvCache(self.block_ptrs.data_ptr(), self.cu_block_nums.data_ptr(), layer * L.layer_size,
                            L.kv_heads, L.head_dim, L.block_len, L.bits)

    def upload(self, oracle_cache: o.PagedKVCache):
        self.pool.copy_(torch.from_numpy(oracle_cache.pool).cuda())

    def download(self) -> np.ndarray:
        return host(self.pool)


def rope_table(tm, max_pos, p: o.RopeParam) -> np.ndarray:
    tab = np.zeros((max_pos, p.dim // 2, 2), np.float16)
    rt = {'default': 0, 'linear': 1, 'llama3': 2}[p.type]
    _ffi.check(tm.tm_rope_table(tab.ctypes.data, max_pos, p.dim, p.base, rt, p.factor, p.low_freq_factor,
                                p.high_freq_factor, p.original_max_position_embeddings))
    return tab


def ulp_diff_f16(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """distance in fp16 representable steps (sign-magnitude aware)."""
    ai = a.astype(np.float16).view(np.int16).astype(np.int32)
    bi = b.astype(np.float16).view(np.int16).astype(np.int32)
    ai = np.where(ai < 0, -(ai & 0x7fff), ai)
    bi = np.where(bi < 0, -(bi & 0x7fff), bi)
    return np.abs(ai - bi)
