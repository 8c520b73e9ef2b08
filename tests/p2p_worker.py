"""Driver of the native P2P all-reduce + residual + RMSNorm parity runs (not a test module: tests/test_gpu_p2p.py starts it
in a subprocess with a hard timeout, because a protocol bug shows up as ranks waiting for each other).

  (append `2` to either form: the TWO-SHOT all-reduce for large messages -- reduce-scatter, norm on the owned row slice,
   all-gather by push -- instead of the one-shot exchange; the residual is then checked on every rank's own slice)
  streams <tp> <M> <H> <calls>      one process: tp "ranks" = tp segments on cuda:0, one stream each, peers addressed through
                                    the same pointer tables a multi-GPU run uses (needs GPU_MAX_HW_QUEUES >= tp so that the tp
                                    kernels really run concurrently)
  ipc <dir> <rank> <tp> <M> <H> <calls>   one process per rank (all on cuda:0): segments exchanged as IPC handles through
                                    files in <dir>, peers mapped with tm_p2p_segment_open -- the multi-process plumbing of
                                    a real TP run minus the xGMI hop
Prints one JSON line {"ok": bool, ...}."""
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lmdeploy_amd import _ffi            # noqa: E402
from oracle import tm_oracle as o        # noqa: E402

f16 = np.float16


def _inputs(tp, M, H, call):
    rng = np.random.default_rng(1000 * tp + 10 * M + call)
    parts = [(0.5 * rng.standard_normal((M, H))).astype(f16) for _ in range(tp)]
    return parts


def _tables(ptrs):
    return (ctypes.c_void_p * len(ptrs))(*ptrs)


def _check(tag, c, state, resid, y, r_ref, y_ref, gathered, g_ref, rows=None, rowform=0):
    """rows: (lo, hi) of the residual rows this rank keeps current (two-shot form), None = all; the two-shot call takes two epochs"""
    st = state.cpu().numpy()
    if st[3] != 0:
        return f'{tag} timed out waiting in call {st[3]}'
    per_call = 1 if rowform else (3 if rows else 2)       # row-flag form: only the all-gather uses the shared call counter
    if st[0] != per_call * (c + 1) or st[1] != 0 or st[2] != 0:
        return f'{tag} state after call {c}: {st[:4].tolist()}'
    if rowform and not np.all(st[4:4 + rowform] == c + 1):
        return f'{tag} per-row call counters after call {c}: {st[4:4 + rowform].tolist()}'
    lo, hi = rows or (0, r_ref.shape[0])
    if not np.array_equal(resid.cpu().numpy()[lo:hi].view(np.uint16), r_ref[lo:hi].view(np.uint16)):
        return f'{tag} residual differs from the oracle in call {c}'
    # y = h(h(r * inv) * w): the kernel's fp32 sum of squares associates differently from numpy's, so n = h(r * inv) may sit
    # one fp16 ulp off on a rounding boundary, and one ulp of n is up to two ulps of y when n * w drops a binade
    d = np.abs(y.cpu().numpy().view(np.int16).astype(np.int32) - y_ref.view(np.int16).astype(np.int32))
    if d.max() > 2 or (d > 0).mean() > 1e-3:
        return f'{tag} normed output off by {d.max()} ulp ({(d > 0).mean():.2e} of the elements) in call {c}'
    if not np.array_equal(gathered.cpu().numpy(), g_ref):
        return f'{tag} all-gather differs in call {c}'
    return None


def _gather_src(tp, words, c):
    return [np.random.default_rng(77 * c + r).integers(0, 2**31, words).astype(np.int32) for r in range(tp)]


def _slice(tp, M, r):
    sl = (M + tp - 1) // tp
    return (min(M, r * sl), min(M, (r + 1) * sl))


def run_streams(tp, M, H, calls, two=0):
    tm = _ffi.load()
    rng = np.random.default_rng(5)
    w = (1 + 0.02 * rng.standard_normal(H)).astype(f16)
    resid0 = rng.standard_normal((M, H)).astype(f16)
    w_d = torch.from_numpy(w).cuda()
    rowform = two == 3                        # the row-flag one-shot form (tm_p2p_allreduce_norm_rows): own tiles, own flags, per-row counters
    two = two if two != 3 else 0
    rows = (M + 3) if not two else 8          # segment capacity need not equal M; two-shot: small one-shot tiles, big regions
    rows2 = M + 5
    words = 2 * M if not two else 2 * rows
    nbytes = tm.tm_p2p_segment_bytes_rows(rows, 7, H) if rowform else tm.tm_p2p_segment_bytes2(rows, rows2 if two else 0, H)
    seg = [torch.zeros(nbytes, dtype=torch.uint8, device='cuda') for _ in range(tp)]
    state = [torch.zeros(4 + (rows if rowform else 0), dtype=torch.int32, device='cuda') for _ in range(tp)]
    resid = [torch.from_numpy(resid0).cuda() for _ in range(tp)]
    y = [torch.empty((M, H), dtype=torch.float16, device='cuda') for _ in range(tp)]
    gathered = [torch.zeros((tp, words), dtype=torch.int32, device='cuda') for _ in range(tp)]
    streams = [torch.cuda.Stream() for _ in range(tp)]
    torch.cuda.synchronize()
    segs = _tables([s.data_ptr() for s in seg])
    r_ref = resid0
    for c in range(calls):
        parts = _inputs(tp, M, H, c)
        gsrc = _gather_src(tp, words, c)
        part_d = [torch.from_numpy(p).cuda() for p in parts]
        gsrc_d = [torch.from_numpy(g).cuda() for g in gsrc]
        torch.cuda.synchronize()
        for r in range(tp):                   # every rank: fused all-reduce + norm, then the all-gather, back to back
            if two:
                _ffi.check(tm.tm_p2p_allreduce_norm_2shot(segs, tp, r, state[r].data_ptr(), rows, rows2, part_d[r].data_ptr(), y[r].data_ptr(),
                                                         resid[r].data_ptr(), w_d.data_ptr(), 1e-5, M, H, streams[r].cuda_stream))
            elif rowform:
                _ffi.check(tm.tm_p2p_allreduce_norm_rows(segs, tp, r, state[r].data_ptr(), rows, 7, part_d[r].data_ptr(), y[r].data_ptr(),
                                                        resid[r].data_ptr(), w_d.data_ptr(), 1e-5, M, H, streams[r].cuda_stream))
            else:
                _ffi.check(tm.tm_p2p_allreduce_norm(segs, tp, r, state[r].data_ptr(), rows, part_d[r].data_ptr(), y[r].data_ptr(),
                                                   resid[r].data_ptr(), w_d.data_ptr(), 1e-5, M, H, streams[r].cuda_stream))
            _ffi.check(tm.tm_p2p_allgather(segs, tp, r, state[r].data_ptr(), rows, H, gsrc_d[r].data_ptr(), gathered[r].data_ptr(),
                                          words, streams[r].cuda_stream))
        torch.cuda.synchronize()
        r_ref, y_ref = o.p2p_allreduce_norm(parts, r_ref, w, 1e-5)
        for r in range(tp):
            why = _check(f'rank {r}', c, state[r], resid[r], y[r], r_ref, y_ref, gathered[r], np.stack(gsrc), _slice(tp, M, r) if two else None,
                         rowform=M if rowform else 0)
            if why:
                return {'ok': False, 'why': why}
            if not torch.equal(y[r], y[0]):
                return {'ok': False, 'why': f'rank {r} and rank 0 disagree in call {c}'}
    return {'ok': True, 'tp': tp, 'calls': calls}


def _wait_file(path, timeout=60.0):
    t0 = time.time()
    while not os.path.exists(path):
        if time.time() - t0 > timeout:
            raise TimeoutError(path)
        time.sleep(0.01)


def _barrier(d, rank, tp, tag):
    open(os.path.join(d, f'{tag}.{rank}.tmp'), 'w').close()
    os.rename(os.path.join(d, f'{tag}.{rank}.tmp'), os.path.join(d, f'{tag}.{rank}'))
    for r in range(tp):
        _wait_file(os.path.join(d, f'{tag}.{r}'))


def run_ipc(d, rank, tp, M, H, calls, two=0):
    tm = _ffi.load()
    torch.cuda.init()
    torch.zeros(1, device='cuda')
    rng = np.random.default_rng(5)
    w = (1 + 0.02 * rng.standard_normal(H)).astype(f16)
    resid0 = rng.standard_normal((M, H)).astype(f16)
    rowform = two == 3
    two = two if two != 3 else 0
    rows, words = (M, 2 * M) if not two else (8, 16)
    rows2 = M
    mine = ctypes.c_void_p()
    handle = ctypes.create_string_buffer(64)
    nbytes = tm.tm_p2p_segment_bytes_rows(rows, 0, H) if rowform else tm.tm_p2p_segment_bytes2(rows, rows2 if two else 0, H)
    _ffi.check(tm.tm_p2p_segment_create(nbytes, ctypes.byref(mine), handle))
    with open(os.path.join(d, f'handle.{rank}.tmp'), 'wb') as f:
        f.write(handle.raw)
    os.rename(os.path.join(d, f'handle.{rank}.tmp'), os.path.join(d, f'handle.{rank}'))
    base = []
    for r in range(tp):
        if r == rank:
            base.append(mine.value)
            continue
        _wait_file(os.path.join(d, f'handle.{r}'))
        peer = ctypes.c_void_p()
        _ffi.check(tm.tm_p2p_segment_open(open(os.path.join(d, f'handle.{r}'), 'rb').read(), ctypes.byref(peer)))
        base.append(peer.value)
    _barrier(d, rank, tp, 'mapped')
    w_d = torch.from_numpy(w).cuda()
    state = torch.zeros(4 + (rows if rowform else 0), dtype=torch.int32, device='cuda')
    resid = torch.from_numpy(resid0).cuda()
    y = torch.empty((M, H), dtype=torch.float16, device='cuda')
    gathered = torch.zeros((tp, words), dtype=torch.int32, device='cuda')
    segs = _tables(base)
    r_ref = resid0
    stream = torch.cuda.current_stream().cuda_stream
    out = {'ok': True, 'tp': tp, 'rank': rank}
    for c in range(calls):
        parts = _inputs(tp, M, H, c)
        gsrc = _gather_src(tp, words, c)
        part_d = torch.from_numpy(parts[rank]).cuda()
        gsrc_d = torch.from_numpy(gsrc[rank]).cuda()
        if two:
            _ffi.check(tm.tm_p2p_allreduce_norm_2shot(segs, tp, rank, state.data_ptr(), rows, rows2, part_d.data_ptr(), y.data_ptr(),
                                                     resid.data_ptr(), w_d.data_ptr(), 1e-5, M, H, stream))
        elif rowform:
            _ffi.check(tm.tm_p2p_allreduce_norm_rows(segs, tp, rank, state.data_ptr(), rows, 0, part_d.data_ptr(), y.data_ptr(), resid.data_ptr(),
                                                    w_d.data_ptr(), 1e-5, M, H, stream))
        else:
            _ffi.check(tm.tm_p2p_allreduce_norm(segs, tp, rank, state.data_ptr(), rows, part_d.data_ptr(), y.data_ptr(), resid.data_ptr(),
                                               w_d.data_ptr(), 1e-5, M, H, stream))
        _ffi.check(tm.tm_p2p_allgather(segs, tp, rank, state.data_ptr(), rows, H, gsrc_d.data_ptr(), gathered.data_ptr(), words,
                                      stream))
        torch.cuda.synchronize()
        r_ref, y_ref = o.p2p_allreduce_norm(parts, r_ref, w, 1e-5)
        why = _check(f'rank {rank}', c, state, resid, y, r_ref, y_ref, gathered, np.stack(gsrc), _slice(tp, M, rank) if two else None,
                     rowform=M if rowform else 0)
        if why:
            out = {'ok': False, 'why': why}
            break
    _barrier(d, rank, tp, 'done')            # nobody unmaps / frees while a peer may still read
    for r in range(tp):
        if r != rank:
            _ffi.check(tm.tm_p2p_segment_close(ctypes.c_void_p(base[r]), 1))
    _barrier(d, rank, tp, 'closed')
    _ffi.check(tm.tm_p2p_segment_close(mine, 0))
    return out


if __name__ == '__main__':
    try:
        if sys.argv[1] == 'streams':
            res = run_streams(*[int(a) for a in sys.argv[2:7]])
        else:
            res = run_ipc(sys.argv[2], *[int(a) for a in sys.argv[3:9]])
    except Exception as e:      # noqa: BLE001
        res = {'ok': False, 'why': f'{type(e).__name__}: {e}'}
    print(json.dumps(res), flush=True)
