#!/usr/bin/env python
"""Time the paged decode attention kernel at the metric shape (B=64, 32 q / 8 kv heads, int8 KV) on random cache
contents.  python tools/bench_attention.py [--ctx 1536] [--bits 8] [--splits 1,2,4]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lmdeploy_amd import _ffi  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--ctx', type=int, default=1536)
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--hq', type=int, default=32)
    ap.add_argument('--hkv', type=int, default=8)
    ap.add_argument('--bits', type=int, default=8)
    ap.add_argument('--splits', default='1,2,4')
    ap.add_argument('--iters', type=int, default=20)
    ap.add_argument('--trace', action='store_true')
    ap.add_argument('--layers', type=int, default=4, help='layers stored in the pool (32 = the engine\'s block stride)')
    ap.add_argument('--layout', default='block', choices=['block', 'layer'],
                    help='block: [block][layer] (reference: all layers of a block contiguous); layer: [layer][block]')
    a = ap.parse_args()
    tm = _ffi.load()
    B, Hq, Hkv, ctx, bits = a.batch, a.hq, a.hkv, a.ctx, a.bits
    layers = a.layers   # several layers so that consecutive launches touch different (cold) cache bytes
    lsz = tm.tm_kv_layer_size(Hkv, 128, 64, bits)
    nblk = (ctx + 63) // 64
    pool = torch.randint(0, 255, (B * nblk, layers * lsz), dtype=torch.uint8, device='cuda')
    # make the per-token (scale, zero) params finite fp16 values
    pv = pool.view(B * nblk, layers, lsz) if a.layout == 'block' else pool.view(-1).view(layers, B * nblk, lsz)
    if bits < 16:
        nparam = Hkv * 2 * 64 * 4
        prm = torch.empty((pv.shape[0], pv.shape[1], nparam // 2), dtype=torch.float16, device='cuda').uniform_(0.01, 0.05)
        pv[:, :, lsz - nparam:] = prm.view(torch.uint8).view(pv.shape[0], pv.shape[1], nparam)
    perm = torch.randperm(B * nblk)
    if a.layout == 'block':
        ptrs_of = lambda layer: (pool.data_ptr() + perm.to(torch.int64) * layers * lsz).cuda()   # + layer offset in the view
        off_of = lambda layer: layer * lsz
    else:   # layer-major: the same bytes, block b of layer l at (l * nblocks + b) * lsz
        ptrs_of = lambda layer: (pool.data_ptr() + (layer * B * nblk + perm.to(torch.int64)) * lsz).cuda()
        off_of = lambda layer: 0
    tables = [ptrs_of(l) for l in range(layers)]
    ptrs = tables[0]
    cu = torch.arange(0, (B + 1) * nblk, nblk, dtype=torch.int32, device='cuda')
    klen = torch.full((B,), ctx, dtype=torch.int32, device='cuda')
    q = torch.randn((B, Hq * 128), device='cuda').half()
    out = torch.empty((B, Hq * 128), device='cuda').half()
    st = torch.cuda.current_stream().cuda_stream
    bytes_per_launch = B * ctx * 2 * Hkv * (128 * bits // 8 + (4 if bits < 16 else 0))
    if a.trace:
        # in-kernel phase stamps (100 MHz): start / prologue done (q ready, first block issued) / wave 0 done /
        # all waves done / end, per workgroup
        splits = int(a.splits.split(',')[0])
        ws = torch.empty(max(1, tm.tm_decode_attention_workspace(B, Hq, splits)), dtype=torch.uint8, device='cuda')
        dbg = torch.zeros((8192, 8), dtype=torch.int64, device='cuda')
        for it in range(4):
            dbg.zero_()
            torch.cuda.synchronize()
            tm.tm_debug_set_gemm_trace(dbg.data_ptr())
            view = _ffi.KvCache(tables[it % layers].data_ptr(), cu.data_ptr(), off_of(it % layers), Hkv, 128, 64, bits)
            _ffi.check(tm.tm_decode_attention(out.data_ptr(), q.data_ptr(), Hq * 128, klen.data_ptr(), B, Hq, 0.0, splits,
                                              ws.data_ptr(), view, st))
            tm.tm_debug_set_gemm_trace(None)
            torch.cuda.synchronize()
            raw = dbg.cpu().numpy()
            t = raw[raw[:, 0] > 0].astype(np.float64) / 100.0
            t0 = t[:, 0].min()
            print(f'launch {it}: {len(t)} workgroups, span {t[:, 3].max() - t0:.2f} us')
            for name, v in (('start skew', t[:, 0] - t0), ('prologue', t[:, 1] - t[:, 0]), ('wave0 loop', t[:, 2] - t[:, 1]),
                            ('wait waves', t[:, 5] - t[:, 2]), ('merge+store', t[:, 3] - t[:, 5]), ('end time', t[:, 3] - t0)):
                print(f'  {name:11s} min {v.min():6.2f}  mean {v.mean():6.2f}  p90 {np.percentile(v, 90):6.2f}  max {v.max():6.2f} us')
            # where do the late workgroups sit?  end time by XCD, by CU, by (sequence, kv head)
            live = raw[raw[:, 0] > 0]
            end = live[:, 3].astype(np.float64) / 100.0 - t0
            xcc = ((live[:, 4] >> 32) & 0xf).astype(int)
            hw = (live[:, 4] & 0xffffffff).astype(np.int64)
            cuid = ((hw >> 8) & 0xf) | (((hw >> 13) & 0x7) << 4) | (xcc << 7)      # CU_ID | SE_ID << 4 | XCC << 7
            start = live[:, 0].astype(np.float64) / 100.0 - t0
            print('  end time by XCD: ' + '  '.join(f'{x}: {end[xcc == x].mean():5.1f} (n {int((xcc == x).sum())})' for x in range(8)))
            print('  start    by XCD: ' + '  '.join(f'{x}: {start[xcc == x].mean():5.2f}' for x in range(8)))
            print('  max end  by XCD: ' + '  '.join(f'{x}: {end[xcc == x].max():5.1f}' for x in range(8)))
            per_cu = {}
            for c, e_ in zip(cuid, end):
                per_cu.setdefault(int(c), []).append(e_)
            occ = np.bincount([len(v) for v in per_cu.values()])
            print(f'  CUs used {len(per_cu)}; workgroups per CU histogram {occ.tolist()}; mean end of CUs with 1 / 2 / 3+ WGs: ' +
                  ' / '.join(f"{np.mean([np.max(v) for v in per_cu.values() if (len(v) if len(v) < 3 else 3) == k]) if any((len(v) if len(v) < 3 else 3) == k for v in per_cu.values()) else float('nan'):.1f}" for k in (1, 2, 3)))
            order = np.argsort(live[:, 0])
            nq4 = len(order) // 4
            print('  end time by launch-order quartile: ' + '  '.join(f'{end[order[i * nq4:(i + 1) * nq4]].mean():5.1f}' for i in range(4)))
        return
    for splits in [int(s) for s in a.splits.split(',')]:
        ws = torch.empty(max(1, tm.tm_decode_attention_workspace(B, Hq, splits)), dtype=torch.uint8, device='cuda')
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.iters)]
        for i, (e0, e1) in enumerate(ev):
            view = _ffi.KvCache(tables[i % layers].data_ptr(), cu.data_ptr(), off_of(i % layers), Hkv, 128, 64, bits)
            e0.record()
            _ffi.check(tm.tm_decode_attention(out.data_ptr(), q.data_ptr(), Hq * 128, klen.data_ptr(), B, Hq, 0.0, splits,
                                              ws.data_ptr(), view, st))
            e1.record()
        torch.cuda.synchronize()
        ts = sorted(x.elapsed_time(y) for x, y in ev)
        med = ts[len(ts) // 2]
        print(f'ctx={ctx} bits={bits} splits={splits}: {med*1e3:8.1f} us  {bytes_per_launch/(med*1e-3)/1e9:7.0f} GB/s '
              f'({bytes_per_launch/1e6:.1f} MB/launch)  layout={a.layout} layers={layers}', flush=True)


if __name__ == '__main__':
    main()
