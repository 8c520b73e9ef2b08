#!/usr/bin/env python
"""Per-launch fixed-cost table of the captured decode step: in-kernel s_memrealtime stamps (100 MHz, one clock
for the whole device) of EVERY workgroup of every kernel of the hipGraph-replayed step, through the trace arena
(tm_debug_trace_arena): each launch of the graph keeps its own stamp region, one replay fills them all.

  python tools/fixed_cost_table.py [--batch 64] [--prompt-len 1024] [--steps 24] [--layers 32] [--tune 1]

Columns (us, mean over the traced layers; per-workgroup phases are means over workgroups, `max` = slowest workgroup):
  gap      last store of the PREVIOUS launch -> first workgroup of this launch starts (the kernel boundary as the device sees it)
  skew     first -> last workgroup start
  issue    workgroup start -> its prologue loads are issued (kernarg, descriptors, address set-up)
  x_in     ... -> activations of stage 0 landed in LDS + first barrier (GEMMs) / q + first block ready (attention)
  w_in     ... -> wave 0's first weight unit landed: the first MFMA can issue
  loop     main loop (first stage barrier -> loop exit)
  merge    k-phase merge through LDS (GEMMs) / waves' tail + 4-wave merge (attention)
  store    epilogue stores drained
  tail     mean workgroup end -> last workgroup end
  span     first workgroup start -> last workgroup end        wall = gap + span
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--prompt-len', type=int, default=1024)
    ap.add_argument('--steps', type=int, default=24)
    ap.add_argument('--layers', type=int, default=32)
    ap.add_argument('--tune', type=int, default=1)
    ap.add_argument('--quant-policy', type=int, default=8)
    ap.add_argument('--per-layer', action='store_true', help='also print the wall of every layer')
    ap.add_argument('--attn-detail', action='store_true', help='straggler analysis of the decode attention workgroups')
    args = ap.parse_args()
    import ctypes as C

    import torch

    import bench
    from lmdeploy_amd import _ffi
    from lmdeploy_amd.turbomind.engine import Engine

    model = dict(bench.LLAMA3_8B)
    model['layers'] = args.layers
    B, S = args.batch, args.prompt_len
    eng = Engine.from_model_config(bench._Cfg(model), max_batch_size=B, session_len=S + args.steps + 16, quant_policy=args.quant_policy,
                                   max_prefill_token_num=8192, use_graph=1)
    eng.init_synthetic(seed=0)
    eng.start()
    if args.tune:
        eng.tune_gemm(B)
    gen = torch.Generator().manual_seed(0)
    prompts = torch.randint(0, model['vocab'], (B, S), generator=gen, dtype=torch.int32).numpy()
    eng.prefill(list(prompts), max_new_tokens=args.steps + 8)
    eng.sync()
    tm = _ffi.load()
    cap = 400000
    arena = torch.zeros((cap, 8), dtype=torch.int64, device='cuda')
    _ffi.check(tm.tm_debug_trace_arena(arena.data_ptr(), cap))
    eng.decode(args.steps)          # one eager step (its own records), the capture (the records we read), steps - 1 replays
    eng.sync()
    torch.cuda.synchronize()
    need = tm.tm_debug_trace_records(None, 0)
    buf = C.create_string_buffer(int(need) + 16)
    tm.tm_debug_trace_records(buf, need + 16)
    recs = []
    for line in buf.value.decode().splitlines():
        i, tag, gx, gy, gz, off = line.split()
        recs.append((tag, int(gx), int(gy), int(gz), int(off)))
    raw = arena.cpu().numpy().astype(np.int64)
    tm.tm_debug_trace_arena(None, 0)
    ctx = S + args.steps
    # the captured step = the records after the first lm_head (the eager step ends with it)
    heads = [i for i, r in enumerate(recs) if r[0] == 'lm_head']
    assert len(heads) >= 2, f'expected an eager and a captured step in the records, got {len(heads)} lm_head launches'
    step = recs[heads[-2] + 1:heads[-1] + 1]
    launches = []
    for tag, gx, gy, gz, off in step:
        n = gx * gy * gz
        d = raw[off:off + n].astype(np.float64) / 100.0      # us
        d[raw[off:off + n] == 0] = np.nan
        launches.append(dict(tag=tag, grid=(gx, gy, gz), d=d))
    t_origin = np.nanmin(launches[0]['d'][:, 0])
    rows = {}
    prev_end = None
    order = []
    layer_marks = []
    for L in launches:
        d = L['d']
        st, en = d[:, 0], d[:, 3]
        key = (L['tag'], L['grid'])
        if L['tag'] == 'w_qkv':
            layer_marks.append(np.nanmin(st))
        r = dict(gap=np.nan if prev_end is None else np.nanmin(st) - prev_end, skew=np.nanmax(st) - np.nanmin(st),
                 span=np.nanmax(en) - np.nanmin(st), tail=np.nanmax(en) - np.nanmean(en), wg=np.nanmean(en - st), wg_max=np.nanmax(en - st))
        if L['tag'] in ('w_qkv', 'wo', 'w1w3', 'w2'):
            r.update(issue=np.nanmean(d[:, 5] - st), x_in=np.nanmean(d[:, 1] - st), w_in=np.nanmean(d[:, 6] - st),
                     loop=np.nanmean(d[:, 2] - d[:, 1]), loop_max=np.nanmax(d[:, 2] - d[:, 1]), merge=np.nanmean(d[:, 7] - d[:, 2]),
                     store=np.nanmean(en - d[:, 7]))
        elif L['tag'] == 'attn':
            w_done = np.nanmax(np.stack([d[:, 2], d[:, 6], d[:, 7]]), axis=0)
            r.update(x_in=np.nanmean(d[:, 1] - st), loop=np.nanmean(d[:, 2] - d[:, 1]), loop_max=np.nanmax(d[:, 2] - d[:, 1]),
                     w1=np.nanmean(d[:, 6] - d[:, 1]), w3=np.nanmean(d[:, 7] - d[:, 1]), merge=np.nanmean(d[:, 5] - d[:, 2]),
                     store=np.nanmean(en - d[:, 5]), wave_spread=np.nanmean(w_done - np.nanmin(np.stack([d[:, 2], d[:, 6], d[:, 7]]), axis=0)))
        rows.setdefault(key, []).append(r)
        if key not in order:
            order.append(key)
        prev_end = np.nanmax(en)
    walls = np.diff(layer_marks)
    print(f'# fixed cost by launch: Llama-3-8B W4A16 + int{args.quant_policy} KV, batch {B}, ctx {ctx}, {args.layers} layers, hipGraph replay, '
          f's_memrealtime stamps of every workgroup (tools/fixed_cost_table.py)')
    print(f'# layer wall (w_qkv start -> next w_qkv start): mean {walls.mean():.2f} us, min {walls.min():.2f}, max {walls.max():.2f}; '
          f'step (first start -> last end) {np.nanmax(launches[-1]["d"][:, 3]) - t_origin:.1f} us')
    cols = ['gap', 'skew', 'issue', 'x_in', 'w_in', 'loop', 'loop_max', 'merge', 'store', 'tail', 'span', 'wg', 'wg_max']
    print(f'{"launch":12s} {"grid":>11s} {"n":>3s} ' + ' '.join(f'{c:>8s}' for c in cols) + '   wall=gap+span')
    tot = 0.0
    for key in order:
        rs = rows[key]
        sel = rs[2:-2] if len(rs) > 8 else rs      # drop the first / last layers (embedding / head neighbours)
        m = {c: np.nanmean([r.get(c, np.nan) for r in sel]) for c in cols + ['w1', 'w3', 'wave_spread']}
        wall = m['gap'] + m['span']
        if len(rs) > 8:
            tot += wall
        print(f'{key[0]:12s} {"x".join(map(str, key[1])):>11s} {len(rs):3d} ' + ' '.join(f'{m[c]:8.2f}' for c in cols) + f'   {wall:8.2f}')
        if key[0] == 'attn':
            print(f'{"":28s} attention waves: wave 0 loop {m["loop"]:.2f}, wave 1 {m["w1"]:.2f}, wave 3 {m["w3"]:.2f} us after the prologue; '
                  f'first -> last wave done {m["wave_spread"]:.2f} us')
    print(f'# sum of per-layer launches (wall): {tot:.2f} us')
    if args.attn_detail:
        # who are the stragglers of the decode attention?  per-workgroup duration by XCC, by kv head (blockIdx.x), by CU occupancy
        atts = [(tag, gx, gy, gz, off) for tag, gx, gy, gz, off in step if tag == 'attn']
        dur_all, xcc_all, cu_all, bx_all, st_all = [], [], [], [], []
        for tag, gx, gy, gz, off in atts[4:-4]:
            n = gx * gy * gz
            r = raw[off:off + n]
            dur_all.append((r[:, 3] - r[:, 0]) / 100.0)
            st_all.append((r[:, 0] - r[:, 0].min()) / 100.0)
            xcc_all.append((r[:, 4] >> 32) & 0xf)
            hwid = r[:, 4] & 0xffffffff
            cu_all.append(((hwid >> 8) & 0xff) | (((r[:, 4] >> 32) & 0xf) << 8))   # HW_ID bits 8..15 (CU_ID, SH_ID, SE_ID) | XCC
            bx_all.append(np.arange(n) % gx)
        dur, xcc, cu, bx, stt = map(np.concatenate, (dur_all, xcc_all, cu_all, bx_all, st_all))
        print(f'# attention workgroups ({len(dur)} over {len(atts) - 8} launches): duration mean {dur.mean():.2f} p50 {np.percentile(dur, 50):.2f} '
              f'p90 {np.percentile(dur, 90):.2f} p99 {np.percentile(dur, 99):.2f} max {dur.max():.2f} us')
        print('#   by XCC:      ' + ' '.join(f'{dur[xcc == k].mean():6.2f}' for k in range(8)))
        print('#   by kv head:  ' + ' '.join(f'{dur[bx == k].mean():6.2f}' for k in range(int(bx.max()) + 1)))
        for tag, gx, gy, gz, off in atts[8:9]:
            n = gx * gy * gz
            r = raw[off:off + n]
            hwid = r[:, 4] & 0xffffffff
            key = ((hwid >> 8) & 0xff) | (((r[:, 4] >> 32) & 0xf) << 8)
            cnt = np.bincount(np.unique(key, return_inverse=True)[1])
            print(f'#   one launch: workgroups per (XCC, SE, CU) slot: ' + ', '.join(f'{c} on {np.sum(cnt == c)} CUs' for c in np.unique(cnt)))
            dd = (r[:, 3] - r[:, 0]) / 100.0
            order = np.argsort(dd)[-8:]
            print('#   slowest workgroups of that launch (dur us, start us, xcc, kv head, seq): ' +
                  '; '.join(f'{dd[i]:.1f} {((r[i, 0] - r[:, 0].min()) / 100.0):.2f} {int((r[i, 4] >> 32) & 0xf)} {i % gx} {i // gx}' for i in order))
            # per CU: sum of its workgroups' spans vs the launch
            per_cu_end = {}
            for i in range(n):
                per_cu_end[key[i]] = max(per_cu_end.get(key[i], 0), r[i, 3])
            ends = (np.array(list(per_cu_end.values())) - r[:, 0].min()) / 100.0
            print(f'#   per-CU finish time: mean {ends.mean():.2f} p10 {np.percentile(ends, 10):.2f} p90 {np.percentile(ends, 90):.2f} max {ends.max():.2f} us '
                  f'over {len(ends)} CUs')
    if args.per_layer:
        print('# layer walls: ' + ' '.join(f'{w:.1f}' for w in walls))
    eng.close()


if __name__ == '__main__':
    main()
