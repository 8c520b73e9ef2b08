#!/usr/bin/env python
"""Request-stream throughput through the engine scheduler (continuous batching), Llama-3-8B W4A16 + int8 KV shapes,
synthetic weights: N requests with random prompt / output lengths are submitted at once and the scheduler is stepped
until all have finished (the shape of the reference's benchmark/profile_throughput.py, without a dataset or HTTP).
Prints one JSON line: output tokens/s, request/s, mean slot occupancy.
  python tools/bench_continuous.py [--requests 256] [--batch 64] [--prompt 64,1024] [--out 32,256]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import LLAMA3_8B, _Cfg  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--requests', type=int, default=256)
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--prompt', default='64,1024')
    ap.add_argument('--out', default='32,256')
    ap.add_argument('--tune', type=int, default=1)
    ap.add_argument('--modes', default='0', help='comma list of TM_ASYNC_STEP values: one timed session per entry on ONE engine (A/B of the '
                    'two-phase schedule / forward overlap on one box); the LAST entry is the line\'s headline value')
    ap.add_argument('--export-table', default='', help='write the measured GEMM dispatch table here')
    ap.add_argument('--import-table', default='', help='load a dispatch table instead of measuring (profiling runs: no tuner launches)')
    args = ap.parse_args()
    from lmdeploy_amd.turbomind.engine import Engine
    p0, p1 = map(int, args.prompt.split(','))
    o0, o1 = map(int, args.out.split(','))
    rng = np.random.default_rng(0)
    plen = rng.integers(p0, p1 + 1, args.requests)
    olen = rng.integers(o0, o1 + 1, args.requests)
    eng = Engine.from_model_config(_Cfg(dict(LLAMA3_8B)), max_batch_size=args.batch, session_len=p1 + o1 + 1, quant_policy=8,
                                   max_prefill_token_num=8192)
    eng.init_synthetic(seed=0)
    eng.start()
    if args.import_table:
        eng.import_gemm_table(args.import_table)
    elif args.tune:   # measured GEMM dispatch: the decode batch and every prefill size class an admission can produce
        for m in (args.batch, 512, 1024, 2048, 4096, 8192):
            eng.tune_gemm(m, args.export_table)
    prompts = [rng.integers(0, LLAMA3_8B['vocab'], n).astype(np.int32) for n in plen]
    # warm-up: one short session (graph capture, lazy module loads)
    for p in prompts[:2]:
        eng.submit(p[:16], 4)
    while eng.step() != (0, 0):
        pass
    eng.release()
    eng.sync()
    sessions = []
    for mode in args.modes.split(','):
        os.environ['TM_ASYNC_STEP'] = mode      # read by the engine when a continuous-batching session starts
        ov0 = eng.overlapped_steps()
        mx0 = eng.mixed_steps()
        t0 = time.perf_counter()
        ids = [eng.submit(p, int(n)) for p, n in zip(prompts, olen)]
        steps, occ = 0, 0
        while True:
            na, nw = eng.step()
            steps += 1
            occ += na
            if na == 0 and nw == 0:
                break
        eng.sync()
        dt = time.perf_counter() - t0
        got = sum(len(eng.poll(r)[1]) for r in ids)
        assert got == int(olen.sum()), (got, int(olen.sum()))
        sessions.append({'TM_ASYNC_STEP': mode, 'value': round(got / dt, 1), 'wall_s': round(dt, 3), 'scheduler_steps': steps,
                         'mean_active_slots': round(occ / steps, 1), 'mixed_steps': eng.mixed_steps() - mx0,
                         'overlapped_steps': eng.overlapped_steps() - ov0})
        eng.release()
    last = sessions[-1]
    print(json.dumps({'metric': 'continuous batching, output tokens/s (request stream, all submitted at t=0)',
                      'value': last['value'], 'unit': 'tokens/s', 'requests': args.requests, 'requests_per_s': round(args.requests / last['wall_s'], 2),
                      'batch_slots': args.batch, 'prompt_len': [p0, p1], 'output_len': [o0, o1], 'prompt_tokens': int(plen.sum()),
                      'output_tokens': int(olen.sum()), 'wall_s': last['wall_s'], 'scheduler_steps': last['scheduler_steps'],
                      'mean_active_slots': last['mean_active_slots'],
                      'total_tokens_per_s': round((int(olen.sum()) + int(plen.sum())) / last['wall_s'], 1),
                      'mixed_steps': last['mixed_steps'], 'overlapped_steps': last['overlapped_steps'],
                      'TM_MIXED_STEP': os.environ.get('TM_MIXED_STEP', '1'), 'TM_ASYNC_STEP': last['TM_ASYNC_STEP'], 'sessions': sessions,
                      'gemm_dispatch': ('imported table' if args.import_table else 'measured (decode batch + prefill size classes 512 .. 8192)')
                                       if (args.tune or args.import_table) else 'heuristic'}))
    eng.close()


if __name__ == '__main__':
    main()
