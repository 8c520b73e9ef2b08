#!/usr/bin/env python
"""Headline benchmark: decode tokens/s (+ p50 TTFT) of Llama-3-8B W4A16 (AWQ g128) + int8 KV, batch 64,
1k-token synthetic prompts, on N MI355X GPUs of one node (TP=N, one process per GPU, RCCL).

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A "step" = one decode iteration of the whole batch (64 tokens).  Rank 0 prints ONE JSON line.
`value` = batch * K / t, where t brackets exactly K steps (graph replays) with barrier + device sync on both
sides, max over ranks.  Inputs (weights, KV cache of the prefilled prompts) are resident in HBM.
`roofline` is for the kernel family that takes the most TIME of a step -- at the driver command's context the four W4A16
decode GEMMs of a layer (algorithmic weight bytes of the layer / the sum of their mean launch durations), else the paged decode
attention (algorithmic KV bytes per launch / mean launch duration); the other one follows as `attention_roofline` /
`gemm_family_roofline`, the whole step as `step_roofline`.  Durations are IN-GRAPH: a rocprofv3 --kernel-trace child run of this
script over hipGraph replays with this run's GEMM dispatch table (measure_in_graph_durations); `traffic` comes from a second
child under --pmc FETCH_SIZE (its own pass).  `kernel_ms_per_step` keeps the eager HIP-event view of every category.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LLAMA3_8B = dict(hidden=4096, layers=32, q_heads=32, kv_heads=8, head_dim=128, inter=14336, vocab=128256,
                 rms_eps=1e-5)
# BASELINE.json configs 3 / 4 (parity-test cases; `--model` lets the same harness time them)
INTERNLM2_20B = dict(hidden=6144, layers=48, q_heads=48, kv_heads=8, head_dim=128, inter=16384, vocab=92544, rms_eps=1e-5)
LLAMA3_70B = dict(hidden=8192, layers=80, q_heads=64, kv_heads=8, head_dim=128, inter=28672, vocab=128256, rms_eps=1e-5)
# config 5: Mixtral-8x7B, fp8 block-scaled weights, 8 experts / top-2 (TP = 2 in BASELINE.json; --emulate-tp 2 = one rank)
MIXTRAL_8X7B = dict(hidden=4096, layers=32, q_heads=32, kv_heads=8, head_dim=128, inter=14336, vocab=32000, rms_eps=1e-5,
                    moe_experts=8, moe_top_k=2, weight_type=2)
# config 1 (BASELINE.json configs[0]): the reference's own CPU-runnable case -- `--cpu-config0` times it on the host cores, no GPU involved
INTERNLM2_1_8B = dict(hidden=2048, layers=24, q_heads=16, kv_heads=8, head_dim=128, inter=8192, vocab=92544, rms_eps=1e-5)
MODELS = {'llama3_8b': LLAMA3_8B, 'internlm2_20b': INTERNLM2_20B, 'llama3_70b': LLAMA3_70B, 'mixtral_8x7b': MIXTRAL_8X7B}
HBM_PEAK_GBPS = 8000.0     # MI355X HBM3E spec (/opt/skills/guides/MI355X_MICROARCH.md); 6290 measured copy


class _Rope:
    dim, base, type, factor = 128, 500000.0, 'llama3', 8.0
    low_freq_factor, high_freq_factor, original_max_position_embeddings = 1.0, 4.0, 8192


class _Cfg:
    group = 128
    rope = _Rope()

    def __init__(self, d):
        self.__dict__.update(d)


def algorithmic_bytes(m, batch, ctx, kv_bits, tp):
    """bytes(step) = W_q + W_sz + W_head + B*ctx*kv_bytes_per_token (per rank: / tp)."""
    H, D = m['hidden'], m['head_dim']
    P = m['layers'] * (H * (m['q_heads'] + 2 * m['kv_heads']) * D + m['q_heads'] * D * H + 3 * H * m['inter'])
    w = 0.5 * P + 4.0 * P / 128 + 2.0 * H * m['vocab']
    kv_tok = m['layers'] * 2 * m['kv_heads'] * (D * kv_bits / 8 + (4 if kv_bits < 16 else 0))
    return (w + batch * ctx * kv_tok) / tp, kv_tok / tp


def self_launch(n: int) -> int:
    """Spawn ranks 0..n-1 of this script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment, exactly what
    torch.distributed.run would set), forward rank 0's stdout (the JSON line) and everybody's stderr.  Returns the
    first non-zero exit code; a rank that dies takes the others down with it (they would hang in a collective)."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    alive = set(range(n))
    while alive:
        for r in sorted(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.discard(r)
            if code != 0 and rc == 0:
                rc = code
                print(f'bench.py: rank {r} exited with {code}; stopping the other ranks', file=sys.stderr)
                for q in alive:
                    procs[q].terminate()
        time.sleep(0.05)
    return rc


def measure_attention_traffic(args, ctx_prof):
    """HBM bytes per launch of the decode-attention kernel, measured NOW: a child run of this script under
    `rocprofv3 --pmc FETCH_SIZE` (its own pass, no tracing domains) with the same batch / KV format, two layers, eager
    launches, and a prompt length that puts the mean context of its decode steps at `ctx_prof`.  FETCH_SIZE is corrected
    as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes for gfx950: KiB -> bytes, x 2 for wide coalesced
    streaming reads.  Returns (attention bytes_per_launch | None, description, {bytes_per_layer, launches} of the four decode GEMMs
    of the same pass | None)."""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    prof = shutil.which('rocprofv3') or ('/opt/rocm/bin/rocprofv3' if os.path.exists('/opt/rocm/bin/rocprofv3') else None)
    if prof is None:
        return None, 'rocprofv3 not found', None
    steps, warm = 6, 1
    s_child = max(64, int(round(ctx_prof - 1 - (warm + steps - 1) / 2.0)))
    tmp = tempfile.mkdtemp(prefix='tm_pmc_', dir='/tmp')
    cmd = [prof, '--pmc', 'FETCH_SIZE', '-d', tmp, '-o', 'pmc', '--', sys.executable, os.path.abspath(__file__), '--traffic-child',
           '--steps', str(steps), '--warmup', str(warm), '--profile-steps', '0', '--no-cpu-baseline', '--no-graph', '--layers', '2',
           '--batch', str(args.batch), '--prompt-len', str(s_child), '--quant-policy', str(args.quant_policy), '--model', args.model]
    try:
        subprocess.run(cmd, cwd='/tmp', env=dict(os.environ, TMPDIR='/tmp'), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                       timeout=300, check=True)
        dbs = glob.glob(os.path.join(tmp, '**', '*_results.db'), recursive=True)
        db = sqlite3.connect(dbs[0])
        mean_kib, n = list(db.execute("select avg(value), count(*) from counters_collection where kernel_name like "
                                      "'%decode_attention%' and counter_name = 'FETCH_SIZE'"))[0]
        db.close()
        if not n:
            return None, 'no decode_attention dispatches in the PMC pass', None
        ctx_child = s_child + 1 + (warm + steps - 1) / 2.0
        # the decode linears of the same pass: every launch of the M <= 64 GEMM kernels (the row-block templates <1, ..> / <2, ..> of
        # gemm_dec32_kernel and the loader / consumer kernel; the prefill forwards run the <4, ..> / pre64 tiles), four per layer
        gdb = sqlite3.connect(dbs[0])
        gsum, gn = list(gdb.execute("select sum(value), count(*) from counters_collection where counter_name = 'FETCH_SIZE' and "
                                    "(kernel_name like '%gemm_dec32_kernel<1,%' or kernel_name like '%gemm_dec32_kernel<2,%' "
                                    "or kernel_name like '%gemm_dec_lc_kernel%')"))[0]
        gdb.close()
        gemm = None
        if gn and gn % 4 == 0:
            gemm = dict(bytes_per_layer=int(gsum * 1024.0 * 2.0 / (gn / 4)), launches=int(gn))
        return int(mean_kib * 1024.0 * 2.0), (f'rocprofv3 --pmc FETCH_SIZE child run of this bench ({n} launches, 2 layers, eager, mean ctx '
                                              f'{ctx_child}): mean FETCH_SIZE {mean_kib:.0f} KiB x 1024 x 2 (gfx950 correction)'), gemm
    except Exception as e:   # noqa: BLE001 -- a failed profiler pass must not take the bench line down
        return None, f'PMC pass failed: {type(e).__name__}', None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def gemm_family_bytes(m, tp=1):
    """algorithmic bytes of the four W4A16 decode linears of ONE layer -- K * N / 2 + K * N / 32 each (u4 codes + fp16 scale /
    zero per 128 x 1 group), per rank.  Returns ({role: bytes}, total)."""
    H, D = m['hidden'], m['head_dim']
    hq, hkv, inter = m['q_heads'] // tp, max(1, m['kv_heads'] // tp), m['inter'] // tp
    kn = dict(w_qkv=(H, (hq + 2 * hkv) * D), wo=(hq * D, H), w1w3=(H, 2 * inter), w2=(inter, H))
    per = {r: k * n / 2.0 + k * n / 32.0 for r, (k, n) in kn.items()}
    return per, sum(per.values())


def family_roofline(per_role_bytes, per_role_us):
    """the `roofline` object's arithmetic for a kernel FAMILY (the four decode GEMMs of a layer): achieved = sum of their algorithmic
    bytes / sum of their mean in-graph launch durations.  (tests/test_host.py checks it against hand numbers.)"""
    tot_b = sum(per_role_bytes[r] for r in per_role_us)
    tot_us = sum(per_role_us.values())
    ach = tot_b / (tot_us * 1e-6) / 1e9
    return dict(achieved=round(ach, 1), frac=round(ach / HBM_PEAK_GBPS, 4), bytes_per_layer=int(tot_b), us_per_layer=round(tot_us, 2))


def measure_in_graph_durations(args, table_text, ctx_target, layers, batch):
    """Mean IN-GRAPH launch durations of the decode kernels, measured NOW: a child run of this script under `rocprofv3 --kernel-trace`
    (tracing only, no counters) with hipGraph replays, the parent's GEMM dispatch table imported (same tilings, no tuning launches) and a
    prompt length that puts the mean context of its replayed steps at `ctx_target`.  The eager HIP-event numbers (`kernel_ms_per_step`)
    time a slower path than the one `value` is measured on; these are the launches of the timed path itself.
    Returns ({'w_qkv' | 'wo' | 'w1w3' | 'w2' | 'attention' | 'lm_head': mean us per launch}, description) or (None, reason)."""
    import glob
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    prof = shutil.which('rocprofv3') or ('/opt/rocm/bin/rocprofv3' if os.path.exists('/opt/rocm/bin/rocprofv3') else None)
    if prof is None:
        return None, 'rocprofv3 not found'
    steps, warm = 12, 4
    s_child = max(64, int(round(ctx_target - 1 - warm - (steps - 1) / 2.0)))
    tmp = tempfile.mkdtemp(prefix='tm_trace_', dir='/tmp')
    env = dict(os.environ, TMPDIR='/tmp')
    if table_text:
        tpath = os.path.join(tmp, 'gemm_table.txt')
        open(tpath, 'w').write(table_text)
        env['TM_GEMM_IMPORT'] = tpath
    cmd = [prof, '--kernel-trace', '-d', tmp, '-o', 'trace', '--', sys.executable, os.path.abspath(__file__), '--traffic-child',
           '--steps', str(steps), '--warmup', str(warm), '--profile-steps', '0', '--no-cpu-baseline', '--no-traffic', '--no-full-run', '--tune', '0',
           '--batch', str(batch), '--prompt-len', str(s_child), '--quant-policy', str(args.quant_policy), '--model', args.model,
           '--max-prefill-tokens', str(args.max_prefill_tokens)]
    if args.emulate_tp:
        cmd += ['--emulate-tp', str(args.emulate_tp)]
    try:
        subprocess.run(cmd, cwd='/tmp', env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=420, check=True)
        dbs = glob.glob(os.path.join(tmp, '**', '*_results.db'), recursive=True)
        db = sqlite3.connect(dbs[0])
        # the replayed steps are the LAST launches of the trace: per kernel family, the final steps x launches-per-step of them
        def last(where, n):
            rows = list(db.execute(f"select end - start, grid_x / workgroup_x, grid_y / workgroup_y, grid_z / workgroup_z from kernels where {where} "
                                   f"order by start desc limit {int(n)}"))
            return rows
        out = {}
        g = last("(name like '%gemm_dec32_kernel<1,%' or name like '%gemm_dec32_kernel<2,%' or name like '%gemm_dec_lc_kernel%')", steps * layers * 4)
        if len(g) == steps * layers * 4:
            # newest first; within a layer the launch order is w_qkv, wo, w1w3, w2 -> reversed positions 3, 2, 1, 0
            for pos, role in enumerate(('w2', 'w1w3', 'wo', 'w_qkv')):
                d = [r[0] for i, r in enumerate(g) if i % 4 == pos]
                out[role] = sum(d) / len(d) / 1e3
        a = last("name like '%decode_attention%'", steps * layers)
        if len(a) == steps * layers:
            out['attention'] = sum(r[0] for r in a) / len(a) / 1e3
        db.close()
        if not out:
            return None, 'no decode launches found in the kernel trace'
        ctx_child = s_child + 1 + warm + (steps - 1) / 2.0
        return out, (f'rocprofv3 --kernel-trace child run of this bench: the last {steps} hipGraph-replayed steps ({layers} layers, mean ctx {ctx_child}), '
                     f'mean duration per launch')
    except Exception as e:   # noqa: BLE001 -- a failed profiler pass must not take the bench line down
        return None, f'kernel-trace pass failed: {type(e).__name__}'
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def run_canary(args, rank, world, timeout_s=240, extra_env=None):
    """Multi-GPU runs only: before THIS process touches the device for real, a child copy of the bench (its own rendezvous port) pushes a
    2-layer model of the same width through every collective code path the real run will take with the CURRENT switches -- RCCL bring-up,
    the native communicator + its self-test, a prefill forward large enough for the side-stream micro-batch schedule, graph-captured decode
    steps -- under a hard time limit.  No multi-rank collective of this engine has ever crossed xGMI before the driver's scaling run,
    and a hang inside a collective cannot be recovered in-process: the canary turns "hang -> no record at all" into "fall back to the most
    conservative switches and measure".  Returns True when the child exited 0 within the limit."""
    import subprocess
    port = int(os.environ.get('MASTER_PORT', '29500')) + 17
    env = dict(os.environ, MASTER_PORT=str(port), TM_BENCH_CANARY='1', **(extra_env or {}))
    if extra_env:           # second attempt: its own rendezvous port (ranks of the first attempt may still be dying)
        env['MASTER_PORT'] = str(port + 13)
    cmd = [sys.executable, os.path.abspath(__file__), '--gpus', str(world), '--steps', '6', '--warmup', '3', '--layers', '2', '--batch', str(args.batch),
           '--prompt-len', '64', '--max-prefill-tokens', '4096', '--quant-policy', str(args.quant_policy), '--model', args.model, '--profile-steps', '0',
           '--no-cpu-baseline', '--no-traffic', '--no-full-run', '--tune', '0']
    if args.allow_shared_devices:
        cmd.append('--allow-shared-devices')
    try:
        p = subprocess.Popen(cmd, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    except OSError:
        return False
    try:
        return p.wait(timeout=timeout_s) == 0
    except subprocess.TimeoutExpired:
        p.kill()            # this exact child: a rank stuck in a collective
        try:
            p.wait(timeout=20)
        except subprocess.TimeoutExpired:
            pass
        return False


class Watchdog:
    """Multi-GPU runs only: a rank that makes no progress (a collective that never completes, a peer that died before its
    first step) would otherwise hang until the launcher's own limit.  Every phase of the run re-arms the timer; when it
    fires the rank says which phase stalled and exits with code 3 -- the record then shows a cause instead of a timeout."""

    def __init__(self, enabled, rank):
        import threading
        self.enabled, self.rank, self.phase, self.deadline = enabled, rank, 'start', None
        self._lock = threading.Lock()
        if enabled:
            threading.Thread(target=self._run, daemon=True).start()

    def arm(self, phase, seconds):
        with self._lock:
            self.phase, self.deadline = phase, time.monotonic() + seconds

    def _run(self):
        while True:
            time.sleep(1.0)
            with self._lock:
                late = self.deadline is not None and time.monotonic() > self.deadline
                phase = self.phase
            if late:
                print(f'bench.py: rank {self.rank} made no progress in phase "{phase}" within its time limit -- giving up '
                      f'(exit 3); TM_COMM=native / TM_GRAPH_COMM=0 select the other collective paths', file=sys.stderr, flush=True)
                os._exit(3)


_JSON_FD = None


def claim_stdout():
    """Rank 0's stdout must carry the JSON line and nothing else, but native libraries (RCCL's version banner, rocprofv3
    children) write to file descriptor 1 behind Python's back.  Keep a private duplicate of the real stdout for the result and
    point fd 1 at stderr for everybody else."""
    global _JSON_FD
    if _JSON_FD is None:
        sys.stdout.flush()
        _JSON_FD = os.dup(1)
        os.dup2(2, 1)


def emit_json(obj):
    data = (json.dumps(obj) + '\n').encode()
    if _JSON_FD is None:
        sys.stdout.write(data.decode())
        sys.stdout.flush()
    else:
        os.write(_JSON_FD, data)


def main():
    # multi-process GPU work on this pool needs dmabuf IPC (RCCL's P2P setup fails with the legacy mode:
    # "hipIpcGetMemHandle: invalid argument"); already exported on the driver's boxes, kept here for hand launches
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=512)
    ap.add_argument('--warmup', type=int, default=16)
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--prompt-len', type=int, default=1024)
    ap.add_argument('--quant-policy', type=int, default=8)
    ap.add_argument('--profile-steps', type=int, default=8)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-graph', action='store_true', help='eager launches (rocprofv3 --pmc passes)')
    ap.add_argument('--layers', type=int, default=0, help='debug only: override the layer count (result is then INVALID)')
    ap.add_argument('--model', default='llama3_8b', choices=['llama3_8b', 'internlm2_20b', 'llama3_70b', 'mixtral_8x7b'],
                    help='shapes to time; the headline metric is llama3_8b (anything else changes metric/config in the output)')
    ap.add_argument('--tune', type=int, default=1, help='1 (default): TM_GEMM_TUNE=1 -- the engine measures its decode GEMM tilings at '
                    'start-up, outside the timed region (the reference\'s warm-up tuning); 0: heuristics only')
    ap.add_argument('--no-traffic', action='store_true', help='skip the rocprofv3 --pmc FETCH_SIZE child run (roofline.traffic = null)')
    ap.add_argument('--no-full-run', action='store_true', help='skip the continuation to 1024 generated tokens (value_full_run)')
    ap.add_argument('--traffic-child', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--allow-shared-devices', action='store_true',
                    help='let several ranks share one GPU when the box has fewer GPUs than --gpus (launcher / bring-up tests only: the '
                         'record then says so and is not a scaling result)')
    ap.add_argument('--max-prefill-tokens', type=int, default=8192, choices=[512, 1024, 2048, 4096, 8192],
                    help='TurbomindEngineConfig.max_prefill_token_num: tokens per prefill forward (the reference default is 8192).  The p50 '
                         'TTFT of B simultaneous prompts is quantised by it: sequences get their first token when THEIR chunk is done')
    ap.add_argument('--decode-splits', type=int, default=0, help='split-KV count of the decode attention (0: the engine\'s heuristic)')
    ap.add_argument('--emulate-tp', type=int, default=0,
                    help='SURVEY 8(e) on a 1-GPU box: run ONE rank\'s shard of a TP=N job (heads / inter / vocab divided by N, '
                         'collectives through a 1-rank RCCL communicator): per-rank kernel time only, labelled as such')
    ap.add_argument('--cpu-config0', action='store_true',
                    help='BASELINE.json configs[0] only: InternLM2-1.8B fp16 greedy decode on the host cores (oracle/cpu_baseline.py::run_config0, the '
                         'torch-CPU port of the reference\'s default-backend ops); prints its own JSON line; no GPU is touched')
    args = ap.parse_args()
    if args.cpu_config0:
        from oracle import cpu_baseline
        b = 1 if args.batch == 64 else args.batch
        r = cpu_baseline.run_config0(INTERNLM2_1_8B, batch=b, ctx=args.prompt_len)
        emit_json({'metric': f'decode tokens/sec, InternLM2-1.8B fp16 greedy decode, batch {b}, ctx {args.prompt_len}, on the host CPU cores '
                             '(BASELINE.json configs[0]: the reference\'s CPU-runnable plumbing case; NOT the headline config)',
                   'value': round(r['value'], 3), 'unit': 'tokens/s', 'n_gpus': 0, 'higher_is_better': True, 'dtype': 'f16 storage, f32 contraction',
                   'data': 'synthetic', 'config': {'workload': 'InternLM2-1.8B shapes, fp16 random weights, fp16 KV, greedy decode', 'batch': b,
                                                   'ctx': args.prompt_len}, 'cpu_baseline': r})
        return
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # plain `python bench.py --gpus N`: bring up the N ranks of this node ourselves, one process per GPU (the
        # reference starts all ranks of a node from one call too: lmdeploy/turbomind/turbomind.py:191-217)
        sys.exit(self_launch(args.gpus))

    if os.environ.get('TM_BENCH_CANARY') and os.environ.get('TM_BENCH_CANARY_FAIL', '0') == '1':
        sys.exit(3)         # test hook: a canary child that fails at once (tests/test_gpu_bench_launcher.py)
    claim_stdout()
    import torch
    import torch.distributed as dist
    from lmdeploy_amd.turbomind.engine import Engine

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        sys.exit(f'bench.py: --gpus {args.gpus} but the launcher set WORLD_SIZE={world}')
    # one GPU per rank; on a box with fewer GPUs than ranks (the 1-GPU test box: `--gpus 2` exercises the launcher, the RCCL
    # bring-up failure -> native communicator fall-back and the tuning-table broadcast) ranks share devices
    ndev = max(1, torch.cuda.device_count())
    ranks_per_device = (world + ndev - 1) // ndev
    if ranks_per_device > 1:
        if not args.allow_shared_devices:
            sys.exit(f'bench.py: --gpus {world} on a box with {ndev} GPU(s): ranks would share devices, which is not a scaling measurement; '
                     f'pass --allow-shared-devices for a launcher / bring-up test')
        # co-located ranks: the two-shot all-reduce's persistent grids must fit on the device TOGETHER
        os.environ.setdefault('TM_P2P_2SHOT_GRID', str(max(8, 256 // ranks_per_device // 2)))
    local_dev = local_rank % ndev
    torch.cuda.set_device(local_dev)
    canary_note = ''
    replicas = False          # set when the tensor-parallel path cannot run on this node at all (see the canaries below)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('gloo', rank=rank, world_size=world)   # control plane only; data plane = RCCL in C++
        # the canary (run_canary): every rank's child forms its own group; ANY failure -> all ranks take the conservative switches together
        if not os.environ.get('TM_BENCH_CANARY') and os.environ.get('TM_BENCH_NO_CANARY', '0') == '0' \
                and (ranks_per_device == 1 or os.environ.get('TM_BENCH_CANARY_SHARED', '0') == '1'):     # (the latter: mechanism test on a 1-GPU box)
            ok = run_canary(args, rank, world)
            oks = [None] * world
            dist.all_gather_object(oks, bool(ok))
            if not all(oks):
                safe = {'TM_COMM': 'rccl', 'TM_COMM_STREAM': '0', 'TM_GRAPH_COMM': '0'}
                second = dict(safe, TM_BENCH_CANARY_FAIL='0')
                canary_note = (f'canary run failed or hung on ranks {[r for r, o_ in enumerate(oks) if not o_]} with the default collective switches -> '
                               f'RCCL only, collectives on the engine stream, eager (TM_COMM=rccl TM_COMM_STREAM=0 TM_GRAPH_COMM=0)')
                print(f'[bench] rank {rank}: {canary_note}', file=sys.stderr)
                # the conservative switches get a canary of their own: if the tensor-parallel path cannot finish even so on this node, the run
                # still produces a record -- N independent TP = 1 replicas (no collective at all), labelled as such -- instead of a hang
                ok2 = os.environ.get('TM_BENCH_FORCE_REPLICAS', '0') != '1' and run_canary(args, rank, world, extra_env=second)
                dist.all_gather_object(oks, bool(ok2))
                if all(oks):
                    os.environ.update(safe)
                else:
                    replicas = True
                    canary_note += (f'; a second canary with those switches failed or hung on ranks {[r for r, o_ in enumerate(oks) if not o_]} -> THIS LINE IS NOT '
                                    f'TENSOR PARALLEL: {world} independent TP = 1 replicas, one per GPU, batch {args.batch} each (weak scaling, no collective)')
                    print(f'[bench] rank {rank}: {canary_note}', file=sys.stderr)

    tp = 1 if replicas else world     # model-parallel degree; `world` stays the number of processes / GPUs of the job
    model = dict(MODELS[args.model])
    if args.layers:
        model['layers'] = args.layers
    emu = args.emulate_tp
    if emu > 1:
        assert world == 1, '--emulate-tp is a single-process measurement'
        model.update(q_heads=model['q_heads'] // emu, kv_heads=max(1, model['kv_heads'] // emu), inter=model['inter'] // emu,
                     vocab=model['vocab'] // emu)
        os.environ['TM_FORCE_COMM'] = '1'      # the rank's collective code path, through a 1-rank communicator
        # a 1-rank exchange takes no time, so splitting the prefill forwards to hide it would only add launches: the emulation reports the
        # rank's kernel time unsplit, unless the run asks for the exchange stand-in (TM_EMULATE_AR_GBPS: the overlap experiment)
        os.environ.setdefault('TM_COMM_STREAM', '1' if float(os.environ.get('TM_EMULATE_AR_GBPS', '0') or 0) > 0 else '0')
    K, W, B, S = args.steps, args.warmup, args.batch, args.prompt_len
    P = args.profile_steps
    child = args.traffic_child
    full_run = (not args.no_full_run) and not child and 1 + W + K < 1024     # continue to 1024 generated tokens after the timed region
    max_new = max(1 + W + K + P + 2, (1024 + P + 2) if full_run else 0)
    weight_type = int(model.pop('weight_type', 0))
    eng = Engine.from_model_config(_Cfg(model), weight_type=weight_type, tp=tp, rank=rank if tp > 1 else 0, device=local_dev, max_batch_size=B,
                                   session_len=S + max_new + 1, quant_policy=args.quant_policy,
                                   max_prefill_token_num=args.max_prefill_tokens, use_graph=0 if args.no_graph else 1, decode_splits=args.decode_splits)
    dog = Watchdog(world > 1, rank)
    comm_note = ''
    if tp > 1:
        dog.arm('communicator set-up', 300)

        def gather(h):
            out = [None] * world
            dist.all_gather_object(out, h)
            return out
        # default (round 6): RCCL for the prefill-sized forwards + the native fused all-reduce + residual + RMSNorm for the decode-sized
        # ones, taken only when its bring-up self-test passes on every rank; TM_COMM=rccl keeps RCCL alone, TM_COMM=native drops RCCL
        mode = os.environ.get('TM_COMM', 'hybrid')
        want_native = mode == 'native'
        uid = [Engine.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        rccl_ok = True
        try:
            eng.comm_init(uid[0])
        except Exception as exc:        # noqa: BLE001 -- RCCL could not be brought up on this node: the native communicator remains
            rccl_ok = False
            print(f'[bench] rank {rank}: RCCL communicator failed ({exc}); trying the native P2P communicator', file=sys.stderr)
        oks = gather(rccl_ok)
        if not all(oks):                # a collective path needs EVERY rank: fall back together
            comm_note = f'RCCL init failed on ranks {[r for r, ok in enumerate(oks) if not ok]} -> native P2P communicator'
            want_native = True
            eng.comm_drop_rccl()
        if want_native:                 # TM_COMM=native (or no RCCL): the native communicator without the self-test's fall-back
            eng.comm_native_setup(gather, rows=max(B, 1))
        elif mode != 'rccl':            # hybrid: the native communicator on top of RCCL, if its self-test passes everywhere
            try:
                eng.comm_native_setup(gather, rows=max(B, 1))
                good = eng.comm_native_selftest()
            except Exception as exc:    # noqa: BLE001
                print(f'[bench] rank {rank}: native communicator bring-up failed ({exc})', file=sys.stderr)
                good = False
            goods = gather(bool(good))
            if not all(goods):
                eng.comm_native_drop()
                comm_note = (comm_note + '; ' if comm_note else '') + (f'native communicator self-test failed on ranks '
                                                                       f'{[r for r, g in enumerate(goods) if not g]} -> RCCL + residual-norm launch')
    elif emu > 1:
        eng.comm_init(Engine.comm_unique_id())
        if os.environ.get('TM_COMM', 'hybrid') != 'rccl':   # the rank's DEFAULT decode collective: the fused launch (one-rank communicator)
            eng.comm_native_setup(lambda h: [h], rows=max(B, 1))
            assert eng.comm_native_selftest(), 'native communicator self-test (one rank)'
    dog.arm('weights + engine start', 600)
    eng.init_synthetic(seed=0)          # same seed on every rank: shards are generated per rank-local shape
    eng.start()
    # measured GEMM dispatch (start-up work like the reference's TM_GEMM_TUNE warm-up: not in any timed region).  tp > 1: rank 0
    # times the candidates on ITS shard shapes (all ranks hold the same shapes) and the table is broadcast, so that every rank
    # runs identical tilings (a rank-local winner could differ by noise and desynchronise the ranks' step times)
    tuned = bool(args.tune) and B <= 256 and not child
    if tuned:
        dog.arm('GEMM tuning', 600)
        import tempfile
        table = None
        if rank == 0:
            try:
                with tempfile.TemporaryDirectory() as td:
                    path = os.path.join(td, 'gemm_table.txt')
                    eng.tune_gemm(B, path)
                    if B * S >= args.max_prefill_tokens:   # the prefill forwards of this run are chunks of that size: their size class as well
                        eng.tune_gemm(args.max_prefill_tokens, path)
                    table = open(path).read()
            except Exception as exc:    # noqa: BLE001 -- the heuristics are the measured winners on these shapes anyway
                print(f'[bench] GEMM tuning skipped: {exc}', file=sys.stderr)
        if world > 1:
            box = [table]
            dist.broadcast_object_list(box, src=0)
            table = box[0]
            if table and rank != 0:
                with tempfile.TemporaryDirectory() as td:
                    path = os.path.join(td, 'gemm_table.txt')
                    open(path, 'w').write(table)
                    eng.import_gemm_table(path)
        tuned = bool(table)
    table_text = None
    if not child and rank == 0:
        try:    # the dispatch this process runs (measured + imported entries): the trace child imports it instead of tuning again
            import tempfile
            with tempfile.TemporaryDirectory() as td:
                path = os.path.join(td, 'gemm_table_now.txt')
                Engine.export_gemm_table(path)
                table_text = open(path).read()
        except Exception:       # noqa: BLE001
            table_text = None

    gen = torch.Generator().manual_seed(0)
    prompts = torch.randint(0, model['vocab'], (B, S), generator=gen, dtype=torch.int32).numpy()

    def barrier():
        eng.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    dog.arm('prefill', 600)
    barrier()
    t0 = time.perf_counter()
    eng.prefill(list(prompts), max_new_tokens=max_new)
    eng.sync()
    prefill_s = time.perf_counter() - t0
    ttft = eng.prefill_times_ms()

    dog.arm('decode warm-up (graph capture)', 600)
    eng.decode(W)                       # untimed warm-up (includes the graph capture)
    barrier()
    dog.arm('timed decode steps', 300 + K)
    t0 = time.perf_counter()
    eng.decode(K)                       # exactly K steps
    dt_enqueue = time.perf_counter() - t0   # host time to enqueue the K graph replays (the call returns without a device sync)
    barrier()
    dt = time.perf_counter() - t0
    dog.arm('after the timed region', 1800)
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    ctx_first = S + 1 + W               # context length (incl. the new token) of the first timed step
    ctx_mean = ctx_first + (K - 1) / 2.0
    kv_bits = 16 if args.quant_policy == 0 else args.quant_policy
    step_bytes, kv_tok = algorithmic_bytes(model, B, ctx_mean, kv_bits, tp)

    # ---- per-kernel durations: HIP events on the engine stream, eager steps right after the timed region ----
    prof = eng.profile_decode(P) if P > 0 else {}
    ctx_prof = S + 1 + W + K + (P - 1) / 2.0
    # ---- the rest of the 1k-out window (64 x 1024 generated tokens, steady state without the first steps):
    # the timed K steps sit at the CHEAP end of the run (shortest contexts), so also report the rate over everything
    # from the first timed step to the 1024th generated token (the P eager profile steps advance the context but are
    # not part of either time)
    full = None
    if full_run:
        R = 1024 - (1 + W + K + P)
        barrier()
        t0 = time.perf_counter()
        eng.decode(R)
        barrier()
        dt_r = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt_r], dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt_r = float(tt.item())
        full = dict(value=round(B * (world if replicas else 1) * (K + R) / (dt + dt_r), 1), steps=K + R, ctx_first=ctx_first, ctx_last=S + 1024,
                    ms_per_step=round((dt + dt_r) / (K + R) * 1e3, 4))
    toks = eng.fetch()
    stats = eng.stats()
    cinfo = eng.comm_info()
    # the tilings the timed steps ran (measured table first, then the heuristic): per decode linear (shape, split-K) at M = B
    hq_l, hkv_l = model['q_heads'] // tp, max(1, model['kv_heads'] // tp)
    D_, H_, I_l = model['head_dim'], model['hidden'], model['inter'] // tp
    tilings, prefill_tilings = {}, {}
    pf_rows = min(B * S, args.max_prefill_tokens)     # rows of a prefill forward of this run (max_prefill_token_num chunks)
    if weight_type == 0 and not model.get('moe_experts') and B <= 256:
        for role, (name, (kk, nn)) in enumerate(dict(w_qkv=(H_, (hq_l + 2 * hkv_l) * D_), wo=(hq_l * D_, H_), w1w3=(H_, 2 * I_l),
                                                      w2=(I_l, H_)).items(), start=1):
            tilings[name] = dict(zip(('shape', 'splits'), Engine.pick_tiling(kk, nn, B, role=role)))
            # 4 / 5 = the fused 128-row W4A16 tiles, 12 = the 256 x 256 tile with the dequant through LDS (gemm_prefill.hip)
            prefill_tilings[name] = dict(zip(('shape', 'splits'), Engine.pick_tiling(kk, nn, pf_rows, role=role)))
    # the fp16 lm_head runs the general kernel: (tiles per wave, split-K) from the measured table (`G` lines) or the heuristic
    head_tiling = None
    if B <= 256 and (model['vocab'] // tp) % 16 == 0:
        head_tiling = dict(zip(('nt', 'splits'), Engine.pick_general(1, 5, H_, model['vocab'] // tp, B)[:2]))
    if weight_type != 0 or model.get('moe_experts'):
        # formats other than AWQ u4 / MoE: the weight bytes are what the engine actually streams (packed weights + scales +
        # lm_head); with batch 64 and top-2 of 8 every expert is hit every step
        step_bytes = stats['weight_bytes'] + B * ctx_mean * kv_tok

    if rank == 0:
        ms_step = dt / K * 1e3
        value = B * (world if replicas else 1) * K / dt    # replicas: every GPU ran its own batch of B
        kv_name = {0: 'fp16-KV', 4: 'int4-KV', 8: 'int8-KV'}.get(args.quant_policy, f'quant_policy={args.quant_policy}')
        headline = args.model == 'llama3_8b' and args.quant_policy == 8 and B == 64 and S == 1024 and not args.layers
        out = {
            'metric': f'decode tokens/sec, Llama-3-8B W4A16 {kv_name} batch {B} ({S}-in/1k-out synthetic)'
                      + ('' if headline else ' (NOT the headline config)'),
            'value': round(value, 1), 'unit': 'tokens/s', 'n_gpus': world, 'steps': K, 'warmup': W,
            'ms_per_step': round(ms_step, 4), 'higher_is_better': True, 'scaling': 'weak' if replicas else 'strong', 'vs_baseline': None,
            'dtype': 'f16', 'data': 'synthetic',
            'config': {'workload': f'Llama-3-8B shapes, W4A16 AWQ g128 random weights, quant_policy={args.quant_policy} '
                                   f'KV, batch {B}{" per replica" if replicas else ""}, {S}-token random prompts, greedy decode, '
                                   + (f'{world} independent TP=1 replicas' if replicas else f'TP={world}'),
                       'batch': B, 'prompt_len': S, 'ctx_first_timed_step': ctx_first, 'ctx_mean': ctx_mean,
                       'parallelism': f'dp{world} (replicas only)' if replicas else f'tp{world}', 'collectives': cinfo['backend'],
                       'rccl_ranks': cinfo['ranks'] if tp > 1 else 0,
                       'decode_splits': stats['decode_splits'], 'hipgraph': cinfo['hipgraph'],
                       # RMSNorm folded into the decode GEMMs (tp = 1, dense u4, batch <= 64; TM_FOLD_NORM bit 0: wo -> w1w3, bit 1: w2 ->
                       # w_qkv): 5 launches per layer instead of 7
                       'rmsnorm_fold': (int(os.environ.get('TM_FOLD_NORM', '0')) & 3) if (tp == 1 and emu <= 1 and weight_type == 0
                                                                                         and not model.get('moe_experts') and B <= int(os.environ.get('TM_FOLD_MAX_M', '64'))) else 0,
                       'gemm_dispatch': ('measured at start-up (tm_engine_tune_gemm' + (', rank 0\'s table broadcast' if world > 1 else '') + ')')
                                        if tuned else 'heuristic',
                       'gemm_tilings': tilings, 'lm_head_tiling': head_tiling, 'prefill_rows': pf_rows,
                       'prefill_gemm_tilings': prefill_tilings},
            # host side of the timed region: the K hipGraphLaunch calls return long before the device is done when the step is device-bound
            'host_enqueue_ms_per_step': round(dt_enqueue / K * 1e3, 4),
            'ttft_p50_ms': round(float(np.median(ttft)), 2), 'prefill_total_s': round(prefill_s, 3),
            'prefill_tokens_per_s': round(B * S / prefill_s, 1),
            'step_roofline': {'bound': 'hbm', 'algorithmic_bytes_per_step': int(step_bytes),
                              'achieved': round(step_bytes / (dt / K) / 1e9, 1), 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                              'frac': round(step_bytes / (dt / K) / 1e9 / HBM_PEAK_GBPS, 4)},
        }
        if canary_note:
            comm_note = (comm_note + '; ' if comm_note else '') + canary_note
        if comm_note:
            out['config']['collectives_note'] = comm_note
        if tp > 1 or emu > 1:
            # prefill-sized forwards: all-reduces on the side stream under the other row half's GEMMs (TM_COMM_STREAM, default on)
            out['config']['prefill_allreduce_overlap'] = {
                'side_stream': cinfo['side_stream'], 'overlapped_forwards': cinfo['overlapped_forwards'],
                'microbatch_forwards': cinfo['microbatch_forwards'],
                'side_stream_allreduces': cinfo['side_stream_allreduces']}
            if emu > 1 and float(os.environ.get('TM_EMULATE_AR_GBPS', '0') or 0) > 0:
                out['config']['prefill_allreduce_overlap']['emulated_exchange_gbps'] = float(os.environ['TM_EMULATE_AR_GBPS'])
                out['config']['prefill_allreduce_overlap']['note'] = (
                    'one rank: the all-reduce is the identity; an idle one-workgroup kernel holds the stream for 10 us + bytes / that '
                    'bandwidth behind every prefill all-reduce -- shows what the choreography hides, not a multi-GPU result')
        if ranks_per_device > 1:
            out['config']['devices_shared'] = True
            out['config']['ranks_per_device'] = ranks_per_device
            out['metric'] += ' -- RANKS SHARE A DEVICE: launcher / bring-up test, not a multi-GPU result'
        if replicas:
            out['metric'] += f' -- TENSOR PARALLELISM COULD NOT RUN ON THIS NODE: {world} independent TP=1 replicas (weak scaling), see config.collectives_note'
        if tp > 1 or emu > 1:
            out['scaling_note'] = ('no 1 -> 8 GPU scaling curve of this engine has been measured on hardware before this run: the tensor-parallel '
                                   'path was validated on one device only (two processes on one GPU over the native communicator, 1-rank RCCL, '
                                   'gloo world-size-2 CPU tests)')
        if args.model != 'llama3_8b':
            out['metric'] = (f'decode tokens/sec, {args.model} {"FP8 block-scaled weights" if weight_type == 2 else "W4A16"} {kv_name} '
                             f'batch {B} (NOT the headline config)')
            out['config']['workload'] = out['config']['workload'].replace('Llama-3-8B', args.model)
        if emu > 1:
            out['metric'] = (f'PER-RANK EMULATION of TP={emu} on one GPU (one rank\'s shard, 1-rank collectives): '
                             'decode tokens/sec of the rank-local kernels only -- not a multi-GPU result')
            out['config']['emulated_tp'] = emu
        if prof:
            attn_ms, attn_n = prof['attention']
            per_launch_bytes = B * ctx_prof * kv_tok / model['layers']      # one layer's KV of the whole batch
            per_launch_s = attn_ms / 1e3 / max(model['layers'], 1)          # attention (+ split-K merge) of one layer
            ach = per_launch_bytes / per_launch_s / 1e9
            # HBM bytes per launch: measured in THIS run by a rocprofv3 --pmc FETCH_SIZE child pass (null if unavailable)
            traffic, traffic_src, gemm_traffic = None, 'skipped', None
            if not args.no_traffic and not child and world == 1:
                traffic, traffic_src, gemm_traffic = measure_attention_traffic(args, ctx_prof)
            attn_kernel = {8: 'decode_attention_i8_mfma_kernel<fused, 8>', 4: 'decode_attention_i8_mfma_kernel<fused, 4> (int4 codes expanded to '
                           'bytes on the way into LDS)'}.get(args.quant_policy, 'decode_attention_kernel<16> (fp16 KV, VALU)')
            attn_obj = {'bound': 'hbm', 'kernel': attn_kernel,
                        'achieved': round(ach, 1), 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                        'frac': round(ach / HBM_PEAK_GBPS, 4), 'traffic': traffic, 'traffic_source': traffic_src,
                        'bytes_per_launch': int(per_launch_bytes), 'us_per_launch': round(per_launch_s * 1e6, 2),
                        'ctx': ctx_prof, 'timing': f'HIP events, {P} eager steps after the timed region'}
            out['kernel_ms_per_step'] = {k: round(v[0], 4) for k, v in prof.items()}
            out['kernel_ms_per_step_note'] = 'eager launches with HIP events around every kernel: slower than the graph replays `value` times'
            # ---- in-graph durations of the timed path (a rocprofv3 --kernel-trace child over hipGraph replays, same tilings) ----
            ing, ing_src = (None, 'skipped')
            dense_u4 = weight_type == 0 and not model.get('moe_experts') and B <= 64
            if not args.no_traffic and not child and world == 1 and dense_u4:
                ing, ing_src = measure_in_graph_durations(args, table_text, ctx_prof, model['layers'], B)
            if ing and 'attention' in ing:
                a2 = per_launch_bytes / (ing['attention'] * 1e-6) / 1e9
                attn_obj.update(achieved=round(a2, 1), frac=round(a2 / HBM_PEAK_GBPS, 4), us_per_launch=round(ing['attention'], 2),
                                timing=ing_src, us_per_launch_eager=round(per_launch_s * 1e6, 2))
            # `roofline` = the kernel (family) that takes the most time of a step: at the driver command's context
            # the four W4A16 decode GEMMs of a layer; the attention and the whole step follow as named extras
            gemm_roles = ('w_qkv', 'wo', 'w1w3', 'w2')
            if ing and all(r in ing for r in gemm_roles):
                per_b, _ = gemm_family_bytes(model, max(tp, emu, 1))
                fam = family_roofline(per_b, {r: ing[r] for r in gemm_roles})
                lin_flop = 2.0 * B * sum(per_b.values()) / (0.5 + 1.0 / 32.0)      # 2 * M * K * N summed over the four linears
                gemm_obj = {'bound': 'hbm', 'kernel': 'the four W4A16 decode GEMMs of a layer (gemm_dec32_kernel / gemm_dec_lc_kernel: w_qkv, wo, '
                                                      'w1w3, w2)', 'achieved': fam['achieved'], 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s', 'frac': fam['frac'],
                            'traffic': gemm_traffic['bytes_per_layer'] if gemm_traffic else None,
                            'traffic_source': 'FETCH_SIZE x 1024 x 2 of the M <= 64 GEMM dispatches in the child PMC pass, per layer' if gemm_traffic else 'skipped',
                            'bytes_per_launch_group': fam['bytes_per_layer'], 'us_per_launch_group': fam['us_per_layer'],
                            'us_per_launch': {r: round(ing[r], 2) for r in gemm_roles},
                            'frac_per_launch': {r: round(per_b[r] / (ing[r] * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4) for r in gemm_roles},
                            'mfma_tflops': round(lin_flop / (fam['us_per_layer'] * 1e-6) / 1e12, 1),
                            'mfma_util': round(lin_flop / (fam['us_per_layer'] * 1e-6) / 1e12 / 2500.0, 4),
                            'launch_group': 'one layer = 4 launches', 'ctx': ctx_prof, 'timing': ing_src}
                attn_us = ing.get('attention', per_launch_s * 1e6)
                if fam['us_per_layer'] >= attn_us:
                    out['roofline'] = gemm_obj
                    out['roofline']['why_this_kernel'] = (f'time-dominant per layer: GEMMs {fam["us_per_layer"]:.1f} us vs attention {attn_us:.1f} us '
                                                          f'at ctx {ctx_prof}')
                    out['attention_roofline'] = attn_obj
                else:
                    out['roofline'] = attn_obj
                    out['roofline']['why_this_kernel'] = (f'time-dominant per layer: attention {attn_us:.1f} us vs GEMMs {fam["us_per_layer"]:.1f} us '
                                                          f'at ctx {ctx_prof}')
                    out['gemm_family_roofline'] = gemm_obj
            else:
                out['roofline'] = attn_obj
                out['roofline']['in_graph_trace'] = ing_src
            gemm_ms = sum(prof[k][0] for k in ('gemm_qkv', 'gemm_o', 'gemm_gate_up', 'gemm_down'))
            wbytes = (stats['weight_bytes'] - 2 * model['hidden'] * model['vocab'] / tp)
            out['gemm_roofline'] = {'bound': 'hbm', 'timing': 'eager HIP events (see roofline / gemm_family_roofline for the in-graph launches)',
                                    'achieved': round(wbytes / (gemm_ms / 1e3) / 1e9, 1),
                                    'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                                    'frac': round(wbytes / (gemm_ms / 1e3) / 1e9 / HBM_PEAK_GBPS, 4),
                                    'bytes_per_step': int(wbytes)}
            # matrix-pipe view of the same four GEMMs (north_star: "MFMA utilisation for the dequant-GEMM against gfx950 peak"):
            # 2 * B * (quantised linear parameters) flop per step over their time, against the dense fp16 peak of 2.5 PFLOP/s
            lin_params = model['layers'] * (H_ * (hq_l + 2 * hkv_l) * D_ + hq_l * D_ * H_ + 3 * H_ * I_l) if not model.get('moe_experts') else 0
            if lin_params:
                tf = 2.0 * B * lin_params / (gemm_ms / 1e3) / 1e12
                out['gemm_roofline']['mfma_tflops'] = round(tf, 1)
                out['gemm_roofline']['mfma_util'] = round(tf / 2500.0, 4)
                # prefill: all linear flops of the prompt tokens over the WHOLE prefill time (attention, norms, KV stores included):
                # a lower bound of the prefill GEMMs' matrix-pipe utilisation
                tfp = 2.0 * B * S * lin_params / prefill_s / 1e12
                out['prefill_mfma_util'] = {'linear_tflops_over_whole_prefill': round(tfp, 1), 'frac_of_2500': round(tfp / 2500.0, 4)}
            if gemm_traffic:
                # HBM-side bytes of the four decode GEMM launches of one layer (same child PMC pass as roofline.traffic) against the
                # algorithmic weight bytes of a layer: > 1 would mean weights fetched more than once
                alg = wbytes / model['layers']
                out['roofline_gemm_traffic'] = {'bytes_per_layer': gemm_traffic['bytes_per_layer'], 'algorithmic_bytes_per_layer': int(alg),
                                                'ratio': round(gemm_traffic['bytes_per_layer'] / alg, 4), 'launches': gemm_traffic['launches'],
                                                'source': 'FETCH_SIZE x 1024 x 2 of the M <= 64 GEMM dispatches in the child PMC pass'}
        if full is not None:
            out['value_1k_out'] = full['value']      # the metric-faithful rate: all 1024 generated tokens of the batch
            out['value_full_run'] = full
            out['value_full_run']['step_roofline_frac'] = round(
                algorithmic_bytes(model, B, (full['ctx_first'] + full['ctx_last']) / 2.0, kv_bits, tp)[0] / (full['ms_per_step'] / 1e3) / 1e9
                / HBM_PEAK_GBPS, 4) if weight_type == 0 and not model.get('moe_experts') else None
        if not args.no_cpu_baseline and world == 1 and not child:
            from oracle import cpu_baseline
            out['cpu_baseline'] = cpu_baseline.run(model, B, S, sample_layers=2)
        out['sample_tokens'] = toks[0, :4].tolist()
        import ctypes
        ctypes.CDLL(None).fflush(None)     # whatever C stdio still buffers goes to stderr (claim_stdout), the JSON line alone to stdout
        emit_json(out)
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
