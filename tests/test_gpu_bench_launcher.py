"""bench.py's multi-rank path on the ONE-GPU test box: `--gpus 2` starts two ranks that share cuda:0.  RCCL refuses two ranks on
one device, which is exactly the failure the launcher must survive: every rank reports it, ALL ranks drop RCCL and continue on
the native P2P communicator (comm_p2p.hip), rank 0's measured GEMM dispatch table is broadcast, the decode step (collectives
included) is captured in a hipGraph, and the JSON line says what actually ran (`tm_engine_comm_info`), not what was intended.
(the first multi-GPU run must be survivable.  Reference: one process per rank + NCCL id exchange,
lmdeploy/turbomind/turbomind.py:187-217, src/turbomind/comm/nccl/nccl.cu:505-518.)"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(600)
def test_bench_two_ranks_on_one_gpu_falls_back_to_the_native_communicator(cuda):
    env = dict(os.environ, GPU_MAX_HW_QUEUES='8', HSA_ENABLE_IPC_MODE_LEGACY='0', TM_P2P_2SHOT_GRID='96')
    env.pop('WORLD_SIZE', None)
    pr = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '6', '--warmup', '3', '--layers', '2',
                         '--batch', '8', '--prompt-len', '96', '--no-traffic', '--no-cpu-baseline', '--no-full-run', '--profile-steps', '0',
                         '--allow-shared-devices'],
                        env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=480)
    assert pr.returncode == 0, pr.stderr[-3000:]
    lines = pr.stdout.strip().splitlines()
    assert len(lines) == 1 and lines[0].startswith('{'), f'stdout must be the JSON line alone (native banners go to stderr): {lines[:3]}'
    d = json.loads(lines[0])
    cfg = d['config']
    assert cfg['devices_shared'] is True and cfg['ranks_per_device'] == 2 and 'RANKS SHARE A DEVICE' in d['metric']
    assert d['n_gpus'] == 2 and d['value'] > 0 and cfg['parallelism'] == 'tp2'
    assert cfg['collectives'] == 'native-p2p' and cfg['rccl_ranks'] == 2, cfg
    assert 'RCCL init failed' in cfg['collectives_note']
    assert cfg['hipgraph'] is True
    assert set(cfg['gemm_tilings']) == {'w_qkv', 'wo', 'w1w3', 'w2'} and 'broadcast' in cfg['gemm_dispatch']
    assert 'scaling_note' in d
    assert 'RCCL communicator failed' in pr.stderr


@pytest.mark.timeout(900)
def test_bench_refuses_shared_devices_unless_asked(cuda):
    """`--gpus 2` on a one-GPU box silently put both ranks on cuda:0 and reported tp2 as if it were a scaling run."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.pop('WORLD_SIZE', None)
    pr = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--layers', '1',
                         '--batch', '4', '--prompt-len', '64', '--no-traffic', '--no-cpu-baseline', '--no-full-run', '--profile-steps', '0'],
                        env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert pr.returncode != 0 and 'ranks would share devices' in pr.stderr and pr.stdout.strip() == ''


@pytest.mark.timeout(1200)
def test_bench_eight_ranks_on_one_gpu_bring_up(cuda):
    """the first real 8-GPU run must not die in bring-up.  All eight ranks of the TP = 8 job (one kv head and
    four q heads per rank, vocabulary / inter sharded eight ways) as eight processes on the one GPU: launcher, RCCL refusal ->
    native communicator on every rank, table broadcast, graph capture with the collectives inside, greedy tokens out."""
    env = dict(os.environ, GPU_MAX_HW_QUEUES='8', HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.pop('WORLD_SIZE', None)
    pr = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--steps', '4', '--warmup', '2', '--layers', '2',
                         '--batch', '8', '--prompt-len', '96', '--no-traffic', '--no-cpu-baseline', '--no-full-run', '--profile-steps', '0',
                         '--allow-shared-devices', '--tune', '0'],
                        env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1100)
    assert pr.returncode == 0, pr.stderr[-3000:]
    lines = pr.stdout.strip().splitlines()
    assert len(lines) == 1, lines[:3]
    d = json.loads(lines[0])
    cfg = d['config']
    assert d['n_gpus'] == 8 and d['value'] > 0 and cfg['parallelism'] == 'tp8' and cfg['ranks_per_device'] == 8
    assert cfg['collectives'] == 'native-p2p' and cfg['rccl_ranks'] == 8, cfg


@pytest.mark.timeout(900)
@pytest.mark.parametrize('force_replicas', [0, 1])
def test_bench_canary_fallbacks(cuda, force_replicas):
    """The driver's scaling run is the first time this engine's collectives cross xGMI.  bench.py sends a canary child through the collective
    paths first; when it fails (forced here) ALL ranks take the conservative switches after a second canary with those switches passed --
    and when that fails too (forced) the job still produces a record: N independent TP = 1 replicas, labelled as such (weak scaling, no
    collective), instead of a hang."""
    env = dict(os.environ, GPU_MAX_HW_QUEUES='8', HSA_ENABLE_IPC_MODE_LEGACY='0', TM_P2P_2SHOT_GRID='96', TM_BENCH_CANARY_SHARED='1',
               TM_BENCH_CANARY_FAIL='1', TM_BENCH_FORCE_REPLICAS=str(force_replicas))
    env.pop('WORLD_SIZE', None)
    pr = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '6', '--warmup', '3', '--layers', '2',
                         '--batch', '8', '--prompt-len', '96', '--no-traffic', '--no-cpu-baseline', '--no-full-run', '--profile-steps', '0',
                         '--allow-shared-devices'],
                        env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=800)
    assert pr.returncode == 0, pr.stderr[-3000:]
    lines = pr.stdout.strip().splitlines()
    assert len(lines) == 1 and lines[0].startswith('{'), lines[:3]
    d = json.loads(lines[0])
    cfg = d['config']
    assert 'canary run failed or hung on ranks [0, 1]' in cfg['collectives_note']
    if force_replicas:
        assert d['scaling'] == 'weak' and cfg['parallelism'] == 'dp2 (replicas only)' and 'TENSOR PARALLELISM COULD NOT RUN' in d['metric']
        assert cfg['collectives'] == 'none' and cfg['rccl_ranks'] == 0 and 'independent TP = 1 replicas' in cfg['collectives_note']
        assert d['n_gpus'] == 2 and d['value'] == pytest.approx(2 * 8 * d['steps'] / (d['ms_per_step'] * d['steps'] / 1e3), rel=1e-3)
    else:
        # the second canary ran with the conservative switches (two ranks on one GPU: RCCL refuses -> the native communicator, as in the first test)
        assert d['scaling'] == 'strong' and cfg['parallelism'] == 'tp2' and 'second canary' not in cfg['collectives_note']
        assert cfg['hipgraph'] is False, 'TM_GRAPH_COMM=0 of the fall-back must reach the run'
