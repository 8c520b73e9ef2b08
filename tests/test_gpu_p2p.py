"""the native communicator's fused all-reduce + residual + RMSNorm (comm_p2p.hip) against the oracle.

One GPU is all a test box has, so the tp ranks are (a) tp segments on the same device driven from tp streams, and (b) tp
PROCESSES on the same device that exchange IPC handles and map each other's segments -- the same C-ABI calls, pointer
tables and flag protocol a one-process-per-GPU run uses, minus the xGMI hop.  Every run is a subprocess under a hard
timeout: a protocol bug means ranks waiting for each other (the kernel's spins are bounded, ~1 s).
The stream emulation stops at tp = 4: the HIP runtime multiplexes streams onto 4 hardware queues, and two "ranks" that
share a queue cannot run concurrently (the second kernel sits behind the first one's successor) -- an artefact of
emulating ranks with streams; tp = 8 runs as 8 processes instead."""
import json
import os
import subprocess
import sys
import tempfile

import pytest

pytestmark = pytest.mark.gpu
WORKER = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'p2p_worker.py')


def _env():
    env = dict(os.environ)
    env['GPU_MAX_HW_QUEUES'] = '8'            # tp concurrent kernels need tp hardware queues
    env['HSA_ENABLE_IPC_MODE_LEGACY'] = '0'
    env['TM_P2P_2SHOT_GRID'] = '48'           # the tp ranks share ONE device here: their persistent grids must fit together
    return env


def _last_json(text):
    for line in reversed(text.strip().splitlines()):
        if line.startswith('{'):
            return json.loads(line)
    raise AssertionError('no result line:\n' + text[-2000:])


@pytest.mark.timeout(400)
@pytest.mark.parametrize('tp,M,H', [(2, 64, 4096), (4, 64, 4096), (3, 16, 8192), (2, 1, 2048), (4, 33, 5120)])
def test_p2p_allreduce_norm_streams(cuda, tp, M, H):
    p = subprocess.run([sys.executable, WORKER, 'streams', str(tp), str(M), str(H), '6'], env=_env(), capture_output=True,
                       text=True, timeout=300)
    res = _last_json(p.stdout + '\n' + p.stderr)
    assert res['ok'], res


@pytest.mark.timeout(400)
@pytest.mark.parametrize('tp,M,H', [(2, 64, 4096), (4, 8, 4096), (8, 8, 4096)])
def test_p2p_allreduce_norm_ipc_processes(cuda, tp, M, H):
    with tempfile.TemporaryDirectory() as d:
        procs = [subprocess.Popen([sys.executable, WORKER, 'ipc', d, str(r), str(tp), str(M), str(H), '5'], env=_env(),
                                  stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(tp)]
        outs = []
        try:
            for pr in procs:
                outs.append(pr.communicate(timeout=300)[0])
        finally:
            for pr in procs:
                if pr.poll() is None:
                    pr.kill()
        for r, text in enumerate(outs):
            res = _last_json(text)
            assert res['ok'], (r, res)


@pytest.mark.timeout(400)
@pytest.mark.parametrize('tp,M,H', [(2, 300, 4096), (4, 1000, 4096), (3, 130, 8192), (2, 1, 2048), (4, 3, 5120)])
def test_p2p_allreduce_norm_two_shot_streams(cuda, tp, M, H):
    """the two-shot form (reduce-scatter, norm on the owned row slice, all-gather by push; reference
    comm/cuda_ipc/fused_allreduce.cu:25-405): all M normed rows on every rank equal the oracle and each other, the residual
    stream is current on every rank's own slice; ragged last slice, slices of zero rows (M < tp), two vectors per thread"""
    p = subprocess.run([sys.executable, WORKER, 'streams', str(tp), str(M), str(H), '5', '2'], env=_env(), capture_output=True,
                       text=True, timeout=300)
    res = _last_json(p.stdout + '\n' + p.stderr)
    assert res['ok'], res


@pytest.mark.timeout(400)
@pytest.mark.parametrize('tp,M,H', [(2, 520, 4096), (8, 260, 4096)])
def test_p2p_allreduce_norm_two_shot_ipc_processes(cuda, tp, M, H):
    with tempfile.TemporaryDirectory() as d:
        procs = [subprocess.Popen([sys.executable, WORKER, 'ipc', d, str(r), str(tp), str(M), str(H), '4', '2'], env=_env(),
                                  stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(tp)]
        outs = []
        try:
            for pr in procs:
                outs.append(pr.communicate(timeout=300)[0])
        finally:
            for pr in procs:
                if pr.poll() is None:
                    pr.kill()
        for r, text in enumerate(outs):
            res = _last_json(text)
            assert res['ok'], (r, res)


@pytest.mark.timeout(400)
@pytest.mark.parametrize('tp,M,H', [(2, 64, 4096), (4, 64, 4096), (3, 16, 8192), (2, 1, 2048), (4, 33, 5120), (4, 300, 4096)])
def test_p2p_allreduce_norm_row_flags_streams(cuda, tp, M, H):
    """Round 6: the one-shot form whose workgroups do not wait for each other (tm_p2p_allreduce_norm_rows: one flag word per (sender, row),
    per-row call counters, write-through system-scope stores and system-scope loads, no tickets / no fences / no residency requirement --
    M = 300 rows exceeds what the ticket form may launch) interleaved with the all-gather of the shared call sequence: the oracle's bits on
    every rank, per-row counters advanced by one per call, the shared counter by the all-gather alone."""
    p = subprocess.run([sys.executable, WORKER, 'streams', str(tp), str(M), str(H), '6', '3'], env=_env(), capture_output=True,
                       text=True, timeout=300)
    res = _last_json(p.stdout + '\n' + p.stderr)
    assert res['ok'], res


@pytest.mark.timeout(400)
@pytest.mark.parametrize('tp,M,H', [(2, 64, 4096), (4, 8, 4096), (8, 8, 4096)])
def test_p2p_allreduce_norm_row_flags_ipc_processes(cuda, tp, M, H):
    with tempfile.TemporaryDirectory() as d:
        procs = [subprocess.Popen([sys.executable, WORKER, 'ipc', d, str(r), str(tp), str(M), str(H), '5', '3'], env=_env(),
                                  stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(tp)]
        outs = []
        try:
            for pr in procs:
                outs.append(pr.communicate(timeout=300)[0])
        finally:
            for pr in procs:
                if pr.poll() is None:
                    pr.kill()
        for r, text in enumerate(outs):
            res = _last_json(text)
            assert res['ok'], (r, res)
