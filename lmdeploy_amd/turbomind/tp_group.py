"""`pipeline(model, backend_config=TurbomindEngineConfig(tp=N))` as ONE call: the ranks of the node are started from here.

Reference: lmdeploy/turbomind/turbomind.py:187-217 starts one thread per GPU of the node from the single `TurboMind(...)` call
(create_context / process_weight / create_engine concurrently), src/turbomind/turbomind.cc:206-279 builds the communicator groups
with host barriers between the phases.  The MI355X engine is one PROCESS per GPU (one HIP context, one RCCL rank, IPC-mapped
peer segments for the native communicator), so the same call here owns N - 1 worker processes next to the caller's own rank 0:

    parent (rank 0)                                    worker r = 1 .. N-1 (`python -m lmdeploy_amd.turbomind.tp_group`)
    ----------------------------------------------     ---------------------------------------------------------
    Pipeline.__init__                                  Pipeline.__init__(..., rank=r, _tp_link=WorkerLink)
      Engine(tp=N, rank=0)                               Engine(tp=N, rank=r)
      link.setup_comm(engine)   <- unique id, oks ->     link.setup_comm(engine)      RCCL; every rank falls back to the native
                                <- IPC handles    ->                                  P2P communicator TOGETHER when any rank fails
      weights (rank 0's shard), start                    weights (rank r's shard), start
      engine = TpEngine(engine, link)                    link.serve(engine): executes the mirrored engine calls until close

Every state-changing engine call of the parent (prefill / decode / submit / step / cancel / release / set_sampling / ...) is sent
to the workers BEFORE the parent executes it itself (the collectives inside need all ranks in the call), replies are collected
afterwards; a worker's error is raised in the parent with its status code.  Reads (poll / fetch / stats) are rank 0's: the sampled
token is identical on every rank (candidate all-gather / identical Philox draw), so the schedulers of all ranks take the same
decisions from the same call sequence.  The expert path (`rank=`, `comm_unique_id=`, one externally launched process per GPU, e.g.
torch.distributed.run) is unchanged.
"""
from __future__ import annotations

import os
import traceback

from .. import _ffi
from .engine import Engine

# engine methods every rank must execute, in the same order
MIRRORED = ('prefill', 'decode', 'release', 'set_sampling', 'set_logits_params', 'set_logprobs', 'submit', 'step', 'step_many', 'cancel', 'forget', 'tune_gemm',
            'import_gemm_table', 'sync')
# reads rank 0 answers alone: they touch no collective and change no state the other ranks would have to follow
RANK0_READS = ('poll', 'poll_logprobs', 'fetch', 'fetch_logprobs', 'fetch_logits', 'fetch_residual', 'fetch_kv_block', 'stats', 'comm_info', 'prefill_times_ms', 'mixed_steps',
               'overlapped_steps', 'pick_tiling', 'pick_general', 'export_gemm_table', 'stream', 'cfg', 'batch', 'max_new', 'PROF_CATEGORIES')
# everything else of Engine either runs a forward or changes engine state on ONE rank (serve_start / serve_stop / wait: the engine thread;
# profile_decode; init_synthetic / load_weights / start / comm_*: construction): rank 0 alone would enter collectives the workers never
# join (RCCL: hangs without bound; native communicator: its 30 s give-up, terminal) -- refused loudly
_TIMEOUT_S = float(os.environ.get('TM_TP_GROUP_TIMEOUT_S', '600'))


def _setup_comm(eng: Engine, want_rccl: bool, uid: bytes, all_gather, rows: int, want_native: bool = False) -> str:
    """The same sequence on every rank: RCCL unless the ranks share a device (RCCL refuses duplicate devices); if ANY rank failed,
    ALL ranks drop RCCL and bring up the native P2P communicator (fused all-reduce + residual + RMSNorm, comm_p2p.hip).
    want_native (`TurbomindEngineConfig.communicator` = 'native' / 'cuda-ipc', the reference's names for its own in-house communicator,
    src/turbomind/comm/device_comm.cc:14-30): the native communicator serves the decode-sized forwards -- one fused launch per
    exchange, as the reference's AllreduceResidualBiasRMSnorm -- and RCCL, where it came up on every rank, keeps the prefill-sized
    ones (side-stream overlap)."""
    # the switches that decide HOW MANY collectives a forward issues are read by every rank from its own environment (engine_comm.hip:
    # TM_COMM_STREAM / TM_PIPE_MIN_ROWS / TM_PIPE_MICROBATCH select the two-micro-batch prefill schedule): a mismatch would change the
    # collective sequence on one rank only and deadlock the group -- compare them before any communicator exists
    knobs = tuple(os.environ.get(k, '') for k in ('TM_COMM_STREAM', 'TM_PIPE_MIN_ROWS', 'TM_PIPE_MICROBATCH', 'TM_COMM', 'TM_GRAPH_COMM',
                                                  'TM_FOLD_NORM', 'TM_P2P_DECODE_ROWS'))
    seen = all_gather(knobs)
    if any(k != seen[0] for k in seen):
        raise _ffi.TmError(1, f'tensor-parallel ranks disagree on the collective schedule switches (TM_COMM_STREAM, TM_PIPE_MIN_ROWS, '
                              f'TM_PIPE_MICROBATCH, TM_COMM, TM_GRAPH_COMM, TM_FOLD_NORM, TM_P2P_DECODE_ROWS) per rank: {seen}')
    ok = False
    if want_rccl:
        try:
            eng.comm_init(uid)
            ok = True
        except Exception:       # noqa: BLE001 -- reported through the gather below, every rank takes the same branch
            ok = False
    oks = all_gather(ok)
    if want_rccl and all(oks):
        # The DEFAULT arrangement (round 6): RCCL serves the prefill-sized forwards (side-stream overlap), the native
        # communicator the decode-sized ones -- wo / w2 -> ONE fused all-reduce + residual + RMSNorm launch per exchange, as the
        # reference's AllreduceResidualBiasRMSnorm (fused_allreduce.cu:406-500, unified_decoder.cc:278-285,328-335) -- provided its bring-up
        # self-test passes on EVERY rank (the hop between devices is what no single-GPU test covers); else RCCL + the norm launch.
        # TM_COMM=rccl (or TM_NATIVE_DECODE=0) keeps RCCL alone; `communicator='native' / 'cuda-ipc'` asks for the same arrangement.
        if os.environ.get('TM_COMM', '') == 'rccl' or os.environ.get('TM_NATIVE_DECODE', '1') == '0':
            return 'rccl'
        try:
            eng.comm_native_setup(all_gather, rows=rows)
            good = eng.comm_native_selftest()
        except Exception:       # noqa: BLE001 -- reported through the gather: every rank takes the same branch
            good = False
        if all(all_gather(bool(good))):
            return 'native-p2p (decode) + rccl (large forwards)'
        eng.comm_native_drop()
        if want_native:
            raise _ffi.TmError(5, 'communicator="native" / "cuda-ipc": the native communicator failed its bring-up self-test on at least one rank')
        return 'rccl (native communicator failed its bring-up self-test)'
    if want_rccl and ok:
        eng.comm_drop_rccl()
    eng.comm_native_setup(all_gather, rows=rows)
    return 'native-p2p'


class ParentLink:
    """rank 0's end: one authenticated localhost connection per worker process.  The workers are fresh interpreters
    (`python -m lmdeploy_amd.turbomind.tp_group`), not forks / multiprocessing children: no HIP context is inherited and the caller's
    script is not re-imported (pipeline() may be called from unguarded top-level code, as with the reference's in-process ranks)."""

    def __init__(self, tp: int, model_path: str, backend_config, conns=None):
        import subprocess
        import sys
        from multiprocessing.connection import Listener
        self.tp = tp
        self.conns, self.procs = [], []
        if conns is not None:       # established connections (tests drive the protocol with stub engines on threads)
            self.conns = list(conns)
            return
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC: what RCCL and the native segments need on this pool
        key = os.urandom(16)
        root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        env = dict(os.environ, TM_TP_GROUP_KEY=key.hex(),
                   PYTHONPATH=root + (os.pathsep + os.environ['PYTHONPATH'] if os.environ.get('PYTHONPATH') else ''))
        with Listener(('127.0.0.1', 0), authkey=key) as srv:
            port = srv.address[1]
            for r in range(1, tp):
                self.procs.append(subprocess.Popen([sys.executable, '-m', 'lmdeploy_amd.turbomind.tp_group', str(port), str(r)], env=env,
                                                   stdout=sys.stderr))        # a worker never writes to the caller's stdout
            by_rank = {}
            try:
                srv._listener._socket.settimeout(_TIMEOUT_S)      # accept() must not wait forever for a worker that died at import
            except AttributeError:
                pass
            try:
                for _ in range(1, tp):
                    c = srv.accept()
                    by_rank[c.recv()] = c
            except OSError as e:       # socket.timeout: a worker never connected
                self._kill()
                raise _ffi.TmError(5, f'tensor-parallel workers did not connect within {_TIMEOUT_S:.0f} s: {e}') from None
            except BaseException:      # e.g. multiprocessing.AuthenticationError from accept(): no worker may outlive the failed constructor
                self._kill()
                raise
        self.conns = [by_rank[r] for r in range(1, tp)]
        self.broadcast((model_path, backend_config))

    def _kill(self):
        for p in self.procs:
            if p.poll() is None:
                p.terminate()

    def _recv(self, i):
        c = self.conns[i]
        if not c.poll(_TIMEOUT_S):
            raise _ffi.TmError(5, f'tensor-parallel rank {i + 1} did not answer within {_TIMEOUT_S:.0f} s')
        try:
            return c.recv()
        except EOFError:
            code = self.procs[i].poll() if i < len(self.procs) else None
            raise _ffi.TmError(5, f'tensor-parallel rank {i + 1} exited (exit code {code})') from None

    def broadcast(self, obj):
        for c in self.conns:
            c.send(obj)

    def all_gather(self, mine):
        """host-side all-gather in rank order; the workers call WorkerLink.all_gather at the same point of the same sequence"""
        vals = [mine]
        for i in range(len(self.conns)):
            tag, v = self._recv(i)
            if tag != 'gather':
                raise _ffi.TmError(5, f'tensor-parallel rank {i + 1}: {v}' if tag == 'error' else f'protocol: expected gather, got {tag}')
            vals.append(v)
        self.broadcast(vals)
        return vals

    def setup_comm(self, eng: Engine, want_rccl: bool, rows: int, want_native: bool = False) -> str:
        uid = Engine.comm_unique_id()
        self.broadcast(('comm', want_rccl, uid, rows, want_native))
        return _setup_comm(eng, want_rccl, uid, self.all_gather, rows, want_native)

    def wait_ready(self):
        """every worker reports ('ready', rank) once its engine has its weights and is started"""
        for i in range(len(self.conns)):
            rep = self._recv(i)
            if rep[0] != 'ready':
                raise _ffi.TmError(5, f'tensor-parallel rank {i + 1} failed to start: {rep[1] if len(rep) > 1 else rep}')

    def collect(self):
        """one reply per worker for the call just mirrored: ('ok', value) | ('error', status, message)"""
        out = []
        for i in range(len(self.conns)):
            out.append(self._recv(i))
        return out

    def close(self):
        for c in self.conns:
            try:
                c.send(None)
            except (BrokenPipeError, OSError):
                pass
        for p in self.procs:
            try:
                p.wait(timeout=30)
            except Exception:       # noqa: BLE001 -- subprocess.TimeoutExpired
                p.terminate()
                try:
                    p.wait(timeout=5)
                except Exception:   # noqa: BLE001
                    p.kill()
        for c in self.conns:
            c.close()
        self.conns, self.procs = [], []


class WorkerLink:
    """a worker's end of its pipe to rank 0"""

    def __init__(self, conn):
        self.conn = conn

    def all_gather(self, mine):
        self.conn.send(('gather', mine))
        return self.conn.recv()

    def setup_comm(self, eng: Engine, want_rccl: bool, rows: int, want_native: bool = False) -> str:
        tag, want, uid, rows0, native0 = self.conn.recv()     # rank 0's decisions, not this rank's guesses
        assert tag == 'comm', tag
        return _setup_comm(eng, want, uid, self.all_gather, rows0, native0)

    def serve(self, eng: Engine):
        """execute the parent's mirrored calls until it closes the group"""
        while True:
            try:
                msg = self.conn.recv()
            except EOFError:
                break
            if msg is None:
                break
            name, args, kwargs = msg
            try:
                self.conn.send(('ok', getattr(eng, name)(*args, **kwargs)))
            except _ffi.TmError as e:
                self.conn.send(('error', e.status, str(e)))
            except Exception as e:      # noqa: BLE001 -- the parent must hear about it instead of waiting for the timeout
                self.conn.send(('error', 5, f'{type(e).__name__}: {e}'))


class TpEngine:
    """rank 0's Engine + the mirrored workers behind the interface Pipeline uses"""

    def __init__(self, eng: Engine, link: ParentLink, backend: str):
        self._eng, self._link, self.comm_backend = eng, link, backend

    def __getattr__(self, name):
        attr = getattr(self._eng, name)
        if name in RANK0_READS or name.startswith('_'):
            return attr
        if name not in MIRRORED:
            if not callable(attr):
                return attr

            def refuse(*_a, **_k):
                raise NotImplementedError(f'Engine.{name} on a one-call tensor-parallel pipeline: it would run on rank 0 alone while the other '
                                          f'ranks wait in a collective; mirrored calls: {", ".join(MIRRORED)}')
            return refuse

        def call(*args, **kwargs):
            self._link.broadcast((name, args, kwargs))
            err = None
            try:
                out = attr(*args, **kwargs)
            except BaseException as e:      # noqa: BLE001 -- ANY failure of rank 0's own call (a ValueError from argument conversion as well as
                err, out = e, None          # a TmError): the workers ran the call and queued a reply each -- drain them, or every later call
            try:                            # reads the previous call's replies and the ranks drift apart
                replies = self._link.collect()
            except BaseException:           # noqa: BLE001 -- a worker died / timed out: rank 0's own error is the more useful one
                if err is not None:
                    raise err
                raise
            if err is not None:
                raise err
            for r, rep in enumerate(replies, start=1):
                if rep[0] == 'error':
                    raise _ffi.TmError(rep[1], f'tensor-parallel rank {r}: {rep[2]}')
            return out
        return call

    def close(self):
        try:
            self._link.close()
        finally:
            self._eng.close()


def worker_main(port: int, rank: int):
    """a worker process: the same Pipeline construction as the caller's, for rank `rank`, then the serve loop"""
    from multiprocessing.connection import Client
    conn = Client(('127.0.0.1', port), authkey=bytes.fromhex(os.environ['TM_TP_GROUP_KEY']))
    conn.send(rank)
    try:
        model_path, backend_config = conn.recv()
        from ..pipeline import Pipeline
        link = WorkerLink(conn)
        pipe = Pipeline(model_path, backend_config, rank=rank, _tp_link=link)
        conn.send(('ready', rank))
        link.serve(pipe.engine)
        pipe.engine.close()
    except Exception as e:      # noqa: BLE001
        try:
            conn.send(('error', f'{type(e).__name__}: {e}\n{traceback.format_exc()}'))
        except (BrokenPipeError, OSError):
            pass
    finally:
        conn.close()


if __name__ == '__main__':
    import sys
    worker_main(int(sys.argv[1]), int(sys.argv[2]))
