"""Continuous-batching scheduler policy (lmdeploy_amd/csrc/scheduler.h), driven on the CPU through the C-ABI hooks
tm_sched_* exactly as the engine drives it (reference: engine/engine.cc:434-470, scheduler.cc:1018-1078):
arrival-order admission, slot reuse, 64-token block accounting for prompt + max_new_tokens, prefill token budget,
finish on EOS / length / cancel, reference status codes."""
import ctypes as C
import os

import numpy as np
import pytest

from lmdeploy_amd import _ffi

FINISH, CANCEL, INVALID, TOO_LONG, OOM = 7, 8, 1, 6, 11


class Sched:
    def __init__(self, max_batch, num_blocks, session_len):
        self.lib = _ffi.load()
        self.h = C.c_void_p()
        _ffi.check(self.lib.tm_sched_create(C.byref(self.h), max_batch, num_blocks, session_len))

    def submit(self, n, max_new, eos=-1):
        ids = np.arange(n, dtype=np.int32)
        rid = C.c_int64(0)
        rc = self.lib.tm_sched_submit(self.h, ids.ctypes.data, n, max_new, eos, C.byref(rid))
        return rc, rid.value

    def admit(self, budget=1 << 30):
        ids, slots, n = (C.c_int64 * 64)(), (C.c_int * 64)(), C.c_int(0)
        _ffi.check(self.lib.tm_sched_admit(self.h, budget, ids, slots, 64, C.byref(n)))
        return [(ids[i], slots[i]) for i in range(n.value)]

    def on_token(self, slot, tok):
        f = C.c_int(0)
        _ffi.check(self.lib.tm_sched_on_token(self.h, slot, tok, C.byref(f)))
        return bool(f.value)

    def cancel(self, rid):
        s = C.c_int(-1)
        rc = self.lib.tm_sched_cancel(self.h, rid, C.byref(s))
        return rc, s.value

    def query(self, rid):
        st, slot, ng, nb = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
        rc = self.lib.tm_sched_query(self.h, rid, C.byref(st), C.byref(slot), C.byref(ng), C.byref(nb))
        return rc, st.value, slot.value, ng.value, nb.value

    def admit_ready(self):
        r = C.c_int(0)
        _ffi.check(self.lib.tm_sched_admit_ready(self.h, C.byref(r)))
        return bool(r.value)

    def counts(self):
        a, w, f = C.c_int(0), C.c_int(0), C.c_int(0)
        _ffi.check(self.lib.tm_sched_counts(self.h, C.byref(a), C.byref(w), C.byref(f)))
        return a.value, w.value, f.value

    def abort_all(self, status):
        return self.lib.tm_sched_abort_all(self.h, status)

    def forget(self, rid):
        return self.lib.tm_sched_forget(self.h, rid)

    def __del__(self):
        self.lib.tm_sched_destroy(self.h)


def test_submit_status_codes():
    s = Sched(max_batch=2, num_blocks=4, session_len=200)
    assert s.submit(0, 5)[0] == INVALID
    assert s.submit(5, 0)[0] == INVALID
    assert s.submit(150, 51)[0] == TOO_LONG           # 201 > session_len
    s2 = Sched(max_batch=2, num_blocks=2, session_len=1000)
    assert s2.submit(100, 100)[0] == OOM              # 4 blocks can never fit a 2-block pool
    rc, rid = s.submit(100, 28)
    assert rc == 0 and rid == 1
    assert s.query(99)[0] == INVALID


def test_fifo_admission_slots_and_blocks():
    s = Sched(max_batch=2, num_blocks=10, session_len=4096)
    r = [s.submit(n, m)[1] for n, m in ((100, 28), (64, 1), (10, 10), (300, 20))]   # 2, 2, 1, 5 blocks
    adm = s.admit()
    assert adm == [(r[0], 0), (r[1], 1)]                     # arrival order, lowest free slot first
    assert s.counts() == (2, 2, 10 - 2 - 2)
    assert s.query(r[0])[4] == 2 and s.query(r[1])[4] == 2    # ceil(128/64), ceil(65/64)
    assert s.admit() == []                                    # no free slot
    assert s.on_token(1, 7) is True                           # max_new = 1: finished on its first token
    assert s.query(r[1])[:4] == (0, FINISH, -1, 1)
    assert s.counts() == (1, 2, 8)
    assert s.admit() == [(r[2], 1)]                           # the freed slot is reused
    # r[3] needs 5 blocks, 7 are free, but no slot: head-of-line blocked
    assert s.admit() == []
    for t in range(27):
        assert s.on_token(0, t) is False
    assert s.on_token(0, 99) is True                          # 28th token
    assert s.admit() == [(r[3], 0)]
    assert s.counts() == (2, 0, 10 - 1 - 5)


def test_block_shortage_blocks_the_queue_in_order():
    s = Sched(max_batch=4, num_blocks=4, session_len=4096)
    a = s.submit(100, 28)[1]      # 2 blocks
    b = s.submit(130, 62)[1]      # 3 blocks: does not fit next to a
    c = s.submit(10, 10)[1]       # 1 block: would fit, but order is never changed (scheduler.cc:1076-1078)
    assert s.admit() == [(a, 0)]
    assert s.admit() == []
    for _ in range(28):
        fin = s.on_token(0, 1)
    assert fin
    assert s.admit() == [(b, 0), (c, 1)]


def test_prefill_token_budget():
    s = Sched(max_batch=8, num_blocks=100, session_len=4096)
    r = [s.submit(n, 4)[1] for n in (300, 300, 300, 2000, 10)]
    assert [x[0] for x in s.admit(budget=700)] == r[:2]       # third would exceed 700 prompt tokens
    assert [x[0] for x in s.admit(budget=700)] == r[2:3]      # 300, then 2000 does not fit behind it
    assert [x[0] for x in s.admit(budget=700)] == r[3:4]      # larger than the budget: admitted alone (engine chunks it)
    assert [x[0] for x in s.admit(budget=700)] == r[4:]


def test_eos_and_cancel():
    s = Sched(max_batch=2, num_blocks=8, session_len=512)
    a = s.submit(5, 50, eos=42)[1]
    b = s.submit(5, 50)[1]                                    # eos < 0: ignore_eos
    c = s.submit(5, 50)[1]
    s.admit()
    assert not s.on_token(0, 1) and not s.on_token(1, 42)     # 42 is not b's stop token
    assert s.on_token(0, 42)                                  # EOS ends a (token included in the output)
    assert s.query(a)[:4] == (0, FINISH, -1, 2)
    assert s.cancel(c) == (0, -1)                             # cancelled while waiting: never admitted
    assert s.admit() == []
    assert s.query(c)[1] == CANCEL
    assert s.cancel(b) == (0, 1)                              # running: slot 1 is released
    assert s.query(b)[1] == CANCEL and s.counts() == (0, 0, 8)
    assert s.cancel(12345)[0] == INVALID
    assert not s.on_token(1, 3)                               # tokens of a parked slot are ignored


def test_abort_all_fails_everything_unfinished():
    """Device error in the engine thread: running and queued requests end with kFail (5), finished ones keep their
    status, slots and blocks return to the pool and new requests are admitted again."""
    FAIL = 5
    s = Sched(max_batch=2, num_blocks=6, session_len=256)
    ids = [s.submit(10, 5)[1] for _ in range(4)]
    assert [a[0] for a in s.admit()] == ids[:2]
    for _ in range(5):
        s.on_token(0, 1)                                  # request 0 finishes by length
    assert s.query(ids[0])[1] == FINISH
    assert s.abort_all(0) != 0                            # status must be non-zero
    assert s.abort_all(FAIL) == 0
    assert [s.query(i)[1] for i in ids] == [FINISH, FAIL, FAIL, FAIL]
    assert s.counts() == (0, 0, 6)
    assert not s.on_token(1, 3)                           # the aborted slot no longer belongs to a request
    rc, rid = s.submit(70, 5)
    assert rc == 0 and s.admit() == [(rid, 0)] and s.counts() == (1, 0, 4)


def test_forget_drops_only_finished_requests():
    """Long-lived sessions: a finished (or cancelled) request's record can be dropped, a queued or running one cannot;
    ids are never reused."""
    s = Sched(max_batch=1, num_blocks=4, session_len=128)
    a, b = s.submit(5, 2)[1], s.submit(5, 2)[1]
    assert s.admit() == [(a, 0)]
    assert s.forget(a) == INVALID and s.forget(b) == INVALID and s.forget(12345) == INVALID
    s.on_token(0, 1)
    assert s.on_token(0, 1)                                # a finished by length
    assert s.forget(a) == 0 and s.query(a)[0] == INVALID   # gone
    assert s.cancel(b)[0] == 0 and s.forget(b) == 0        # cancelled while waiting
    rc, c = s.submit(5, 1)
    assert rc == 0 and c not in (a, b) and s.admit() == [(c, 0)]


@pytest.mark.parametrize('seed', range(6))
def test_random_operation_sequences_keep_the_invariants(seed):
    """Randomised submit / admit / token / cancel / forget sequences against a Python model of the policy: block
    conservation (free + held = pool), at most one request per slot, arrival-order admission, terminal statuses never
    change, forgotten ids disappear, the counters agree with the model at every step."""
    rng = np.random.default_rng(seed)
    B, NB, SL = int(rng.integers(1, 5)), int(rng.integers(3, 12)), 256
    s = Sched(max_batch=B, num_blocks=NB, session_len=SL)
    blocks = lambda n: (n + 63) // 64
    model = {}                       # id -> dict(n, max_new, eos, status, slot, out, blocks)
    order, slots = [], [None] * B    # waiting ids in arrival order; slot -> id
    for step in range(400):
        op = rng.choice(['submit', 'admit', 'token', 'token', 'token', 'cancel', 'forget'])
        if op == 'submit':
            n, mx = int(rng.integers(1, 200)), int(rng.integers(1, 80))
            eos = int(rng.integers(0, 4)) if rng.random() < 0.5 else -1
            rc, rid = s.submit(n, mx, eos)
            if n + mx > SL:
                assert rc == TOO_LONG
            elif blocks(n + mx) > NB:
                assert rc == OOM
            else:
                assert rc == 0 and rid not in model
                model[rid] = dict(n=n, max_new=mx, eos=eos, status=0, slot=None, out=0, blocks=0)
                order.append(rid)
        elif op == 'admit':
            budget = int(rng.integers(1, 400))
            got = s.admit(budget)
            want, tokens = [], 0
            free = NB - sum(m['blocks'] for m in model.values())
            while order:
                rid = order[0]
                if rid not in model or model[rid]['status'] != 0:
                    order.pop(0)
                    continue
                m = model[rid]
                need = blocks(m['n'] + m['max_new'])
                slot = next((i for i, v in enumerate(slots) if v is None), None)
                if slot is None or need > free or (want and tokens + m['n'] > budget):
                    break
                slots[slot], m['slot'], m['blocks'] = rid, slot, need
                free -= need
                tokens += m['n']
                want.append((rid, slot))
                order.pop(0)
            assert got == want, (step, got, want)
        elif op == 'token':
            slot = int(rng.integers(0, B))
            tok = int(rng.integers(0, 4))
            fin = s.on_token(slot, tok)
            rid = slots[slot]
            if rid is None:
                assert not fin
            else:
                m = model[rid]
                m['out'] += 1
                done = (m['eos'] >= 0 and tok == m['eos']) or m['out'] >= m['max_new']
                assert fin == done
                if done:
                    m.update(status=FINISH, slot=None, blocks=0)
                    slots[slot] = None
        elif op == 'cancel' and model:
            rid = int(rng.choice(list(model)))
            rc, rel = s.cancel(rid)
            m = model[rid]
            assert rc == 0
            if m['status'] == 0:
                assert rel == (m['slot'] if m['slot'] is not None else -1)
                if m['slot'] is not None:
                    slots[m['slot']] = None
                m.update(status=CANCEL, slot=None, blocks=0)
            else:
                assert rel == -1
        elif op == 'forget' and model:
            rid = int(rng.choice(list(model)))
            rc = s.forget(rid)
            if model[rid]['status'] == 0:
                assert rc == INVALID
            else:
                assert rc == 0
                del model[rid]
        # the counters and every record agree with the model
        na, nw, nf = s.counts()
        assert na == sum(v is not None for v in slots) <= B
        assert nw == sum(1 for r in order if r in model and model[r]['status'] == 0)
        assert nf == NB - sum(m['blocks'] for m in model.values())
        for rid, m in model.items():
            rc, st, slot, ng, nb = s.query(rid)
            assert (rc, st, slot, ng, nb) == (0, m['status'], -1 if m['slot'] is None else m['slot'], m['out'], m['blocks']), (step, rid)


def test_admit_ready_is_admit_without_side_effects():
    s = Sched(max_batch=2, num_blocks=4, session_len=4096)
    assert s.admit_ready() is False                           # nothing queued
    a = s.submit(100, 28)[1]                                  # 2 blocks
    b = s.submit(130, 62)[1]                                  # 3 blocks
    c = s.submit(10, 10)[1]
    assert s.admit_ready() is True and s.counts() == (0, 3, 4)
    assert s.admit() == [(a, 0)]
    assert s.admit_ready() is False                           # b needs 3 of the 2 free blocks: the head of the line decides
    s.cancel(b)                                               # cancelled while waiting: admit() drops it, c is the new head
    assert s.admit_ready() is True
    assert s.admit() == [(c, 1)]
    d = s.submit(10, 10)[1]
    assert s.admit_ready() is False                           # no free slot
    for _ in range(28):
        s.on_token(0, 1)
    assert s.admit_ready() is True and s.admit() == [(d, 0)]


class _DeviceModel:
    """What the decode step does to the per-slot device state of the engine (active flag, position in the sequence): a token
    is the pair (request the slot was prefilled for, index of the token in that request's output)."""
    def __init__(self, slots):
        self.active, self.req, self.k = [0] * slots, [-1] * slots, [0] * slots

    def prefill(self, slot, rid):          # first token = (rid, 0); the slot joins the decode table
        self.active[slot], self.req[slot], self.k[slot] = 1, rid, 1
        return (rid, 0)

    def step(self):                        # every active slot produces its next token (parked slots: garbage)
        out = []
        for b in range(len(self.active)):
            out.append((self.req[b], self.k[b]) if self.active[b] else None)
            self.k[b] += self.active[b]
        return out

    def park(self, slot):
        self.active[slot] = 0


@pytest.mark.parametrize('async_on', [True, False])
@pytest.mark.parametrize('slots,seed', [(1, 0), (3, 1), (3, 2), (8, 3)])
def test_two_phase_step_protocol_model(slots, seed, async_on):
    """The issue / retire protocol of the engine's scheduler step (engine_serve.hip: step_locked, cb_issue, cb_retire; reference: the
    two alternating phases of turbomind.cc:171) replayed on the CPU against the REAL scheduler and a model of the device state:
    a pure decode step N+1 is issued before step N is retired; a slot's token counts only if the slot still runs the request it
    ran at issue time.  Invariants: every request receives exactly its tokens 0 .. n-1, once, in order (EOS / length / cancel
    included); tokens of dead rows (a sequence that ended one step earlier, a cancelled one) are dropped; no step stays in flight
    across an admission; with the overlap off the same token streams come out."""
    rng = np.random.default_rng(seed)
    n_req = 24
    s = Sched(max_batch=slots, num_blocks=64, session_len=1024)
    spec = {}
    for _ in range(n_req):
        n, m = int(rng.integers(1, 200)), int(rng.integers(1, 20))
        rid = s.submit(n, m)[1]
        spec[rid] = dict(max_new=m, eos_at=int(rng.integers(0, m)) if rng.random() < 0.3 else None,
                         cancel_at=int(rng.integers(1, m)) if m > 1 and rng.random() < 0.2 else None)
    dev = _DeviceModel(slots)
    slot_req, h_active = [-1] * slots, [0] * slots
    out = {rid: [] for rid in spec}
    status = {rid: 0 for rid in spec}
    pending, calls, overlapped, issued = None, 0, 0, 0

    # (the scheduler's own EOS test needs an eos id: model EOS by cancelling nothing -- use on_token's length rule and an
    #  explicit finish through a stop token: submit() above passed eos = -1, so EOS is modelled on this side)
    def on_token(slot, tok):
        rid = slot_req[slot]
        assert tok == (rid, len(out[rid])), f'slot {slot}: token {tok} handed to request {rid} with {len(out[rid])} tokens'
        out[rid].append(tok)
        fin = s.on_token(slot, 1)
        if not fin and spec[rid]['eos_at'] == tok[1]:
            assert s.cancel(rid) == (0, slot)                # stands in for an EOS hit: the request ends here
            status[rid] = FINISH
            fin = True
        elif fin:
            status[rid] = FINISH
        if fin:
            slot_req[slot], h_active[slot] = -1, 0
            dev.park(slot)
        return fin

    def issue(skip=()):
        nonlocal issued
        issued += 1
        toks = dev.step()
        return dict(ids=[slot_req[b] if h_active[b] and b not in skip else -1 for b in range(slots)], toks=toks)

    def retire(p):
        if p is None:
            return
        for b in range(slots):
            rid = p['ids'][b]
            if rid < 0 or not h_active[b] or slot_req[b] != rid:
                continue
            on_token(b, p['toks'][b])

    def more_needed():
        for b in range(slots):
            rid = slot_req[b]
            if rid >= 0 and h_active[b]:
                underway = 1 if pending is not None and pending['ids'][b] == rid else 0
                if len(out[rid]) + underway < spec[rid]['max_new']:
                    return True
        return False

    while any(st == 0 for st in status.values()):
        calls += 1
        assert calls < 2000, 'no progress'
        # API calls between two steps: cancels (as the pipeline / a client would issue them after polling)
        for rid, sp in spec.items():
            if status[rid] == 0 and sp['cancel_at'] is not None and len(out[rid]) >= sp['cancel_at']:
                rc, slot = s.cancel(rid)
                assert rc == 0
                status[rid] = CANCEL
                if slot >= 0:
                    slot_req[slot], h_active[slot] = -1, 0
                    dev.park(slot)
        if pending is not None and s.admit_ready():
            retire(pending)
            pending = None
        n_act_before = sum(r >= 0 for r in slot_req)
        admits = s.admit(96)
        if admits:
            assert pending is None, 'a step is in flight across an admission'
            fresh = []
            for rid, slot in admits:
                slot_req[slot] = rid
            for rid, slot in admits:                          # prefill: first tokens
                tok = dev.prefill(slot, rid)
                h_active[slot] = 1
                fresh.append(slot)
                on_token(slot, tok)
            merged = n_act_before > 0
            if any(r >= 0 for r in slot_req):
                if merged:
                    # the decode rows rode on the prefill forward: the fresh slots did not decode in it
                    for b in fresh:
                        dev.k[b] -= dev.active[b]
                now = issue(skip=fresh if merged else ())
                retire(now)
        elif any(r >= 0 for r in slot_req) and more_needed():
            nxt = issue()
            overlapped += pending is not None
            retire(pending)
            pending = nxt if async_on else None
            if not async_on:
                retire(nxt)
        else:
            retire(pending)
            pending = None
    for rid, sp in spec.items():
        n = len(out[rid])
        if status[rid] == CANCEL:
            assert sp['cancel_at'] <= n <= sp['max_new']
        elif sp['eos_at'] is not None:
            assert n == sp['eos_at'] + 1
        else:
            assert n == sp['max_new']
        assert out[rid] == [(rid, k) for k in range(n)]
    assert s.counts()[0] == 0 and s.counts()[2] == 64
    assert (overlapped > 0) == (async_on and True)


def test_prefill_microbatch_split():
    """Tensor-parallel prefill: the forward splits at the sequence boundary nearest to half its rows into two micro-batches
    whose all-reduces run under the other one's kernels; no usable boundary -> 0 (row halves inside every layer).  Pure host logic of
    the engine (scheduler.h: prefill_microbatch_split), driven through the C-ABI without a GPU."""
    lib = _ffi.load()

    def split(lens, min_rows):
        cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        sa, ra = C.c_int(-1), C.c_int(-1)
        _ffi.check(lib.tm_prefill_split(cu.ctypes.data, len(lens), min_rows, C.byref(sa), C.byref(ra)))
        assert ra.value == (int(cu[sa.value]) if sa.value else 0)
        return sa.value, ra.value

    assert split([1024] * 8, 1024) == (4, 4096)                 # the bench's prefill forward: 8 x 1024 tokens
    assert split([300, 77, 190, 33], 256) == (1, 300)           # 300 | 300
    assert split([300, 77, 71], 64) == (1, 300)                 # 300 | 148: nearer to half than 377 | 71
    assert split([119, 33], 64) == (1, 119)                     # smaller side 33 rows: 66 >= 64
    assert split([119, 31], 64) == (0, 0)                       # 62 < 64: lopsided -> row halves
    assert split([560, 20], 64) == (0, 0)
    assert split([600], 64) == (0, 0)                           # one sequence
    assert split([10, 10, 10, 10, 2000], 64) == (4, 40)         # the only boundaries are far from the middle: the nearest one, 80 >= 64
    assert split([10, 10, 10, 10, 2000], 128) == (0, 0)
    assert split([500, 500, 500, 500], 64) == (2, 1000)         # exact middle
    assert split([400, 200, 200, 400], 64) == (2, 600)          # ties (400 | 800 vs 800 | 400 do not occur here): the middle boundary
    assert split([100, 400, 100], 64) == (1, 100)               # |200 - 600| = |1000 - 600|: the earlier boundary
    # malformed offsets are refused
    bad = np.array([0, 5, 5], np.int32)
    sa, ra = C.c_int(), C.c_int()
    assert lib.tm_prefill_split(bad.ctypes.data, 2, 64, C.byref(sa), C.byref(ra)) != 0


def test_scheduler_under_sanitizers(tmp_path):
    """scheduler.h is pure host C++: a seeded random stream of the engine's calls (submit / admit / logprob record + token / cancel / forget, abort)
    with the block, slot and logprob-record invariants checked after every call, built with AddressSanitizer + UndefinedBehaviorSanitizer
    (tests/cpp/scheduler_sanitizer.cpp; the GPU side cannot be sanitised on this pool)."""
    import shutil
    import subprocess
    if shutil.which('g++') is None:
        pytest.skip('g++ not available')
    src = os.path.join(os.path.dirname(__file__), 'cpp', 'scheduler_sanitizer.cpp')
    exe = str(tmp_path / 'sched_san')
    build = subprocess.run(['g++', '-std=c++17', '-O1', '-g', '-fsanitize=address,undefined', '-fno-sanitize-recover=all', '-o', exe, src],
                           capture_output=True, text=True)
    assert build.returncode == 0, build.stderr
    for args in (['1'], ['2', 'abort'], ['77']):
        run = subprocess.run([exe] + args, capture_output=True, text=True, timeout=120)
        assert run.returncode == 0 and run.stdout.startswith('ok seed'), run.stdout + run.stderr
