// Continuous batching and the engine thread (Engine::Impl::Schedule / InternalThreadEntry, src/turbomind/engine/engine.cc:434-470,770-870;
// ModelRequest::Forward / Cancel, model_request.cc:36-135): submit / step / poll / cancel, mixed steps, the two-phase issue / retire schedule.
#include "engine_internal.h"

namespace tmk {

__global__ void park_slot_kernel(int* active, int* k_len, uint64_t* block_row, int slot, uint64_t dummy_block_ptr)
{
    active[slot] = 0;
    k_len[slot]  = 1;
    block_row[0] = dummy_block_ptr;
}

}  // namespace tmk

extern "C" {

// ------------------------------------------------------------------------------------------------------------------
// Continuous batching: request queue + slot scheduler (scheduler.h) on top of the same forward().
// Every decode step runs all max_batch_size slots (one graph); a free slot is parked on a scratch block with
// k_len = 1 and its token is ignored.  A scheduler step = admit waiting requests (prefill, chunked) + one decode step
// for everything that is running (prefill-priority, like the reference's default when new requests arrive).
// ------------------------------------------------------------------------------------------------------------------
static int cb_enter(tm_engine* e)
{
    if (e->sched) {
        return 0;
    }
    TM_REQUIRE(e->started, "engine not started");
    TM_REQUIRE(e->batch == 0, "a static batch is admitted (release it first)");
    TM_REQUIRE(e->num_blocks >= 2, "continuous batching needs at least two KV blocks");
    TM_HIP_CHECK(hipSetDevice(e->cfg.device));
    const int B = e->cfg.max_batch_size;
    if (!e->d_active) {
        TM_TRY(dmalloc(&e->d_active, (size_t)B));
        TM_TRY(dmalloc(&e->d_pf_k_len, (size_t)B));
        TM_TRY(dmalloc(&e->d_pf_cu_q, (size_t)B + 1));
        TM_TRY(dmalloc(&e->d_pf_block_ptrs, (size_t)B * e->max_blocks_per_seq));
        TM_TRY(dmalloc(&e->d_first_ids, (size_t)B));
    }
    if (!e->aux_stream) {
        const char* ts       = getenv("TM_MIXED_2STREAM");
        e->mixed_two_streams = !(ts && !atoi(ts));
        TM_HIP_CHECK(hipStreamCreateWithFlags(&e->aux_stream, hipStreamNonBlocking));
        TM_HIP_CHECK(hipEventCreateWithFlags(&e->ev_aux_fork, hipEventDisableTiming));
        TM_HIP_CHECK(hipEventCreateWithFlags(&e->ev_aux_join, hipEventDisableTiming));
    }
    e->dummy_block = (int)e->num_blocks - 1;  // parking block of the free slots; the scheduler owns the others
    e->sched.reset(new BatchScheduler(B, (int)e->num_blocks - 1, e->cfg.session_len, e->cfg.cache_block_seq_len));
    e->free_blocks.clear();
    e->h_active.assign(B, 0);
    // read when a continuous-batching session starts.  Default OFF -- measured (profiles/r04_request_stream_device_busy.txt): the device
    // is 97.9 % busy over the request-stream benchmark with synchronous steps, so the overlap has no idle time to hide, while a
    // sequence that ends rides one dead row and every admission waits one more step: 7 899 vs 7 932 output tok/s (A/B on one engine)
    const char* as   = getenv("TM_ASYNC_STEP");
    e->async_step_on = as && atoi(as);
    if (!e->h_step_pin[0]) {
        for (int i = 0; i < 2; ++i) {
            TM_HIP_CHECK(hipHostMalloc((void**)&e->h_step_pin[i], ((size_t)B + 1) * 4, hipHostMallocDefault));
            TM_HIP_CHECK(hipEventCreateWithFlags(&e->ev_step[i], hipEventDisableTiming));
        }
    }
    e->pending.valid = false;
    std::vector<int>      ones(B, 1), zeros(B, 0), cu_q(B + 1);
    std::vector<uint64_t> ptrs((size_t)B * e->max_blocks_per_seq,
                               (uint64_t)(e->pool + (int64_t)e->dummy_block * e->block_bytes));
    for (int b = 0; b <= B; ++b) {
        cu_q[b] = b;
    }
    TM_HIP_CHECK(hipMemcpyAsync(e->d_block_ptrs, ptrs.data(), ptrs.size() * 8, hipMemcpyHostToDevice, e->stream));
    TM_HIP_CHECK(hipMemcpyAsync(e->d_pf_block_ptrs, ptrs.data(), ptrs.size() * 8, hipMemcpyHostToDevice, e->stream));
    TM_HIP_CHECK(hipMemcpyAsync(e->d_k_len, ones.data(), B * 4, hipMemcpyHostToDevice, e->stream));
    TM_HIP_CHECK(hipMemcpyAsync(e->d_active, zeros.data(), B * 4, hipMemcpyHostToDevice, e->stream));
    TM_HIP_CHECK(hipMemcpyAsync(e->d_ids, zeros.data(), B * 4, hipMemcpyHostToDevice, e->stream));
    TM_HIP_CHECK(hipMemcpyAsync(e->d_cu_q, cu_q.data(), (B + 1) * 4, hipMemcpyHostToDevice, e->stream));
    TM_HIP_CHECK(hipStreamSynchronize(e->stream));
    e->batch = B;  // the decode step always covers every slot
    setup_decode(e, B);
    return 0;
}

static int decode_step_cb(tm_engine* e)
{
    const int B = e->cfg.max_batch_size;
    TM_TRY(launch_advance_active(e->d_k_len, e->d_active, B, e->stream));
    TM_TRY(forward(e, e->d_ids, B, B, true, 1, 0, 0, 0));
    TM_HIP_CHECK(hipMemcpyAsync(e->d_ids, e->d_next_ids, (size_t)B * 4, hipMemcpyDeviceToDevice, e->stream));
    return 0;
}

// park a slot again after its sequence finished / was cancelled
static int cb_park_slot(tm_engine* e, int slot)
{
    const uint64_t dp = (uint64_t)(e->pool + (int64_t)e->dummy_block * e->block_bytes);
    e->h_active[slot] = 0;
    park_slot_kernel<<<1, 1, 0, e->stream>>>(e->d_active, e->d_k_len, e->d_block_ptrs + (size_t)slot * e->max_blocks_per_seq, slot, dp);
    TM_HIP_CHECK(hipGetLastError());
    return 0;
}

// prefill the newly admitted requests (contiguous slot runs share one chunked prefill), hand over their first tokens
// `merged` != nullptr: the decode step of this scheduler step may ride on the last prefill iteration (mixed forward);
// *merged says whether it did, `fresh` receives the slots that were prefilled by that forward (they did not decode in it)
static int cb_prefill_admitted(tm_engine* e, const std::vector<SchedAdmit>& admits, std::vector<StepUpdate>* updates,
                               bool* merged = nullptr, std::vector<int>* fresh = nullptr)
{
    std::vector<SchedAdmit> sorted = admits;
    std::sort(sorted.begin(), sorted.end(), [](const SchedAdmit& a, const SchedAdmit& b) { return a.slot < b.slot; });
    size_t i = 0;
    while (i < sorted.size()) {
        size_t j = i + 1;
        while (j < sorted.size() && sorted[j].slot == sorted[j - 1].slot + 1) {
            ++j;
        }
        const int               n     = (int)(j - i);
        const int               slot0 = sorted[i].slot;
        std::vector<const int*> ids(n);
        std::vector<int>        lens(n);
        std::vector<uint64_t>   ptrs((size_t)n * e->max_blocks_per_seq,
                                     (uint64_t)(e->pool + (int64_t)e->dummy_block * e->block_bytes));
        for (int k = 0; k < n; ++k) {
            const SchedRequest* r = e->sched->find(sorted[i + k].id);
            TM_REQUIRE(r && r->running, "internal: admitted request vanished");
            ids[k]  = r->prompt.data();
            lens[k] = (int)r->prompt.size();
            if (e->sampling_on) {  // greedy rows are top_k = 1 rows of the sampling kernels
                auto              it = e->cb_sampling.find(r->id);
                const tm_sampling sp = it == e->cb_sampling.end() ? tm_sampling{1.f, 1, 1.f, 0.f, 0} : it->second;
                TM_TRY(sampling_upload(e, &sp, slot0 + k, 1));
            }
            if (e->logits_on) {  // slots without parameters run the processors as no-ops
                auto                  it = e->cb_logits.find(r->id);
                const tm_logits_param lp = it == e->cb_logits.end() ? kNoLogitsParam : it->second;
                TM_TRY(logits_upload(e, &lp, &lens[k], &r->eos, slot0 + k, 1));
            }
            TM_REQUIRE((int)r->blocks.size() <= e->max_blocks_per_seq, "internal: block table row too short");
            for (size_t q = 0; q < r->blocks.size(); ++q) {
                ptrs[(size_t)k * e->max_blocks_per_seq + q] = (uint64_t)(e->pool + (int64_t)r->blocks[q] * e->block_bytes);
            }
        }
        TM_HIP_CHECK(hipMemcpyAsync(e->d_pf_block_ptrs + (size_t)slot0 * e->max_blocks_per_seq, ptrs.data(), ptrs.size() * 8,
                                    hipMemcpyHostToDevice, e->stream));
        // prefill uses iteration-local k_len / cu_q arrays and its own block table: the decode state of the running slots
        // stays untouched, the new slots' decode rows stay parked until the prefill is done
        const bool      last_run = merged && j == sorted.size();
        bool            did      = false;
        const MixedStep mix{e->cfg.max_batch_size, e->d_k_len, e->d_active, e->d_ids, e->d_block_ptrs, e->d_cu_q, &did};
        std::swap(e->d_k_len, e->d_pf_k_len);
        std::swap(e->d_cu_q, e->d_pf_cu_q);
        std::swap(e->d_block_ptrs, e->d_pf_block_ptrs);
        const int rc = prefill_slots(e, ids.data(), lens.data(), n, slot0, nullptr, last_run ? &mix : nullptr);
        std::swap(e->d_k_len, e->d_pf_k_len);
        std::swap(e->d_cu_q, e->d_pf_cu_q);
        std::swap(e->d_block_ptrs, e->d_pf_block_ptrs);
        if (rc) {
            return rc;
        }
        // prefilled: the rows join the decode table
        TM_HIP_CHECK(hipMemcpyAsync(e->d_block_ptrs + (size_t)slot0 * e->max_blocks_per_seq, ptrs.data(), ptrs.size() * 8,
                                    hipMemcpyHostToDevice, e->stream));
        if (did) {
            *merged = true;
            for (int k = 0; k < n; ++k) {
                fresh->push_back(slot0 + k);
            }
        }
        // decode state of the new slots: context length, current token; first tokens go to the host
        std::vector<int> first(n), ones(n, 1);
        std::vector<float> lp_vals, lp_sel;
        std::vector<int>   lp_idx, lp_num;
        const int          lpw = e->cb_logprobs_on ? e->cb_lp_used : 0;  // the first tokens' logprob records (written by the prefill's head)
        if (lpw > 0) {
            lp_vals.resize((size_t)n * lpw), lp_idx.resize((size_t)n * lpw), lp_num.resize(n), lp_sel.resize(n);
            TM_HIP_CHECK(hipMemcpy2DAsync(lp_vals.data(), (size_t)lpw * 4, e->d_cb_lp_vals + (size_t)slot0 * kMaxLogProb, (size_t)kMaxLogProb * 4,
                                          (size_t)lpw * 4, n, hipMemcpyDeviceToHost, e->stream));
            TM_HIP_CHECK(hipMemcpy2DAsync(lp_idx.data(), (size_t)lpw * 4, e->d_cb_lp_idx + (size_t)slot0 * kMaxLogProb, (size_t)kMaxLogProb * 4,
                                          (size_t)lpw * 4, n, hipMemcpyDeviceToHost, e->stream));
            TM_HIP_CHECK(hipMemcpyAsync(lp_num.data(), e->d_cb_lp_num + slot0, n * 4, hipMemcpyDeviceToHost, e->stream));
            TM_HIP_CHECK(hipMemcpyAsync(lp_sel.data(), e->d_cb_lp_sel + slot0, n * 4, hipMemcpyDeviceToHost, e->stream));
        }
        TM_HIP_CHECK(hipMemcpyAsync(e->d_k_len + slot0, lens.data(), n * 4, hipMemcpyHostToDevice, e->stream));
        TM_HIP_CHECK(hipMemcpyAsync(e->d_ids + slot0, e->d_next_ids + slot0, n * 4, hipMemcpyDeviceToDevice, e->stream));
        TM_HIP_CHECK(hipMemcpyAsync(first.data(), e->d_next_ids + slot0, n * 4, hipMemcpyDeviceToHost, e->stream));
        TM_HIP_CHECK(hipMemcpyAsync(e->d_active + slot0, ones.data(), n * 4, hipMemcpyHostToDevice, e->stream));
        TM_HIP_CHECK(hipStreamSynchronize(e->stream));
        for (int k = 0; k < n; ++k) {
            e->h_active[slot0 + k] = 1;
            const int64_t rid      = e->sched->slot_request(slot0 + k);
            if (lpw > 0) {
                e->sched->on_logprobs(slot0 + k, lp_vals.data() + (size_t)k * lpw, lp_idx.data() + (size_t)k * lpw, lp_num[k], lp_sel[k]);
            }
            const bool    finished = e->sched->on_token(slot0 + k, first[k]);
            if (updates) {
                const SchedRequest* r = e->sched->find(rid);
                updates->push_back({rid, r->status, (int)r->out.size()});
            }
            if (finished) {  // finished on its first token
                TM_TRY(cb_park_slot(e, slot0 + k));
            }
        }
        i = j;
    }
    return 0;
}

static int submit_locked(tm_engine* e, const int* host_ids, int n, int max_new_tokens, int eos_id, int64_t* req_id)
{
    if (e->comm_failed) {  // terminal for the communicator (device_marks_check): no new work on this engine
        return device_marks_check(e);
    }
    TM_TRY(cb_enter(e));
    const int rc = e->sched->submit(host_ids, n, max_new_tokens, eos_id, req_id);
    if (rc == TM_TOO_LONG) {
        set_last_error("prompt + max_new_tokens exceeds session_len");
    }
    else if (rc == TM_OOM) {
        set_last_error("request can never fit the KV block pool");
    }
    else if (rc) {
        set_last_error("invalid request (empty prompt or max_new_tokens < 1)");
    }
    return rc;
}

int tm_engine_submit_ex(tm_engine* e, const int* host_ids, int n, int max_new_tokens, int eos_id, const tm_sampling* sampling,
                        int64_t* req_id)
{
    return tm_engine_submit_gen(e, host_ids, n, max_new_tokens, eos_id, sampling, nullptr, req_id);
}

int tm_engine_submit_gen(tm_engine* e, const int* host_ids, int n, int max_new_tokens, int eos_id, const tm_sampling* sampling,
                         const tm_logits_param* logits_param, int64_t* req_id)
{
    TM_REQUIRE(e && host_ids && req_id, "null pointer");
    if (logits_param) {
        TM_TRY(logits_param_check(*logits_param));
    }
    if (sampling) {
        TM_REQUIRE(!e->use_comm || e->comm || e->p2p_ready,
               "stochastic sampling with tp > 1 gathers the logits: tm_engine_comm_init or the native communicator first");
        TM_REQUIRE(sampling->temperature > 0.f, "sampling: temperature must be > 0");
    }
    {
        ApiLock lock(e);
        TM_HIP_CHECK(hipSetDevice(e->cfg.device));
        TM_TRY(submit_locked(e, host_ids, n, max_new_tokens, eos_id, req_id));
        if (sampling) {
            e->cb_sampling[*req_id] = *sampling;
            if (!e->sampling_on) {  // the first stochastic request switches the decode step to the sampling kernels
                const int                B = e->cfg.max_batch_size;
                std::vector<tm_sampling> greedy(B, tm_sampling{1.f, 1, 1.f, 0.f, 0});
                TM_TRY(sampling_upload(e, greedy.data(), 0, B));
                e->sampling_on = true;
            }
        }
        if (logits_param) {
            e->cb_logits[*req_id] = *logits_param;
            e->sched->set_stop_ids(*req_id, logits_param->stop_ids, logits_param->n_stop_ids);
            if (!e->logits_on) {  // the first such request switches the decode step to the processor kernels
                const int                    B = e->cfg.max_batch_size;
                std::vector<tm_logits_param> none(B, kNoLogitsParam);
                std::vector<int>             zeros(B, 0);
                TM_TRY(logits_upload(e, none.data(), zeros.data(), nullptr, 0, B));
                e->logits_on = true;
            }
        }
    }
    e->cv_work.notify_one();
    return 0;
}

int tm_engine_request_logprobs(tm_engine* e, int64_t req_id, int n)
{
    TM_REQUIRE(e, "null pointer");
    TM_REQUIRE(n >= 1 && n <= kMaxLogProb, "1 <= logprobs <= TM_MAX_LOGPROBS");
    TM_REQUIRE(!e->use_comm || e->comm || e->p2p_ready,
               "logprobs with tp > 1 gather the logits like stochastic sampling: tm_engine_comm_init or the native communicator first");
    ApiLock lock(e);
    TM_REQUIRE(e->sched, "no continuous-batching session (submit first)");
    TM_HIP_CHECK(hipSetDevice(e->cfg.device));
    if (e->sched->set_logprobs(req_id, n)) {
        set_last_error("unknown request id, or the request is already running (ask for logprobs right after the submit)");
        return TM_INVALID;
    }
    const int B = e->cfg.max_batch_size;
    if (!e->d_cb_lp_vals) {
        TM_TRY(dmalloc(&e->d_cb_lp_vals, (size_t)B * kMaxLogProb));
        TM_TRY(dmalloc(&e->d_cb_lp_idx, (size_t)B * kMaxLogProb));
        TM_TRY(dmalloc(&e->d_cb_lp_num, (size_t)B));
        TM_TRY(dmalloc(&e->d_cb_lp_sel, (size_t)B));
        for (int i = 0; i < 2; ++i) {
            TM_HIP_CHECK(hipHostMalloc((void**)&e->h_cb_lp_vals[i], (size_t)B * kMaxLogProb * 4));
            TM_HIP_CHECK(hipHostMalloc((void**)&e->h_cb_lp_idx[i], (size_t)B * kMaxLogProb * 4));
            TM_HIP_CHECK(hipHostMalloc((void**)&e->h_cb_lp_num[i], (size_t)B * 4));
            TM_HIP_CHECK(hipHostMalloc((void**)&e->h_cb_lp_sel[i], (size_t)B * 4));
        }
    }
    if (!e->d_kept) {
        TM_TRY(dmalloc(&e->d_kept, (size_t)B));
    }
    if (!e->sampling_on) {  // the records come out of the sampling kernels: greedy rows are their top_k = 1 rows
        std::vector<tm_sampling> greedy(B, tm_sampling{1.f, 1, 1.f, 0.f, 0});
        TM_TRY(sampling_upload(e, greedy.data(), 0, B));
        e->sampling_on = true;
    }
    e->cb_lp_used     = std::max(e->cb_lp_used, n);  // columns copied to the host behind every step from now on
    e->cb_logprobs_on = true;                        // (a captured step without the records is re-captured by cb_launch_decode)
    return 0;
}

int tm_engine_poll_logprobs(tm_engine* e, int64_t req_id, float* host_vals, int* host_idx, int* host_num, float* host_sel, int max_tokens,
                            int* n_tokens, int* n_per_token)
{
    TM_REQUIRE(e && n_tokens && n_per_token, "null pointer");
    ApiLock lock(e);
    TM_REQUIRE(e->sched, "no continuous-batching session (submit first)");
    const SchedRequest* r = e->sched->find(req_id);
    if (!r) {
        set_last_error("unknown request id");
        return TM_INVALID;
    }
    *n_per_token = r->lp_n;
    *n_tokens    = (int)r->lp_num.size();
    const size_t t = (size_t)std::max(0, std::min(max_tokens, *n_tokens));
    if (host_vals && host_idx && host_num && host_sel && t > 0) {
        memcpy(host_vals, r->lp_vals.data(), t * r->lp_n * 4);
        memcpy(host_idx, r->lp_idx.data(), t * r->lp_n * 4);
        memcpy(host_num, r->lp_num.data(), t * 4);
        memcpy(host_sel, r->lp_sel.data(), t * 4);
    }
    return 0;
}

int tm_engine_submit(tm_engine* e, const int* host_ids, int n, int max_new_tokens, int eos_id, int64_t* req_id)
{
    return tm_engine_submit_ex(e, host_ids, n, max_new_tokens, eos_id, nullptr, req_id);
}

// ---- issue / retire of a decode step (two-phase overlap, see tm_engine::PendingStep) ----
// the decode step of every slot, as a graph replay when graphs are on (captured on first use)
static int cb_launch_decode(tm_engine* e)
{
    if (e->graph_cb && (e->graph_cb_sampling != e->sampling_on || e->graph_cb_logits != e->logits_on || e->graph_cb_logprobs != e->cb_logprobs_on)) {
        TM_HIP_CHECK(hipStreamSynchronize(e->stream));  // (a replay of the old graph may still be running)
        (void)hipGraphExecDestroy(e->graph_cb);
        e->graph_cb = nullptr;
    }
    if (graph_enabled(e) && !e->graph_cb) {
        TM_TRY(decode_step_cb(e));  // one eager step first (lazy module loading must not happen inside a capture)
        TM_HIP_CHECK(hipStreamSynchronize(e->stream));
        TM_TRY(capture_step(e, decode_step_cb, &e->graph_cb));
        e->graph_cb_sampling = e->sampling_on;
        e->graph_cb_logits   = e->logits_on;
        e->graph_cb_logprobs = e->cb_logprobs_on;
        return 0;
    }
    if (graph_enabled(e) && e->graph_cb) {
        TM_HIP_CHECK(hipGraphLaunch(e->graph_cb, e->stream));
        return 0;
    }
    return decode_step_cb(e);
}

// behind a launched step: its tokens (d_ids after the step) and the communicator's give-up mark go to a pinned buffer, an event
// marks the hand-over.  `skip`: slots that were prefilled by this very forward (their first token was handed over already)
static int cb_issue(tm_engine* e, tm_engine::PendingStep* p, const std::vector<int>& skip)
{
    const int B = e->cfg.max_batch_size;
    p->buf      = e->issue_count++ & 1;
    p->ids.assign(B, -1);
    for (int b = 0; b < B; ++b) {
        if (e->h_active[b] && std::find(skip.begin(), skip.end(), b) == skip.end()) {
            p->ids[b] = e->sched->slot_request(b);
        }
    }
    int* const h = e->h_step_pin[p->buf];
    h[B]         = 0;
    TM_HIP_CHECK(hipMemcpyAsync(h, e->d_ids, (size_t)B * 4, hipMemcpyDeviceToHost, e->stream));
    if (e->p2p_state) {
        TM_HIP_CHECK(hipMemcpyAsync(h + B, e->p2p_state + 3, 4, hipMemcpyDeviceToHost, e->stream));
    }
    p->lp = e->cb_logprobs_on && e->cb_lp_used > 0;
    if (p->lp) {  // this step's logprob records: the first cb_lp_used candidates of every slot
        const size_t w = (size_t)e->cb_lp_used * 4, pitch = (size_t)kMaxLogProb * 4;
        TM_HIP_CHECK(hipMemcpy2DAsync(e->h_cb_lp_vals[p->buf], pitch, e->d_cb_lp_vals, pitch, w, B, hipMemcpyDeviceToHost, e->stream));
        TM_HIP_CHECK(hipMemcpy2DAsync(e->h_cb_lp_idx[p->buf], pitch, e->d_cb_lp_idx, pitch, w, B, hipMemcpyDeviceToHost, e->stream));
        TM_HIP_CHECK(hipMemcpyAsync(e->h_cb_lp_num[p->buf], e->d_cb_lp_num, (size_t)B * 4, hipMemcpyDeviceToHost, e->stream));
        TM_HIP_CHECK(hipMemcpyAsync(e->h_cb_lp_sel[p->buf], e->d_cb_lp_sel, (size_t)B * 4, hipMemcpyDeviceToHost, e->stream));
    }
    TM_HIP_CHECK(hipEventRecord(e->ev_step[p->buf], e->stream));
    p->valid = true;
    return 0;
}

// wait for an issued step, hand its tokens to the scheduler, park the slots whose sequence ended.  A slot counts only if it still
// runs the request it ran when the step was issued (finished one step earlier / cancelled / re-admitted since: token dropped)
static int cb_retire(tm_engine* e, tm_engine::PendingStep* p, std::vector<StepUpdate>* updates)
{
    if (!p->valid) {
        return 0;
    }
    p->valid    = false;
    const int B = e->cfg.max_batch_size;
    TM_HIP_CHECK(hipEventSynchronize(e->ev_step[p->buf]));
    const int* const h = e->h_step_pin[p->buf];
    if (e->p2p_state && h[B]) {
        e->h_mark = (unsigned)h[B];
    }
    TM_TRY(device_marks_check(e));  // -> the serve loop ends every unfinished request with kFail
    for (int b = 0; b < B; ++b) {
        const int64_t id = p->ids[b];
        if (id < 0 || !e->h_active[b] || e->sched->slot_request(b) != id) {
            continue;
        }
        if (p->lp) {
            e->sched->on_logprobs(b, e->h_cb_lp_vals[p->buf] + (size_t)b * kMaxLogProb, e->h_cb_lp_idx[p->buf] + (size_t)b * kMaxLogProb,
                                  e->h_cb_lp_num[p->buf][b], e->h_cb_lp_sel[p->buf][b]);
        }
        const bool finished = e->sched->on_token(b, h[b]);
        if (updates) {
            const SchedRequest* r = e->sched->find(id);
            updates->push_back({id, r->status, (int)r->out.size()});
        }
        if (finished) {
            TM_TRY(cb_park_slot(e, b));
        }
    }
    return 0;
}

// does any running sequence need a token beyond the ones that are already on their way (the unretired step)?
static bool cb_more_tokens_needed(const tm_engine* e)
{
    const int B = e->cfg.max_batch_size;
    for (int b = 0; b < B; ++b) {
        const int64_t id = e->sched->slot_request(b);
        if (id < 0 || !e->h_active[b]) {
            continue;
        }
        const SchedRequest* r        = e->sched->find(id);
        const int           underway = e->pending.valid && e->pending.ids[b] == id ? 1 : 0;
        if (r && (int)r->out.size() + underway < r->max_new) {
            return true;
        }
    }
    return false;
}

// one scheduler step; the caller holds e->mu.  `updates` (optional): requests that produced a token / finished
static int step_locked(tm_engine* e, int* n_active, int* n_waiting, std::vector<StepUpdate>* updates)
{
    TM_TRY(cb_enter(e));
    TM_HIP_CHECK(hipSetDevice(e->cfg.device));
    const int B = e->cfg.max_batch_size;
    // 0. an admission is due: everything from here to the end of this call is synchronous (the admission's first tokens are read
    //    back, the block accounting of the scheduler must be current) -- retire the step that is still in flight first
    if (e->pending.valid && e->sched->admit_ready()) {
        TM_TRY(cb_retire(e, &e->pending, updates));
    }
    // 1. admission + prefill (budget = max_prefill_token_num tokens of prompts per step)
    // Mixed steps (TM_MIXED_STEP, default on): when something is already decoding, the decode step rides on the admission's
    // last prefill forward -- one weight stream for both (reference: the unified batch of unified_attention_layer.cc:310-311).
    const bool        mixed_on = e->mixed_steps_on;  // TM_MIXED_STEP, read when the engine was created
    // Every configuration mixes: tp > 1 (the row-parallel reductions of the merged forward take the large-message path),
    // logits processors (the seen-mask update skips decode rows whose slot holds no running sequence), fp16 KV (the decode
    // rows' K/V go through kv_rope_store instead of the fused prologue), admissions of any size (see prefill_slots).
    // (not with logprob records: the merged forward's decode head would overwrite the records of the admission's earlier iterations)
    const bool        can_mix  = mixed_on && e->sched->n_active() > 0 && e->max_tokens - B >= 16 && !e->cb_logprobs_on;
    bool             merged = false;
    std::vector<int> fresh;
    const std::vector<SchedAdmit> admits = e->sched->admit(e->max_tokens);
    if (!admits.empty()) {
        TM_TRY(cb_retire(e, &e->pending, updates));  // (admit_ready() said so above; kept for the invariant: no step in flight here)
        TM_TRY(cb_prefill_admitted(e, admits, updates, can_mix ? &merged : nullptr, &fresh));
        // 2a. the decode step of this call: rode on the admission's forward, or a launch of its own; retired at once
        if (e->sched->n_active() > 0) {
            if (merged) {
                ++e->mixed_steps;
            }
            else {
                TM_TRY(cb_launch_decode(e));
            }
            tm_engine::PendingStep now;
            TM_TRY(cb_issue(e, &now, fresh));
            TM_TRY(cb_retire(e, &now, updates));
        }
    }
    else if (e->sched->n_active() > 0 && cb_more_tokens_needed(e)) {
        // 2b. pure decode step: issue step N+1, THEN retire step N (the device runs N+1 under the host's bookkeeping)
        TM_TRY(cb_launch_decode(e));
        tm_engine::PendingStep next;
        TM_TRY(cb_issue(e, &next, fresh));
        if (e->pending.valid) {
            ++e->overlapped_steps;
        }
        TM_TRY(cb_retire(e, &e->pending, updates));
        if (e->async_step_on) {
            e->pending = std::move(next);
        }
        else {
            TM_TRY(cb_retire(e, &next, updates));
        }
    }
    else {
        TM_TRY(cb_retire(e, &e->pending, updates));  // nothing to issue: the tokens on their way end every running sequence
    }
    if (n_active) {
        *n_active = e->sched->n_active();
    }
    if (n_waiting) {
        *n_waiting = e->sched->n_waiting();
    }
    return 0;
}

int tm_engine_step(tm_engine* e, int* n_active, int* n_waiting)
{
    TM_REQUIRE(e, "null pointer");
    if (e->loop_on.load()) {
        set_last_error("the engine thread owns the scheduler loop (tm_engine_serve_stop first)");
        return TM_CONFLICT;
    }
    ApiLock lock(e);
    return step_locked(e, n_active, n_waiting, nullptr);
}

// Up to `max_steps` scheduler iterations in ONE call; stops early when nothing runs and nothing waits.  What a tensor-parallel rank group
// mirrors instead of single steps: the ranks' schedulers take the same decisions from the same call sequence, so a burst of steps needs one
// host round trip, not one per step (reference: every rank's engine thread loops on its own, src/turbomind/engine/engine.cc:770-870 --
// the host exchanges admissions, not steps).  Also the cheaper host loop at tp = 1: a session polls once per burst.
int tm_engine_step_many(tm_engine* e, int max_steps, int* steps_done, int* n_active, int* n_waiting)
{
    TM_REQUIRE(e && steps_done && max_steps >= 1, "arguments");
    if (e->loop_on.load()) {
        set_last_error("the engine thread owns the scheduler loop (tm_engine_serve_stop first)");
        return TM_CONFLICT;
    }
    ApiLock lock(e);
    int     na = 0, nw = 0;
    *steps_done = 0;
    for (int i = 0; i < max_steps; ++i) {
        const int rc = step_locked(e, &na, &nw, nullptr);
        if (rc) {
            return rc;
        }
        ++*steps_done;
        if (na == 0 && nw == 0) {
            break;
        }
    }
    if (n_active) {
        *n_active = na;
    }
    if (n_waiting) {
        *n_waiting = nw;
    }
    return 0;
}

static int poll_locked(tm_engine* e, int64_t req_id, int* status, int* host_tokens, int cap, int* n_tokens)
{
    TM_REQUIRE(e->sched, "no continuous-batching session (submit first)");
    const SchedRequest* r = e->sched->find(req_id);
    if (!r) {
        set_last_error("unknown request id");
        return TM_INVALID;
    }
    *status   = r->status;
    *n_tokens = (int)r->out.size();
    if (host_tokens) {
        memcpy(host_tokens, r->out.data(), (size_t)std::max(0, std::min(cap, *n_tokens)) * 4);
    }
    return 0;
}

int tm_engine_poll(tm_engine* e, int64_t req_id, int* status, int* host_tokens, int cap, int* n_tokens)
{
    TM_REQUIRE(e && status && n_tokens, "null pointer");
    ApiLock lock(e);
    return poll_locked(e, req_id, status, host_tokens, cap, n_tokens);
}

int tm_engine_cancel(tm_engine* e, int64_t req_id)
{
    TM_REQUIRE(e, "null pointer");
    {
        ApiLock lock(e);
        TM_REQUIRE(e->sched, "no continuous-batching session (submit first)");
        int       slot = -1;
        const int rc   = e->sched->cancel(req_id, &slot);
        if (rc) {
            set_last_error("unknown request id");
            return TM_INVALID;
        }
        if (slot >= 0) {
            TM_HIP_CHECK(hipSetDevice(e->cfg.device));
            TM_TRY(cb_park_slot(e, slot));
        }
    }
    e->cv_out.notify_all();
    return 0;
}

int tm_engine_forget(tm_engine* e, int64_t req_id)
{
    TM_REQUIRE(e, "null pointer");
    ApiLock lock(e);
    TM_REQUIRE(e->sched, "no continuous-batching session (submit first)");
    if (!e->sched->erase(req_id)) {
        set_last_error("unknown or unfinished request id");
        return TM_INVALID;
    }
    e->cb_sampling.erase(req_id);
    e->cb_logits.erase(req_id);
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// The engine thread: schedule -> forward -> update while requests exist, asleep otherwise.
// ------------------------------------------------------------------------------------------------------------------
static void serve_loop(tm_engine* e)
{
    (void)hipSetDevice(e->cfg.device);
    std::vector<StepUpdate> updates;
    for (;;) {
        while (e->api_waiting.load() > 0) {  // callers queue on the mutex: let them in before the next step
            std::this_thread::yield();
        }
        std::unique_lock<std::mutex> lk(e->mu);
        e->cv_work.wait(lk, [&] { return e->loop_stop || (e->sched && e->sched->n_active() + e->sched->n_waiting() > 0); });
        if (e->loop_stop) {
            break;
        }
        updates.clear();
        const int rc = step_locked(e, nullptr, nullptr, &updates);
        if (rc) {  // device error: nothing that is queued or running can finish
            e->loop_rc  = rc;
            e->loop_err = tm_last_error();
            updates.clear();
            if (e->sched) {
                const int B = e->cfg.max_batch_size;
                for (int b = 0; b < B; ++b) {
                    e->h_active[b] = 0;
                }
                e->pending.valid = false;
                e->sched->abort_all(TM_FAIL);
            }
            lk.unlock();
            e->cv_out.notify_all();
            break;
        }
        lk.unlock();
        e->cv_out.notify_all();
        if (e->on_update) {
            for (const StepUpdate& u : updates) {
                e->on_update(e->on_update_user, u.id, u.status, u.n_tokens);
            }
        }
    }
}

int tm_engine_serve_start(tm_engine* e, tm_request_cb on_update, void* user)
{
    TM_REQUIRE(e, "null pointer");
    TM_REQUIRE(e->started, "engine not started");
    if (e->loop_on.load()) {
        set_last_error("the engine thread is already running");
        return TM_CONFLICT;
    }
    {
        ApiLock lock(e);
        TM_HIP_CHECK(hipSetDevice(e->cfg.device));
        TM_TRY(cb_enter(e));  // fails while a static batch is admitted
        e->on_update      = on_update;
        e->on_update_user = user;
        e->loop_stop      = false;
        e->loop_rc        = 0;
        e->loop_err.clear();
    }
    e->loop = std::thread(serve_loop, e);
    e->loop_on.store(true);
    return 0;
}

int tm_engine_serve_stop(tm_engine* e)
{
    TM_REQUIRE(e, "null pointer");
    if (!e->loop_on.load()) {
        return 0;
    }
    if (e->loop.joinable() && std::this_thread::get_id() == e->loop.get_id()) {
        set_last_error("tm_engine_serve_stop called from the engine thread (inside the on_update callback)");
        return TM_CONFLICT;  // the thread cannot join itself
    }
    {
        ApiLock lock(e);
        e->loop_stop = true;
    }
    e->cv_work.notify_all();
    if (e->loop.joinable()) {
        e->loop.join();
    }
    e->loop_on.store(false);
    e->cv_out.notify_all();
    if (e->loop_rc) {
        set_last_error("engine thread: " + e->loop_err);
        return e->loop_rc;
    }
    return 0;
}

int tm_engine_wait(tm_engine* e, int64_t req_id, int have_tokens, int timeout_ms, int* status, int* n_tokens)
{
    TM_REQUIRE(e && status && n_tokens, "null pointer");
    TM_REQUIRE(e->loop_on.load(), "tm_engine_wait needs the engine thread (tm_engine_serve_start)");
    ApiLock    lock(e);
    const auto ready = [&] {
        const SchedRequest* r = e->sched ? e->sched->find(req_id) : nullptr;
        return !r || r->status != 0 || (int)r->out.size() > have_tokens || e->loop_rc != 0 || e->loop_stop;
    };
    if (timeout_ms < 0) {
        e->cv_out.wait(lock.lk, ready);
    }
    else {
        e->cv_out.wait_for(lock.lk, std::chrono::milliseconds(timeout_ms), ready);
    }
    return poll_locked(e, req_id, status, nullptr, 0, n_tokens);
}

}  // extern "C"
