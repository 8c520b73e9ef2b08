// Continuous-batching scheduler (host side, no device code): request queue, batch slots, KV block accounting.
//
// Replaces, at the level this engine needs: Engine::Impl::Schedule / InternalThreadEntry
// (src/turbomind/engine/engine.cc:434-470,770-870) and Scheduler::Schedule (engine/scheduler.cc:1018-1078):
//   * requests are admitted in arrival order (the reference sorts by unique_id, scheduler.cc:1076-1078);
//   * a request is admitted when a batch slot is free and the block pool can hold prompt + max_new_tokens
//     (blocks of 64 tokens, all reserved at admission: no preemption / eviction, no prefix cache -- SURVEY 8f-1
//     names the prefix-free variant);
//   * the prefill token budget of one scheduler step is max_prefill_token_num (ForwardTokenResource,
//     engine.cc:462-464); at least one waiting request is admitted per step if it fits at all;
//   * a sequence finishes on EOS (unless ignore_eos), at max_new_tokens, or when cancelled; its slot and blocks are
//     free for the NEXT step (status codes: Request::kFinish = 7, kCancel = 8, engine/request.h:120-131).
// The class is pure bookkeeping so that it can be driven from CPU tests through tm_sched_* (c_api.hip) exactly as the
// engine drives it.
#pragma once

#include <cstddef>
#include <cstdint>
#include <deque>
#include <map>
#include <vector>

namespace tmk {

struct SchedRequest {
    int64_t          id       = 0;
    std::vector<int> prompt;
    int              max_new  = 0;
    int              eos      = -1;  // < 0: never stop on a token (ignore_eos)
    std::vector<int> stops;          // further ids that end the sequence (GenerationConfig.stop_token_ids)
    int              status   = 0;   // Request::k*: 0 = waiting / running, 7 finished, 8 cancelled
    int              slot     = -1;  // batch slot while running
    bool             running  = false;
    std::vector<int> blocks;         // KV blocks owned while running
    std::vector<int> out;            // generated tokens so far
    // GenerationConfig.logprobs (set_logprobs): per generated token the first lp_n kept candidates (entries beyond lp_num[t] are
    // padding), their count and the token's own logprob -- the request's logprob_vals / logprob_indexes / logprob_nums outputs
    // (src/turbomind/engine/model_request.cc:88-90)
    int                lp_n = 0;
    std::vector<float> lp_vals, lp_sel;
    std::vector<int>   lp_idx, lp_num;
};

struct SchedAdmit {
    int64_t id;
    int     slot;
};

class BatchScheduler {
public:
    BatchScheduler(int max_batch, int num_blocks, int session_len, int block_len = 64):
        max_batch_(max_batch), session_len_(session_len), block_len_(block_len), slot_req_(max_batch, -1)
    {
        free_blocks_.resize(num_blocks);
        for (int i = 0; i < num_blocks; ++i) {
            free_blocks_[i] = num_blocks - 1 - i;
        }
        total_blocks_ = num_blocks;
    }

    // 0 = queued; 6 (kTooLong) = prompt + max_new exceeds session_len; 11 (kOutOfMemory) = can never fit the pool;
    // 1 (kInvalid) = empty prompt / max_new < 1
    int submit(const int* ids, int n, int max_new, int eos, int64_t* id)
    {
        if (n < 1 || max_new < 1 || ids == nullptr) {
            return 1;
        }
        if (n + max_new > session_len_) {
            return 6;
        }
        if (blocks_for(n + max_new) > total_blocks_) {
            return 11;
        }
        SchedRequest r;
        r.id      = next_id_++;
        r.prompt.assign(ids, ids + n);
        r.max_new = max_new;
        r.eos     = eos;
        waiting_.push_back(r.id);
        if (id) {
            *id = r.id;
        }
        reqs_[r.id] = std::move(r);
        return 0;
    }

    // Admit waiting requests (arrival order) into free slots while blocks and the prefill token budget last.
    // A request larger than the budget is admitted alone (the engine chunks its prefill).
    std::vector<SchedAdmit> admit(int token_budget)
    {
        std::vector<SchedAdmit> out;
        int                     tokens = 0;
        while (!waiting_.empty()) {
            auto it = reqs_.find(waiting_.front());
            if (it == reqs_.end() || it->second.status != 0) {  // cancelled (and possibly forgotten) while waiting
                waiting_.pop_front();
                continue;
            }
            SchedRequest& r = it->second;
            const int n    = (int)r.prompt.size();
            const int need = blocks_for(n + r.max_new);
            int       slot = -1;
            for (int b = 0; b < max_batch_; ++b) {
                if (slot_req_[b] < 0) {
                    slot = b;
                    break;
                }
            }
            if (slot < 0 || need > (int)free_blocks_.size()) {
                break;  // head-of-line blocking, like the reference: order is never changed
            }
            if (!out.empty() && tokens + n > token_budget) {
                break;
            }
            for (int i = 0; i < need; ++i) {
                r.blocks.push_back(free_blocks_.back());
                free_blocks_.pop_back();
            }
            r.slot          = slot;
            r.running       = true;
            slot_req_[slot] = r.id;
            tokens += n;
            out.push_back({r.id, slot});
            waiting_.pop_front();
        }
        return out;
    }

    // would admit() admit anything right now?  (const: the engine asks before it decides to keep a decode step in flight)
    bool admit_ready() const
    {
        for (int64_t id : waiting_) {
            auto it = reqs_.find(id);
            if (it == reqs_.end() || it->second.status != 0) {
                continue;  // cancelled while waiting: admit() drops it
            }
            const SchedRequest& r    = it->second;
            bool                slot = false;
            for (int b = 0; b < max_batch_ && !slot; ++b) {
                slot = slot_req_[b] < 0;
            }
            return slot && blocks_for((int)r.prompt.size() + r.max_new) <= (int)free_blocks_.size();  // head of the line decides
        }
        return false;
    }

    // a token was produced for the sequence in `slot`; returns true if the sequence finished (slot + blocks released)
    bool on_token(int slot, int token)
    {
        const int64_t id = slot_req_[slot];
        if (id < 0) {
            return false;
        }
        SchedRequest& r = reqs_[id];
        r.out.push_back(token);
        bool stop = r.eos >= 0 && token == r.eos;
        for (int sid : r.stops) {
            stop = stop || token == sid;
        }
        if (stop || (int)r.out.size() >= r.max_new) {
            finish(r, 7);
            return true;
        }
        return false;
    }

    // the logprob record of the token that on_token() is about to hand over for `slot` (call it first: on_token may free the slot);
    // vals / idx hold at least min(num, lp_n) entries
    void on_logprobs(int slot, const float* vals, const int* idx, int num, float sel)
    {
        const int64_t id = slot_req_[slot];
        if (id < 0) {
            return;
        }
        SchedRequest& r = reqs_[id];
        if (r.lp_n <= 0) {
            return;
        }
        const int n = num < 0 ? 0 : (num < r.lp_n ? num : r.lp_n);
        r.lp_vals.insert(r.lp_vals.end(), vals, vals + n);
        r.lp_idx.insert(r.lp_idx.end(), idx, idx + n);
        r.lp_vals.resize(r.lp_vals.size() + (r.lp_n - n), 0.f);
        r.lp_idx.resize(r.lp_idx.size() + (r.lp_n - n), -1);
        r.lp_num.push_back(n);
        r.lp_sel.push_back(sel);
    }

    // logprobs of a QUEUED request (nothing generated yet); 1 = unknown id / already running
    int set_logprobs(int64_t id, int n)
    {
        auto it = reqs_.find(id);
        if (it == reqs_.end() || it->second.running || !it->second.out.empty() || it->second.status != 0) {
            return 1;
        }
        it->second.lp_n = n;
        return 0;
    }

    // additional stop ids of a queued / running request; 1 = unknown id
    int set_stop_ids(int64_t id, const int* ids, int n)
    {
        auto it = reqs_.find(id);
        if (it == reqs_.end()) {
            return 1;
        }
        it->second.stops.assign(ids, ids + (n > 0 ? n : 0));
        return 0;
    }

    // 0 ok; 1 unknown id.  A running sequence is released immediately (the caller deactivates its slot).
    int cancel(int64_t id, int* released_slot)
    {
        auto it = reqs_.find(id);
        if (released_slot) {
            *released_slot = -1;
        }
        if (it == reqs_.end()) {
            return 1;
        }
        SchedRequest& r = it->second;
        if (r.status != 0) {
            return 0;
        }
        if (r.running) {
            if (released_slot) {
                *released_slot = r.slot;
            }
            finish(r, 8);
        }
        else {
            r.status = 8;
        }
        return 0;
    }

    // the engine cannot continue (device error): every unfinished request ends with `status` (Request::kFail = 5);
    // slots and blocks return to the pool
    void abort_all(int status)
    {
        for (auto& kv : reqs_) {
            SchedRequest& r = kv.second;
            if (r.status != 0) {
                continue;
            }
            if (r.running) {
                finish(r, status);
            }
            else {
                r.status = status;
            }
        }
        waiting_.clear();
    }

    const SchedRequest* find(int64_t id) const
    {
        auto it = reqs_.find(id);
        return it == reqs_.end() ? nullptr : &it->second;
    }
    // forget a finished request (poll consumed it); false: unknown id or the request is still queued / running
    bool erase(int64_t id)
    {
        auto it = reqs_.find(id);
        if (it == reqs_.end() || it->second.status == 0) {
            return false;
        }
        reqs_.erase(it);
        for (auto w = waiting_.begin(); w != waiting_.end(); ++w) {  // a request cancelled while it was waiting
            if (*w == id) {
                waiting_.erase(w);
                break;
            }
        }
        return true;
    }
    size_t n_records() const { return reqs_.size(); }
    int64_t slot_request(int slot) const { return slot_req_[slot]; }
    int     n_active() const
    {
        int n = 0;
        for (int64_t v : slot_req_) {
            n += v >= 0;
        }
        return n;
    }
    int n_waiting() const
    {
        int n = 0;
        for (int64_t id : waiting_) {
            auto it = reqs_.find(id);
            n += it != reqs_.end() && it->second.status == 0;
        }
        return n;
    }
    int n_free_blocks() const { return (int)free_blocks_.size(); }
    int blocks_for(int tokens) const { return (tokens + block_len_ - 1) / block_len_; }

private:
    void finish(SchedRequest& r, int status)
    {
        r.status = status;
        for (int b : r.blocks) {
            free_blocks_.push_back(b);
        }
        r.blocks.clear();
        if (r.slot >= 0) {
            slot_req_[r.slot] = -1;
        }
        r.running = false;
        r.slot    = -1;
    }

    int                             max_batch_, session_len_, block_len_, total_blocks_ = 0;
    int64_t                         next_id_ = 1;
    std::vector<int>                free_blocks_;
    std::vector<int64_t>            slot_req_;
    std::deque<int64_t>             waiting_;
    std::map<int64_t, SchedRequest> reqs_;
};

// Tensor-parallel prefill forward of `nseq` sequences with row offsets cu_q[0 .. nseq] (cu_q[0] = 0): the sequence boundary nearest to
// half the rows, so that the forward can run as two micro-batches that leapfrog through the layers (engine_forward.hip:
// forward_layers_two_microbatches; each one's all-reduce on the side stream under the other's kernels).  Returns the number of
// sequences of the first micro-batch, 0 when there is no usable boundary: fewer than two sequences, or the smaller side would fall
// below min_rows / 2 rows (a lopsided split hides little and halves nothing; the forward then splits by rows inside every layer).
// Ties go to the earlier boundary.  Pure host logic: tm_prefill_split exports it for the CPU tests.
inline int prefill_microbatch_split(const int* cu_q, int nseq, int min_rows)
{
    if (nseq < 2) {
        return 0;
    }
    const int tokens = cu_q[nseq];
    int       best   = 1;
    for (int s = 2; s < nseq; ++s) {
        const int d0 = 2 * cu_q[best] - tokens, d1 = 2 * cu_q[s] - tokens;
        if ((d1 < 0 ? -d1 : d1) < (d0 < 0 ? -d0 : d0)) {
            best = s;
        }
    }
    const int small = cu_q[best] < tokens - cu_q[best] ? cu_q[best] : tokens - cu_q[best];
    return 2 * small >= min_rows && small > 0 ? best : 0;
}

}  // namespace tmk
