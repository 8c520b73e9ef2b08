// Internal header of the engine translation units (engine.hip: configuration, weights, start / destroy; engine_forward.hip: the
// forward pass, static-batch prefill / decode, hipGraph capture; engine_comm.hip: tensor-parallel collectives and communicator set-up;
// engine_tune.hip: the measured GEMM dispatch; engine_serve.hip: continuous batching and the engine thread).  Not part of the C-ABI.
#pragma once
#include "../../include/tm_mi355x.h"
#include "scheduler.h"
#include "tm_common.h"
#include "tm_kernels.h"
#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <rccl/rccl.h>
#include <thread>
#include <chrono>
#include <string>
#include <tuple>
#include <vector>


namespace tmk {

int build_rope_table(half_t* out, int max_pos, int dim, float base, int type, float factor, float low, float high,
                     int orig_max_pos);

#define TM_NCCL_CHECK(expr)                                                                        \
    do {                                                                                           \
        ncclResult_t _r = (expr);                                                                  \
        if (_r != ncclSuccess) {                                                                   \
            ::tmk::set_last_error(std::string(#expr) + ": " + ncclGetErrorString(_r));             \
            return 5;                                                                              \
        }                                                                                          \
    } while (0)

#define TM_TRY(expr)                                                                               \
    do {                                                                                           \
        int _rc = (expr);                                                                          \
        if (_rc) {                                                                                 \
            return _rc;                                                                            \
        }                                                                                          \
    } while (0)

// ---------------------------------------------------------------------------------------------
struct Slot {
    void*   dev   = nullptr;  // staging (boundary layout) -- freed by process_weights for linears
    int64_t bytes = 0;
    bool    filled = false;
};

struct LinearSlots {
    LinearWeight w;
    std::string  prefix;
};

struct Layer {
    LinearSlots qkv, wo, w13, w2;
    half_t*     attn_norm = nullptr;
    half_t*     ffn_norm  = nullptr;
    // mixture of experts: the dense w13 / w2 are unused; gate slot + per-expert slots feed `moe`
    bool                     is_moe = false;
    std::vector<LinearSlots> ex13, ex2;
    MoeBlock                 moe;
};

}  // namespace tmk

using namespace tmk;

struct tm_engine {
    tm_engine_config cfg{};
    // local (per-rank) dims
    int q_heads = 0, kv_heads = 0, inter = 0, vocab_local = 0, hidden = 0, D = 128;
    int qkv_n = 0;

    hipStream_t  stream = nullptr;
    ncclComm_t   comm   = nullptr;
    // TM_COMM_STREAM (default 1): the all-reduces of PREFILL-sized tensor-parallel forwards run on their own stream under the other
    // row half's GEMMs (engine_comm.hip: allreduce_rows_side; engine_forward.hip: forward_tail_two_halves).  Decode-sized forwards
    // keep every collective on the engine stream: a decode layer is one dependent chain, a side stream had nothing to run beside it
    // (profiles/r02_comm_stream_arms.txt; the fork / join arm of rounds 2-4 is gone)
    hipStream_t  comm_stream = nullptr;
    std::vector<hipEvent_t> pipe_events;      // (ready, done) pairs of the side-stream all-reduces of ONE forward
    size_t       pipe_events_used = 0;
    int          pipe_min_rows    = 1024;     // TM_PIPE_MIN_ROWS: smallest row half (forwards below twice this stay unsplit)
    int64_t      pipe_forwards = 0, pipe_mb_forwards = 0, pipe_allreduces = 0;   // tm_engine_comm_overlap_info
    // set by prefill_slots() around ONE forward: the forward's sequences [0, seqs_a) own its rows [0, rows_a) -- two micro-batches that
    // share nothing between the embedding and the lm_head (forward_layers_two_microbatches); seqs_a = 0: no such split (one sequence,
    // a lopsided boundary, a mixed forward) -> the row-half schedule inside every layer (forward_tail_two_halves)
    struct { int seqs_a = 0, rows_a = 0; } mb;
    float        emulate_ar_gbps = 0.f;       // TM_EMULATE_AR_GBPS, one-rank emulation only: stand-in for an exchange's duration (engine_comm.hip)
    // mixed forwards: the decode rows' attention runs on this stream beside the prefill rows' K/V store -> flatten -> attention
    // on the engine stream (reference: aux_stream_ + event fork / join, unified_attention_layer.cc:613-651)
    hipStream_t  aux_stream = nullptr;
    hipEvent_t   ev_aux_fork = nullptr, ev_aux_join = nullptr;
    bool         mixed_two_streams = true;  // TM_MIXED_2STREAM=0: back to back on the engine stream
    bool         mixed_steps_on    = true;  // TM_MIXED_STEP=0: prefill forwards and decode steps alternate
    bool         graph_comm        = true;  // TM_GRAPH_COMM=0: tensor-parallel decode steps stay eager (collectives not captured)
    // native communicator (TM_COMM=native, comm_p2p.hip): this rank's symmetric segment [flags 256 B | tile 0 | tile 1] and the
    // peers' mappings of theirs; serves the row-parallel all-reduces of forwards with M <= p2p_rows, RCCL the rest
    void*        p2p_seg = nullptr;
    void*        p2p_peer[8] = {};
    uint32_t*    p2p_state = nullptr;
    int          p2p_rows = 0;
    int          p2p_rows2 = 0;  // rows of the segment's two-shot regions (in2 / out2): forwards larger than p2p_rows (prefill) without RCCL
    bool         p2p_ready = false;
    bool         comm_overlap = false;   // TM_COMM_STREAM != 0 and an RCCL communicator: prefill all-reduces on comm_stream
    bool         graph_comm_failed = false;  // capturing the RCCL calls failed once: stay eager
    bool         use_comm = false;  // collectives on the data path: tp > 1 (or TM_FORCE_COMM=1: single-rank communicator,
                                    // exercises the RCCL code path on a 1-GPU box)
    std::map<std::string, Slot> slots;
    std::vector<Layer>          layers;
    half_t*      tok_embeddings = nullptr;
    half_t*      final_norm     = nullptr;
    LinearSlots  output;
    bool         weights_ready = false;
    bool         started       = false;

    // KV cache
    KvLayout  layout{};
    char*     pool        = nullptr;
    int64_t   block_bytes = 0;
    int64_t   num_blocks  = 0;
    std::vector<int> free_blocks;
    int       max_blocks_per_seq = 0;
    uint64_t* d_block_ptrs    = nullptr;  // [max_batch][max_blocks_per_seq]
    int*      d_cu_block_nums = nullptr;  // [max_batch+1]

    // activations
    int     max_tokens = 0;
    half_t *d_resid = nullptr, *d_x = nullptr, *d_qkv = nullptr, *d_attn = nullptr, *d_act = nullptr, *d_tmp = nullptr;
    half_t* d_logits = nullptr;
    half_t* d_last   = nullptr;
    float*  d_gemm_ws = nullptr;
    size_t  gemm_ws_bytes = 0;
    // RMSNorm folded into the decode GEMMs (NormFold, tm_kernels.h): per-tile sums of squares [hidden / 64][64] and the split-K
    // arrival counters of the producing GEMM; TM_FOLD_NORM=0 keeps the reduce-norm launches
    float*    d_ss      = nullptr;
    unsigned* d_tickets = nullptr;
    int       fold_max_rows = 64;  // rows of a forward up to which the folded layer runs (TM_FOLD_MAX_M: 64 | 128)
    int       fold_norm = 0;  // bit 0: wo -> w1w3, bit 1: w2 -> the next layer's w_qkv
    unsigned  h_mark = 0;             // host copy of the native communicator's give-up mark (device_marks_fetch)
    bool      comm_failed = false;    // a give-up mark was seen: the ranks' call sequences may have diverged (sticky, see device_marks_check)
    float*  d_attn_ws = nullptr;
    half_t *d_kflat = nullptr, *d_vflat = nullptr;
    int     kflat_stride = 0;
    half2_t* d_rope = nullptr;
    int      rope_max_pos = 0;

    // batch state (device)
    int *d_next_ids = nullptr;
    int *d_ids = nullptr, *d_k_len = nullptr, *d_cu_q = nullptr, *d_cu_koff = nullptr, *d_rows = nullptr;
    int *d_cu_q_b = nullptr;   // cu_q of the second micro-batch of a tensor-parallel prefill forward, counted from ITS first row
    int *d_generated = nullptr, *d_step = nullptr;
    int *d_prefill_ids = nullptr;
    half_t* d_argmax_val = nullptr;
    float*  d_cand = nullptr;
    float*  d_cand_all = nullptr;

    // batch state (host)
    int              batch = 0, max_new = 0;
    std::vector<int> h_len;
    std::vector<std::vector<int>> h_blocks;
    int              steps_done = 0;
    int              steps_fetched = 0;  // steps_done at the last fetch that saw no communicator give-up mark
    int              steps_valid = -1;   // >= 0 after a give-up: the columns of d_generated known to be valid

    int            decode_splits = 1;
    bool           fuse_qkv      = false;  // decode: qkv GEMM output -> attention kernel directly (int8 KV, MFMA kernel)
    hipGraphExec_t graph = nullptr;
    // per-kernel-category HIP event profiling (tm_engine_profile_decode)
    bool                                            prof_on = false;
    std::vector<hipEvent_t>                         prof_pool;
    size_t                                          prof_used = 0;
    std::vector<std::tuple<int, size_t, size_t>>    prof_spans;  // (category, start event, stop event)
    std::vector<float>                              h_ttft_ms;
    int            graph_batch = 0;
    int            graph_max_new = 0;  // record_kernel's bound is a captured kernel argument

    // continuous batching (tm_engine_submit / step / poll / cancel): slot-based, every decode step runs all
    // max_batch_size slots; free slots are parked on a scratch block with k_len = 1 and never advance
    std::unique_ptr<tmk::BatchScheduler> sched;
    int64_t          mixed_steps = 0;        // scheduler steps whose decode rows rode on a prefill forward
    int*             d_active    = nullptr;  // [max_batch] 1 = slot holds a running sequence
    uint64_t*        d_pf_block_ptrs = nullptr;  // [max_batch][max_blocks_per_seq] the table an admission's prefill walks: a new
                                                 // slot's row reaches the decode table (d_block_ptrs) only once it is prefilled,
                                                 // until then its decode row stays parked on the dummy block (a mixed forward runs
                                                 // the parked decode row and the real prefill of the same slot side by side)
    int*             d_pf_k_len  = nullptr;  // prefill-local arrays (the decode arrays stay live during an admission)
    int*             d_pf_cu_q   = nullptr;
    int*             d_first_ids = nullptr;  // [max_batch] first tokens of an admission's earlier prefill iterations (mixed steps)
    std::vector<int> h_active;
    int              dummy_block = -1;
    hipGraphExec_t   graph_cb    = nullptr;
    // Two-phase schedule / forward overlap (reference: the two alternating batch phases of turbomind.cc:171, engine.cc:770-870):
    // a pure decode step is ISSUED (graph launch + result copy into a pinned buffer + event) and RETIRED (event wait, tokens to
    // the scheduler, finished slots parked) by different scheduler steps -- step N+1 is issued before step N is retired, so the
    // host's bookkeeping, the caller's polling and the next launch run under the device's step N+1.  A sequence that ends in
    // step N rides one more step as a dead row (its token is dropped: the slot's request id no longer matches).
    // TM_ASYNC_STEP=1 switches the overlap on; by default every step is retired by the call that issued it (see cb_enter).
    struct PendingStep {
        bool                 valid = false;
        bool                 lp    = false;  // the step's logprob records were copied into the pinned buffers of `buf` as well
        int                  buf   = 0;
        std::vector<int64_t> ids;  // request of every slot whose token this step produces (-1: free, parked, prefilled by this step)
    } pending;
    bool       async_step_on = false;
    int*       h_step_pin[2] = {nullptr, nullptr};  // pinned [max_batch + 1]: next ids of the slots, then the communicator's give-up mark
    hipEvent_t ev_step[2]    = {nullptr, nullptr};
    int        issue_count   = 0;
    int64_t    overlapped_steps = 0;  // decode steps issued while the previous one was still unretired

    // stochastic sampling (tm_engine_set_sampling / tm_engine_submit_ex); off = arg-max
    bool      sampling_on = false, graph_sampling = false, graph_cb_sampling = false;
    float *   d_temp = nullptr, *d_topp = nullptr, *d_minp = nullptr, *d_u = nullptr;
    int*      d_topk = nullptr;
    uint64_t* d_seed = nullptr;
    void*     d_sample_ws = nullptr;
    half_t*   d_logits_gather = nullptr;  // tp > 1 + sampling: [tp][max_batch][vocab / tp] all-gathered shards ...
    half_t*   d_logits_full   = nullptr;  // ... and the full rows [max_batch][vocab] every rank samples from
    void*     d_moe_ws    = nullptr;  // routing tables + expert activations of one forward (moe_workspace_bytes)
    // logprobs of the generated tokens (tm_engine_set_logprobs, static batch): records [batch][max_new] x [cap] next to d_generated
    int    logprobs_next = 0, logprobs_n = 0;
    bool   logprobs_on = false, graph_logprobs = false;
    float *d_lpr_vals = nullptr, *d_lpr_sel = nullptr;
    int *  d_lpr_idx = nullptr, *d_lpr_num = nullptr, *d_kept = nullptr;
    size_t lpr_records = 0, lpr_entries = 0;  // allocated records / records x cap
    // continuous batching (tm_engine_request_logprobs): the record of the CURRENT step per batch slot, [max_batch][kMaxLogProb] on the
    // device, copied behind every step into the pinned buffer of the step's parity (first cb_lp_used columns) and appended to the
    // requests that asked for logprobs when the step is retired
    bool   cb_logprobs_on = false, graph_cb_logprobs = false;
    int    cb_lp_used = 0;
    float *d_cb_lp_vals = nullptr, *d_cb_lp_sel = nullptr, *h_cb_lp_vals[2] = {nullptr, nullptr}, *h_cb_lp_sel[2] = {nullptr, nullptr};
    int *  d_cb_lp_idx = nullptr, *d_cb_lp_num = nullptr, *h_cb_lp_idx[2] = {nullptr, nullptr}, *h_cb_lp_num[2] = {nullptr, nullptr};
    std::vector<tm_sampling>       h_sampling;      // static batch: parameters of the next prefill
    std::map<int64_t, tm_sampling> cb_sampling;     // continuous batching: per request

    // logits processors (tm_engine_set_logits_params / tm_engine_submit_gen): repetition penalty, bad ids, min length;
    // off = arg-max / sampling see the raw lm_head output.  d_seen = persistent "token occurs in the sequence" bitmask
    // per batch slot over the GLOBAL vocabulary.
    bool      logits_on = false, graph_logits = false, graph_cb_logits = false;
    uint32_t* d_seen     = nullptr;
    int       seen_words = 0;
    float*    d_lp_rep    = nullptr;
    int *     d_lp_minlen = nullptr, *d_lp_ban = nullptr, *d_lp_end = nullptr;
    std::vector<tm_logits_param>       h_logits;   // static batch: parameters of the next prefill
    std::map<int64_t, tm_logits_param> cb_logits;  // continuous batching: per request

    // engine thread (tm_engine_serve_start): runs step_locked() while requests exist.  `mu` serialises the scheduler
    // and every device-side effect of submit / step / poll / cancel; API callers announce themselves in api_waiting so
    // that the loop (which re-locks immediately) lets them in between two steps.
    std::mutex              mu;
    std::condition_variable cv_work, cv_out;
    std::thread             loop;
    std::atomic<int>        api_waiting{0};
    std::atomic<bool>       loop_on{false};
    bool                    loop_stop = false;
    int                     loop_rc   = 0;
    std::string             loop_err;
    tm_request_cb           on_update      = nullptr;
    void*                   on_update_user = nullptr;
};

namespace tmk {

// lock of an API call: counted, so that the engine thread yields to callers between steps
struct ApiLock {
    tm_engine*                   e;
    std::unique_lock<std::mutex> lk;
    explicit ApiLock(tm_engine* eng): e(eng), lk(eng->mu, std::defer_lock)
    {
        e->api_waiting.fetch_add(1);
        lk.lock();
        e->api_waiting.fetch_sub(1);
    }
};

struct StepUpdate {
    int64_t id;
    int     status, n_tokens;
};

enum ProfCat { P_EMBED = 0, P_GEMM_QKV, P_KV_STORE, P_ATTN, P_GEMM_O, P_RES_NORM, P_GEMM_GATE_UP, P_GEMM_DOWN, P_LM_HEAD,
               P_SAMPLE, P_ALLREDUCE, P_NUM };

#define TM_PROF(cat, stmt)                                                                         \
    do {                                                                                           \
        size_t _a = 0;                                                                             \
        if (e->prof_on) {                                                                          \
            _a = prof_event(e);                                                                    \
        }                                                                                          \
        stmt;                                                                                      \
        if (e->prof_on) {                                                                          \
            e->prof_spans.emplace_back((int)(cat), _a, prof_event(e));                             \
        }                                                                                          \
    } while (0)

// the decode rows of a mixed forward (forward(), engine_forward.hip)
struct MixedDecode {
    int             rows;        // decode rows = batch slots
    const int*      k_len;       // [rows] context lengths including this step's token
    const uint64_t* block_ptrs;  // the unshifted block table
    const int*      cu_q;        // [rows + 1] = 0 .. rows (one token per decode row; kv_rope_store of the fp16-KV path)
    const int*      active;      // [rows] 1 = the slot holds a running sequence (logits processors skip the others)
};

// the decode step that rides on the last prefill forward of a continuous-batching admission (prefill_slots(), engine_forward.hip)
struct MixedStep {
    int        rows;     // batch slots = decode rows
    int*       k_len;    // the decode state arrays (NOT the iteration-local e->d_k_len)
    const int* active;
    int*       ids;      // current token of every slot
    const uint64_t* block_ptrs;  // the decode block table
    const int* cu_q;     // 0 .. rows (the decode step's own array)
    bool*      done;
};

template<class T>
inline int dmalloc(T** p, size_t n)
{
    TM_HIP_CHECK(hipMalloc((void**)p, n * sizeof(T)));
    return 0;
}

// ---- shared between the engine translation units ----
int launch_advance_active(int* k_len, const int* active, int n, hipStream_t st);
size_t prof_event(tm_engine* e);
void p2p_tables(tm_engine* e, half_t** data, uint32_t** flags);
int reduce_residual_norm(tm_engine* e, int M, const half_t* norm_w);
bool prefill_pipe_ok(const tm_engine* e, int M);
int allreduce_rows_side(tm_engine* e, int r0, int rows, hipEvent_t* done);
int pipe_wait(tm_engine* e, hipEvent_t done);
int forward(tm_engine* e, const int* d_ids, int M, int nseq, bool decode, int max_q_len, int max_k_len, int kflat_stride, int slot0,
            const MixedDecode* md = nullptr);
int device_marks_fetch(tm_engine* e, bool async);
int device_marks_check(tm_engine* e);

}  // namespace tmk

extern "C" {

extern const tm_logits_param kNoLogitsParam;  // every logits processor off (engine_forward.hip)

int tune_decode_gemms(tm_engine* e, int M, bool verbose);
int sampling_upload(tm_engine* e, const tm_sampling* p, int slot0, int n);
int logits_param_check(const tm_logits_param& p);
int logits_upload(tm_engine* e, const tm_logits_param* p, const int* prompt_len, const int* eos, int slot0, int n);
void setup_decode(tm_engine* e, int batch);
int prefill_slots(tm_engine* e, const int* const* seq_ids, const int* host_lens, int batch, int slot0, float* ttft_ms,
                  const MixedStep* mix = nullptr);
bool graph_enabled(const tm_engine* e);
int capture_step(tm_engine* e, int (*step)(tm_engine*), hipGraphExec_t* exec);

}  // extern "C"
