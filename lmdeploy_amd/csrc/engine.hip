// Static-batch decode engine around the HIP kernels: the MI355X-native counterpart of
//   TurboMind::CreateEngine / ModelExecutor::Run / LanguageModel::Forward / UnifiedDecoder::Forward
//   (src/turbomind/turbomind.cc:281-361, engine/model_executor.cc:68-101, models/language_model.cc:493-542,
//    models/llama/unified_decoder.cc:163-380, unified_attention_layer.cc:365-441, LlamaFfnLayer.cc:28-91).
//
// One process per GPU.  All per-step state (context lengths, current token ids, generated tokens, step
// counter) lives in device memory so a whole decode step is a fixed kernel sequence that is captured ONCE in
// a hipGraph and replayed (the reference keeps ids on the device too: language_model.cc:540).  Tensor
// parallelism: column-parallel w_qkv / w1w3, row-parallel wo / w2 followed by an RCCL all-reduce of the
// fp16 [M,H] partial sums (comm/nccl/nccl.cu:356-398), lm_head sharded over the vocabulary with a
// (value, index) all-gather instead of gathering logits.
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
// device-side helpers for the engine
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// out[i] = h(mean + std * N(0,1)), counter-based (Box-Muller on splitmix64), 2 values per thread
__global__ void fill_normal_kernel(half_t* out, size_t n, float mean, float stddev, uint64_t seed)
{
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (i >= n) {
        return;
    }
    const uint64_t r  = splitmix64(seed ^ (i * 0x9E3779B97F4A7C15ull + 0x1234567ull));
    const float    u1 = ((float)(uint32_t)(r >> 40) + 1.0f) * (1.0f / 16777217.0f);  // (0,1)
    const float    u2 = (float)(uint32_t)((r >> 8) & 0xffffffu) * (1.0f / 16777216.0f);
    const float    rr = sqrtf(-2.0f * __logf(u1));
    float          s, c;
    __sincosf(6.283185307179586f * u2, &s, &c);
    out[i] = (half_t)(mean + stddev * rr * c);
    if (i + 1 < n) {
        out[i + 1] = (half_t)(mean + stddev * rr * s);
    }
}

__global__ void fill_fp8_kernel(uint8_t* out, size_t n, uint64_t seed)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        uint64_t z = (i + 1) * 0x9E3779B97F4A7C15ull + seed * 0xD1B54A32D192ED03ull;
        z          = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z          = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        uint8_t b  = (uint8_t)(z >> 33);
        if ((b & 0x7f) == 0x7f) {
            b &= 0xf7;  // never NaN
        }
        out[i] = b;
    }
}

__global__ void fill_const_f32_kernel(float* out, size_t n, float v)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        out[i] = v;
    }
}

__global__ void advance_kernel(int* k_len, int batch)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < batch) {
        k_len[b] += 1;
    }
}

// continuous batching: only slots that hold a running sequence advance
__global__ void advance_active_kernel(int* k_len, const int* active, int batch)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < batch && active[b]) {
        k_len[b] += 1;
    }
}

// continuous batching: a finished / cancelled slot goes back to the scratch block (no host memory involved: the launch needs
// no synchronisation, so it can queue up behind a decode step that is still running)
static int launch_advance_active(int* k_len, const int* active, int n, hipStream_t st)
{
    advance_active_kernel<<<(n + 63) / 64, 64, 0, st>>>(k_len, active, n);
    TM_HIP_CHECK(hipGetLastError());
    return 0;
}

__global__ void park_slot_kernel(int* active, int* k_len, uint64_t* block_row, int slot, uint64_t dummy_block_ptr)
{
    active[slot] = 0;
    k_len[slot]  = 1;
    block_row[0] = dummy_block_ptr;
}

// ids -> generated[b][step]; step++ (single thread does the counter after everybody read it)
__global__ void record_kernel(const int* ids, int* generated, int* step_counter, int batch, int max_new)
{
    const int b    = blockIdx.x * blockDim.x + threadIdx.x;
    const int step = *step_counter;
    if (b < batch && step < max_new) {
        generated[(size_t)b * max_new + step] = ids[b];
    }
    __syncthreads();
    if (b == 0) {
        *step_counter = step + 1;
    }
}

// pick the global arg-max out of tp (value, index) candidates per sequence
__global__ void pick_kernel(int* out_ids, const float* cand /*[tp][B][2]*/, int tp, int batch)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) {
        return;
    }
    float best = -INFINITY;
    int   bi   = 0;
    for (int r = 0; r < tp; ++r) {
        const float v = cand[((size_t)r * batch + b) * 2];
        const int   i = __float_as_int(cand[((size_t)r * batch + b) * 2 + 1]);
        if (v > best || (v == best && i < bi)) {
            best = v;
            bi   = i;
        }
    }
    out_ids[b] = bi;
}

// all-gathered vocabulary shards [tp][n][vl] -> full rows [n][tp * vl] (16-byte vectors; vl % 8 == 0)
__global__ void gather_vocab_kernel(half_t* __restrict__ full, const half_t* __restrict__ shards, int n, int vl, int tp)
{
    const size_t nvec = (size_t)n * tp * (vl / 8);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
        const int    v = (int)(i % (vl / 8));
        const int    r = (int)((i / (vl / 8)) % tp);
        const size_t b = i / ((size_t)(vl / 8) * tp);
        *(u32x4*)(full + (b * tp + r) * vl + (size_t)v * 8) = *(const u32x4*)(shards + ((size_t)r * n + b) * vl + (size_t)v * 8);
    }
}

__global__ void pack_candidates_kernel(float* cand, const int* ids, const half_t* vals, int batch)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < batch) {
        cand[b * 2]     = (float)vals[b];
        cand[b * 2 + 1] = __int_as_float(ids[b]);
    }
}

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
    // tensor-parallel collectives run on their own stream, forked from / joined to the engine stream by events (inside
    // a hipGraph capture the pair becomes a parallel branch): while RCCL moves the partial sums over xGMI the engine
    // stream pulls the NEXT linear's weights towards the Infinity Cache (weight_prefetch_kernel)
    hipStream_t  comm_stream = nullptr;
    hipEvent_t   ev_fork = nullptr, ev_join = nullptr;
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
    bool         comm_overlap = false;   // TM_COMM_STREAM=1: collectives on a side stream (fork / join around each)
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

namespace {
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
}  // namespace

namespace tmk {

static int64_t slot_bytes_linear(const tm_engine* e, int K, int N, const char* part)
{
    if (!strcmp(part, "qweight")) return (int64_t)K * N / 2;
    if (!strcmp(part, "scales") || !strcmp(part, "zeros")) return (int64_t)(K / e->cfg.model.group_size) * N * 2;
    if (!strcmp(part, "weight")) return (int64_t)K * N * 2;
    if (!strcmp(part, "weight_fp8")) return (int64_t)K * N;
    if (!strcmp(part, "scales_fp8")) return (int64_t)(K / 128) * ((N + 127) / 128) * 4;
    return 0;
}

static void add_linear(tm_engine* e, LinearSlots& l, const std::string& prefix, int K, int N, int type, int role = 0)
{
    l.prefix  = prefix;
    l.w.role  = role;  // dispatch-table key next to (K, N, M): the tuner times each role with its own consumer
    l.w.K     = K;
    l.w.N     = N;
    l.w.group = e->cfg.model.group_size;
    l.w.type  = type;
    if (type == TM_WEIGHT_U4) {
        for (const char* part : {"qweight", "scales", "zeros"}) {
            e->slots[prefix + "." + part].bytes = slot_bytes_linear(e, K, N, part);
        }
    }
    else if (type == TM_WEIGHT_FP8) {  // e4m3 [K][N] + fp32 128x128 block scales (weight_format.py:349-393)
        e->slots[prefix + ".weight"].bytes = slot_bytes_linear(e, K, N, "weight_fp8");
        e->slots[prefix + ".scales"].bytes = slot_bytes_linear(e, K, N, "scales_fp8");
    }
    else {
        e->slots[prefix + ".weight"].bytes = slot_bytes_linear(e, K, N, "weight");
    }
}

static int ensure_slot(tm_engine* e, const std::string& name, Slot** out)
{
    auto it = e->slots.find(name);
    TM_REQUIRE(it != e->slots.end(), "unknown weight slot: " + name);
    Slot& s = it->second;
    if (!s.dev) {
        TM_HIP_CHECK(hipMalloc(&s.dev, s.bytes));
    }
    *out = &s;
    return 0;
}

static int fill_normal(tm_engine* e, void* dst, size_t n, float mean, float stddev, uint64_t seed)
{
    const size_t thr = (n + 1) / 2;
    fill_normal_kernel<<<(thr + 255) / 256, 256, 0, e->stream>>>((half_t*)dst, n, mean, stddev, seed);
    TM_HIP_CHECK(hipGetLastError());
    return 0;
}

static int prepare_linear(tm_engine* e, LinearSlots& l)
{
    if (l.w.type == TM_WEIGHT_U4) {
        Slot &q = e->slots[l.prefix + ".qweight"], &s = e->slots[l.prefix + ".scales"], &z = e->slots[l.prefix + ".zeros"];
        TM_REQUIRE(q.filled && s.filled && z.filled, "weight not loaded: " + l.prefix);
        // dense linears whose every M goes to gemm_decode.hip keep ONE HBM image (P32); MoE experts run the grouped mode of
        // gemm_kernel and keep the 16-column image as well
        const bool p32_only = l.prefix.find(".experts.") == std::string::npos && dec32_serves_every_m(l.w.K, l.w.N);
        TM_TRY(linear_weight_prepare_u4(l.w, (const int32_t*)q.dev, (const half_t*)s.dev, (const half_t*)z.dev, e->stream, p32_only));
        TM_HIP_CHECK(hipStreamSynchronize(e->stream));
        for (Slot* p : {&q, &s, &z}) {
            TM_HIP_CHECK(hipFree(p->dev));
            p->dev = nullptr;
        }
    }
    else if (l.w.type == TM_WEIGHT_FP8) {
        Slot &w = e->slots[l.prefix + ".weight"], &s = e->slots[l.prefix + ".scales"];
        TM_REQUIRE(w.filled && s.filled, "weight not loaded: " + l.prefix);
        const bool gated = l.prefix.size() >= 5 && l.prefix.compare(l.prefix.size() - 5, 5, ".w1w3") == 0;
        TM_TRY(linear_weight_prepare_fp8(l.w, (const uint8_t*)w.dev, (const float*)s.dev, gated, e->stream));
        TM_HIP_CHECK(hipStreamSynchronize(e->stream));
        for (Slot* p : {&w, &s}) {
            TM_HIP_CHECK(hipFree(p->dev));
            p->dev = nullptr;
        }
    }
    else {
        Slot& w = e->slots[l.prefix + ".weight"];
        TM_REQUIRE(w.filled, "weight not loaded: " + l.prefix);
        TM_TRY(linear_weight_prepare_f16(l.w, (const half_t*)w.dev, e->stream));
        TM_HIP_CHECK(hipStreamSynchronize(e->stream));
        TM_HIP_CHECK(hipFree(w.dev));
        w.dev = nullptr;
    }
    return 0;
}

static KvCacheView cache_view(const tm_engine* e, int layer)
{
    KvCacheView v{};
    v.block_ptrs    = e->d_block_ptrs;
    v.cu_block_nums = e->d_cu_block_nums;
    static const bool rect = !getenv("TM_ATTN_RECT") || atoi(getenv("TM_ATTN_RECT")) != 0;  // A/B switch
    v.block_stride  = rect ? e->max_blocks_per_seq : 0;  // cu_block_nums[b] = b * max_blocks_per_seq (create())
    v.layer_offset  = (int64_t)layer * e->layout.layer_size();
    v.layout        = e->layout;
    return v;
}

// fp16 sum of the row-parallel partial outputs over the TP group (comm/nccl/nccl.cu:356-398 calls ncclAllReduce on the
// compute stream; TM_COMM_STREAM=1: on a side stream between a fork / join pair)
static int allreduce_hidden(tm_engine* e, half_t* buf, int M)
{
    if (!e->use_comm) {
        return 0;
    }
    TM_REQUIRE(e->comm != nullptr, "tm_engine_comm_init was not called");
    if (!e->comm_overlap || !e->comm_stream) {
        TM_NCCL_CHECK(ncclAllReduce(buf, buf, (size_t)M * e->hidden, ncclHalf, ncclSum, e->comm, e->stream));
        return 0;
    }
    TM_HIP_CHECK(hipEventRecord(e->ev_fork, e->stream));
    TM_HIP_CHECK(hipStreamWaitEvent(e->comm_stream, e->ev_fork, 0));
    TM_NCCL_CHECK(ncclAllReduce(buf, buf, (size_t)M * e->hidden, ncclHalf, ncclSum, e->comm, e->comm_stream));
    TM_HIP_CHECK(hipEventRecord(e->ev_join, e->comm_stream));
    TM_HIP_CHECK(hipStreamWaitEvent(e->stream, e->ev_join, 0));
    return 0;
}

enum ProfCat { P_EMBED = 0, P_GEMM_QKV, P_KV_STORE, P_ATTN, P_GEMM_O, P_RES_NORM, P_GEMM_GATE_UP, P_GEMM_DOWN, P_LM_HEAD,
               P_SAMPLE, P_ALLREDUCE, P_NUM };

static size_t prof_event(tm_engine* e)
{
    if (e->prof_used == e->prof_pool.size()) {
        hipEvent_t ev;
        (void)hipEventCreate(&ev);
        e->prof_pool.push_back(ev);
    }
    (void)hipEventRecord(e->prof_pool[e->prof_used], e->stream);
    return e->prof_used++;
}

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

static void p2p_tables(tm_engine* e, half_t** data, uint32_t** flags)
{
    for (int r = 0; r < e->cfg.tp; ++r) {
        char* base = (char*)(r == e->cfg.rank ? e->p2p_seg : e->p2p_peer[r]);
        flags[r]   = (uint32_t*)base;
        data[r]    = (half_t*)(base + 256);
    }
}

// d_x = RMSNorm(d_resid += sum over ranks of d_tmp): one fused P2P launch per <= p2p_rows rows on the native communicator
// (any M when there is no RCCL communicator to fall back to), else RCCL all-reduce + the residual-norm kernel
static int reduce_residual_norm(tm_engine* e, int M, const half_t* norm_w)
{
    if (e->p2p_ready && (M <= e->p2p_rows || !e->comm)) {
        half_t*   data[8];
        uint32_t* flags[8];
        p2p_tables(e, data, flags);
        if (M > e->p2p_rows && M <= e->p2p_rows2) {
            // a prefill-sized forward on the native communicator alone: ONE two-shot launch (reduce-scatter, norm on the owned row
            // slice, all-gather) instead of a chain of one-shot launches over row chunks, each reading tp x its bytes
            half_t *in2[8], *out2[8];
            for (int r = 0; r < e->cfg.tp; ++r) {
                in2[r]  = data[r] + 2 * (size_t)e->p2p_rows * e->hidden;
                out2[r] = in2[r] + (size_t)e->p2p_rows2 * e->hidden;
            }
            TM_PROF(P_ALLREDUCE, TM_TRY(launch_p2p_allreduce_norm_2shot(in2, out2, flags, e->cfg.tp, e->cfg.rank, e->p2p_state,
                                                                        (size_t)e->p2p_rows2 * e->hidden, e->d_tmp, e->d_x, e->d_resid, norm_w,
                                                                        e->cfg.model.rms_eps, M, e->hidden, e->stream)));
            return 0;
        }
        const size_t tile = (size_t)e->p2p_rows * e->hidden;
        for (int m0 = 0; m0 < M; m0 += e->p2p_rows) {
            const int    rows = std::min(e->p2p_rows, M - m0);
            const size_t off  = (size_t)m0 * e->hidden;
            TM_PROF(P_ALLREDUCE, TM_TRY(launch_p2p_allreduce_norm(data, flags, e->cfg.tp, e->cfg.rank, e->p2p_state, tile, e->d_tmp + off,
                                                                  e->d_x + off, e->d_resid + off, norm_w, e->cfg.model.rms_eps, rows,
                                                                  e->hidden, e->stream)));
        }
        return 0;
    }
    TM_PROF(P_ALLREDUCE, TM_TRY(allreduce_hidden(e, e->d_tmp, M)));
    TM_PROF(P_RES_NORM, TM_TRY(launch_residual_rmsnorm(e->d_x, e->d_resid, e->d_tmp, nullptr, 0, nullptr, norm_w,
                                                       e->cfg.model.rms_eps, M, e->hidden, e->stream)));
    return 0;
}

static GemmConfig pick_config(tm_engine*, const LinearWeight& w, int M, bool)
{
    return gemm_pick_config(w, M);
}

// row-parallel linear followed by (all-reduce +) residual + RMSNorm
static int linear_residual_norm(tm_engine* e, LinearSlots& l, const half_t* x, int ldx, int M, const half_t* norm_w,
                                int gemm_cat)
{
    GemmConfig cfg = pick_config(e, l.w, M, false);
    const bool can_defer = !e->use_comm && cfg.splits > 1
                           && gemm_workspace_bytes(M, l.w.N, cfg.splits) <= e->gemm_ws_bytes;
    if (gemm_workspace_bytes(M, l.w.N, cfg.splits) > e->gemm_ws_bytes) {
        cfg.splits = 1;
    }
    int slabs = 1;
    TM_PROF(gemm_cat, TM_TRY(launch_linear(l.w, x, ldx, e->d_tmp, e->hidden, M, false, cfg, e->d_gemm_ws, can_defer, &slabs,
                                           e->stream)));
    if (can_defer && slabs > 1) {
        TM_PROF(P_RES_NORM, TM_TRY(launch_residual_rmsnorm(e->d_x, e->d_resid, nullptr, e->d_gemm_ws, slabs, nullptr, norm_w,
                                                           e->cfg.model.rms_eps, M, e->hidden, e->stream)));
        return 0;
    }
    TM_REQUIRE(!can_defer || slabs == 1, "internal: deferred reduce without slabs");
    return reduce_residual_norm(e, M, norm_w);
}

static int linear_plain(tm_engine* e, LinearSlots& l, const half_t* x, int ldx, half_t* y, int ldy, int M, bool gated)
{
    GemmConfig cfg = pick_config(e, l.w, M, gated);
    if (gemm_workspace_bytes(M, l.w.N, cfg.splits) > e->gemm_ws_bytes) {
        cfg.splits = 1;
    }
    return launch_linear(l.w, x, ldx, y, ldy, M, gated, cfg, e->d_gemm_ws, false, nullptr, e->stream);
}

// ---- RMSNorm folded into the decode GEMMs (tp = 1, dense u4 layers, M <= 64; NormFold in tm_kernels.h) ----------------------------
// the tiling of a folded launch: the measured / heuristic pick when its kernel carries the folded epilogue, else the heuristic's
static void fold_tiling(tm_engine* e, const LinearWeight& w, int M, int* shape, int* splits)
{
    dec32_pick(w, M, shape, splits);
    if (!dec32_fold_shape(*shape)) {
        dec32_pick_ex(w, M, shape, splits, false);
    }
    if (!dec32_fold_shape(*shape)) {
        *shape  = 0;
        *splits = 1;
    }
    if (gemm_workspace_bytes(M, w.N, *splits) > e->gemm_ws_bytes) {
        *splits = 1;
    }
}

static bool fold_ok(const tm_engine* e, const Layer& L, int M)
{
    return e->fold_norm != 0 && !L.is_moe && M <= 64 && dec32_supported(L.qkv.w, M) && dec32_supported(L.wo.w, M) && dec32_supported(L.w13.w, M)
           && dec32_supported(L.w2.w, M);
}

// consumer: y = (x . W) * inv[m] (x = r . g of the producing GEMM); ss_tiles == 0: x is already normalised (plain GEMM)
static int linear_fold_consume(tm_engine* e, LinearSlots& l, const half_t* x, int ldx, half_t* y, int ldy, int M, bool gated, int ss_tiles,
                               bool slabs_ok, int* slabs)
{
    int shape, splits;
    fold_tiling(e, l.w, M, &shape, &splits);
    if (!slabs_ok) {
        splits = 1;
    }
    NormFold nf{};
    nf.ss_in    = ss_tiles > 0 ? e->d_ss : nullptr;
    nf.ss_tiles = ss_tiles;
    nf.inv_h    = 1.0f / (float)e->hidden;
    nf.eps      = e->cfg.model.rms_eps;
    int nslab   = 1;
    TM_TRY(launch_linear_dec32(l.w, x, ldx, y, ldy, M, gated, shape, splits, e->d_gemm_ws, &nslab, e->stream, ss_tiles > 0 ? &nf : nullptr));
    if (slabs) {
        *slabs = nslab;
    }
    return 0;
}

// producer: d_resid += x . W ; d_x = d_resid . norm_w (not normalised) ; d_ss = per-tile sums of squares -> *ss_tiles
static int linear_fold_produce(tm_engine* e, LinearSlots& l, const half_t* x, int ldx, int M, const half_t* norm_w, int* ss_tiles)
{
    int shape, splits;
    fold_tiling(e, l.w, M, &shape, &splits);
    NormFold nf{};
    nf.resid   = e->d_resid;
    nf.norm_w  = norm_w;
    nf.ss_out  = e->d_ss;
    nf.tickets = e->d_tickets;
    TM_TRY(launch_linear_dec32(l.w, x, ldx, e->d_x, e->hidden, M, false, shape, splits, e->d_gemm_ws, nullptr, e->stream, &nf));
    *ss_tiles = nf.tiles_out;
    return 0;
}

// One forward over M tokens.  decode: one token per sequence (cu_q = 0..B); prefill: nseq sequences.
// Mixed forward (continuous batching, the reference's unified batch: unified_attention_layer.cc:310-311 puts the decode rows
// first): `md` != nullptr and !decode -> rows [0, md->rows) are the decode tokens of batch slots 0 .. md->rows-1 (their KV
// through the fused decode attention on the engine's own, UNSHIFTED block table and md->k_len), rows [md->rows, M) are the
// prefill tokens of `nseq` sequences described by e->d_cu_q / d_k_len / d_rows (d_rows already counts from row 0 of the
// forward).  Every linear, norm and (MoE) FFN runs ONCE over all M rows -- one weight stream serves both.
struct MixedDecode {
    int             rows;        // decode rows = batch slots
    const int*      k_len;       // [rows] context lengths including this step's token
    const uint64_t* block_ptrs;  // the unshifted block table
    const int*      cu_q;        // [rows + 1] = 0 .. rows (one token per decode row; kv_rope_store of the fp16-KV path)
    const int*      active;      // [rows] 1 = the slot holds a running sequence (logits processors skip the others)
};

static int forward(tm_engine* e, const int* d_ids, int M, int nseq, bool decode, int max_q_len, int max_k_len,
                   int kflat_stride, int slot0, const MixedDecode* md = nullptr)
{
    const tm_model_config& m = e->cfg.model;
    hipStream_t            st = e->stream;
    const int              nd = md ? md->rows : 0;  // leading decode rows of a mixed forward
    TM_REQUIRE(!md || (!decode && nd > 0 && nd < M), "internal: mixed forward");
    half_t* const qkv_p  = e->d_qkv + (size_t)nd * e->qkv_n;            // first prefill row
    half_t* const attn_p = e->d_attn + (size_t)nd * e->q_heads * e->D;
    TM_PROF(P_EMBED, TM_TRY(launch_embedding(e->d_resid, e->tok_embeddings, d_ids, M, e->hidden, m.vocab, st)));
    TM_PROF(P_RES_NORM, TM_TRY(launch_rmsnorm(e->d_x, e->d_resid, e->layers[0].attn_norm, m.rms_eps, M, e->hidden, st)));
    const float scale_log2 = (1.0f / std::sqrt((float)e->D)) * 1.4426950408889634f;
    int         ss_tiles   = 0;  // > 0: d_x holds r . g of a folded producer, d_ss its sums of squares (the next GEMM applies the row factor)
    for (int li = 0; li < m.layers; ++li) {
        Layer& L = e->layers[li];
        KvCacheView cv = cache_view(e, li);
        const bool fold = decode && fold_ok(e, L, M);
        TM_REQUIRE(ss_tiles == 0 || fold, "internal: folded norm without a folded consumer");
        // decode + int8 KV: the attention kernel consumes the qkv GEMM's raw output (fp32 split-K slabs or fp16),
        // applies RoPE and quantises/stores the new K/V itself -> no splitk_reduce, no kv_rope_store launch
        const bool fuse_qkv = decode && e->fuse_qkv;
        int        qkv_slabs = 1;
        if (fold) {
            TM_PROF(P_GEMM_QKV, TM_TRY(linear_fold_consume(e, L.qkv, e->d_x, e->hidden, e->d_qkv, e->qkv_n, M, false, ss_tiles, fuse_qkv, &qkv_slabs)));
            ss_tiles = 0;
            if (!fuse_qkv) {
                TM_PROF(P_KV_STORE, TM_TRY(launch_kv_rope_store(qkv_p, e->q_heads, e->d_cu_q, e->d_k_len, nseq, M - nd, e->d_rope,
                                                               e->rope_max_pos, cv, st)));
            }
        }
        else if (fuse_qkv) {
            GemmConfig cfg = gemm_pick_config(L.qkv.w, M);
            if (gemm_workspace_bytes(M, L.qkv.w.N, cfg.splits) > e->gemm_ws_bytes) {
                cfg.splits = 1;
            }
            TM_PROF(P_GEMM_QKV, TM_TRY(launch_linear(L.qkv.w, e->d_x, e->hidden, e->d_qkv, e->qkv_n, M, false, cfg,
                                                     e->d_gemm_ws, cfg.splits > 1, &qkv_slabs, st)));
        }
        else {
            TM_PROF(P_GEMM_QKV, TM_TRY(linear_plain(e, L.qkv, e->d_x, e->hidden, e->d_qkv, e->qkv_n, M, false)));
            TM_PROF(P_KV_STORE, TM_TRY(launch_kv_rope_store(qkv_p, e->q_heads, e->d_cu_q, e->d_k_len, nseq, M - nd, e->d_rope,
                                                           e->rope_max_pos, cv, st)));
        }
        if (md) {
            // decode rows: (fused prologue: RoPE + K/V quantise-store) + attention on the fp16 projection rows; they share
            // nothing with the prefill rows' path below, so they run beside it on the aux stream (fork here, join behind the
            // prefill attention) -- the reference's decode / prefill split of a unified batch
            hipStream_t dst = st;
            if (e->mixed_two_streams && e->aux_stream) {
                TM_HIP_CHECK(hipEventRecord(e->ev_aux_fork, st));
                TM_HIP_CHECK(hipStreamWaitEvent(e->aux_stream, e->ev_aux_fork, 0));
                dst = e->aux_stream;
            }
            KvCacheView cvd = cv;
            cvd.block_ptrs  = md->block_ptrs;
            DecodeAttnParams p{};
            if (e->fuse_qkv) {
                p.qkv_f16 = e->d_qkv;
                p.qkv_n   = e->qkv_n;
                p.cos_sin = e->d_rope;
                p.max_pos = e->rope_max_pos;
            }
            else {  // fp16 KV (no fused prologue): RoPE + store of the decode rows' K/V first
                TM_PROF(P_KV_STORE, TM_TRY(launch_kv_rope_store(e->d_qkv, e->q_heads, md->cu_q, md->k_len, nd, nd, e->d_rope,
                                                               e->rope_max_pos, cvd, dst)));
            }
            p.q              = e->d_qkv;
            p.q_stride       = e->qkv_n;
            p.out            = e->d_attn;
            p.k_len          = md->k_len;
            p.batch          = nd;
            p.q_heads        = e->q_heads;
            p.scale_log2     = scale_log2;
            p.splits         = e->decode_splits;
            p.partial_o      = e->d_attn_ws;
            p.partial_ml     = e->d_attn_ws + (size_t)nd * e->q_heads * e->decode_splits * e->D;
            p.cache          = cvd;
            TM_PROF(P_ATTN, TM_TRY(launch_decode_attention(p, dst)));
            if (dst != st) {
                TM_HIP_CHECK(hipEventRecord(e->ev_aux_join, dst));
            }
        }
        if (decode) {
            DecodeAttnParams p{};
            if (fuse_qkv) {
                p.qkv_slabs  = qkv_slabs > 1 ? e->d_gemm_ws : nullptr;
                p.qkv_f16    = qkv_slabs > 1 ? nullptr : e->d_qkv;
                p.qkv_splits = qkv_slabs > 1 ? qkv_slabs : 0;
                p.qkv_n      = e->qkv_n;
                p.cos_sin    = e->d_rope;
                p.max_pos    = e->rope_max_pos;
            }
            p.q          = e->d_qkv;
            p.q_stride   = e->qkv_n;
            p.out        = e->d_attn;
            p.k_len      = e->d_k_len;
            p.batch      = nseq;
            p.q_heads    = e->q_heads;
            p.scale_log2 = scale_log2;
            p.splits     = e->decode_splits;
            p.partial_o  = e->d_attn_ws;
            p.partial_ml = e->d_attn_ws + (size_t)nseq * e->q_heads * e->decode_splits * e->D;
            p.cache      = cv;
            TM_PROF(P_ATTN, TM_TRY(launch_decode_attention(p, st)));
        }
        else {
            TM_PROF(P_KV_STORE, TM_TRY(launch_flatten_kv(e->d_kflat, e->d_vflat, 1, e->d_cu_koff, e->d_k_len, nseq, max_k_len,
                                                        kflat_stride, cv, st)));
            PrefillAttnParams p{};
            p.q          = qkv_p;
            p.q_stride   = e->qkv_n;
            p.out        = attn_p;
            p.k          = e->d_kflat;
            p.vt         = e->d_vflat;
            p.k_stride   = kflat_stride;
            p.cu_q_len   = e->d_cu_q;
            p.cu_k_off   = e->d_cu_koff;
            p.k_len      = e->d_k_len;
            p.batch      = nseq;
            p.max_q_len  = max_q_len;
            p.q_heads    = e->q_heads;
            p.kv_heads   = e->kv_heads;
            p.scale_log2 = scale_log2;
            TM_PROF(P_ATTN, TM_TRY(launch_prefill_attention(p, st)));
            if (md && e->mixed_two_streams && e->aux_stream) {
                TM_HIP_CHECK(hipStreamWaitEvent(st, e->ev_aux_join, 0));  // join: wo reads the decode rows' attention output too
            }
        }
        const half_t* next_norm = li + 1 < m.layers ? e->layers[li + 1].attn_norm : e->final_norm;
        if (fold) {
            // wo's epilogue updates the residual stream and hands r . g + sums of squares to w1w3, whose accumulators take the row
            // factor before the gated SiLU; w2 does the same for the next layer's w_qkv (the last layer's w2 feeds the final norm
            // and the fp16 lm_head: reduce-norm launch as before).  5 launches per layer instead of 7.
            if (e->fold_norm & 1) {
                TM_PROF(P_GEMM_O, TM_TRY(linear_fold_produce(e, L.wo, e->d_attn, e->q_heads * e->D, M, L.ffn_norm, &ss_tiles)));
            }
            else {
                TM_TRY(linear_residual_norm(e, L.wo, e->d_attn, e->q_heads * e->D, M, L.ffn_norm, P_GEMM_O));
            }
            TM_PROF(P_GEMM_GATE_UP, TM_TRY(linear_fold_consume(e, L.w13, e->d_x, e->hidden, e->d_act, e->inter, M, true, ss_tiles, false, nullptr)));
            ss_tiles = 0;
            if ((e->fold_norm & 2) && li + 1 < m.layers && fold_ok(e, e->layers[li + 1], M)) {
                TM_PROF(P_GEMM_DOWN, TM_TRY(linear_fold_produce(e, L.w2, e->d_act, e->inter, M, next_norm, &ss_tiles)));
            }
            else {
                TM_TRY(linear_residual_norm(e, L.w2, e->d_act, e->inter, M, next_norm, P_GEMM_DOWN));
            }
            continue;
        }
        TM_TRY(linear_residual_norm(e, L.wo, e->d_attn, e->q_heads * e->D, M, L.ffn_norm, P_GEMM_O));
        if (L.is_moe) {
            // router + grouped expert FFNs + combine -> d_tmp, then (all-reduce +) residual + RMSNorm as for the dense FFN
            TM_PROF(P_GEMM_GATE_UP, TM_TRY(moe_forward(L.moe, e->d_tmp, e->hidden, e->d_x, e->hidden, M, e->d_moe_ws, nullptr,
                                                       nullptr, st)));
            TM_TRY(reduce_residual_norm(e, M, next_norm));
            continue;
        }
        TM_PROF(P_GEMM_GATE_UP, TM_TRY(linear_plain(e, L.w13, e->d_x, e->hidden, e->d_act, e->inter, M, true)));
        TM_TRY(linear_residual_norm(e, L.w2, e->d_act, e->inter, M, next_norm, P_GEMM_DOWN));
    }
    // last-token hidden states -> logits -> next ids, for `n` sequences whose logits / next ids land in the batch slots
    // [slot, slot + n): hx = their hidden rows, ids / cu_q (nullptr: one token per sequence) / ntok = the tokens this forward
    // consumed for them, k_len = their context lengths
    auto head = [&](const half_t* hx, int n, int slot, const int* ids_in, const int* cu_q, int ntok, const int* k_len,
                    const int* active = nullptr) -> int {
        half_t* logits = e->d_logits + (size_t)slot * e->vocab_local;
        int*    ids    = e->d_next_ids + slot;
        TM_PROF(P_LM_HEAD, TM_TRY(linear_plain(e, e->output, hx, e->hidden, logits, e->vocab_local, n, false)));
        if (e->logits_on) {
            // the tokens this forward consumed join the slots' seen masks, then penalty / bans on the (local) logits
            uint32_t* seen = e->d_seen + (size_t)slot * e->seen_words;
            TM_PROF(P_SAMPLE, TM_TRY(launch_seen_update(seen, e->seen_words, ids_in, cu_q, n, ntok, m.vocab, st, active)));
            TM_PROF(P_SAMPLE, TM_TRY(launch_logits_process(logits, n, e->vocab_local, e->vocab_local,
                                                           e->vocab_local < m.vocab ? e->cfg.rank * e->vocab_local : 0, seen,
                                                           e->seen_words, e->d_lp_rep + slot, e->d_lp_ban + slot * kMaxBadIds,
                                                           e->d_lp_end + slot * kMaxEndIds, k_len, e->d_lp_minlen + slot, st)));
        }
        if (e->sampling_on && (!e->use_comm || ((e->comm || e->p2p_ready) && e->d_logits_full))) {
            // parameters are indexed by batch slot, the counter (context length) by the row of this forward
            const half_t* lg = logits;
            int           V  = e->vocab_local;
            if (e->use_comm) {
                // tp > 1: the vocabulary shards are all-gathered into full rows (the reference gathers the logits too,
                // models/language_model.cc:304-333) and EVERY rank draws from the same distribution with the same Philox
                // number -> the same token everywhere, no further exchange
                const size_t cnt = (size_t)n * e->vocab_local;
                if (!e->comm) {
                    // native communicator alone: the shards travel through the P2P segments, as many whole rows per exchange as
                    // one segment buffer holds; rank q's rows land in its [n][vocab / tp] plane of d_logits_gather
                    half_t*   data[8];
                    uint32_t* flags[8];
                    p2p_tables(e, data, flags);
                    const size_t tile = (size_t)e->p2p_rows * e->hidden;  // fp16 elements of one segment buffer
                    const int    per  = (int)std::min<size_t>(n, tile / e->vocab_local);
                    TM_REQUIRE(per >= 1, "native communicator: one logits row does not fit a segment buffer (export more rows)");
                    for (int r0 = 0; r0 < n; r0 += per) {
                        const int rows = std::min(per, n - r0);
                        TM_TRY(launch_p2p_allgather(data, flags, e->cfg.tp, e->cfg.rank, e->p2p_state, tile, logits + (size_t)r0 * e->vocab_local,
                                                    e->d_logits_gather + (size_t)r0 * e->vocab_local, rows * e->vocab_local / 2, st,
                                                    cnt / 2));
                    }
                }
                else if (e->comm_overlap && e->comm_stream) {
                    TM_HIP_CHECK(hipEventRecord(e->ev_fork, st));
                    TM_HIP_CHECK(hipStreamWaitEvent(e->comm_stream, e->ev_fork, 0));
                    TM_NCCL_CHECK(ncclAllGather(logits, e->d_logits_gather, cnt, ncclHalf, e->comm, e->comm_stream));
                    TM_HIP_CHECK(hipEventRecord(e->ev_join, e->comm_stream));
                    TM_HIP_CHECK(hipStreamWaitEvent(st, e->ev_join, 0));
                }
                else {
                    TM_NCCL_CHECK(ncclAllGather(logits, e->d_logits_gather, cnt, ncclHalf, e->comm, st));
                }
                V = e->vocab_local * e->cfg.tp;
                gather_vocab_kernel<<<std::min<size_t>(1024, (cnt * e->cfg.tp / 8 + 255) / 256), 256, 0, st>>>(
                    e->d_logits_full, e->d_logits_gather, n, e->vocab_local, e->cfg.tp);
                TM_HIP_CHECK(hipGetLastError());
                lg = e->d_logits_full;
            }
            TM_PROF(P_SAMPLE, TM_TRY(launch_sample_uniform(e->d_u + slot, e->d_seed + slot, k_len, n, st)));
            TM_PROF(P_SAMPLE, TM_TRY(launch_sample(ids, nullptr, lg, n, V, V, e->d_temp + slot, e->d_topk + slot, e->d_topp + slot,
                                                   e->d_minp + slot, e->d_u + slot, e->d_sample_ws, st)));
        }
        else if (!e->use_comm) {
            TM_PROF(P_SAMPLE, TM_TRY(launch_argmax(ids, nullptr, logits, n, e->vocab_local, e->vocab_local, 0, st)));
        }
        else {
            TM_TRY(launch_argmax(ids, e->d_argmax_val, logits, n, e->vocab_local, e->vocab_local, e->cfg.rank * e->vocab_local, st));
            pack_candidates_kernel<<<(n + 63) / 64, 64, 0, st>>>(e->d_cand, ids, e->d_argmax_val, n);
            TM_HIP_CHECK(hipGetLastError());
            if (e->p2p_ready) {
                half_t*   data[8];
                uint32_t* flags[8];
                p2p_tables(e, data, flags);
                TM_TRY(launch_p2p_allgather(data, flags, e->cfg.tp, e->cfg.rank, e->p2p_state, (size_t)e->p2p_rows * e->hidden, e->d_cand,
                                            e->d_cand_all, n * 2, st));
            }
            // the same communicator is only ever driven from ONE stream (the side stream when it exists)
            else if (e->comm_overlap && e->comm_stream) {
                TM_HIP_CHECK(hipEventRecord(e->ev_fork, st));
                TM_HIP_CHECK(hipStreamWaitEvent(e->comm_stream, e->ev_fork, 0));
                TM_NCCL_CHECK(ncclAllGather(e->d_cand, e->d_cand_all, (size_t)n * 2, ncclFloat, e->comm, e->comm_stream));
                TM_HIP_CHECK(hipEventRecord(e->ev_join, e->comm_stream));
                TM_HIP_CHECK(hipStreamWaitEvent(st, e->ev_join, 0));
            }
            else {
                TM_NCCL_CHECK(ncclAllGather(e->d_cand, e->d_cand_all, (size_t)n * 2, ncclFloat, e->comm, st));
            }
            pick_kernel<<<(n + 63) / 64, 64, 0, st>>>(ids, e->d_cand_all, e->cfg.tp, n);
            TM_HIP_CHECK(hipGetLastError());
        }
        return 0;
    };
    if (decode) {
        return head(e->d_x, nseq, slot0, d_ids, nullptr, M, e->d_k_len);
    }
    if (md) {  // the decode rows first: their slots are 0 .. nd-1; the prefilled slots' entries are overwritten right after
        TM_TRY(head(e->d_x, nd, 0, d_ids, nullptr, nd, md->k_len, md->active));
    }
    TM_TRY(launch_gather_rows(e->d_last, e->d_x, e->d_rows, nseq, e->hidden, st));
    return head(e->d_last, nseq, slot0, d_ids + nd, e->d_cu_q, M - nd, e->d_k_len);
}

// Device-side give-up mark of the native P2P communicator: a bounded wait for a peer expired (p2p_state[3] = the call number a
// peer missed; comm_p2p.hip carries on with wrong numbers instead of hanging).  Read at the host's synchronisation points.
// `async`: enqueue the copy on the engine stream (the caller syncs).
static int device_marks_fetch(tm_engine* e, bool async)
{
    if (e->p2p_state) {
        if (async) {
            TM_HIP_CHECK(hipMemcpyAsync(&e->h_mark, e->p2p_state + 3, 4, hipMemcpyDeviceToHost, e->stream));
        }
        else {
            TM_HIP_CHECK(hipMemcpy(&e->h_mark, e->p2p_state + 3, 4, hipMemcpyDeviceToHost));
        }
    }
    return 0;
}

// A mark means: this rank stopped waiting for a peer in collective call `h_mark` and continued with whatever the peer's buffer
// held.  What was computed from that call on is invalid on THIS rank, and the ranks may no longer agree on the call sequence
// (a late peer is still inside the call this rank left), so the condition is terminal for the communicator: the step fails with
// TM_FAIL, the serve loop ends the requests in flight with kFail, and every later step / submit fails with the same status until the engine is
// recreated (tm_engine_destroy + tm_engine_create on every rank).  Tokens fetched BEFORE the failing step stay readable
// (tm_engine_fetch does not check the mark).  The wait bound is TM_P2P_TIMEOUT_MS (default 30 s: RCCL has no bound at all; a
// one-sided stall -- graph capture, a first-use library load, a descheduled host thread -- must not kill the job).
static int device_marks_check(tm_engine* e)
{
    if (e->h_mark) {
        e->comm_failed = true;
    }
    if (e->comm_failed) {
        set_last_error("native communicator: a peer did not arrive within TM_P2P_TIMEOUT_MS (call " + std::to_string(e->h_mark)
                       + "); results from that call on are invalid and the ranks may have diverged: recreate the engine on every rank");
        return TM_FAIL;
    }
    return 0;
}

// next ids become the current ids and are appended to generated[b][step]
static int commit_tokens(tm_engine* e)
{
    TM_HIP_CHECK(hipMemcpyAsync(e->d_ids, e->d_next_ids, (size_t)e->batch * 4, hipMemcpyDeviceToDevice, e->stream));
    record_kernel<<<1, std::max(64, ((e->batch + 63) / 64) * 64), 0, e->stream>>>(e->d_ids, e->d_generated, e->d_step,
                                                                                  e->batch, e->max_new);
    TM_HIP_CHECK(hipGetLastError());
    return 0;
}

static int decode_step(tm_engine* e)
{
    advance_kernel<<<(e->batch + 63) / 64, 64, 0, e->stream>>>(e->d_k_len, e->batch);
    TM_HIP_CHECK(hipGetLastError());
    TM_TRY(forward(e, e->d_ids, e->batch, e->batch, true, 1, 0, 0, 0));
    return commit_tokens(e);
}

template<class T>
static int dmalloc(T** p, size_t n)
{
    TM_HIP_CHECK(hipMalloc((void**)p, n * sizeof(T)));
    return 0;
}

}  // namespace tmk

extern "C" {

int tm_engine_create(tm_engine** out, const tm_engine_config* c)
{
    TM_REQUIRE(out && c, "null pointer");
    const tm_model_config& m = c->model;
    TM_REQUIRE(m.head_dim == 128, "head_dim must be 128");
    TM_REQUIRE(m.group_size == 128, "AWQ group size must be 128");
    TM_REQUIRE(c->tp >= 1 && c->rank >= 0 && c->rank < c->tp, "0 <= rank < tp");
    TM_REQUIRE(m.q_heads % c->tp == 0 && m.inter % c->tp == 0 && m.vocab % c->tp == 0, "heads/inter/vocab % tp");
    TM_REQUIRE(m.kv_heads % c->tp == 0 || c->tp % m.kv_heads == 0, "kv_heads vs tp");
    TM_REQUIRE(c->quant_policy == 0 || c->quant_policy == 4 || c->quant_policy == 8,
               "quant_policy in {0,4,8} (lmdeploy/messages.py:351-358)");
    TM_REQUIRE(c->cache_block_seq_len == 64, "cache_block_seq_len must be 64");
    TM_REQUIRE(m.weight_type == TM_WEIGHT_U4 || m.weight_type == TM_WEIGHT_F16 || m.weight_type == TM_WEIGHT_FP8, "weight_type");
    TM_REQUIRE(c->max_batch_size >= 1 && c->max_batch_size <= 1024 && c->session_len >= 1,
               "1 <= max_batch_size <= 1024, session_len >= 1");
    TM_HIP_CHECK(hipSetDevice(c->device));

    auto* e        = new tm_engine();
    e->cfg         = *c;
    e->hidden      = m.hidden;
    e->q_heads     = m.q_heads / c->tp;
    e->kv_heads    = std::max(1, m.kv_heads / c->tp);
    e->inter       = m.inter / c->tp;
    e->vocab_local = m.vocab / c->tp;
    {
        const char* fc = getenv("TM_FORCE_COMM");
        e->use_comm    = c->tp > 1 || (fc && atoi(fc));
        const char* ms    = getenv("TM_MIXED_STEP");
        const char* gc    = getenv("TM_GRAPH_COMM");
        e->mixed_steps_on = !(ms && !atoi(ms));
        e->graph_comm     = !(gc && !atoi(gc));
    }
    e->qkv_n       = (e->q_heads + 2 * e->kv_heads) * e->D;
    TM_REQUIRE((e->inter * 1) % 128 == 0 && (e->q_heads * e->D) % 128 == 0 && m.hidden % 128 == 0,
               "K dims must be multiples of 128 after TP sharding");
    TM_REQUIRE(m.moe_experts == 0 || (m.moe_experts <= 64 && m.moe_top_k >= 1 && m.moe_top_k <= 8 && m.moe_top_k <= m.moe_experts
                                      && m.weight_type != TM_WEIGHT_F16),
               "moe: 1 <= top_k <= experts <= 64, top_k <= 8, u4 or fp8 expert weights");
    TM_HIP_CHECK(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));

    e->layers.resize(m.layers);
    for (int i = 0; i < m.layers; ++i) {
        const std::string p = "layers." + std::to_string(i);
        Layer&            L = e->layers[i];
        add_linear(e, L.qkv, p + ".attention.w_qkv", m.hidden, e->qkv_n, m.weight_type, 1);
        add_linear(e, L.wo, p + ".attention.wo", e->q_heads * e->D, m.hidden, m.weight_type, 2);
        if (m.moe_experts > 0) {
            L.is_moe = true;
            L.ex13.resize(m.moe_experts);
            L.ex2.resize(m.moe_experts);
            e->slots[p + ".moe_ffn.gate.weight"].bytes = (int64_t)m.hidden * m.moe_experts * 2;  // fp16 [H][E], replicated
            for (int x = 0; x < m.moe_experts; ++x) {
                const std::string q = p + ".moe_ffn.experts." + std::to_string(x);
                add_linear(e, L.ex13[x], q + ".w1w3", m.hidden, 2 * e->inter, m.weight_type);
                add_linear(e, L.ex2[x], q + ".w2", e->inter, m.hidden, m.weight_type);
            }
        }
        else {
            add_linear(e, L.w13, p + ".feed_forward.w1w3", m.hidden, 2 * e->inter, m.weight_type, 3);
            add_linear(e, L.w2, p + ".feed_forward.w2", e->inter, m.hidden, m.weight_type, 4);
        }
        e->slots[p + ".attention_norm.weight"].bytes = (int64_t)m.hidden * 2;
        e->slots[p + ".ffn_norm.weight"].bytes       = (int64_t)m.hidden * 2;
    }
    e->slots["tok_embeddings.weight"].bytes = (int64_t)m.vocab * m.hidden * 2;  // replicated
    e->slots["norm.weight"].bytes           = (int64_t)m.hidden * 2;
    add_linear(e, e->output, "output", m.hidden, e->vocab_local, TM_WEIGHT_F16, 5);
    *out = e;
    return 0;
}

int tm_comm_unique_id(void* host_out128)
{
    TM_REQUIRE(host_out128, "null pointer");
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId size");
    ncclUniqueId id;
    TM_NCCL_CHECK(ncclGetUniqueId(&id));
    memcpy(host_out128, &id, sizeof(id));
    return 0;
}

int tm_engine_comm_init(tm_engine* e, const void* host_id128)
{
    TM_REQUIRE(e && host_id128, "null pointer");
    if (!e->use_comm) {
        return 0;
    }
    ncclUniqueId id;
    memcpy(&id, host_id128, sizeof(id));
    TM_HIP_CHECK(hipSetDevice(e->cfg.device));
    TM_NCCL_CHECK(ncclCommInitRank(&e->comm, e->cfg.tp, id, e->cfg.rank));
    const char* cs   = getenv("TM_COMM_STREAM");
    // Measured on MI355X (per-rank emulation of Llama-3-70B TP = 8, 160 collectives per step, profiles/r02_comm_stream_arms.txt):
    // engine stream 6.81 ms/step; side stream without the prefetch 6.81 ms (a fork/join inside a hipGraph is free, but costs
    // ~30 us per collective on eager launches); side stream + weight prefetch 7.80 ms (the prefetch kernel costs 6 us and the
    // next GEMM gains nothing from L2 / Infinity-Cache resident weights).  Default: engine stream; the arms stay reachable.
    e->comm_overlap  = cs && atoi(cs);
    if (e->comm_overlap && !e->comm_stream) {
        TM_HIP_CHECK(hipStreamCreateWithFlags(&e->comm_stream, hipStreamNonBlocking));
        TM_HIP_CHECK(hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
        TM_HIP_CHECK(hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming));
    }
    return 0;
}

// Forget the RCCL communicator (a tensor-parallel job whose ranks could not ALL bring RCCL up continues on the native P2P
// communicator alone: a rank that kept its communicator would call ncclAllReduce for large forwards while its peers do not)
int tm_engine_comm_drop_rccl(tm_engine* e)
{
    TM_REQUIRE(e, "null pointer");
    if (e->comm) {
        (void)ncclCommDestroy(e->comm);
        e->comm = nullptr;
    }
    return 0;
}

// Native communicator set-up, two calls around one host-side exchange (any transport: the caller's torch.distributed /
// MPI / files): export allocates this rank's segment and returns its 64-byte IPC handle; import takes the tp handles in rank
// order, maps the peers' segments and switches the row-parallel all-reduces with M <= rows (every M when tm_engine_comm_init
// was not called: no RCCL communicator to fall back to) and the candidate all-gather to comm_p2p.hip.  Replaces the buffer
// registration of comm/cuda_ipc (cuda_ipc_comm.cu Register / the symmetric allocator) for this path.
int tm_engine_comm_native_export(tm_engine* e, int rows, void* handle64)
{
    TM_REQUIRE(e && handle64, "null pointer");
    TM_REQUIRE(e->use_comm, "native communicator: the engine runs with tp = 1");
    TM_REQUIRE(!e->p2p_seg, "native communicator: already exported");
    TM_REQUIRE(rows >= 1 && rows <= 1024, "native communicator: 1 <= rows <= 1024 (one-shot exchange; larger batches stay on RCCL)");
    TM_HIP_CHECK(hipSetDevice(e->cfg.device));
    {   // one workgroup per token row, all resident at once (comm_p2p.hip): clamp to what this device holds
        const int nvec = e->hidden / 8;
        const int t    = std::min(512, (nvec + 63) / 64 * 64);
        const int cap  = p2p_allreduce_capacity(t, (nvec + t - 1) / t == 1);
        TM_REQUIRE(cap >= 1, "native communicator: occupancy query failed");
        rows = std::min(rows, cap);
    }
    // two-shot regions for everything a forward can carry (TM_P2P_2SHOT=0: one-shot row chunks only)
    const char* ts = getenv("TM_P2P_2SHOT");
    const int   rows2 = (ts && !atoi(ts)) ? 0 : std::max(e->cfg.max_prefill_token_num, e->cfg.max_batch_size);
    TM_TRY(tm_p2p_segment_create(tm_p2p_segment_bytes2(rows, rows2, e->hidden), &e->p2p_seg, handle64));
    e->p2p_rows2 = rows2;
    TM_HIP_CHECK(hipMalloc((void**)&e->p2p_state, 4 * sizeof(uint32_t)));
    TM_HIP_CHECK(hipMemset(e->p2p_state, 0, 4 * sizeof(uint32_t)));
    e->p2p_rows = rows;
    return 0;
}

int tm_engine_comm_native_import(tm_engine* e, const void* handles, int count)
{
    TM_REQUIRE(e && handles, "null pointer");
    TM_REQUIRE(e->p2p_seg && !e->p2p_ready, "native communicator: export first, import once");
    TM_REQUIRE(count == e->cfg.tp, "native communicator: one handle per rank");
    TM_HIP_CHECK(hipSetDevice(e->cfg.device));
    for (int r = 0; r < e->cfg.tp; ++r) {
        if (r != e->cfg.rank) {
            TM_TRY(tm_p2p_segment_open((const char*)handles + 64 * (size_t)r, &e->p2p_peer[r]));
        }
    }
    e->p2p_ready = true;
    return 0;
}

int64_t tm_engine_weight_bytes(tm_engine* e, const char* name)
{
    if (!e || !name) {
        return 0;
    }
    auto it = e->slots.find(name);
    if (it == e->slots.end()) {
        set_last_error(std::string("unknown weight slot: ") + name);
        return 0;
    }
    return it->second.bytes;
}

int tm_engine_weight_copy(tm_engine* e, const char* name, const void* host_src, int64_t bytes)
{
    TM_REQUIRE(e && name && host_src, "null pointer");
    TM_REQUIRE(!e->weights_ready, "weights already processed");
    Slot* s = nullptr;
    TM_TRY(ensure_slot(e, name, &s));
    TM_REQUIRE(bytes == s->bytes, std::string("byte size mismatch for ") + name + ": got " + std::to_string(bytes)
                                      + " expected " + std::to_string(s->bytes));
    TM_HIP_CHECK(hipMemcpy(s->dev, host_src, bytes, hipMemcpyHostToDevice));
    s->filled = true;
    return 0;
}

int tm_engine_init_synthetic(tm_engine* e, uint64_t seed)
{
    TM_REQUIRE(e, "null pointer");
    TM_REQUIRE(!e->weights_ready, "weights already processed");
    const tm_model_config& m = e->cfg.model;
    uint64_t               sd = seed * 1000003ull + 17;
    half_t*                master = nullptr;  // fp16 master of the largest linear, reused
    size_t                 master_elems = 0;
    auto                   linear = [&](LinearSlots& l) -> int {
        const size_t n = (size_t)l.w.K * l.w.N;
        if (l.w.type == TM_WEIGHT_F16) {
            Slot* w = nullptr;
            TM_TRY(ensure_slot(e, l.prefix + ".weight", &w));
            TM_TRY(fill_normal(e, w->dev, n, 0.f, 0.1f / std::sqrt((float)l.w.K), ++sd));
            w->filled = true;
            return 0;
        }
        if (l.w.type == TM_WEIGHT_FP8) {
            // random e4m3 codes (no NaN), one constant block scale: E|code value| ~ 30 -> weights ~ 0.1 / sqrt(K)
            Slot *w = nullptr, *sc = nullptr;
            TM_TRY(ensure_slot(e, l.prefix + ".weight", &w));
            TM_TRY(ensure_slot(e, l.prefix + ".scales", &sc));
            fill_fp8_kernel<<<(n + 255) / 256, 256, 0, e->stream>>>((uint8_t*)w->dev, n, ++sd);
            TM_HIP_CHECK(hipGetLastError());
            const size_t nsc = (size_t)sc->bytes / 4;
            fill_const_f32_kernel<<<(nsc + 255) / 256, 256, 0, e->stream>>>((float*)sc->dev, nsc, 0.0027f / std::sqrt((float)l.w.K));
            TM_HIP_CHECK(hipGetLastError());
            w->filled = sc->filled = true;
            TM_TRY(prepare_linear(e, l));
            return 0;
        }
        if (n > master_elems) {
            if (master) {
                TM_HIP_CHECK(hipStreamSynchronize(e->stream));
                TM_HIP_CHECK(hipFree(master));
            }
            TM_HIP_CHECK(hipMalloc((void**)&master, n * 2));
            master_elems = n;
        }
        TM_TRY(fill_normal(e, master, n, 0.f, 0.1f / std::sqrt((float)l.w.K), ++sd));
        Slot *q = nullptr, *s = nullptr, *z = nullptr;
        TM_TRY(ensure_slot(e, l.prefix + ".qweight", &q));
        TM_TRY(ensure_slot(e, l.prefix + ".scales", &s));
        TM_TRY(ensure_slot(e, l.prefix + ".zeros", &z));
        TM_TRY(launch_quantize_groupwise_u4((int32_t*)q->dev, (half_t*)s->dev, (half_t*)z->dev, nullptr, master, l.w.K,
                                            l.w.N, l.w.group, e->stream));
        q->filled = s->filled = z->filled = true;
        // repack right away so that staging never holds more than one linear
        TM_TRY(prepare_linear(e, l));
        return 0;
    };
    auto vec = [&](const std::string& name, size_t n, float mean, float stddev) -> int {
        Slot* s = nullptr;
        TM_TRY(ensure_slot(e, name, &s));
        TM_TRY(fill_normal(e, s->dev, n, mean, stddev, ++sd));
        s->filled = true;
        return 0;
    };
    for (int i = 0; i < m.layers; ++i) {
        const std::string p = "layers." + std::to_string(i);
        Layer&            L = e->layers[i];
        TM_TRY(linear(L.qkv));
        TM_TRY(linear(L.wo));
        if (L.is_moe) {
            TM_TRY(vec(p + ".moe_ffn.gate.weight", (size_t)m.hidden * m.moe_experts, 0.f, 0.05f));
            for (int x = 0; x < m.moe_experts; ++x) {
                TM_TRY(linear(L.ex13[x]));
                TM_TRY(linear(L.ex2[x]));
            }
        }
        else {
            TM_TRY(linear(L.w13));
            TM_TRY(linear(L.w2));
        }
        TM_TRY(vec(p + ".attention_norm.weight", m.hidden, 1.f, 0.02f));
        TM_TRY(vec(p + ".ffn_norm.weight", m.hidden, 1.f, 0.02f));
    }
    TM_TRY(vec("tok_embeddings.weight", (size_t)m.vocab * m.hidden, 0.f, 0.02f));
    TM_TRY(vec("norm.weight", m.hidden, 1.f, 0.02f));
    TM_TRY(linear(e->output));
    TM_HIP_CHECK(hipStreamSynchronize(e->stream));
    if (master) {
        TM_HIP_CHECK(hipFree(master));
    }
    return 0;
}

int tm_engine_process_weights(tm_engine* e)
{
    TM_REQUIRE(e, "null pointer");
    if (e->weights_ready) {
        return 0;
    }
    TM_HIP_CHECK(hipSetDevice(e->cfg.device));
    const tm_model_config& m = e->cfg.model;
    auto norm = [&](const std::string& name, half_t** dst) -> int {
        Slot& s = e->slots[name];
        TM_REQUIRE(s.filled, "weight not loaded: " + name);
        *dst = (half_t*)s.dev;  // used in place
        return 0;
    };
    for (int i = 0; i < m.layers; ++i) {
        const std::string p = "layers." + std::to_string(i);
        Layer&            L = e->layers[i];
        for (LinearSlots* l : {&L.qkv, &L.wo}) {
            if (!l->w.packed && !l->w.packed32) {
                TM_TRY(prepare_linear(e, *l));
            }
        }
        if (L.is_moe) {
            L.moe.hidden       = m.hidden;
            L.moe.inter        = e->inter;
            L.moe.experts      = m.moe_experts;
            L.moe.top_k        = m.moe_top_k;
            L.moe.norm_topk    = m.moe_norm_topk != 0;
            L.moe.routed_scale = m.moe_routed_scale > 0.f ? m.moe_routed_scale : 1.f;
            L.moe.w13.resize(m.moe_experts);
            L.moe.w2.resize(m.moe_experts);
            for (int x = 0; x < m.moe_experts; ++x) {
                for (LinearSlots* l : {&L.ex13[x], &L.ex2[x]}) {
                    if (!l->w.packed && !l->w.packed32) {
                        TM_TRY(prepare_linear(e, *l));
                    }
                }
                L.moe.w13[x] = L.ex13[x].w;  // the MoE block owns the packed weights from here on
                L.moe.w2[x]  = L.ex2[x].w;
                L.ex13[x].w.packed = nullptr, L.ex13[x].w.sz = nullptr, L.ex13[x].w.packed32 = nullptr, L.ex13[x].w.packed8 = nullptr;
                L.ex2[x].w.packed = nullptr, L.ex2[x].w.sz = nullptr, L.ex2[x].w.packed32 = nullptr, L.ex2[x].w.packed8 = nullptr;
            }
            Slot& g = e->slots[p + ".moe_ffn.gate.weight"];
            TM_REQUIRE(g.filled, "weight not loaded: " + p + ".moe_ffn.gate.weight");
            L.moe.gate = (half_t*)g.dev;  // used in place (freed by moe_free)
            g.dev      = nullptr;
            TM_TRY(moe_prepare(L.moe, e->stream));
        }
        else {
            for (LinearSlots* l : {&L.w13, &L.w2}) {
                if (!l->w.packed && !l->w.packed32) {
                    TM_TRY(prepare_linear(e, *l));
                }
            }
        }
        TM_TRY(norm(p + ".attention_norm.weight", &L.attn_norm));
        TM_TRY(norm(p + ".ffn_norm.weight", &L.ffn_norm));
    }
    TM_TRY(norm("tok_embeddings.weight", &e->tok_embeddings));
    TM_TRY(norm("norm.weight", &e->final_norm));
    if (!e->output.w.packed) {
        TM_TRY(prepare_linear(e, e->output));
    }
    e->weights_ready = true;
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// Measured GEMM dispatch (reference: the warm-up tuning of turbomind.cc:363-487 -> gemm::Gemm::Run's DispatchCache,
// kernels/gemm/gemm.cu:92-224; TM_GEMM_TUNE / TM_GEMM_EXPORT / TM_GEMM_IMPORT).  For the decode batch M <= 256 every
// dense linear role of the model (w_qkv, wo, w1w3, w2) is timed with every (workgroup shape, split-K) candidate of the
// decode kernel as ONE hipGraph over the model's own layers -- distinct weights per node, more bytes than the Infinity
// Cache holds, as in a decode step -- and each node is followed by the kernel that consumes its result (the fused split-K
// reduce + residual + RMSNorm for wo / w2, the slab reduce standing in for the attention prologue for w_qkv): a split-K
// GEMM looks cheap in isolation and pays at the kernel boundary (profiles/r02_gemm_boundary_gap.txt).  The winner enters
// the (K, N, M) table that dec32_pick consults first; it replaces the heuristic only when it is >= 7 % faster.
// ------------------------------------------------------------------------------------------------------------------
static int tune_decode_gemms(tm_engine* e, int M, bool verbose)
{
    TM_REQUIRE(M >= 1 && M <= e->max_tokens && M == dec32_m_bucket(M),
               "tuning: 1 <= M <= 256 (a decode batch) or a prefill size class 512, 1024, ... 8192, within max_prefill_token_num");
    half_t* const norm_out = M <= e->cfg.max_batch_size ? e->d_last : e->d_x;  // (d_x is the INPUT of w_qkv / w1w3 only)
    hipStream_t st = e->stream;
    struct Role {
        const char*   name;
        int           which;  // 0 qkv, 1 wo, 2 w13, 3 w2
        const half_t* x;
        int           ldx;
        half_t*       y;
        int           ldy;
        bool          gated;
    };
    const Role roles[4] = {{"w_qkv", 0, e->d_x, e->hidden, e->d_qkv, e->qkv_n, false},
                           {"wo", 1, e->d_attn, e->q_heads * e->D, e->d_tmp, e->hidden, false},
                           {"w1w3", 2, e->d_x, e->hidden, e->d_act, e->inter, true},
                           {"w2", 3, e->d_act, e->inter, e->d_tmp, e->hidden, false}};
    hipEvent_t e0, e1;
    TM_HIP_CHECK(hipEventCreate(&e0));
    TM_HIP_CHECK(hipEventCreate(&e1));
    // stand-in activations (the buffers are scratch before the first forward): pseudo-random values of the magnitude a
    // normed hidden state / an attention output / a gated activation has -- NOT zeros, which let the power-limited matrix pipe
    // run ~45 % faster than on real data and mis-rank the compute-bound candidates (launch_fill_uniform_f16)
    TM_TRY(launch_fill_uniform_f16(e->d_x, (size_t)M * e->hidden, 1.7f, 1u, st));
    TM_TRY(launch_fill_uniform_f16(e->d_attn, (size_t)M * e->q_heads * e->D, 0.5f, 2u, st));
    TM_TRY(launch_fill_uniform_f16(e->d_act, (size_t)M * e->inter, 0.5f, 3u, st));
    TM_HIP_CHECK(hipMemsetAsync(e->d_resid, 0, (size_t)M * e->hidden * 2, st));
    TM_HIP_CHECK(hipMemsetAsync(e->d_ss, 0x3f, (size_t)(e->hidden / 64) * 64 * sizeof(float), st));  // finite stand-in sums of squares (0.747)
    int rc = 0;
    for (const Role& r : roles) {
        std::vector<const LinearWeight*> ws;
        for (Layer& L : e->layers) {
            if (r.which >= 2 && L.is_moe) {
                continue;
            }
            const LinearWeight* w = r.which == 0 ? &L.qkv.w : r.which == 1 ? &L.wo.w : r.which == 2 ? &L.w13.w : &L.w2.w;
            if (dec32_supported(*w, M)) {
                ws.push_back(w);
            }
        }
        if (ws.size() < 2) {
            continue;
        }
        const LinearWeight& w0 = *ws[0];
        int                 hs, hp;
        if (dec32_table_get(w0.K, w0.N, M, &hs, &hp, r.which + 1)) {
            continue;  // imported / tuned already
        }
        dec32_pick_ex(w0, M, &hs, &hp, false);
        int       cand[96][2];
        int       nc = dec32_candidates(w0, M, cand, 95);
        bool      has = false;
        for (int i = 0; i < nc; ++i) {
            has = has || (cand[i][0] == hs && cand[i][1] == hp);
        }
        if (!has) {
            cand[nc][0] = hs;
            cand[nc][1] = hp;
            ++nc;
        }
        float best = 1e30f, heur = 1e30f;
        int   bs = hs, bp = hp;
        for (int i = 0; i < nc && !rc; ++i) {
            GemmConfig cfg{};
            cfg.nt        = 2;
            cfg.waves     = 16;
            cfg.kphases   = 1;
            cfg.d32_shape = cand[i][0];
            cfg.splits    = cand[i][1];
            if (gemm_workspace_bytes(M, w0.N, cfg.splits) > e->gemm_ws_bytes) {
                continue;
            }
            const bool norm_consumer = (r.which == 1 || r.which == 3) && !e->use_comm;
            // decode batches of an engine that folds the RMSNorm into the GEMMs (linear_fold_*): the candidates are timed as they
            // will run -- wo / w2 with the residual / sums-of-squares epilogue (and the in-launch slab merge) instead of the
            // reduce-norm launch, w_qkv / w1w3 with the row factor from d_ss -- and only tiles whose kernel carries that code
            // (which == 1 / 2: the wo -> w1w3 pair, bit 0; which == 3 / 0: the w2 -> w_qkv pair, bit 1)
            const bool folded = M <= 64 && !e->layers[0].is_moe && (e->fold_norm & ((r.which == 1 || r.which == 2) ? 1 : 2)) != 0;
            const bool slabs_ok = norm_consumer || (r.which == 0 && e->fuse_qkv);  // folded: who can take fp32 slabs
            if (folded && (!dec32_fold_shape(cand[i][0]) || (cfg.splits > 1 && !slabs_ok))) {
                continue;
            }
            auto chain = [&]() -> int {
                if (folded) {
                    const int tiles = e->hidden / 64;
                    for (const LinearWeight* w : ws) {
                        NormFold nf{};
                        if (norm_consumer) {
                            nf.resid = e->d_resid, nf.norm_w = e->final_norm, nf.ss_out = e->d_ss, nf.tickets = e->d_tickets;
                        }
                        else {
                            nf.ss_in = e->d_ss, nf.ss_tiles = tiles, nf.inv_h = 1.0f / (float)e->hidden, nf.eps = e->cfg.model.rms_eps;
                        }
                        TM_TRY(launch_linear_dec32(*w, r.x, r.ldx, norm_consumer ? norm_out : r.y, norm_consumer ? e->hidden : r.ldy, M, r.gated,
                                                   cfg.d32_shape, cfg.splits, e->d_gemm_ws, nullptr, st, &nf));
                    }
                    return 0;
                }
                for (const LinearWeight* w : ws) {
                    int slabs = 1;
                    TM_TRY(launch_linear(*w, r.x, r.ldx, r.y, r.ldy, M, r.gated, cfg, e->d_gemm_ws, norm_consumer && cfg.splits > 1, &slabs, st));
                    if (norm_consumer) {
                        TM_TRY(launch_residual_rmsnorm(norm_out, e->d_resid, slabs > 1 ? nullptr : e->d_tmp, slabs > 1 ? e->d_gemm_ws : nullptr,
                                                       slabs, nullptr, e->final_norm, e->cfg.model.rms_eps, M, e->hidden, st));
                    }
                }
                return 0;
            };
            if (norm_consumer) {  // the chain accumulates into the residual stream: every candidate starts from zero
                TM_HIP_CHECK(hipMemsetAsync(e->d_resid, 0, (size_t)M * e->hidden * 2, st));
            }
            if ((rc = chain())) {  // eager once: lazy module loading, function attributes
                break;
            }
            hipGraph_t     g  = nullptr;
            hipGraphExec_t ge = nullptr;
            float          us = 1e30f;
            // every error path below ends the capture and destroys what was created: a failed candidate must not leave the
            // engine stream capturing (every later launch would fail) nor leak the graph
            auto timed = [&]() -> int {
                TM_HIP_CHECK(hipStreamSynchronize(st));
                TM_HIP_CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
                const int        crc = chain();
                const hipError_t erc = hipStreamEndCapture(st, &g);
                if (crc) {
                    return crc;
                }
                TM_HIP_CHECK(erc);
                TM_HIP_CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
                for (int rep = 0; rep < 6; ++rep) {
                    TM_HIP_CHECK(hipEventRecord(e0, st));
                    TM_HIP_CHECK(hipGraphLaunch(ge, st));
                    TM_HIP_CHECK(hipEventRecord(e1, st));
                    TM_HIP_CHECK(hipEventSynchronize(e1));
                    float ms = 0.f;
                    TM_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
                    if (rep > 0) {
                        us = std::min(us, ms * 1000.f / (float)ws.size());
                    }
                }
                return 0;
            };
            rc = timed();
            if (ge) {
                (void)hipGraphExecDestroy(ge);
            }
            if (g) {
                (void)hipGraphDestroy(g);
            }
            if (rc) {
                break;
            }
            if (cand[i][0] == hs && cand[i][1] == hp) {
                heur = us;
            }
            if (us < best) {
                best = us;
                bs   = cand[i][0];
                bp   = cand[i][1];
            }
            if (verbose) {
                fprintf(stderr, "[tm tune] %-5s K=%d N=%d M=%d shape %d splits %2d: %7.2f us / layer%s\n", r.name, w0.K, w0.N, M, cand[i][0],
                        cand[i][1], us, (cand[i][0] == hs && cand[i][1] == hp) ? "  <- heuristic" : "");
            }
        }
        if (rc) {
            break;
        }
        // keep the heuristic unless the measurement clearly beats it.  7 %: run-to-run spread of a candidate is +-3 % (the same tiling
        // measured 18.35 / 18.72 / 19.13 us in three starts on one box), and the chain here is not the model (the same kernel back
        // to back, its activations hot in L2): with a 3 % bar w2 of Llama-3-8B once flipped from (3, 4) to (6, 2) -- 12 % slower in
        // the model (gpurun_out/profile_r03c: 0.61 vs 0.54 ms per step)
        if (!(best < 0.93f * heur)) {
            bs = hs;
            bp = hp;
        }
        dec32_table_set(w0.K, w0.N, M, bs, bp, r.which + 1);
        if (verbose) {
            fprintf(stderr, "[tm tune] %-5s K=%d N=%d M=%d -> shape %d splits %d (%.2f us; heuristic shape %d splits %d %.2f us)\n", r.name, w0.K,
                    w0.N, M, bs, bp, best, hs, hp, heur);
        }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    TM_HIP_CHECK(hipMemsetAsync(e->d_resid, 0, (size_t)M * e->hidden * 2, st));
    TM_HIP_CHECK(hipStreamSynchronize(st));
    return rc;
}

// ------------------------------------------------------------------------------------------------------------------
// The same for everything that is not a P32 kernel: the dense linears gemm_kernel serves (e4m3 weight-only
// w_qkv / wo / w1w3 / w2, the fp16 lm_head) and the grouped expert GEMMs (row-tile height; u4 through gemm_kernel<GRP>, e4m3 on
// the fp8 matrix cores).  Same method: one hipGraph per candidate over the model's own weights, min of 5 replays, a measured
// winner replaces the heuristic only when it is >= 7 % faster.  Results enter gen_table (tm_kernels.h) and travel in the same
// export / import file as the P32 entries.
// ------------------------------------------------------------------------------------------------------------------
static int time_graph_us(tm_engine* e, const std::function<int()>& chain, float* us_out)
{
    hipStream_t    st = e->stream;
    hipGraph_t     g  = nullptr;
    hipGraphExec_t ge = nullptr;
    hipEvent_t     e0 = nullptr, e1 = nullptr;
    float          us = 1e30f;
    auto           run = [&]() -> int {
        TM_TRY(chain());  // eager once: lazy module loading, function attributes
        TM_HIP_CHECK(hipEventCreate(&e0));
        TM_HIP_CHECK(hipEventCreate(&e1));
        TM_HIP_CHECK(hipStreamSynchronize(st));
        TM_HIP_CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        const int        crc = chain();
        const hipError_t erc = hipStreamEndCapture(st, &g);  // always ends the capture: a failed candidate must not leave the stream capturing
        if (crc) {
            return crc;
        }
        TM_HIP_CHECK(erc);
        TM_HIP_CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int rep = 0; rep < 6; ++rep) {
            TM_HIP_CHECK(hipEventRecord(e0, st));
            TM_HIP_CHECK(hipGraphLaunch(ge, st));
            TM_HIP_CHECK(hipEventRecord(e1, st));
            TM_HIP_CHECK(hipEventSynchronize(e1));
            float ms = 0.f;
            TM_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0) {
                us = std::min(us, ms * 1000.f);
            }
        }
        return 0;
    };
    const int rc = run();
    if (ge) {
        (void)hipGraphExecDestroy(ge);
    }
    if (g) {
        (void)hipGraphDestroy(g);
    }
    if (e0) {
        (void)hipEventDestroy(e0);
    }
    if (e1) {
        (void)hipEventDestroy(e1);
    }
    *us_out = us;
    return rc;
}

static int tune_aux_gemms(tm_engine* e, int M, bool verbose)
{
    hipStream_t st = e->stream;
    const int   Mb = dec32_m_bucket(M);
    int         tv[4];
    // ---- dense linears of the general kernel: per role over the layers (with the consumer of wo / w2), then the lm_head ----
    struct Role {
        const char*   name;
        int           which;  // 0 qkv, 1 wo, 2 w13, 3 w2, 4 lm_head
        const half_t* x;
        int           ldx;
        half_t*       y;
        int           ldy;
        bool          gated;
    };
    half_t* const norm_out = M <= e->cfg.max_batch_size ? e->d_last : e->d_x;
    const Role roles[5] = {{"w_qkv", 0, e->d_x, e->hidden, e->d_qkv, e->qkv_n, false},
                           {"wo", 1, e->d_attn, e->q_heads * e->D, e->d_tmp, e->hidden, false},
                           {"w1w3", 2, e->d_x, e->hidden, e->d_act, e->inter, true},
                           {"w2", 3, e->d_act, e->inter, e->d_tmp, e->hidden, false},
                           {"lm_head", 4, e->d_last, e->hidden, e->d_logits, e->vocab_local, false}};
    TM_TRY(launch_fill_uniform_f16(e->d_x, (size_t)M * e->hidden, 1.7f, 1u, st));
    TM_TRY(launch_fill_uniform_f16(e->d_attn, (size_t)M * e->q_heads * e->D, 0.5f, 2u, st));
    TM_TRY(launch_fill_uniform_f16(e->d_act, (size_t)M * e->inter, 0.5f, 3u, st));
    for (const Role& r : roles) {
        std::vector<const LinearWeight*> ws;
        if (r.which == 4) {
            if (M <= e->cfg.max_batch_size) {  // logits exist for batch-slot rows only
                TM_TRY(launch_fill_uniform_f16(e->d_last, (size_t)M * e->hidden, 1.7f, 4u, st));
                ws.push_back(&e->output.w);
            }
        }
        else {
            for (Layer& L : e->layers) {
                if (r.which >= 2 && L.is_moe) {
                    continue;
                }
                const LinearWeight* w = r.which == 0 ? &L.qkv.w : r.which == 1 ? &L.wo.w : r.which == 2 ? &L.w13.w : &L.w2.w;
                if (!dec32_supported(*w, M)) {
                    ws.push_back(w);
                }
            }
        }
        if (ws.empty() || (r.which < 4 && ws.size() < 2)) {
            continue;
        }
        const LinearWeight& w0 = *ws[0];
        if (gen_table_get(kGenDense + w0.type, w0.role, w0.K, w0.N, Mb, tv)) {
            continue;  // imported / tuned already
        }
        GemmConfig cand[16];
        int        nc = gen_dense_candidates(w0, M, e->gemm_ws_bytes, cand, 15);
        if (nc == 0) {
            continue;
        }
        const GemmConfig heur = gemm_pick_config_general(w0, M);
        auto same = [](const GemmConfig& a, const GemmConfig& b) { return a.nt == b.nt && a.splits == b.splits; };
        bool has = false;
        for (int i = 0; i < nc; ++i) {
            has = has || same(cand[i], heur);
        }
        if (!has) {
            cand[nc++] = heur;
        }
        const bool norm_consumer = (r.which == 1 || r.which == 3) && !e->use_comm;
        float      best = 1e30f, t_heur = 1e30f;
        GemmConfig bc = heur;
        for (int i = 0; i < nc; ++i) {
            const GemmConfig cfg = cand[i];
            if (gemm_workspace_bytes(M, w0.N, cfg.splits) > e->gemm_ws_bytes) {
                continue;
            }
            auto chain = [&]() -> int {
                for (const LinearWeight* w : ws) {
                    int slabs = 1;
                    TM_TRY(launch_linear(*w, r.x, r.ldx, r.y, r.ldy, M, r.gated, cfg, e->d_gemm_ws, norm_consumer && cfg.splits > 1, &slabs, st));
                    if (norm_consumer) {
                        TM_TRY(launch_residual_rmsnorm(norm_out, e->d_resid, slabs > 1 ? nullptr : e->d_tmp, slabs > 1 ? e->d_gemm_ws : nullptr,
                                                       slabs, nullptr, e->final_norm, e->cfg.model.rms_eps, M, e->hidden, st));
                    }
                    else if (r.which == 4) {  // the head's consumer
                        TM_TRY(launch_argmax(e->d_next_ids, nullptr, e->d_logits, M, e->vocab_local, e->vocab_local, 0, st));
                    }
                }
                return 0;
            };
            if (norm_consumer) {
                TM_HIP_CHECK(hipMemsetAsync(e->d_resid, 0, (size_t)M * e->hidden * 2, st));
            }
            float us = 1e30f;
            TM_TRY(time_graph_us(e, chain, &us));
            us /= (float)ws.size();
            if (same(cfg, heur)) {
                t_heur = us;
            }
            if (us < best) {
                best = us;
                bc   = cfg;
            }
            if (verbose) {
                fprintf(stderr, "[tm tune] %-7s K=%d N=%d M=%d general nt %d splits %d: %8.2f us%s\n", r.name, w0.K, w0.N, M, cfg.nt, cfg.splits, us,
                        same(cfg, heur) ? "  <- heuristic" : "");
            }
        }
        if (!(best < 0.93f * t_heur)) {
            bc = heur;
        }
        const int v[4] = {bc.nt, bc.splits, bc.waves, bc.kphases < 1 ? 1 : bc.kphases};
        gen_table_set(kGenDense + w0.type, w0.role, w0.K, w0.N, Mb, v);
        if (verbose) {
            fprintf(stderr, "[tm tune] %-7s K=%d N=%d M=%d -> general nt %d splits %d (%.2f us; heuristic %.2f us)\n", r.name, w0.K, w0.N, M, bc.nt,
                    bc.splits, best, t_heur);
        }
    }
    // ---- grouped expert GEMMs: the row-tile height, first of w1w3 (w2 on its heuristic), then of w2 ----
    std::vector<Layer*> moe;
    for (Layer& L : e->layers) {
        if (L.is_moe) {
            moe.push_back(&L);
        }
    }
    if (moe.size() >= 2 && e->d_moe_ws) {
        auto chain = [&]() -> int {
            for (Layer* L : moe) {
                TM_TRY(moe_forward(L->moe, e->d_tmp, e->hidden, e->d_x, e->hidden, M, e->d_moe_ws, nullptr, nullptr, st));
            }
            return 0;
        };
        for (int which = 0; which < 2; ++which) {
            const LinearWeight& proto = which == 0 ? moe[0]->moe.w13[0] : moe[0]->moe.w2[0];
            const int           kind  = kGenGrouped + proto.type;
            if (gen_table_get(kind, 0, proto.K, proto.N, Mb, tv)) {
                continue;
            }
            int       rows[4];
            const int nc = gen_grouped_candidates(proto, M, rows, 4);
            if (nc < 2) {
                continue;
            }
            float t_heur = 1e30f, best = 1e30f;
            int   br = 0;
            TM_TRY(time_graph_us(e, chain, &t_heur));  // no entry: the launchers' own rule
            for (int i = 0; i < nc; ++i) {
                gen_grouped_rows_override(rows[i]);  // this thread's launches only: nothing transient enters the shared table
                float     us = 1e30f;
                const int rc = time_graph_us(e, chain, &us);
                gen_grouped_rows_override(0);
                if (rc) {
                    return rc;
                }
                if (verbose) {
                    fprintf(stderr, "[tm tune] experts %s K=%d N=%d tokens=%d rows/tile %2d: %9.2f us per MoE FFN (heuristic %.2f)\n",
                            which == 0 ? "w1w3" : "w2", proto.K, proto.N, M, rows[i], us / (float)moe.size(), t_heur / (float)moe.size());
                }
                if (us < best) {
                    best = us;
                    br   = rows[i];
                }
            }
            if (br && best < 0.93f * t_heur) {  // else: no entry, the heuristic stays
                const int v[4] = {br, 0, 0, 0};
                gen_table_set(kind, 0, proto.K, proto.N, Mb, v);
            }
            if (verbose) {
                fprintf(stderr, "[tm tune] experts %s K=%d N=%d tokens=%d -> %s (best %.2f us, heuristic %.2f us per MoE FFN)\n", which == 0 ? "w1w3" : "w2",
                        proto.K, proto.N, M, (br && best < 0.93f * t_heur) ? "measured tile" : "heuristic", best / (float)moe.size(),
                        t_heur / (float)moe.size());
            }
        }
    }
    TM_HIP_CHECK(hipMemsetAsync(e->d_resid, 0, (size_t)M * e->hidden * 2, st));
    TM_HIP_CHECK(hipStreamSynchronize(st));
    return 0;
}

int tm_engine_tune_gemm(tm_engine* e, int M, const char* export_path)
{
    TM_REQUIRE(e && e->started, "engine not started");
    TM_REQUIRE(e->batch == 0 && !e->sched, "tune before the first batch is admitted");
    TM_HIP_CHECK(hipSetDevice(e->cfg.device));
    const char* v = getenv("TM_GEMM_TUNE_VERBOSE");
    TM_TRY(tune_decode_gemms(e, M, v && atoi(v)));
    TM_TRY(tune_aux_gemms(e, M, v && atoi(v)));
    if (export_path && *export_path) {
        return dec32_table_export(export_path);
    }
    return 0;
}

int tm_gemm_import(const char* path)
{
    TM_REQUIRE(path && *path, "path");
    return dec32_table_import(path);
}

int tm_gemm_export(const char* path)
{
    TM_REQUIRE(path && *path, "path");
    return dec32_table_export(path);
}

int tm_engine_start(tm_engine* e)
{
    TM_REQUIRE(e, "null pointer");
    TM_REQUIRE(e->weights_ready, "process_weights first");
    if (e->started) {
        return 0;
    }
    TM_HIP_CHECK(hipSetDevice(e->cfg.device));
    const tm_engine_config& c = e->cfg;
    const tm_model_config&  m = c.model;
    const int bits = c.quant_policy == 0 ? 16 : c.quant_policy;
    e->layout      = KvLayout{e->kv_heads, e->D, c.cache_block_seq_len, bits};
    e->block_bytes = (int64_t)m.layers * e->layout.layer_size();
    e->max_blocks_per_seq = (c.session_len + 63) / 64;

    const int B   = c.max_batch_size;
    e->max_tokens = std::max(B, std::max(64, c.max_prefill_token_num));
    const size_t T = e->max_tokens;
    TM_TRY(dmalloc(&e->d_resid, T * e->hidden));
    TM_TRY(dmalloc(&e->d_x, T * e->hidden));
    TM_TRY(dmalloc(&e->d_tmp, T * e->hidden));
    TM_TRY(dmalloc(&e->d_qkv, T * e->qkv_n));
    TM_TRY(dmalloc(&e->d_attn, T * e->q_heads * e->D));
    TM_TRY(dmalloc(&e->d_act, T * e->inter));
    TM_TRY(dmalloc(&e->d_logits, (size_t)B * e->vocab_local));
    TM_TRY(dmalloc(&e->d_last, (size_t)B * e->hidden));
    // split-K workspace: decode-sized problems only (M <= 64 rows x widest N x 16 slabs)
    e->gemm_ws_bytes = (size_t)16 * 64 * std::max(std::max(e->qkv_n, 2 * e->inter), e->hidden) * sizeof(float);
    TM_HIP_CHECK(hipMalloc((void**)&e->d_gemm_ws, e->gemm_ws_bytes));
    {
        const size_t tiles = (size_t)(e->hidden + 63) / 64;
        TM_TRY(dmalloc(&e->d_ss, tiles * 64));
        TM_TRY(dmalloc(&e->d_tickets, tiles * 2));
        TM_HIP_CHECK(hipMemset(e->d_tickets, 0, tiles * 2 * sizeof(unsigned)));
        const char* fold = getenv("TM_FOLD_NORM");
        e->fold_norm     = (!e->use_comm && m.weight_type == 0 && m.moe_experts == 0 && e->hidden % 64 == 0) ? (fold ? atoi(fold) & 3 : 3) : 0;
    }
    if (m.moe_experts > 0) {
        TM_HIP_CHECK(hipMalloc(&e->d_moe_ws, moe_workspace_bytes(e->layers[0].moe, e->max_tokens)));
    }
    // prefill scratch: every sequence padded to a multiple of 64 keys
    e->kflat_stride = ((e->max_tokens + c.session_len + 63) / 64) * 64 + 64 * (std::min(B, e->max_tokens) + 1);
    TM_TRY(dmalloc(&e->d_kflat, (size_t)e->kv_heads * e->kflat_stride * e->D));
    TM_TRY(dmalloc(&e->d_vflat, (size_t)e->kv_heads * e->kflat_stride * e->D));

    // RoPE table
    e->rope_max_pos = c.session_len + 1;
    {
        std::vector<half_t> tab((size_t)e->rope_max_pos * e->D);
        TM_TRY(build_rope_table(tab.data(), e->rope_max_pos, e->D, m.rope_base, m.rope_type, m.rope_factor,
                                m.rope_low_freq_factor, m.rope_high_freq_factor, m.rope_original_max_position));
        TM_TRY(dmalloc(&e->d_rope, tab.size() / 2));
        TM_HIP_CHECK(hipMemcpy(e->d_rope, tab.data(), tab.size() * 2, hipMemcpyHostToDevice));
    }

    TM_TRY(dmalloc(&e->d_ids, (size_t)B));
    TM_TRY(dmalloc(&e->d_next_ids, (size_t)B));
    TM_TRY(dmalloc(&e->d_k_len, (size_t)B));
    TM_TRY(dmalloc(&e->d_cu_q, (size_t)B + 1));
    TM_TRY(dmalloc(&e->d_cu_koff, (size_t)B + 1));
    TM_TRY(dmalloc(&e->d_rows, (size_t)B));
    TM_TRY(dmalloc(&e->d_generated, (size_t)B * c.session_len));
    TM_TRY(dmalloc(&e->d_step, (size_t)1));
    TM_TRY(dmalloc(&e->d_prefill_ids, T));
    TM_TRY(dmalloc(&e->d_argmax_val, (size_t)B));
    TM_TRY(dmalloc(&e->d_cand, (size_t)B * 2));
    TM_TRY(dmalloc(&e->d_cand_all, (size_t)B * 2 * c.tp));
    TM_TRY(dmalloc(&e->d_block_ptrs, (size_t)B * e->max_blocks_per_seq));
    TM_TRY(dmalloc(&e->d_cu_block_nums, (size_t)B + 1));
    {
        std::vector<int> cu(B + 1);
        for (int b = 0; b <= B; ++b) {
            cu[b] = b * e->max_blocks_per_seq;
        }
        TM_HIP_CHECK(hipMemcpy(e->d_cu_block_nums, cu.data(), cu.size() * 4, hipMemcpyHostToDevice));
    }
    const int max_splits = 16;
    TM_HIP_CHECK(hipMalloc((void**)&e->d_attn_ws, decode_attention_workspace_bytes(B, e->q_heads, e->D, max_splits)));

    // KV pool
    int64_t blocks = c.cache_blocks;
    if (blocks <= 0) {
        size_t free_b = 0, total_b = 0;
        TM_HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
        const float frac = c.cache_max_entry_count > 0.f ? c.cache_max_entry_count : 0.8f;
        blocks           = (int64_t)((double)free_b * frac / (double)e->block_bytes);
        const int64_t need = (int64_t)B * e->max_blocks_per_seq;
        blocks           = std::min(blocks, need);
    }
    TM_REQUIRE(blocks >= 1, "no memory for KV blocks");
    if (hipMalloc((void**)&e->pool, (size_t)blocks * e->block_bytes) != hipSuccess) {
        (void)hipGetLastError();
        set_last_error("KV pool allocation failed");
        return TM_OOM;
    }
    TM_HIP_CHECK(hipMemsetAsync(e->pool, 0, (size_t)blocks * e->block_bytes, e->stream));
    e->num_blocks = blocks;
    e->free_blocks.resize(blocks);
    for (int64_t i = 0; i < blocks; ++i) {
        e->free_blocks[i] = (int)(blocks - 1 - i);
    }
    TM_HIP_CHECK(hipStreamSynchronize(e->stream));
    e->started = true;
    // the reference's switches: TM_GEMM_IMPORT=<file> loads a dispatch table, TM_GEMM_TUNE=1 measures the decode batch
    // (max_batch_size rows) now, TM_GEMM_EXPORT=<file> writes the table
    if (const char* imp = getenv("TM_GEMM_IMPORT")) {
        if (dec32_table_import(imp)) {
            fprintf(stderr, "[tm] TM_GEMM_IMPORT: nothing read from %s\n", imp);
        }
    }
    const char* tune = getenv("TM_GEMM_TUNE");
    if (tune && atoi(tune) && B <= 256) {
        const char* v = getenv("TM_GEMM_TUNE_VERBOSE");
        if (tune_decode_gemms(e, B, v && atoi(v))) {  // not fatal: the heuristic tilings stay
            fprintf(stderr, "[tm] TM_GEMM_TUNE failed (%s); keeping the heuristic tilings\n", tm_last_error());
        }
    }
    if (const char* exp = getenv("TM_GEMM_EXPORT")) {
        TM_TRY(dec32_table_export(exp));
    }
    return 0;
}

// sampling state: device arrays for all slots (allocated on first use), upload of `n` slots starting at slot0
static int sampling_upload(tm_engine* e, const tm_sampling* p, int slot0, int n)
{
    const int B = e->cfg.max_batch_size;
    if (!e->d_temp) {
        TM_REQUIRE(e->vocab_local % 8 == 0, "sampling needs vocab % 8 == 0");
        TM_TRY(dmalloc(&e->d_temp, (size_t)B));
        TM_TRY(dmalloc(&e->d_topp, (size_t)B));
        TM_TRY(dmalloc(&e->d_minp, (size_t)B));
        TM_TRY(dmalloc(&e->d_u, (size_t)B));
        TM_TRY(dmalloc(&e->d_topk, (size_t)B));
        TM_TRY(dmalloc(&e->d_seed, (size_t)B));
        TM_HIP_CHECK(hipMalloc(&e->d_sample_ws, sample_workspace_bytes(B)));
        TM_HIP_CHECK(hipMemsetAsync(e->d_sample_ws, 0, sample_workspace_bytes(B), e->stream));
        if (e->use_comm) {
            TM_TRY(dmalloc(&e->d_logits_gather, (size_t)B * e->vocab_local * e->cfg.tp));
            TM_TRY(dmalloc(&e->d_logits_full, (size_t)B * e->vocab_local * e->cfg.tp));
        }
        std::vector<float>    one(B, 1.f), zero(B, 0.f);
        std::vector<int>      k1(B, 1);
        std::vector<uint64_t> s0(B, 0);
        TM_HIP_CHECK(hipMemcpyAsync(e->d_temp, one.data(), B * 4, hipMemcpyHostToDevice, e->stream));
        TM_HIP_CHECK(hipMemcpyAsync(e->d_topp, one.data(), B * 4, hipMemcpyHostToDevice, e->stream));
        TM_HIP_CHECK(hipMemcpyAsync(e->d_minp, zero.data(), B * 4, hipMemcpyHostToDevice, e->stream));
        TM_HIP_CHECK(hipMemcpyAsync(e->d_topk, k1.data(), B * 4, hipMemcpyHostToDevice, e->stream));
        TM_HIP_CHECK(hipMemcpyAsync(e->d_seed, s0.data(), B * 8, hipMemcpyHostToDevice, e->stream));
        TM_HIP_CHECK(hipStreamSynchronize(e->stream));
    }
    std::vector<float>    t(n), pp(n), mp(n);
    std::vector<int>      k(n);
    std::vector<uint64_t> sd(n);
    for (int i = 0; i < n; ++i) {
        TM_REQUIRE(p[i].temperature > 0.f, "sampling: temperature must be > 0");
        t[i]  = p[i].temperature;
        k[i]  = p[i].top_k;
        pp[i] = p[i].top_p;
        mp[i] = p[i].min_p;
        sd[i] = p[i].seed;
    }
    TM_HIP_CHECK(hipMemcpyAsync(e->d_temp + slot0, t.data(), n * 4, hipMemcpyHostToDevice, e->stream));
    TM_HIP_CHECK(hipMemcpyAsync(e->d_topk + slot0, k.data(), n * 4, hipMemcpyHostToDevice, e->stream));
    TM_HIP_CHECK(hipMemcpyAsync(e->d_topp + slot0, pp.data(), n * 4, hipMemcpyHostToDevice, e->stream));
    TM_HIP_CHECK(hipMemcpyAsync(e->d_minp + slot0, mp.data(), n * 4, hipMemcpyHostToDevice, e->stream));
    TM_HIP_CHECK(hipMemcpyAsync(e->d_seed + slot0, sd.data(), n * 8, hipMemcpyHostToDevice, e->stream));
    TM_HIP_CHECK(hipStreamSynchronize(e->stream));
    return 0;
}

// logits-processor state of `n` slots starting at slot0: parameters + cleared seen masks.  eos[i] (may be < 0) joins the
// stop ids as an end id; prompt_len[i] turns min_new_tokens into the context-length threshold of the kernel.
static const tm_logits_param kNoLogitsParam = {1.f, 0, 0, {0}, 0, {0}};

static int logits_param_check(const tm_logits_param& p)
{
    TM_REQUIRE(p.repetition_penalty > 0.f, "repetition_penalty must be > 0");
    TM_REQUIRE(p.min_new_tokens >= 0, "min_new_tokens must be >= 0");
    TM_REQUIRE(p.n_bad_ids >= 0 && p.n_bad_ids <= TM_MAX_BAD_IDS, "0 <= n_bad_ids <= TM_MAX_BAD_IDS");
    TM_REQUIRE(p.n_stop_ids >= 0 && p.n_stop_ids <= TM_MAX_STOP_IDS, "0 <= n_stop_ids <= TM_MAX_STOP_IDS");
    return 0;
}

static int logits_upload(tm_engine* e, const tm_logits_param* p, const int* prompt_len, const int* eos, int slot0, int n)
{
    const int B = e->cfg.max_batch_size;
    if (!e->d_seen) {
        TM_REQUIRE(e->vocab_local % 8 == 0, "logits processors need (local) vocab % 8 == 0");
        e->seen_words = (e->cfg.model.vocab + 31) / 32;
        TM_TRY(dmalloc(&e->d_seen, (size_t)B * e->seen_words));
        TM_TRY(dmalloc(&e->d_lp_rep, (size_t)B));
        TM_TRY(dmalloc(&e->d_lp_minlen, (size_t)B));
        TM_TRY(dmalloc(&e->d_lp_ban, (size_t)B * kMaxBadIds));
        TM_TRY(dmalloc(&e->d_lp_end, (size_t)B * kMaxEndIds));
        std::vector<float> one(B, 1.f);
        TM_HIP_CHECK(hipMemcpyAsync(e->d_lp_rep, one.data(), B * 4, hipMemcpyHostToDevice, e->stream));
        TM_HIP_CHECK(hipMemsetAsync(e->d_lp_minlen, 0, (size_t)B * 4, e->stream));
        TM_HIP_CHECK(hipMemsetAsync(e->d_lp_ban, 0xff, (size_t)B * kMaxBadIds * 4, e->stream));
        TM_HIP_CHECK(hipMemsetAsync(e->d_lp_end, 0xff, (size_t)B * kMaxEndIds * 4, e->stream));
        TM_HIP_CHECK(hipMemsetAsync(e->d_seen, 0, (size_t)B * e->seen_words * 4, e->stream));
        TM_HIP_CHECK(hipStreamSynchronize(e->stream));
    }
    std::vector<float> r(n);
    std::vector<int>   ml(n), ban((size_t)n * kMaxBadIds, -1), end((size_t)n * kMaxEndIds, -1);
    for (int i = 0; i < n; ++i) {
        TM_TRY(logits_param_check(p[i]));
        r[i]  = p[i].repetition_penalty;
        ml[i] = p[i].min_new_tokens > 0 ? prompt_len[i] + p[i].min_new_tokens : 0;
        for (int k = 0; k < p[i].n_bad_ids; ++k) {
            ban[(size_t)i * kMaxBadIds + k] = p[i].bad_ids[k];
        }
        end[(size_t)i * kMaxEndIds] = eos ? eos[i] : -1;
        for (int k = 0; k < p[i].n_stop_ids; ++k) {
            end[(size_t)i * kMaxEndIds + 1 + k] = p[i].stop_ids[k];
        }
    }
    TM_HIP_CHECK(hipMemcpyAsync(e->d_lp_rep + slot0, r.data(), n * 4, hipMemcpyHostToDevice, e->stream));
    TM_HIP_CHECK(hipMemcpyAsync(e->d_lp_minlen + slot0, ml.data(), n * 4, hipMemcpyHostToDevice, e->stream));
    TM_HIP_CHECK(hipMemcpyAsync(e->d_lp_ban + (size_t)slot0 * kMaxBadIds, ban.data(), ban.size() * 4, hipMemcpyHostToDevice, e->stream));
    TM_HIP_CHECK(hipMemcpyAsync(e->d_lp_end + (size_t)slot0 * kMaxEndIds, end.data(), end.size() * 4, hipMemcpyHostToDevice, e->stream));
    TM_HIP_CHECK(hipMemsetAsync(e->d_seen + (size_t)slot0 * e->seen_words, 0, (size_t)n * e->seen_words * 4, e->stream));
    TM_HIP_CHECK(hipStreamSynchronize(e->stream));
    return 0;
}

// decode split heuristic: fill >= 2 workgroups per CU (GetSplitCount, kernels/attention/utils.cc:11-46); fused prologue
static void setup_decode(tm_engine* e, int batch)
{
    const tm_engine_config& c = e->cfg;
    int splits = c.decode_splits;
    if (splits <= 0) {
        // int8 / int4 KV run the MFMA kernel: up to 16 query heads of a kv head per workgroup (launch_decode_attention_i8_mfma),
        // four waves = four cache blocks in flight per workgroup -- one workgroup per CU is enough: measured at one rank's head
        // count of TP = 8 (profiles/r04_gemm_experiments_session2.txt, call25): 64 x 4 workgroups 1.485 ms per step against
        // 1.547 .. 1.587 with 64 x 8 (Llama-3-8B shard), 5.06 against 5.26 (Llama-3-70B shard).  fp16 KV (VALU kernel, <= 4
        // heads per workgroup): two workgroups per CU as before.
        const bool mfma  = e->cfg.quant_policy == 8 || e->cfg.quant_policy == 4;
        const int  group = e->q_heads / e->kv_heads;
        int        hpw   = 1;
        if (mfma) {
            hpw = group;
            while (hpw > 16) {
                int d = 2;
                while (hpw % d) {
                    ++d;
                }
                hpw /= d;
            }
        }
        else {
            for (int cand = 4; cand >= 1; --cand) {
                if (group % cand == 0) {
                    hpw = cand;
                    break;
                }
            }
        }
        const int wgs = e->kv_heads * (group / hpw) * batch;
        splits        = 1;
        // (the 256-workgroup target is what was measured: full decode batches; small batches -- typically long contexts per sequence --
        // keep the deeper split)
        while (wgs * splits < (mfma && batch >= 32 ? 256 : 512) && splits < 16) {
            splits *= 2;
        }
    }
    e->decode_splits = std::min(std::max(splits, 1), 16);
    const char* valu = getenv("TM_ATTN_VALU");
    const char* fuse = getenv("TM_FUSE_QKV");
    e->fuse_qkv      = (e->cfg.quant_policy == 8 || e->cfg.quant_policy == 4) && !(valu && atoi(valu)) && !(fuse && !atoi(fuse));
}

// Chunked prefill of `batch` sequences into the batch slots [slot0, slot0 + batch): whole sequences,
// <= max_prefill_token_num tokens per iteration (a sequence longer than the budget is split into history + new tokens).
// Logits / first tokens land in d_logits / d_next_ids at slot0 + i.  Uses e->d_k_len / d_cu_q as iteration-local arrays.
// `mix` (continuous batching): the LAST iteration also carries the decode step of all batch slots as leading rows of the
// same forward (MixedDecode); every iteration then leaves room for those rows.  *mix->done reports that it happened.
struct MixedStep {
    int        rows;     // batch slots = decode rows
    int*       k_len;    // the decode state arrays (NOT the iteration-local e->d_k_len)
    const int* active;
    int*       ids;      // current token of every slot
    const uint64_t* block_ptrs;  // the decode block table
    const int* cu_q;     // 0 .. rows (the decode step's own array)
    bool*      done;
};

static int prefill_slots(tm_engine* e, const int* const* seq_ids, const int* host_lens, int batch, int slot0, float* ttft_ms,
                         const MixedStep* mix = nullptr)
{
    // The decode rows ride on the LAST iteration of the admission (any number of iterations, chunked prompts included -- the
    // reference mixes unconditionally, unified_attention_layer.cc:310-311).  The decode head of that iteration writes the
    // next-id entry of every batch slot, so the first tokens that EARLIER iterations left in d_next_ids are moved to
    // d_first_ids right after each iteration and handed back when the admission is done.
    const int  budget  = e->max_tokens - (mix ? mix->rows : 0);
    const auto t_start = std::chrono::steady_clock::now();
    // Because the batch tables (block_ptrs, cu_block_nums) are indexed by the batch slot, every prefill
    // iteration covers a contiguous range of slots [b0, b1]; the block table is offset accordingly and the
    // logits / first tokens of the iteration land in d_logits / d_next_ids at slot b0 + i.
    int b0 = 0;
    int done_in_b0 = 0;  // tokens of sequence b0 already prefilled (chunked long prompt)
    while (b0 < batch) {
        std::vector<int> cu_q{0}, klen, koff{0}, rows, ids;
        int b1 = b0, tokens = 0, max_q = 0, max_k = 0;
        bool partial_last = false;
        while (b1 < batch) {
            const int start  = (b1 == b0) ? done_in_b0 : 0;
            const int remain = host_lens[b1] - start;
            const int take   = std::min(remain, budget - tokens);
            if (take <= 0) {
                break;
            }
            ids.insert(ids.end(), seq_ids[b1] + start, seq_ids[b1] + start + take);
            tokens += take;
            cu_q.push_back(tokens);
            klen.push_back(start + take);
            koff.push_back(koff.back() + ((start + take + 63) / 64) * 64);
            rows.push_back(tokens - 1);
            max_q = std::max(max_q, take);
            max_k = std::max(max_k, start + take);
            if (take < remain) {  // budget exhausted inside this sequence: it continues in the next iteration
                done_in_b0   = start + take;
                partial_last = true;
                break;
            }
            ++b1;
        }
        const int nseq = (int)klen.size();
        TM_REQUIRE(nseq >= 1, "internal: empty prefill iteration");
        TM_REQUIRE(koff.back() <= e->kflat_stride, "internal: flatten scratch too small");
        // the last iteration of a continuous-batching admission: decode rows of every slot in front of the prefill rows
        const bool merge = mix && b1 == batch && !partial_last;
        const int  nd    = merge ? mix->rows : 0;
        if (merge) {
            TM_TRY(launch_advance_active(mix->k_len, mix->active, nd, e->stream));
            TM_HIP_CHECK(hipMemcpyAsync(e->d_prefill_ids, mix->ids, (size_t)nd * 4, hipMemcpyDeviceToDevice, e->stream));
            for (int& r : rows) {
                r += nd;
            }
        }
        TM_HIP_CHECK(hipMemcpyAsync(e->d_prefill_ids + nd, ids.data(), ids.size() * 4, hipMemcpyHostToDevice, e->stream));
        TM_HIP_CHECK(hipMemcpyAsync(e->d_cu_q, cu_q.data(), cu_q.size() * 4, hipMemcpyHostToDevice, e->stream));
        TM_HIP_CHECK(hipMemcpyAsync(e->d_k_len, klen.data(), klen.size() * 4, hipMemcpyHostToDevice, e->stream));
        TM_HIP_CHECK(hipMemcpyAsync(e->d_cu_koff, koff.data(), koff.size() * 4, hipMemcpyHostToDevice, e->stream));
        TM_HIP_CHECK(hipMemcpyAsync(e->d_rows, rows.data(), rows.size() * 4, hipMemcpyHostToDevice, e->stream));
        // shift the block tables so that slot 0 of this iteration is sequence b0
        uint64_t* saved_ptrs = e->d_block_ptrs;
        e->d_block_ptrs += (size_t)(slot0 + b0) * e->max_blocks_per_seq;
        const MixedDecode md{nd, merge ? mix->k_len : nullptr, merge ? mix->block_ptrs : nullptr, merge ? mix->cu_q : nullptr,
                             merge ? mix->active : nullptr};
        const int rc    = forward(e, e->d_prefill_ids, nd + tokens, nseq, false, max_q, max_k, e->kflat_stride, slot0 + b0,
                                  merge ? &md : nullptr);
        e->d_block_ptrs = saved_ptrs;
        if (rc) {
            return rc;
        }
        if (mix) {
            const int n_done = b1 - b0;  // sequences b0 .. b1-1 got their first token in this iteration
            if (n_done > 0 && !merge) {
                TM_HIP_CHECK(hipMemcpyAsync(e->d_first_ids + slot0 + b0, e->d_next_ids + slot0 + b0, (size_t)n_done * 4,
                                            hipMemcpyDeviceToDevice, e->stream));
            }
        }
        if (merge) {  // as decode_step_cb: the next ids of every slot become its current token ...
            TM_HIP_CHECK(hipMemcpyAsync(mix->ids, e->d_next_ids, (size_t)nd * 4, hipMemcpyDeviceToDevice, e->stream));
            // ... and the first tokens of the admission's earlier iterations return to their d_next_ids entries (the caller
            // reads first tokens from there); the sequences of THIS iteration wrote theirs after the decode head
            if (b0 > 0) {
                TM_HIP_CHECK(hipMemcpyAsync(e->d_next_ids + slot0, e->d_first_ids + slot0, (size_t)b0 * 4, hipMemcpyDeviceToDevice,
                                            e->stream));
            }
            *mix->done = true;
        }
        // the host vectors above are pageable: make sure the async copies are done before they die
        TM_HIP_CHECK(hipStreamSynchronize(e->stream));
        if (ttft_ms) {
            const float ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_start).count();
            for (int b = b0; b < b1; ++b) {
                ttft_ms[b] = ms;  // first token of sequence b exists once its last chunk has been processed
            }
        }
        b0 = b1;  // a partially prefilled sequence (b1) is revisited with done_in_b0 tokens of history
        if (!partial_last) {
            done_in_b0 = 0;
        }
    }
    return 0;
}

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
        TM_HIP_CHECK(hipMemcpyAsync(e->d_k_len + slot0, lens.data(), n * 4, hipMemcpyHostToDevice, e->stream));
        TM_HIP_CHECK(hipMemcpyAsync(e->d_ids + slot0, e->d_next_ids + slot0, n * 4, hipMemcpyDeviceToDevice, e->stream));
        TM_HIP_CHECK(hipMemcpyAsync(first.data(), e->d_next_ids + slot0, n * 4, hipMemcpyDeviceToHost, e->stream));
        TM_HIP_CHECK(hipMemcpyAsync(e->d_active + slot0, ones.data(), n * 4, hipMemcpyHostToDevice, e->stream));
        TM_HIP_CHECK(hipStreamSynchronize(e->stream));
        for (int k = 0; k < n; ++k) {
            e->h_active[slot0 + k] = 1;
            const int64_t rid      = e->sched->slot_request(slot0 + k);
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

int tm_engine_release(tm_engine* e)
{
    if (e && e->loop_on.load()) {
        set_last_error("the engine thread is running (tm_engine_serve_stop first)");
        return TM_CONFLICT;
    }
    TM_REQUIRE(e, "null pointer");
    if (e->stream) {
        TM_HIP_CHECK(hipStreamSynchronize(e->stream));
    }
    e->pending.valid = false;  // a look-ahead decode step of the session that ends here: its tokens belong to nobody
    for (auto& blks : e->h_blocks) {
        for (int b : blks) {
            e->free_blocks.push_back(b);
        }
    }
    e->h_blocks.clear();
    e->h_len.clear();
    e->batch      = 0;
    e->steps_done = 0;
    e->steps_fetched = 0;
    e->h_sampling.clear();
    e->cb_sampling.clear();
    e->sampling_on = false;
    e->h_logits.clear();
    e->cb_logits.clear();
    e->logits_on = false;
    if (e->sched) {  // leave continuous-batching mode: every block goes back to the static free list
        e->sched.reset();
        e->free_blocks.resize(e->num_blocks);
        for (int64_t i = 0; i < e->num_blocks; ++i) {
            e->free_blocks[i] = (int)(e->num_blocks - 1 - i);
        }
    }
    return 0;
}

int tm_engine_prefill(tm_engine* e, const int* host_ids, const int* host_lens, int batch, int max_new_tokens)
{
    if (e && e->loop_on.load()) {
        set_last_error("the engine thread is running (tm_engine_serve_stop first)");
        return TM_CONFLICT;
    }
    TM_REQUIRE(e && host_ids && host_lens, "null pointer");
    TM_REQUIRE(e->started, "engine not started");
    TM_REQUIRE(e->batch == 0 && !e->sched, "a batch is already admitted (release it first)");
    TM_REQUIRE(batch >= 1 && batch <= e->cfg.max_batch_size, "1 <= batch <= max_batch_size");
    TM_REQUIRE(max_new_tokens >= 1, "max_new_tokens >= 1");
    TM_HIP_CHECK(hipSetDevice(e->cfg.device));
    const tm_engine_config& c = e->cfg;

    // ---- admit: reserve blocks for prompt + generation ------------------------------------------
    int64_t need = 0;
    for (int b = 0; b < batch; ++b) {
        TM_REQUIRE(host_lens[b] >= 1, "empty prompt");
        if (host_lens[b] + max_new_tokens > c.session_len) {
            set_last_error("prompt + max_new_tokens exceeds session_len");
            return TM_TOO_LONG;
        }
        need += (host_lens[b] + max_new_tokens + 63) / 64;
    }
    if (need > (int64_t)e->free_blocks.size()) {
        set_last_error("out of KV cache blocks");
        return TM_OOM;
    }
    std::vector<uint64_t> ptrs((size_t)batch * e->max_blocks_per_seq, 0);
    e->h_blocks.assign(batch, {});
    for (int b = 0; b < batch; ++b) {
        const int nb = (host_lens[b] + max_new_tokens + 63) / 64;
        for (int i = 0; i < nb; ++i) {
            const int blk = e->free_blocks.back();
            e->free_blocks.pop_back();
            e->h_blocks[b].push_back(blk);
            ptrs[(size_t)b * e->max_blocks_per_seq + i] = (uint64_t)(e->pool + (int64_t)blk * e->block_bytes);
        }
    }
    TM_HIP_CHECK(hipMemcpyAsync(e->d_block_ptrs, ptrs.data(), ptrs.size() * 8, hipMemcpyHostToDevice, e->stream));
    TM_HIP_CHECK(hipMemsetAsync(e->d_step, 0, 4, e->stream));
    TM_HIP_CHECK(hipStreamSynchronize(e->stream));
    e->batch   = batch;
    e->max_new = max_new_tokens;
    e->h_len.assign(host_lens, host_lens + batch);
    e->steps_done = 0;
    e->steps_fetched = 0;

    e->sampling_on = false;
    if (!e->h_sampling.empty()) {
        TM_REQUIRE((int)e->h_sampling.size() == batch, "tm_engine_set_sampling: batch size differs from the prefill's");
        TM_TRY(sampling_upload(e, e->h_sampling.data(), 0, batch));
        e->sampling_on = true;
    }
    e->logits_on = false;
    if (!e->h_logits.empty()) {
        TM_REQUIRE((int)e->h_logits.size() == batch, "tm_engine_set_logits_params: batch size differs from the prefill's");
        TM_TRY(logits_upload(e, e->h_logits.data(), host_lens, nullptr, 0, batch));
        e->logits_on = true;
    }
    e->h_ttft_ms.assign(batch, 0.f);
    {
        std::vector<const int*> seq_ids(batch);
        int                     off = 0;
        for (int b = 0; b < batch; ++b) {
            seq_ids[b] = host_ids + off;
            off += host_lens[b];
        }
        TM_TRY(prefill_slots(e, seq_ids.data(), host_lens, batch, 0, e->h_ttft_ms.data()));
    }

    // ---- steady-state decode layout: one token per sequence ------------------------------------------
    std::vector<int> cu_q(batch + 1);
    for (int b = 0; b <= batch; ++b) {
        cu_q[b] = b;
    }
    TM_HIP_CHECK(hipMemcpyAsync(e->d_cu_q, cu_q.data(), cu_q.size() * 4, hipMemcpyHostToDevice, e->stream));
    TM_HIP_CHECK(hipMemcpyAsync(e->d_k_len, host_lens, batch * 4, hipMemcpyHostToDevice, e->stream));
    // generated[b][0] = first token; step counter = 1
    TM_HIP_CHECK(hipMemsetAsync(e->d_step, 0, 4, e->stream));
    TM_TRY(commit_tokens(e));
    TM_HIP_CHECK(hipStreamSynchronize(e->stream));
    e->steps_done = 1;

    setup_decode(e, batch);
    if (e->graph && (e->graph_batch != batch || e->graph_max_new != max_new_tokens || e->graph_sampling != e->sampling_on
                     || e->graph_logits != e->logits_on)) {
        (void)hipGraphExecDestroy(e->graph);
        e->graph = nullptr;
    }
    return 0;
}

// hipGraph capture of one decode step.  Steps that contain RCCL calls (tp > 1) are captured too -- a TP = 8 step is
// ~290 launches + 65 collectives, far too many for eager launches -- but defensively: if the capture or the instantiation
// fails (RCCL build without graph support, ...) the engine falls back to eager steps for good instead of failing.
static bool graph_enabled(const tm_engine* e)
{
    if (!e->cfg.use_graph) {
        return false;
    }
    if (!e->use_comm) {
        return true;
    }
    return e->graph_comm && !e->graph_comm_failed;  // TM_GRAPH_COMM=0 (read at create) keeps collectives out of graphs
}

static int capture_step(tm_engine* e, int (*step)(tm_engine*), hipGraphExec_t* exec)
{
    *exec        = nullptr;
    hipGraph_t g = nullptr;
    hipError_t be = hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal);
    int        rc = be == hipSuccess ? step(e) : 0;
    hipError_t ce = be == hipSuccess ? hipStreamEndCapture(e->stream, &g) : be;
    hipError_t ie = hipSuccess;
    if (rc == 0 && ce == hipSuccess) {
        ie = hipGraphInstantiate(exec, g, nullptr, nullptr, 0);
    }
    if (g) {
        (void)hipGraphDestroy(g);
    }
    if (rc == 0 && ce == hipSuccess && ie == hipSuccess) {
        return 0;
    }
    *exec = nullptr;
    if (e->use_comm) {  // collectives inside: give up on graphs, keep running
        (void)hipGetLastError();
        e->graph_comm_failed = true;
        fprintf(stderr, "[tm] hipGraph capture of the tensor-parallel decode step failed (rc %d, capture %s, instantiate %s): "
                        "falling back to eager launches\n", rc, hipGetErrorString(ce), hipGetErrorString(ie));
        return 0;
    }
    if (rc) {
        return rc;
    }
    TM_HIP_CHECK(ce);
    TM_HIP_CHECK(ie);
    return 0;
}

int tm_engine_decode(tm_engine* e, int steps)
{
    if (e && e->loop_on.load()) {
        set_last_error("the engine thread is running (tm_engine_serve_stop first)");
        return TM_CONFLICT;
    }
    TM_REQUIRE(e && e->batch > 0, "no admitted batch");
    TM_REQUIRE(!e->sched, "continuous-batching session active: use tm_engine_step");
    TM_HIP_CHECK(hipSetDevice(e->cfg.device));
    if (e->steps_done + steps > e->max_new) {
        set_last_error("decode past max_new_tokens");
        return TM_TOO_LONG;
    }
    if (graph_enabled(e) && !e->graph) {
        // run one eager step first (lazy module loading etc. must not happen inside a capture)
        if (steps == 0) {
            return 0;
        }
        TM_TRY(decode_step(e));
        TM_HIP_CHECK(hipStreamSynchronize(e->stream));
        e->steps_done += 1;
        steps -= 1;
        TM_TRY(capture_step(e, decode_step, &e->graph));
        e->graph_batch    = e->batch;
        e->graph_max_new  = e->max_new;
        e->graph_sampling = e->sampling_on;
        e->graph_logits   = e->logits_on;
    }
    for (int i = 0; i < steps; ++i) {
        if (graph_enabled(e) && e->graph) {
            TM_HIP_CHECK(hipGraphLaunch(e->graph, e->stream));
        }
        else {
            TM_TRY(decode_step(e));
        }
    }
    e->steps_done += steps;
    return 0;
}

int tm_engine_set_sampling(tm_engine* e, const tm_sampling* host_params, int batch)
{
    TM_REQUIRE(e, "null pointer");
    TM_REQUIRE(e->batch == 0 && !e->sched, "set the sampling parameters before tm_engine_prefill");
    e->h_sampling.clear();
    if (!host_params) {
        return 0;
    }
    TM_REQUIRE(!e->use_comm || e->comm || e->p2p_ready,
               "stochastic sampling with tp > 1 gathers the logits: tm_engine_comm_init or the native communicator first");
    TM_REQUIRE(batch >= 1 && batch <= e->cfg.max_batch_size, "1 <= batch <= max_batch_size");
    for (int i = 0; i < batch; ++i) {
        TM_REQUIRE(host_params[i].temperature > 0.f, "sampling: temperature must be > 0");
    }
    e->h_sampling.assign(host_params, host_params + batch);
    return 0;
}

int tm_engine_set_logits_params(tm_engine* e, const tm_logits_param* host_params, int batch)
{
    TM_REQUIRE(e, "null pointer");
    TM_REQUIRE(e->batch == 0 && !e->sched, "set the logits-processor parameters before tm_engine_prefill");
    e->h_logits.clear();
    if (!host_params) {
        return 0;
    }
    TM_REQUIRE(batch >= 1 && batch <= e->cfg.max_batch_size, "1 <= batch <= max_batch_size");
    for (int i = 0; i < batch; ++i) {
        TM_TRY(logits_param_check(host_params[i]));
    }
    e->h_logits.assign(host_params, host_params + batch);
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

int tm_engine_submit(tm_engine* e, const int* host_ids, int n, int max_new_tokens, int eos_id, int64_t* req_id)
{
    return tm_engine_submit_ex(e, host_ids, n, max_new_tokens, eos_id, nullptr, req_id);
}

// ---- issue / retire of a decode step (two-phase overlap, see tm_engine::PendingStep) ----
// the decode step of every slot, as a graph replay when graphs are on (captured on first use)
static int cb_launch_decode(tm_engine* e)
{
    if (e->graph_cb && (e->graph_cb_sampling != e->sampling_on || e->graph_cb_logits != e->logits_on)) {
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
    const bool        can_mix  = mixed_on && e->sched->n_active() > 0 && e->max_tokens - B >= 16;
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

int tm_engine_prefill_times(tm_engine* e, float* host_ms)
{
    TM_REQUIRE(e && host_ms, "null pointer");
    TM_REQUIRE((int)e->h_ttft_ms.size() == e->batch, "no admitted batch");
    memcpy(host_ms, e->h_ttft_ms.data(), sizeof(float) * e->batch);
    return 0;
}

int tm_engine_profile_decode(tm_engine* e, int steps, float* host_ms_per_step, int* host_launches_per_step)
{
    if (e && e->loop_on.load()) {
        set_last_error("the engine thread is running (tm_engine_serve_stop first)");
        return TM_CONFLICT;
    }
    TM_REQUIRE(e && host_ms_per_step && e->batch > 0 && steps >= 1, "arguments");
    TM_HIP_CHECK(hipSetDevice(e->cfg.device));
    if (e->steps_done + steps > e->max_new) {
        set_last_error("decode past max_new_tokens");
        return TM_TOO_LONG;
    }
    std::vector<double> acc(P_NUM, 0.0);
    std::vector<int>    cnt(P_NUM, 0);
    for (int i = 0; i < steps; ++i) {
        e->prof_on   = true;
        e->prof_used = 0;
        e->prof_spans.clear();
        int rc     = decode_step(e);
        e->prof_on = false;
        if (rc) {
            return rc;
        }
        TM_HIP_CHECK(hipStreamSynchronize(e->stream));
        for (auto& sp : e->prof_spans) {
            float ms = 0.f;
            TM_HIP_CHECK(hipEventElapsedTime(&ms, e->prof_pool[std::get<1>(sp)], e->prof_pool[std::get<2>(sp)]));
            acc[std::get<0>(sp)] += ms;
            cnt[std::get<0>(sp)] += 1;
        }
        e->steps_done += 1;
    }
    for (int c = 0; c < P_NUM; ++c) {
        host_ms_per_step[c] = (float)(acc[c] / steps);
        if (host_launches_per_step) {
            host_launches_per_step[c] = cnt[c] / steps;
        }
    }
    return 0;
}

int tm_engine_sync(tm_engine* e)
{
    TM_REQUIRE(e, "null pointer");
    TM_HIP_CHECK(hipStreamSynchronize(e->stream));
    TM_TRY(device_marks_fetch(e, false));
    return device_marks_check(e);
}

int tm_engine_fetch(tm_engine* e, int* host_out, int* n_generated)
{
    TM_REQUIRE(e && host_out && n_generated, "null pointer");
    TM_HIP_CHECK(hipStreamSynchronize(e->stream));
    TM_TRY(device_marks_fetch(e, false));
    // the tokens are handed over in any case (those of the steps before a communicator give-up are valid); the status says
    // whether every step behind them was
    TM_HIP_CHECK(hipMemcpy(host_out, e->d_generated, (size_t)e->batch * e->max_new * 4, hipMemcpyDeviceToHost));
    const int rc = device_marks_check(e);
    if (rc && e->steps_valid < 0) {
        e->steps_valid = e->steps_fetched;  // the mark was first seen now: what an earlier, clean fetch reported is known to be valid
    }
    // after a communicator give-up *n_generated = the columns known to be valid (the step count of the last clean fetch): the
    // caller can keep those and must discard the rest; the status says so
    *n_generated = rc ? (e->steps_valid < 0 ? 0 : e->steps_valid) : e->steps_done;
    if (!rc) {
        e->steps_fetched = e->steps_done;
    }
    return rc;
}

int tm_engine_fetch_logits(tm_engine* e, void* host_out)
{
    TM_REQUIRE(e && host_out, "null pointer");
    TM_HIP_CHECK(hipStreamSynchronize(e->stream));
    TM_HIP_CHECK(hipMemcpy(host_out, e->d_logits, (size_t)e->batch * e->vocab_local * 2, hipMemcpyDeviceToHost));
    return 0;
}

int tm_engine_debug_read(tm_engine* e, int what, int a, int b, void* host_out, int64_t bytes)
{
    TM_REQUIRE(e && host_out && e->started, "null pointer / engine not started");
    TM_HIP_CHECK(hipStreamSynchronize(e->stream));
    if (what == 0) {  // residual stream of the last forward: rows [0, a)
        TM_REQUIRE(a >= 1 && a <= e->max_tokens && bytes == (int64_t)a * e->hidden * 2, "residual rows / byte count");
        TM_HIP_CHECK(hipMemcpy(host_out, e->d_resid, (size_t)bytes, hipMemcpyDeviceToHost));
        return 0;
    }
    if (what == 1) {  // KV block b of static-batch sequence a (all layers, the reference's block byte layout)
        TM_REQUIRE(a >= 0 && a < (int)e->h_blocks.size() && b >= 0 && b < (int)e->h_blocks[a].size(), "sequence / block index");
        TM_REQUIRE(bytes == (int64_t)e->block_bytes, "byte count must be the block size");
        TM_HIP_CHECK(hipMemcpy(host_out, e->pool + (int64_t)e->h_blocks[a][b] * e->block_bytes, (size_t)bytes, hipMemcpyDeviceToHost));
        return 0;
    }
    if (what == 2) {  // int64: scheduler steps whose decode rows rode on a prefill forward (mixed steps)
        TM_REQUIRE(bytes == 8, "byte count must be 8");
        *(int64_t*)host_out = e->mixed_steps;
        return 0;
    }
    if (what == 3) {  // int64: decode steps issued while the previous one was unretired (two-phase overlap, TM_ASYNC_STEP)
        TM_REQUIRE(bytes == 8, "byte count must be 8");
        *(int64_t*)host_out = e->overlapped_steps;
        return 0;
    }
    set_last_error("tm_engine_debug_read: unknown selector");
    return 1;
}

tm_stream_t tm_engine_stream(tm_engine* e)
{
    return e ? (tm_stream_t)e->stream : nullptr;
}

int tm_engine_stats(tm_engine* e, int64_t* weight_bytes, int64_t* kv_bytes_per_token, int64_t* num_blocks,
                    int* decode_splits)
{
    TM_REQUIRE(e, "null pointer");
    int64_t wb = 0;
    for (auto& L : e->layers) {
        for (const LinearSlots* l : {&L.qkv, &L.wo, &L.w13, &L.w2}) {
            wb += (int64_t)l->w.packed_bytes + (int64_t)l->w.sz_bytes;
        }
        if (L.is_moe) {
            for (int x = 0; x < L.moe.experts; ++x) {
                wb += (int64_t)L.moe.w13[x].packed_bytes + (int64_t)L.moe.w13[x].sz_bytes + (int64_t)L.moe.w2[x].packed_bytes
                      + (int64_t)L.moe.w2[x].sz_bytes;
            }
        }
    }
    wb += (int64_t)e->output.w.packed_bytes;
    if (weight_bytes) *weight_bytes = wb;
    if (kv_bytes_per_token) *kv_bytes_per_token = e->started ? e->block_bytes / 64 : 0;
    if (num_blocks) *num_blocks = e->num_blocks;
    if (decode_splits) *decode_splits = e->decode_splits;
    return 0;
}

int tm_engine_comm_info(tm_engine* e, int* backend, int* ranks, int* graph_captured)
{
    TM_REQUIRE(e, "null pointer");
    int b = 0, n = 1;
    if (e->use_comm) {
        if (e->comm) {
            b = 1;
            TM_NCCL_CHECK(ncclCommCount(e->comm, &n));
        }
        if (e->p2p_ready) {
            b |= 2;
            n = e->cfg.tp;
        }
    }
    if (backend) *backend = b;
    if (ranks) *ranks = n;
    if (graph_captured) *graph_captured = (e->graph || e->graph_cb) ? 1 : 0;
    return 0;
}

int tm_engine_destroy(tm_engine* e)
{
    if (!e) {
        return 0;
    }
    (void)tm_engine_serve_stop(e);
    (void)hipSetDevice(e->cfg.device);
    if (e->stream) {
        (void)hipStreamSynchronize(e->stream);
    }
    if (e->graph) {
        (void)hipGraphExecDestroy(e->graph);
    }
    if (e->graph_cb) {
        (void)hipGraphExecDestroy(e->graph_cb);
    }
    if (e->aux_stream) {
        (void)hipStreamDestroy(e->aux_stream);
        (void)hipEventDestroy(e->ev_aux_fork);
        (void)hipEventDestroy(e->ev_aux_join);
    }
    for (int i = 0; i < 2; ++i) {
        if (e->h_step_pin[i]) {
            (void)hipHostFree(e->h_step_pin[i]);
            (void)hipEventDestroy(e->ev_step[i]);
        }
    }
    for (void* q : {(void*)e->d_active, (void*)e->d_pf_k_len, (void*)e->d_pf_cu_q, (void*)e->d_pf_block_ptrs, (void*)e->d_first_ids, (void*)e->d_temp, (void*)e->d_topp, (void*)e->d_minp,
                    (void*)e->d_u, (void*)e->d_topk, (void*)e->d_seed, e->d_sample_ws, (void*)e->d_logits_gather, (void*)e->d_logits_full,
                    (void*)e->d_seen, (void*)e->d_lp_rep,
                    (void*)e->d_lp_minlen, (void*)e->d_lp_ban, (void*)e->d_lp_end}) {
        if (q) {
            (void)hipFree(q);
        }
    }
    for (auto& L : e->layers) {
        for (LinearSlots* l : {&L.qkv, &L.wo, &L.w13, &L.w2}) {
            linear_weight_free(l->w);
        }
        for (auto& l : L.ex13) {
            linear_weight_free(l.w);
        }
        for (auto& l : L.ex2) {
            linear_weight_free(l.w);
        }
        if (L.is_moe) {
            moe_free(L.moe);
        }
    }
    if (e->d_moe_ws) {
        (void)hipFree(e->d_moe_ws);
    }
    linear_weight_free(e->output.w);
    for (auto& kv : e->slots) {
        if (kv.second.dev) {
            (void)hipFree(kv.second.dev);
        }
    }
    void* bufs[] = {e->pool, e->d_block_ptrs, e->d_cu_block_nums, e->d_resid, e->d_x, e->d_qkv, e->d_attn, e->d_act,
                    e->d_tmp, e->d_logits, e->d_last, e->d_gemm_ws, e->d_attn_ws, e->d_kflat, e->d_vflat, e->d_rope,
                    e->d_ids, e->d_k_len, e->d_cu_q, e->d_cu_koff, e->d_rows, e->d_generated, e->d_step,
                    e->d_prefill_ids, e->d_argmax_val, e->d_cand, e->d_cand_all, e->d_next_ids, e->d_ss, e->d_tickets};
    for (void* p : bufs) {
        if (p) {
            (void)hipFree(p);
        }
    }
    for (int r = 0; r < 8; ++r) {
        (void)tm_p2p_segment_close(e->p2p_peer[r], 1);
    }
    (void)tm_p2p_segment_close(e->p2p_seg, 0);
    if (e->p2p_state) {
        (void)hipFree(e->p2p_state);
    }
    if (e->comm) {
        (void)ncclCommDestroy(e->comm);
        if (e->comm_stream) {
            (void)hipStreamDestroy(e->comm_stream);
            (void)hipEventDestroy(e->ev_fork);
            (void)hipEventDestroy(e->ev_join);
        }
    }
    if (e->stream) {
        (void)hipStreamDestroy(e->stream);
    }
    delete e;
    return 0;
}

}  // extern "C"
