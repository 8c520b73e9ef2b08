// Tensor-parallel collectives of the engine: RCCL all-reduce / all-gather call sites (comm/nccl/nccl.cu:356-398), the native P2P
// communicator's set-up and fused all-reduce + residual + RMSNorm (comm/cuda_ipc/fused_allreduce.cu:406-500), give-up marks.
#include "engine_internal.h"

namespace tmk {

// Stand-in for the duration of an exchange, for the ONE-rank emulation of a tensor-parallel rank only (TM_FORCE_COMM=1 / `bench.py
// --emulate-tp N`: the communicator has one rank, its all-reduce is the identity and takes no time).  TM_EMULATE_AR_GBPS = g > 0: every
// all-reduce of a prefill-sized forward is followed, on the stream it was enqueued on, by one idle workgroup that waits
// 10 us + bytes / g -- what an all-reduce with algorithm bandwidth g GB/s would hold that stream for.  It lets a 1-GPU box show what the
// two-row-half choreography hides and what it leaves exposed (profiles/r05_prefill_overlap_emulated_exchange.txt); it moves no data, is
// never armed with more than one rank, and is not a scaling result.
__global__ void exchange_standin_kernel(unsigned ticks)
{
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
        __builtin_amdgcn_s_sleep(64);
    }
}

static int exchange_standin(tm_engine* e, size_t bytes, hipStream_t st)
{
    if (e->emulate_ar_gbps > 0.f && e->cfg.tp == 1) {
        const double us = 10.0 + (double)bytes / ((double)e->emulate_ar_gbps * 1e3);
        exchange_standin_kernel<<<1, 64, 0, st>>>((unsigned)(us * 100.0));
        TM_HIP_CHECK(hipGetLastError());
    }
    return 0;
}

// fp16 sum of the row-parallel partial outputs over the TP group (comm/nccl/nccl.cu:356-398 calls ncclAllReduce on the
// compute stream).  Decode-sized forwards always take this form: a decode layer is one dependent chain, a collective between two
// of its kernels has nothing to run beside (round 2 measured a side stream with and without a weight prefetch under it:
// profiles/r02_comm_stream_arms.txt).  Prefill-sized forwards overlap for real: allreduce_rows_side below.
static int allreduce_hidden(tm_engine* e, half_t* buf, int M)
{
    if (!e->use_comm) {
        return 0;
    }
    TM_REQUIRE(e->comm != nullptr, "tm_engine_comm_init was not called");
    TM_NCCL_CHECK(ncclAllReduce(buf, buf, (size_t)M * e->hidden, ncclHalf, ncclSum, e->comm, e->stream));
    if (M >= 2 * e->pipe_min_rows) {
        TM_TRY(exchange_standin(e, (size_t)M * e->hidden * 2, e->stream));
    }
    return 0;
}

// ---- prefill forwards of a tensor-parallel rank: all-reduces on the side stream UNDER the other half's kernels ----------------------
// The north star's "RCCL all-reduce over xGMI overlapped on a side HIP stream".  A prefill-sized forward runs as two halves whose
// all-reduces (RCCL, `comm_stream`) sit under the other half's kernels on the engine stream (engine_forward.hip):
//   * two MICRO-BATCHES split at a sequence boundary (prefill_slots -> scheduler.h: prefill_microbatch_split): they share nothing between
//     the embedding and the lm_head, attention included, and leapfrog through all layers (forward_layers_two_microbatches) -- the default;
//   * without a usable boundary, two ROW HALVES of the row-wise part of every layer: wo, all-reduce, residual + RMSNorm, w1w3, w2,
//     all-reduce, residual + RMSNorm, next w_qkv (forward_tail_two_halves).
// The reference hides its collective inside a fused kernel instead (comm/cuda_ipc/fused_allreduce.cu:406-500); at prefill sizes
// (8192 x 4096 fp16 = 64 MB per all-reduce, two per layer) the exchange is bandwidth, not latency, and a second stream is the form that
// hides bandwidth.  One communicator, two streams, never concurrently: the engine stream waits for the side stream's last event before
// it enqueues a collective of its own (lm_head all-gather, decode steps).  Measured against an emulated exchange on one GPU only
// (profiles/r05_prefill_overlap_emulated_exchange.txt, profiles/r05_prefill_overlap_kernel_trace.txt).
// Events come from a per-engine pool (one pair per all-reduce of a forward: nothing is re-recorded while a wait on it may be pending).
static int pipe_event(tm_engine* e, hipEvent_t* ev)
{
    if (e->pipe_events_used == e->pipe_events.size()) {
        hipEvent_t n;
        TM_HIP_CHECK(hipEventCreateWithFlags(&n, hipEventDisableTiming));
        e->pipe_events.push_back(n);
    }
    *ev = e->pipe_events[e->pipe_events_used++];
    return 0;
}

bool prefill_pipe_ok(const tm_engine* e, int M)
{
    // RCCL is what serves a forward of this size (reduce_residual_norm's first branch takes the native communicator), the side
    // stream exists, and the forward is large enough for an exchange to be worth eight cross-stream hand-offs per layer
    return e->use_comm && e->comm && e->comm_overlap && e->comm_stream && !(e->p2p_ready && M <= e->p2p_rows)
           && M >= 2 * e->pipe_min_rows;
}

// rows [r0, r0 + rows) of d_tmp, produced by what the engine stream holds so far: summed over the ranks on the side stream; `*done`
// fires behind the sum (the consumer on the engine stream waits for it: pipe_wait)
int allreduce_rows_side(tm_engine* e, int r0, int rows, hipEvent_t* done)
{
    TM_REQUIRE(e->comm && e->comm_stream, "internal: side-stream all-reduce without a communicator");
    hipEvent_t ready;
    TM_TRY(pipe_event(e, &ready));
    TM_TRY(pipe_event(e, done));
    half_t* buf = e->d_tmp + (size_t)r0 * e->hidden;
    TM_HIP_CHECK(hipEventRecord(ready, e->stream));
    TM_HIP_CHECK(hipStreamWaitEvent(e->comm_stream, ready, 0));
    TM_NCCL_CHECK(ncclAllReduce(buf, buf, (size_t)rows * e->hidden, ncclHalf, ncclSum, e->comm, e->comm_stream));
    TM_TRY(exchange_standin(e, (size_t)rows * e->hidden * 2, e->comm_stream));
    TM_HIP_CHECK(hipEventRecord(*done, e->comm_stream));
    e->pipe_allreduces += 1;
    return 0;
}

int pipe_wait(tm_engine* e, hipEvent_t done)
{
    TM_HIP_CHECK(hipStreamWaitEvent(e->stream, done, 0));
    return 0;
}

void p2p_tables(tm_engine* e, half_t** data, uint32_t** flags)
{
    for (int r = 0; r < e->cfg.tp; ++r) {
        char* base = (char*)(r == e->cfg.rank ? e->p2p_seg : e->p2p_peer[r]);
        flags[r]   = (uint32_t*)base;
        data[r]    = (half_t*)(base + 256);
    }
}

// the row-flag form's own tiles and flags: behind the two-shot regions of every segment (tm_p2p_segment_bytes_rows)
static void p2p_row_tables(tm_engine* e, half_t** rdata, uint32_t** rflags)
{
    const size_t off = tm_p2p_segment_bytes2(e->p2p_rows, e->p2p_rows2, e->hidden);
    for (int r = 0; r < e->cfg.tp; ++r) {
        char* base = (char*)(r == e->cfg.rank ? e->p2p_seg : e->p2p_peer[r]);
        rdata[r]   = (half_t*)(base + off);
        rflags[r]  = (uint32_t*)(rdata[r] + 2 * (size_t)e->p2p_rows * e->hidden);
    }
}

// one decode-sized exchange: ONE launch, its workgroups independent of each other (comm_p2p.hip: p2p_allreduce_norm_rows_kernel);
// TM_P2P_ROWFLAGS=0: the ticket form (p2p_allreduce_norm_kernel)
static int p2p_oneshot(tm_engine* e, const half_t* partial, half_t* y, half_t* resid, const half_t* norm_w, int rows)
{
    static const bool rowflags = [] {
        const char* v = getenv("TM_P2P_ROWFLAGS");
        return !v || atoi(v) != 0;
    }();
    if (rowflags) {
        half_t*   rdata[8];
        uint32_t* rflags[8];
        p2p_row_tables(e, rdata, rflags);
        return launch_p2p_allreduce_norm_rows(rdata, rflags, e->p2p_rows, e->cfg.tp, e->cfg.rank, e->p2p_state, (size_t)e->p2p_rows * e->hidden,
                                              partial, y, resid, norm_w, e->cfg.model.rms_eps, rows, e->hidden, e->stream);
    }
    half_t*   data[8];
    uint32_t* flags[8];
    p2p_tables(e, data, flags);
    return launch_p2p_allreduce_norm(data, flags, e->cfg.tp, e->cfg.rank, e->p2p_state, (size_t)e->p2p_rows * e->hidden, partial, y, resid, norm_w,
                                     e->cfg.model.rms_eps, rows, e->hidden, e->stream);
}

// d_x = RMSNorm(d_resid += sum over ranks of d_tmp): one fused P2P launch per <= p2p_rows rows on the native communicator
// (any M when there is no RCCL communicator to fall back to), else RCCL all-reduce + the residual-norm kernel
int reduce_residual_norm(tm_engine* e, int M, const half_t* norm_w)
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
        for (int m0 = 0; m0 < M; m0 += e->p2p_rows) {
            const int    rows = std::min(e->p2p_rows, M - m0);
            const size_t off  = (size_t)m0 * e->hidden;
            TM_PROF(P_ALLREDUCE, TM_TRY(p2p_oneshot(e, e->d_tmp + off, e->d_x + off, e->d_resid + off, norm_w, rows)));
        }
        return 0;
    }
    TM_PROF(P_ALLREDUCE, TM_TRY(allreduce_hidden(e, e->d_tmp, M)));
    TM_PROF(P_RES_NORM, TM_TRY(launch_residual_rmsnorm(e->d_x, e->d_resid, e->d_tmp, nullptr, 0, nullptr, norm_w,
                                                       e->cfg.model.rms_eps, M, e->hidden, e->stream)));
    return 0;
}

// Device-side give-up mark of the native P2P communicator: a bounded wait for a peer expired (p2p_state[3] = the call number a
// peer missed; comm_p2p.hip carries on with wrong numbers instead of hanging).  Read at the host's synchronisation points.
// `async`: enqueue the copy on the engine stream (the caller syncs).
int device_marks_fetch(tm_engine* e, bool async)
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
int device_marks_check(tm_engine* e)
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

}  // namespace tmk

extern "C" {

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
    // TM_COMM_STREAM (default 1): prefill-sized forwards run their all-reduces on a side stream under the other half's kernels
    // (allreduce_rows_side above); 0 = every collective on the engine stream.  TM_PIPE_MIN_ROWS (1024): a forward splits from 2048 rows
    // (16 MB per all-reduce at hidden 4096): below that the doubled launches and the eight cross-stream hand-offs per layer (measured as
    // + 14 ... 28 % on a prefill whose exchange is free, profiles/r05_prefill_overlap_emulated_exchange.txt) cost more than the exchange
    // they hide; a micro-batch may be as small as half of it (prefill_microbatch_split).
    const char* cs   = getenv("TM_COMM_STREAM");
    const char* pm   = getenv("TM_PIPE_MIN_ROWS");
    const char* em   = getenv("TM_EMULATE_AR_GBPS");
    e->emulate_ar_gbps = em && e->cfg.tp == 1 ? (float)atof(em) : 0.f;
    e->comm_overlap  = !cs || atoi(cs) != 0;
    e->pipe_min_rows = pm && atoi(pm) >= 64 ? atoi(pm) / 64 * 64 : 1024;
    if (e->comm_overlap && !e->comm_stream) {
        TM_HIP_CHECK(hipStreamCreateWithFlags(&e->comm_stream, hipStreamNonBlocking));
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
    TM_TRY(tm_p2p_segment_create(tm_p2p_segment_bytes_rows(rows, rows2, e->hidden), &e->p2p_seg, handle64));
    e->p2p_rows2 = rows2;
    TM_HIP_CHECK(hipMalloc((void**)&e->p2p_state, (4 + (size_t)rows) * sizeof(uint32_t)));  // [0..3] shared call state, [4 + row] per-row call counters
    TM_HIP_CHECK(hipMemset(e->p2p_state, 0, (4 + (size_t)rows) * sizeof(uint32_t)));
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

// Bring-up check of the native communicator, called by EVERY rank right after the import (collective): two fused all-reduce + residual +
// RMSNorm launches (both buffer parities) over `rows` rows of a known pattern -- partial[m][h] = (rank + 1) * (h % 7 + 1) / 8, exact in
// fp16 for tp <= 8 -- the residual stream must come back as the exact sum over the ranks, the normed row as oracle arithmetic of it, and no
// peer wait may have expired.  *ok = 0 tells the caller to fall back (tm_engine_comm_native_drop on every rank): the hop between devices --
// system-scope visibility of the peers' stores over xGMI -- is the one part of this communicator no single-GPU test can exercise, and a
// default path must not depend on it unverified.  The wait bound of the two launches is min(TM_P2P_TIMEOUT_MS, 5 s).
int tm_engine_comm_native_selftest(tm_engine* e, int* ok)
{
    TM_REQUIRE(e && ok, "null pointer");
    TM_REQUIRE(e->p2p_ready, "native communicator: export + import first");
    *ok = 0;
    TM_HIP_CHECK(hipSetDevice(e->cfg.device));
    const int    tp = e->cfg.tp, me = e->cfg.rank, H = e->hidden;
    const int    rows = std::min(e->p2p_rows, 8);
    const size_t n = (size_t)rows * H;
    std::vector<half_t> hp(n), hw(H, (half_t)1.0f), hr(n, (half_t)0.0f), hy(n);
    for (int m = 0; m < rows; ++m) {
        for (int h = 0; h < H; ++h) {
            hp[(size_t)m * H + h] = (half_t)((float)(me + 1) * (float)(h % 7 + 1) / 8.0f);
        }
    }
    half_t *dp = nullptr, *dw = nullptr, *dr = nullptr, *dy = nullptr;
    auto    cleanup = [&]() {
        (void)hipFree(dp), (void)hipFree(dw), (void)hipFree(dr), (void)hipFree(dy);
    };
    int rc = 0;
    auto run = [&]() -> int {
        TM_HIP_CHECK(hipMalloc((void**)&dp, n * 2));
        TM_HIP_CHECK(hipMalloc((void**)&dw, (size_t)H * 2));
        TM_HIP_CHECK(hipMalloc((void**)&dr, n * 2));
        TM_HIP_CHECK(hipMalloc((void**)&dy, n * 2));
        TM_HIP_CHECK(hipMemcpy(dp, hp.data(), n * 2, hipMemcpyHostToDevice));
        TM_HIP_CHECK(hipMemcpy(dw, hw.data(), (size_t)H * 2, hipMemcpyHostToDevice));
        bool         good = true;
        for (int call = 0; call < 2 && good; ++call) {
            TM_HIP_CHECK(hipMemcpy(dr, hr.data(), n * 2, hipMemcpyHostToDevice));
            TM_TRY(p2p_oneshot(e, dp, dy, dr, dw, rows));  // the launch the decode steps will use
            TM_HIP_CHECK(hipStreamSynchronize(e->stream));
            std::vector<half_t> gr(n);
            TM_HIP_CHECK(hipMemcpy(gr.data(), dr, n * 2, hipMemcpyDeviceToHost));
            TM_HIP_CHECK(hipMemcpy(hy.data(), dy, n * 2, hipMemcpyDeviceToHost));
            uint32_t mark = 0;
            TM_HIP_CHECK(hipMemcpy(&mark, e->p2p_state + 3, 4, hipMemcpyDeviceToHost));
            good = mark == 0;
            const float tri = (float)(tp * (tp + 1) / 2);
            for (int m = 0; m < rows && good; ++m) {
                double ss = 0.0;
                for (int h = 0; h < H; ++h) {
                    const float want = tri * (float)(h % 7 + 1) / 8.0f;  // exact: <= 36 * 7 / 8
                    good             = good && (float)gr[(size_t)m * H + h] == want;
                    ss += (double)want * want;
                }
                const float inv = 1.0f / std::sqrt((float)(ss / H) + e->cfg.model.rms_eps);
                for (int h = 0; h < H && good; h += 97) {  // the normed row: a sample, to the rounding of h(h(r * inv) * 1)
                    const float want = tri * (float)(h % 7 + 1) / 8.0f * inv;
                    good             = std::fabs((float)hy[(size_t)m * H + h] - want) <= 2e-3f * std::fabs(want) + 1e-3f;
                }
            }
        }
        // the all-gather of the lm_head's (value, index) candidates runs on this communicator too (ticket form, shared call counter): one call,
        // every rank's 64 words must arrive in rank order
        if (good) {
            uint32_t *dsrc = nullptr, *ddst = nullptr;
            TM_HIP_CHECK(hipMalloc((void**)&dsrc, 64 * 4));
            TM_HIP_CHECK(hipMalloc((void**)&ddst, (size_t)tp * 64 * 4));
            std::vector<uint32_t> hs(64), hd((size_t)tp * 64);
            for (int i = 0; i < 64; ++i) {
                hs[i] = 0x5eed0000u + (uint32_t)me * 256u + (uint32_t)i;
            }
            int rc2 = 0;
            auto body = [&]() -> int {
                TM_HIP_CHECK(hipMemcpy(dsrc, hs.data(), 64 * 4, hipMemcpyHostToDevice));
                TM_HIP_CHECK(hipMemset(ddst, 0, (size_t)tp * 64 * 4));
                half_t*   data[8];
                uint32_t* flags[8];
                p2p_tables(e, data, flags);
                TM_TRY(launch_p2p_allgather(data, flags, tp, me, e->p2p_state, (size_t)e->p2p_rows * H, dsrc, ddst, 64, e->stream));
                TM_HIP_CHECK(hipStreamSynchronize(e->stream));
                TM_HIP_CHECK(hipMemcpy(hd.data(), ddst, (size_t)tp * 64 * 4, hipMemcpyDeviceToHost));
                uint32_t mark = 0;
                TM_HIP_CHECK(hipMemcpy(&mark, e->p2p_state + 3, 4, hipMemcpyDeviceToHost));
                good = mark == 0;
                for (int q = 0; q < tp && good; ++q) {
                    for (int i = 0; i < 64 && good; ++i) {
                        good = hd[(size_t)q * 64 + i] == 0x5eed0000u + (uint32_t)q * 256u + (uint32_t)i;
                    }
                }
                return 0;
            };
            rc2 = body();
            (void)hipFree(dsrc), (void)hipFree(ddst);
            if (rc2) {
                return rc2;
            }
        }
        *ok = good ? 1 : 0;
        return 0;
    };
    p2p_timeout_cap_ms(5000);
    rc = run();
    p2p_timeout_cap_ms(0);
    cleanup();
    return rc;
}

// Every rank calls it when ANY rank's self-test failed: the decode-sized exchanges go back to RCCL + the residual-norm launch.  The peer
// mappings stay until the engine is destroyed (a peer may still be inside the failed call); the call counters and the give-up mark are reset.
int tm_engine_comm_native_drop(tm_engine* e)
{
    TM_REQUIRE(e, "null pointer");
    TM_REQUIRE(e->comm != nullptr || !e->p2p_ready, "native communicator: it is the only communicator of this engine (RCCL was dropped)");
    if (e->p2p_ready) {
        TM_HIP_CHECK(hipSetDevice(e->cfg.device));
        TM_HIP_CHECK(hipStreamSynchronize(e->stream));
        TM_HIP_CHECK(hipMemset(e->p2p_state, 0, (4 + (size_t)e->p2p_rows) * sizeof(uint32_t)));
        if (e->graph) {  // a decode step captured with the fused launches: the next decode captures again
            (void)hipGraphExecDestroy(e->graph);
            e->graph = nullptr;
        }
        if (e->graph_cb) {
            (void)hipGraphExecDestroy(e->graph_cb);
            e->graph_cb = nullptr;
        }
        e->p2p_ready   = false;
        e->h_mark      = 0;
        e->comm_failed = false;
    }
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

int tm_engine_comm_overlap_info(tm_engine* e, int* side_stream, int64_t* forwards, int64_t* microbatch_forwards, int64_t* allreduces)
{
    TM_REQUIRE(e, "null pointer");
    if (side_stream) *side_stream = (e->use_comm && e->comm && e->comm_overlap && e->comm_stream) ? 1 : 0;
    if (forwards) *forwards = e->pipe_forwards;
    if (microbatch_forwards) *microbatch_forwards = e->pipe_mb_forwards;
    if (allreduces) *allreduces = e->pipe_allreduces;
    return 0;
}

}  // extern "C"
