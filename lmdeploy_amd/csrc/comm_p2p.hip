// Fused all-reduce + residual + RMSNorm over peer-mapped buffers (xGMI point-to-point), the TM_COMM=native arm beside RCCL.
//
// Replaces: AllreduceResidualBiasRMSnorm of the native communicator -- src/turbomind/comm/cuda_ipc/fused_allreduce.cu:406-500
//           (one kernel: reduce the ranks' partial [M,H] tiles, add the residual, RMSNorm) with the one-shot exchange of
//           comm/cuda_ipc/allreduce.cu:249-343 (every rank reads every peer's buffer; for messages <= 1 MB the latency of ONE
//           exchange beats reduce-scatter + all-gather).  RCCL's ring needs 2(tp-1) hops of >= 2 us each for a 512 KB message
//           and a separate norm launch; here a decode all-reduce is one launch and one exchange.
// Arithmetic: h = fp16( sum over ranks r = 0..tp-1 (in rank order, fp32) of partial_r )   -- the same association on every
//             rank, so all ranks hold bit-identical hidden states;
//             then exactly rmsnorm_kernel<1>: r = h(resid + h); inv = rsqrt(mean f32(r)^2 + eps); y = h(h(f32(r) * inv) * w).
//
// Protocol (one launch per rank, the same grid everywhere; `partial` is the local [M][H] tile the preceding GEMM wrote):
//   0. every workgroup copies its row of `partial` into buffer (epoch & 1) of the rank's symmetric segment -- the buffer is
//      chosen ON THE DEVICE from the rank's call counter, so hipGraph replays, chunked prefills and interleaved all-gathers
//      keep the strict alternation the reuse argument below needs without any host-side bookkeeping;
//   1. every workgroup makes its XCD's L2 write the tile back at SYSTEM scope (release fence) and takes a ticket; the
//      workgroup that draws the last ticket stores the call's epoch into slot [me] of EVERY peer's flag array;
//   2. every workgroup polls its OWN flag array (relaxed system-scope loads + s_sleep, bounded) until all tp slots show
//      the epoch, then one system-scope acquire;
//   3. one workgroup per token row sums the tp partial rows (peer rows come over xGMI with non-temporal 16-byte loads),
//      adds the residual, normalises, stores residual + normed row locally;
//   4. the workgroup with the last exit ticket advances the rank's epoch word.
// No trailing barrier: a rank can only pass step 2 of call c+1 after every peer has ENTERED call c+1, i.e. finished reading
// buffer (c & 1) -- by the time this rank overwrites that buffer (step 0 of call c+2) nobody reads it any more.
// p2p_allgather_kernel runs the same steps 0-2 and 4 around a copy of every rank's n bytes (the lm_head's (value, index)
// candidates); every rank issues the same sequence of calls, so both kernels share the epoch counter.
// Every spin is bounded (TM_P2P_TIMEOUT_MS, default 30 s): on a timeout the kernel records the epoch in state[3] and carries on
// (wrong numbers, no hang); the engine reads state[3] at its host synchronisation points, fails the step and refuses further
// work until it is recreated (engine_comm.hip: device_marks_check).
//
// Status: protocol and arithmetic run on ONE GPU only in this round -- tests/test_gpu_p2p.py (tp ranks = tp streams, and tp
// PROCESSES that map each other's segments through IPC handles) and tests/test_gpu_tp.py (a tp = 2 engine as two processes
// on cuda:0, no RCCL) -- plus the CPU restatement oracle.p2p_allreduce_norm.  The xGMI hop itself (system-scope visibility
// between devices) has not run on multi-GPU hardware: RCCL stays the default, TM_COMM=native opts in.
#include "tm_common.h"
#include "tm_kernels.h"
#include <algorithm>
#include <stdlib.h>
#include <atomic>

namespace tmk {

struct P2pParams {
    half_t*       data[8];   // the two-tile region of rank r's segment (peer-mapped; [me] = local)
    uint32_t*     flags[8];  // flag array [8] of rank r, peer-mapped
    int           tp, me;
    uint32_t*     state;    // local: [0] epoch of the last finished call, [1] entry tickets, [2] exit tickets, [3] timeout mark
    size_t        tile;     // elements between buffer 0 and buffer 1
    const half_t* partial;  // [M][H] this rank's partial sums (local)
    half_t*       y;        // [M][H] normed output (local)
    half_t*       resid;    // [M][H] residual stream (local, in place)
    const half_t* weight;
    float         eps;
    int           M, H;
    // two-shot form (p2p_allreduce_norm_2shot_kernel): per rank an input region (every rank's partial rows) and an output region
    // (the normed rows the slice owners push), both [rows2][H] fp16, behind the one-shot tiles of the same segment
    half_t*       in2[8];
    half_t*       out2[8];
    int           slice;    // rows per rank: rank r owns rows [r * slice, min(M, (r + 1) * slice))
    const uint32_t* src;    // all-gather: n words of this rank
    uint32_t*       dst;    // all-gather: rank q's n words land at dst + q * dst_stride
    int             n;
    size_t          dst_stride;
    uint64_t        timeout;  // bound of a peer wait in 100 MHz ticks (p2p_timeout_ticks)
    // row-flag form (p2p_allreduce_norm_rows_kernel, round 6): its own two tiles and one flag word per (sender, row) in every segment
    half_t*         rdata[8];
    uint32_t*       rflags[8];  // [8 senders][rows_cap]
    int             rows_cap;
};

// Wait bound of a peer flag, in ticks of the 100 MHz realtime counter.  TM_P2P_TIMEOUT_MS, default 30 s (read once): RCCL has no
// bound at all, and ranks are host-driven one by one -- graph capture, a first-use code load or a descheduled host thread on ONE
// rank must not end the job (the former bound was ~1 s of polling).  On expiry: state[3] = the epoch, the engine
// fails the step and refuses further work (engine_comm.hip: device_marks_check).
static std::atomic<uint64_t> g_p2p_timeout_cap{0};  // > 0: upper bound of the wait while a bring-up self-test runs (tm_engine_comm_native_selftest)
void p2p_timeout_cap_ms(long ms)
{
    g_p2p_timeout_cap.store(ms > 0 ? (uint64_t)ms * 100000ull : 0);
}

static uint64_t p2p_timeout_ticks_base();
static uint64_t p2p_timeout_ticks()
{
    const uint64_t cap = g_p2p_timeout_cap.load(), t = p2p_timeout_ticks_base();
    return cap && cap < t ? cap : t;
}

static uint64_t p2p_timeout_ticks_base()
{
    static const uint64_t t = [] {
        const char* v  = getenv("TM_P2P_TIMEOUT_MS");
        const long  ms = v ? atol(v) : 30000;
        return (uint64_t)(ms < 1 ? 1 : ms) * 100000ull;
    }();
    return t;
}

// steps 1 + 2; the caller's writes into its own segment buffer precede (a __syncthreads in between).  `phase` = how many syncs of
// THIS launch came before: the entry tickets (state[1]) run monotonically through a launch -- sync k publishes at ticket
// (k + 1) * gridDim.x - 1 -- and are reset by p2p_exit only, when every workgroup is past its last sync (a reset
// between two syncs of one launch could wipe an early ticket of the second).
// WT (round 6, the one-shot kernel): the payload went out through WRITE-THROUGH system-scope stores (sc0 sc1) and the consumer reads the
// peers' rows with system-scope loads -- the drained stores are the publish, no L2 write-back (release fence, 1.7 .. 6.5 us) and no
// cache invalidate (acquire fence, ~1.7 us) on the path of a decode-sized exchange (MI355X_MICROARCH.md, valid forms: "{sc0 sc1 stores and
// loads both sides}"; price list: handoff-flag, drained write-through payload vs plain + release).
template<bool WT = false>
__device__ __forceinline__ void p2p_publish_and_wait(const P2pParams& p, uint32_t epoch, uint32_t phase = 0)
{
    const int tid = threadIdx.x;
    // every storing wave drains its own stores before the one release below can cover them (cdna_hip_programming.md Guideline
    // 16 pitfall 14: "only lane 0 drained" is stale under load)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        if constexpr (!WT) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");  // system scope: this XCD's dirty lines of the tile leave the L2
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint32_t t = __hip_atomic_fetch_add(p.state + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == (phase + 1) * gridDim.x - 1) {  // every workgroup (hence every XCD that ran one) has written back
            for (int r = 0; r < p.tp; ++r) {
                __hip_atomic_store(p.flags[r] + p.me, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
    if (tid < p.tp) {
        const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
        uint32_t       spins = 0;
        // ">= epoch" (wrap-safe), not "== epoch": a peer that has already left this call may have published the NEXT epoch
        // before this workgroup got to look (it can be one call ahead, never two: see the reuse argument in the header)
        while ((int32_t)(__hip_atomic_load(p.flags[p.me] + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - epoch) < 0) {
            __builtin_amdgcn_s_sleep(2);
            if ((++spins & 63u) == 0 && __builtin_amdgcn_s_memrealtime() - t0 > p.timeout) {
                __hip_atomic_store(p.state + 3, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
        if constexpr (!WT) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");  // system scope
        }
    }
    __syncthreads();
}

// step 4
__device__ __forceinline__ void p2p_exit(const P2pParams& p, uint32_t epoch)
{
    if (threadIdx.x == 0) {
        const uint32_t t = __hip_atomic_fetch_add(p.state + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == gridDim.x - 1) {
            __hip_atomic_store(p.state + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // entry tickets of this launch's syncs
            __hip_atomic_store(p.state + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(p.state, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

template<int NV>
__global__ __launch_bounds__(512) void p2p_allreduce_norm_kernel(P2pParams p)
{
    __shared__ float    red[8];
    __shared__ uint32_t s_epoch;
    const int tid = threadIdx.x;
    if (tid == 0) {
        s_epoch = __hip_atomic_load(p.state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
    }
    __syncthreads();
    const uint32_t epoch = __builtin_amdgcn_readfirstlane(s_epoch);  // wave-uniform for the compiler too: it becomes part of buffer descriptors
    const size_t   boff  = (size_t)(epoch & 1) * p.tile;

    const int row  = blockIdx.x;
    const int nvec = p.H / 8;
    half8_t   wv[NV], r[NV], mine[NV];
    size_t    off[NV];
    bool      ok[NV];
    // ---- 0. own partial row -> own segment; the loads that do not depend on the peers are issued here as well -------------
    // The row leaves through write-through system-scope stores (16 bytes, `sc0 sc1`; the asm ends with s_nop 1: cdna_hip_programming.md 5.7
    // item 1): once a wave's vmcnt is 0 its part of the row is in memory, where the peers' system-scope loads find it -- no L2 write-back.
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int vi = tid + i * blockDim.x;
        ok[i]        = vi < nvec;
        const int vc = ok[i] ? vi : nvec - 1;
        off[i]       = (size_t)row * p.H + (size_t)vc * 8;
        mine[i]      = *(const half8_t*)(p.partial + off[i]);
        wv[i]        = *(const half8_t*)(p.weight + (size_t)vc * 8);
        r[i]         = *(const half8_t*)(p.resid + off[i]);
        if (ok[i] && p.tp > 1) {  // (one rank: nobody reads the segment)
            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p.data[p.me] + boff + off[i]), "v"(mine[i]) : "memory");
        }
    }
    __syncthreads();
    // ---- 1 + 2 ------------------------------------------------------------------------------------------------------------
    p2p_publish_and_wait<true>(p, epoch);

    // ---- 3. reduce + residual + RMSNorm of row blockIdx.x (rmsnorm_kernel<1> arithmetic) ---------------------------------
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        float acc[8] = {};
        for (int q = 0; q < p.tp; ++q) {  // rank order: the same association on every rank
            // the own row from registers (the bits that were stored); a peer's row by a system-scope load (sc0 sc1: served by memory, not by
            // this device's caches -- what replaces the acquire fence)
            half8_t v = mine[i];
            if (q != p.me) {
                const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.data[q] + boff), 0, (int)(p.tile * 2), 0x00020000);
                v             = bit_cast<half8_t>(__builtin_amdgcn_raw_buffer_load_b128(rs, (int)(off[i] * 2), 0, /*sc0 sc1*/ 17));
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                acc[e] += (float)v[e];
            }
        }
        half8_t h;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            h[e] = (half_t)acc[e];
        }
        r[i] = r[i] + h;
        if (ok[i]) {
            *(half8_t*)(p.resid + off[i]) = r[i];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float f = (float)r[i][e];
                ss            = __builtin_fmaf(f, f, ss);
            }
        }
    }
    ss = group_sum<64>(ss);
    if ((tid & 63) == 0) {
        red[tid >> 6] = ss;
    }
    __syncthreads();
    float tot = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) {
        tot += red[w];
    }
    const float inv = 1.0f / __builtin_sqrtf(tot / (float)p.H + p.eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if (ok[i]) {
            half8_t o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const half_t n = (half_t)((float)r[i][e] * inv);
                o[e]           = n * wv[i][e];
            }
            *(half8_t*)(p.y + off[i]) = o;
        }
    }
    p2p_exit(p, epoch);
}

// ---- one-shot form WITHOUT any coupling between the workgroups of a launch (round 6) -----------------------------------------------
// The kernel above pays, per exchange, a local fan-in (every workgroup's ticket -> the last one publishes the call's epoch), a fan-out
// (every workgroup polls that one flag) and an exit fan-in (the epoch word): three device-scope round trips on the critical path of a
// 64-row decode exchange whatever the peers do (8 us per launch with ONE rank, profiles/r06_tp_default_*).  Here a workgroup owns one
// token row END TO END: it reads the row's call counter (state[4 + row], a local word only it touches), pushes its partial row into its
// own segment with write-through system-scope stores, drains, stores the counter value into flag [me][row] of every peer's segment,
// polls flags [q][row] of its own segment, sums the peers' rows (system-scope loads) in rank order, normalises, and writes the counter
// back.  No ticket, no shared epoch, no requirement that the launch's workgroups are resident together.  Same arithmetic, same bits.
// Buffer reuse: call c of row m uses tile (c & 1); a rank's call c + 1 starts after its call c has finished (stream order), a peer's flag
// for c + 1 is stored after that peer's call c + 1 wrote its row -- so when this rank overwrites tile (c & 1) in call c + 2, every peer
// has finished call c (it published c + 1 before this rank could leave call c + 1's poll).  The tiles and flags are this form's own:
// the all-gather / two-shot calls keep the shared epoch and their regions.
// Replaces: the same AllreduceResidualBiasRMSnorm (fused_allreduce.cu:406-500); per-row flags are what its "flags per block" barrier does.
template<int NV>
__global__ __launch_bounds__(512) void p2p_allreduce_norm_rows_kernel(P2pParams p)
{
    __shared__ float    red[8];
    __shared__ uint32_t s_epoch;
    const int tid = threadIdx.x;
    const int row = blockIdx.x;
    uint32_t* const ctr = p.state + 4 + row;
    if (tid == 0) {
        s_epoch = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
    }
    const int nvec = p.H / 8;
    half8_t   wv[NV], r[NV], mine[NV];
    size_t    off[NV];
    bool      ok[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {  // everything that does not depend on the call counter or the peers
        const int vi = tid + i * blockDim.x;
        ok[i]        = vi < nvec;
        const int vc = ok[i] ? vi : nvec - 1;
        off[i]       = (size_t)row * p.H + (size_t)vc * 8;
        mine[i]      = *(const half8_t*)(p.partial + off[i]);
        wv[i]        = *(const half8_t*)(p.weight + (size_t)vc * 8);
        r[i]         = *(const half8_t*)(p.resid + off[i]);
    }
    __syncthreads();
    const uint32_t epoch = __builtin_amdgcn_readfirstlane(s_epoch);
    const size_t   boff  = (size_t)(epoch & 1) * p.tile;
    if (p.tp > 1) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if (ok[i]) {
                asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p.rdata[p.me] + boff + off[i]), "v"(mine[i]) : "memory");
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave: its part of the row is in memory
        __syncthreads();
        if (tid < p.tp && tid != p.me) {
            __hip_atomic_store(p.rflags[tid] + (size_t)p.me * p.rows_cap + row, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            const uint32_t* f  = p.rflags[p.me] + (size_t)tid * p.rows_cap + row;
            const uint64_t  t0 = __builtin_amdgcn_s_memrealtime();
            uint32_t        spins = 0;
            // ">= epoch" (wrap-safe): the peer may be one call ahead (never two: the reuse argument above)
            while ((int32_t)(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - epoch) < 0) {
                __builtin_amdgcn_s_sleep(1);
                if ((++spins & 63u) == 0 && __builtin_amdgcn_s_memrealtime() - t0 > p.timeout) {
                    __hip_atomic_store(p.state + 3, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
        }
        __syncthreads();
    }
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        float acc[8] = {};
        for (int q = 0; q < p.tp; ++q) {  // rank order: the same association on every rank
            half8_t v = mine[i];
            if (q != p.me) {
                const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.rdata[q] + boff), 0, (int)(p.tile * 2), 0x00020000);
                v             = bit_cast<half8_t>(__builtin_amdgcn_raw_buffer_load_b128(rs, (int)(off[i] * 2), 0, /*sc0 sc1*/ 17));
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                acc[e] += (float)v[e];
            }
        }
        half8_t h;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            h[e] = (half_t)acc[e];
        }
        r[i] = r[i] + h;
        if (ok[i]) {
            *(half8_t*)(p.resid + off[i]) = r[i];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float f = (float)r[i][e];
                ss            = __builtin_fmaf(f, f, ss);
            }
        }
    }
    ss = group_sum<64>(ss);
    if ((tid & 63) == 0) {
        red[tid >> 6] = ss;
    }
    __syncthreads();
    float tot = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) {
        tot += red[w];
    }
    const float inv = 1.0f / __builtin_sqrtf(tot / (float)p.H + p.eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if (ok[i]) {
            half8_t o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const half_t n = (half_t)((float)r[i][e] * inv);
                o[e]           = n * wv[i][e];
            }
            *(half8_t*)(p.y + off[i]) = o;
        }
    }
    if (tid == 0) {
        __hip_atomic_store(ctr, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---- two-shot form for large messages ---------------------------------------------------------------------------------------
// Replaces: AllreduceResidualBiasRMSnorm_Simple_Push / _Pull (src/turbomind/comm/cuda_ipc/fused_allreduce.cu:25-405) and the
// two-shot all-reduce of comm/cuda_ipc/allreduce.cu:22-248: reduce-scatter, residual + RMSNorm on the owned slice, all-gather.
// A prefill forward's [M][H] partial sums are MBs: the one-shot exchange above reads (tp - 1) x M x H x 2 bytes per rank, this
// form reads (tp - 1) / tp of one message and writes as much:
//   0. every rank copies its partial rows into its `in2` region;                                   sync A (epoch e)
//   1. rank r reduces rows [r * slice, (r + 1) * slice) over all ranks' `in2` (rank order, fp32 -- the one-shot kernel's
//      association, so the two forms give the same bits), adds ITS slice of the residual stream, normalises, stores the
//      normed rows locally and PUSHES them into every peer's `out2` region;                         sync B (epoch e + 1)
//   2. every rank copies the other slices' normed rows from its own `out2` into y.
// As in the reference the residual stream is sharded by row slice from here on: a rank keeps only its own rows current
// (nothing else reads them: the next reduction of the same forward owns the same slice; a new forward starts from the embedding).
// The two syncs per call make the regions reusable without the parity alternation the one-shot form needs; the calls share the
// epoch counter and flags with it (two epochs per call), so any interleaving of the two forms keeps its ordering argument.
// Persistent grid: workgroups loop over rows (all of them must be resident: they wait for each other's tickets).
template<int NV>
__global__ __launch_bounds__(512) void p2p_allreduce_norm_2shot_kernel(P2pParams p)
{
    __shared__ float    red[8];
    __shared__ uint32_t s_epoch;
    const int tid = threadIdx.x;
    if (tid == 0) {
        s_epoch = __hip_atomic_load(p.state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
    }
    __syncthreads();
    const uint32_t epoch = s_epoch;
    const int      nvec  = p.H / 8;
    const int      nthr  = blockDim.x;
    // ---- 0. own partial rows -> own in2 ----------------------------------------------------------------------------------------
    for (int row = blockIdx.x; row < p.M; row += gridDim.x) {
        for (int vi = tid; vi < nvec; vi += nthr) {
            const size_t o = (size_t)row * p.H + (size_t)vi * 8;
            *(half8_t*)(p.in2[p.me] + o) = *(const half8_t*)(p.partial + o);
        }
    }
    p2p_publish_and_wait(p, epoch);
    // ---- 1. reduce + residual + RMSNorm of the owned slice; push the normed rows ---------------------------------------------------
    const int r0 = p.me * p.slice, r1 = min(p.M, r0 + p.slice);
    for (int row = r0 + (int)blockIdx.x; row < r1; row += gridDim.x) {
        half8_t wv[NV], r[NV];
        size_t  off[NV];
        bool    ok[NV];
        float   ss = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int vi = tid + i * nthr;
            ok[i]        = vi < nvec;
            const int vc = ok[i] ? vi : nvec - 1;
            off[i]       = (size_t)row * p.H + (size_t)vc * 8;
            wv[i]        = *(const half8_t*)(p.weight + (size_t)vc * 8);
            r[i]         = *(const half8_t*)(p.resid + off[i]);
            float acc[8] = {};
            for (int q = 0; q < p.tp; ++q) {  // rank order: the same association on every rank and in the one-shot form
                const u32x4   raw = __builtin_nontemporal_load((const u32x4*)(p.in2[q] + off[i]));
                const half8_t v   = bit_cast<half8_t>(raw);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    acc[e] += (float)v[e];
                }
            }
            half8_t h;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                h[e] = (half_t)acc[e];
            }
            r[i] = r[i] + h;
            if (ok[i]) {
                *(half8_t*)(p.resid + off[i]) = r[i];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float f = (float)r[i][e];
                    ss            = __builtin_fmaf(f, f, ss);
                }
            }
        }
        ss = group_sum<64>(ss);
        __syncthreads();  // `red` of the previous row has been read by everybody
        if ((tid & 63) == 0) {
            red[tid >> 6] = ss;
        }
        __syncthreads();
        float tot = 0.f;
        for (int w = 0; w < (nthr >> 6); ++w) {
            tot += red[w];
        }
        const float inv = 1.0f / __builtin_sqrtf(tot / (float)p.H + p.eps);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            if (ok[i]) {
                half8_t o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const half_t n = (half_t)((float)r[i][e] * inv);
                    o[e]           = n * wv[i][e];
                }
                *(half8_t*)(p.y + off[i]) = o;
                for (int q = 1; q < p.tp; ++q) {  // start with the next rank: the pushes of the tp ranks fan out over the links
                    const int dst = p.me + q < p.tp ? p.me + q : p.me + q - p.tp;
                    *(half8_t*)(p.out2[dst] + off[i]) = o;
                }
            }
        }
    }
    p2p_publish_and_wait(p, epoch + 1, 1);
    // ---- 2. the other slices' normed rows: own out2 -> y --------------------------------------------------------------------------------
    for (int row = blockIdx.x; row < p.M; row += gridDim.x) {
        if (row >= r0 && row < r1) {
            continue;
        }
        for (int vi = tid; vi < nvec; vi += nthr) {
            const size_t o = (size_t)row * p.H + (size_t)vi * 8;
            const u32x4  v = __builtin_nontemporal_load((const u32x4*)(p.out2[p.me] + o));
            *(u32x4*)(p.y + o) = v;
        }
    }
    p2p_exit(p, epoch + 1);
}

// in2[r] / out2[r]: rank r's two-shot regions ([rows2][H] fp16 each) as mapped by THIS rank; M <= rows2
int launch_p2p_allreduce_norm_2shot(half_t* const* in2, half_t* const* out2, uint32_t* const* flags, int tp, int me, uint32_t* state,
                                    size_t region, const half_t* partial, half_t* y, half_t* resid, const half_t* weight, float eps, int M,
                                    int H, hipStream_t st)
{
    TM_REQUIRE(tp >= 1 && tp <= 8 && me >= 0 && me < tp, "p2p all-reduce: 1 <= tp <= 8");
    TM_REQUIRE(H % 8 == 0 && H <= 8192, "p2p all-reduce: H % 8 == 0, H <= 8192");
    TM_REQUIRE((size_t)M * H <= region, "p2p two-shot all-reduce: the message does not fit the segment region");
    if (M == 0) {
        return 0;
    }
    P2pParams p{};
    for (int r = 0; r < tp; ++r) {
        p.in2[r]   = in2[r];
        p.out2[r]  = out2[r];
        p.flags[r] = flags[r];
    }
    p.tp = tp, p.me = me, p.state = state, p.partial = partial, p.y = y, p.resid = resid, p.weight = weight;
    p.timeout = p2p_timeout_ticks();
    p.eps = eps, p.M = M, p.H = H;
    p.slice        = (M + tp - 1) / tp;
    const int nvec = H / 8;
    int       t    = (nvec + 63) / 64 * 64;
    t              = t > 512 ? 512 : t;
    const bool one = (nvec + t - 1) / t == 1;
    int dev = 0, cus = 0, per_cu = 0;
    TM_HIP_CHECK(hipGetDevice(&dev));
    TM_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const void* k = one ? (const void*)p2p_allreduce_norm_2shot_kernel<1> : (const void*)p2p_allreduce_norm_2shot_kernel<2>;
    TM_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, t, 0));
    per_cu         = per_cu > 1 ? per_cu - 1 : per_cu;  // (see p2p_allreduce_capacity)
    // every workgroup of a rank must be resident at once; ranks that SHARE a device (the one-GPU tests: tp streams / processes on
    // cuda:0) must fit together, so the grid can be capped from outside
    static const int cap_env = getenv("TM_P2P_2SHOT_GRID") ? atoi(getenv("TM_P2P_2SHOT_GRID")) : 0;
    int              grid    = std::max(1, std::min(M, std::min(per_cu, 2) * cus));
    if (cap_env > 0) {
        grid = std::min(grid, cap_env);
    }
    if (one) {
        p2p_allreduce_norm_2shot_kernel<1><<<grid, t, 0, st>>>(p);
    }
    else {
        p2p_allreduce_norm_2shot_kernel<2><<<grid, t, 0, st>>>(p);
    }
    TM_HIP_CHECK(hipGetLastError());
    return 0;
}

// all-gather of n 32-bit words per rank through the same segments (one workgroup)
__global__ __launch_bounds__(256) void p2p_allgather_kernel(P2pParams p)
{
    __shared__ uint32_t s_epoch;
    const int tid = threadIdx.x;
    if (tid == 0) {
        s_epoch = __hip_atomic_load(p.state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
    }
    __syncthreads();
    const uint32_t epoch = s_epoch;
    const size_t   boff  = (size_t)(epoch & 1) * p.tile;
    uint32_t*      mine  = (uint32_t*)(p.data[p.me] + boff);
    for (int i = tid; i < p.n; i += blockDim.x) {
        mine[i] = p.src[i];
    }
    __syncthreads();
    p2p_publish_and_wait(p, epoch);
    for (int q = 0; q < p.tp; ++q) {
        const uint32_t* theirs = (const uint32_t*)(p.data[q] + boff);
        for (int i = tid; i < p.n; i += blockDim.x) {
            p.dst[(size_t)q * p.dst_stride + i] = __builtin_nontemporal_load(theirs + i);
        }
    }
    p2p_exit(p, epoch);
}

// token rows one launch may carry = workgroups of `threads` threads that are resident at once on the current device
int p2p_allreduce_capacity(int threads, bool one_vec)
{
    int dev = 0, cus = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    const void* k = one_vec ? (const void*)p2p_allreduce_norm_kernel<1> : (const void*)p2p_allreduce_norm_kernel<2>;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, threads, 0) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    per_cu = per_cu > 1 ? per_cu - 1 : per_cu;
    return per_cu * cus;
}

// data[r]: the two-tile region (2 * tile elements) of rank r's segment, flags[r]: its flag array, both as mapped by THIS rank
// (index me = local pointers); state: 4 zero-initialised local words; partial: this rank's [M][H] sums, M <= tile / H rows
int launch_p2p_allreduce_norm(half_t* const* data, uint32_t* const* flags, int tp, int me, uint32_t* state, size_t tile,
                              const half_t* partial, half_t* y, half_t* resid, const half_t* weight, float eps, int M, int H,
                              hipStream_t st)
{
    TM_REQUIRE(tp >= 1 && tp <= 8 && me >= 0 && me < tp, "p2p all-reduce: 1 <= tp <= 8");
    TM_REQUIRE(H % 8 == 0 && H <= 8192, "p2p all-reduce: H % 8 == 0, H <= 8192");
    TM_REQUIRE((size_t)M * H <= tile && tile % 8 == 0, "p2p all-reduce: the tile does not fit the segment buffer");
    if (M == 0) {
        return 0;
    }
    P2pParams p{};
    for (int r = 0; r < tp; ++r) {
        p.data[r]  = data[r];
        p.flags[r] = flags[r];
    }
    p.tp = tp, p.me = me, p.state = state, p.tile = tile, p.partial = partial, p.y = y, p.resid = resid, p.weight = weight;
    p.timeout = p2p_timeout_ticks();
    p.eps = eps, p.M = M, p.H = H;
    const int nvec = H / 8;
    int       t    = (nvec + 63) / 64 * 64;
    t              = t > 512 ? 512 : t;
    const bool one = (nvec + t - 1) / t == 1;
    // Every workgroup of the launch waits for the LAST local ticket before the epoch is published, so all M of them must be
    // resident at once: M is capped by what the occupancy query says fits on the chip (one below it per CU when more than one
    // fits: the query is one block per CU high for some SGPR counts, MI355X_MICROARCH.md "Residency and cooperative launch")
    TM_REQUIRE(M <= p2p_allreduce_capacity(t, one), "p2p all-reduce: more token rows than workgroups that are resident at once");
    if (one) {
        p2p_allreduce_norm_kernel<1><<<M, t, 0, st>>>(p);
    }
    else {
        p2p_allreduce_norm_kernel<2><<<M, t, 0, st>>>(p);
    }
    TM_HIP_CHECK(hipGetLastError());
    return 0;
}

// row-flag form: rdata[r] = rank r's two row tiles (`tile` fp16 elements each), rflags[r] = its [8][rows_cap] flag words, state = 4 +
// rows_cap local words (zeroed once); any M <= rows_cap (no residency requirement: the workgroups of a launch do not wait for each other)
int launch_p2p_allreduce_norm_rows(half_t* const* rdata, uint32_t* const* rflags, int rows_cap, int tp, int me, uint32_t* state, size_t tile,
                                   const half_t* partial, half_t* y, half_t* resid, const half_t* weight, float eps, int M, int H, hipStream_t st)
{
    TM_REQUIRE(tp >= 1 && tp <= 8 && me >= 0 && me < tp, "p2p all-reduce: 1 <= tp <= 8");
    TM_REQUIRE(H % 8 == 0 && H <= 8192, "p2p all-reduce: H % 8 == 0, H <= 8192");
    TM_REQUIRE(M <= rows_cap && (size_t)rows_cap * H <= tile && tile % 8 == 0, "p2p all-reduce (row flags): M <= rows, the tile holds rows x H");
    if (M == 0) {
        return 0;
    }
    P2pParams p{};
    for (int r = 0; r < tp; ++r) {
        p.rdata[r]  = rdata[r];
        p.rflags[r] = rflags[r];
    }
    p.rows_cap = rows_cap;
    p.tp = tp, p.me = me, p.state = state, p.tile = tile, p.partial = partial, p.y = y, p.resid = resid, p.weight = weight;
    p.timeout = p2p_timeout_ticks();
    p.eps = eps, p.M = M, p.H = H;
    const int nvec = H / 8;
    int       t    = (nvec + 63) / 64 * 64;
    t              = t > 512 ? 512 : t;
    if ((nvec + t - 1) / t == 1) {
        p2p_allreduce_norm_rows_kernel<1><<<M, t, 0, st>>>(p);
    }
    else {
        p2p_allreduce_norm_rows_kernel<2><<<M, t, 0, st>>>(p);
    }
    TM_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_p2p_allgather(half_t* const* data, uint32_t* const* flags, int tp, int me, uint32_t* state, size_t tile, const void* src,
                         void* dst, int words, hipStream_t st, size_t dst_stride_words)
{
    TM_REQUIRE(tp >= 1 && tp <= 8 && me >= 0 && me < tp, "p2p all-gather: 1 <= tp <= 8");
    TM_REQUIRE((size_t)words * 2 <= tile, "p2p all-gather: message larger than the segment buffer");
    if (words == 0) {
        return 0;
    }
    P2pParams p{};
    for (int r = 0; r < tp; ++r) {
        p.data[r]  = data[r];
        p.flags[r] = flags[r];
    }
    p.tp = tp, p.me = me, p.state = state, p.tile = tile, p.src = (const uint32_t*)src, p.dst = (uint32_t*)dst, p.n = words;
    p.timeout = p2p_timeout_ticks();
    p.dst_stride = dst_stride_words ? dst_stride_words : (size_t)words;
    p2p_allgather_kernel<<<1, 256, 0, st>>>(p);
    TM_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace tmk
