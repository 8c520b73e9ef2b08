// The forward pass of the engine (LanguageModel::Forward / UnifiedDecoder::Forward, src/turbomind/models/language_model.cc:493-542,
// models/llama/unified_decoder.cc:163-380): one forward over M tokens (decode, prefill, mixed), the static-batch prefill / decode entry
// points and the hipGraph capture of the decode step.  See engine.hip for the overview.
#include "engine_internal.h"

namespace tmk {

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
int launch_advance_active(int* k_len, const int* active, int n, hipStream_t st)
{
    advance_active_kernel<<<(n + 63) / 64, 64, 0, st>>>(k_len, active, n);
    TM_HIP_CHECK(hipGetLastError());
    return 0;
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

size_t prof_event(tm_engine* e)
{
    if (e->prof_used == e->prof_pool.size()) {
        hipEvent_t ev;
        (void)hipEventCreate(&ev);
        e->prof_pool.push_back(ev);
    }
    (void)hipEventRecord(e->prof_pool[e->prof_used], e->stream);
    return e->prof_used++;
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
    cfg.tickets = e->d_tickets;
    return launch_linear(l.w, x, ldx, y, ldy, M, gated, cfg, e->d_gemm_ws, false, nullptr, e->stream);
}

// Tensor-parallel prefill forward, dense layer, from the attention output to the next layer's QKV projection, as two row halves
// A = [0, Ma), B = [Ma, M): every operator here is row-wise, so a half's all-reduce (RCCL on the side stream, engine_comm.hip) runs
// under the other half's GEMMs on the engine stream --
//   engine stream:  wo A | wo B | norm A  w1w3 A  w2 A | norm B  w1w3 B  w2 B | norm A  qkv' A | norm B  qkv' B
//   side stream:         | AR A        | AR B                 | AR A                   | AR B
// (qkv' = the NEXT layer's w_qkv: it hides the last exchange; absent behind the last layer).  Same kernels, same per-row arithmetic
// as the unsplit sequence (reduce_residual_norm's RCCL branch): a row's result does not depend on which rows share its launch, as
// long as both forms pick the same tiling -- they pick per launch size, so equality with the unsplit forward is to rounding, and the
// parity gate is the engine's usual one against the oracle.  Reference role: the overlap the reference gets from its fused
// all-reduce + norm kernel (unified_decoder.cc:278,328; comm/cuda_ipc/fused_allreduce.cu:406-500).
static int forward_tail_two_halves(tm_engine* e, Layer& L, int M, int Ma, const half_t* next_norm, Layer* next)
{
    const int    r0[2] = {0, Ma}, nr[2] = {Ma, M - Ma};
    const int    kq = e->q_heads * e->D, H = e->hidden;
    const float  eps = e->cfg.model.rms_eps;
    hipStream_t  st = e->stream;
    hipEvent_t   done[2];
    for (int h = 0; h < 2; ++h) {
        TM_PROF(P_GEMM_O, TM_TRY(linear_plain(e, L.wo, e->d_attn + (size_t)r0[h] * kq, kq, e->d_tmp + (size_t)r0[h] * H, H, nr[h], false)));
        TM_TRY(allreduce_rows_side(e, r0[h], nr[h], &done[h]));
    }
    for (int h = 0; h < 2; ++h) {
        const size_t off = (size_t)r0[h] * H;
        TM_TRY(pipe_wait(e, done[h]));
        TM_PROF(P_RES_NORM, TM_TRY(launch_residual_rmsnorm(e->d_x + off, e->d_resid + off, e->d_tmp + off, nullptr, 0, nullptr, L.ffn_norm, eps,
                                                           nr[h], H, st)));
        TM_PROF(P_GEMM_GATE_UP, TM_TRY(linear_plain(e, L.w13, e->d_x + off, H, e->d_act + (size_t)r0[h] * e->inter, e->inter, nr[h], true)));
        TM_PROF(P_GEMM_DOWN, TM_TRY(linear_plain(e, L.w2, e->d_act + (size_t)r0[h] * e->inter, e->inter, e->d_tmp + off, H, nr[h], false)));
        TM_TRY(allreduce_rows_side(e, r0[h], nr[h], &done[h]));
    }
    for (int h = 0; h < 2; ++h) {
        const size_t off = (size_t)r0[h] * H;
        TM_TRY(pipe_wait(e, done[h]));
        TM_PROF(P_RES_NORM, TM_TRY(launch_residual_rmsnorm(e->d_x + off, e->d_resid + off, e->d_tmp + off, nullptr, 0, nullptr, next_norm, eps,
                                                           nr[h], H, st)));
        if (next) {
            TM_PROF(P_GEMM_QKV, TM_TRY(linear_plain(e, next->qkv, e->d_x + off, H, e->d_qkv + (size_t)r0[h] * e->qkv_n, e->qkv_n, nr[h], false)));
        }
    }
    return 0;
}

// Tensor-parallel prefill forward whose sequences split into two micro-batches A = sequences [0, seqs_a) = rows [0, rows_a) and B = the
// rest (prefill_slots decides; e->mb): between the embedding and the lm_head the two share nothing -- not even the attention -- so they
// leapfrog through ALL layers, every all-reduce of one running on the side stream under a whole block of the other:
//   engine stream:  attn-block A | attn-block B | ffn A | ffn B | attn-block' A | attn-block' B | ...
//   side stream:                 | AR(wo A)     | AR(wo B)  | AR(w2 A) | AR(w2 B)       | ...
// attn-block = (residual + RMSNorm of the previous layer's w2 sum,) w_qkv, RoPE + K/V quantise-store, flatten, attention, wo;
// ffn = residual + RMSNorm of the wo sum, w1w3 + gated SiLU, w2.  The row-half schedule inside a layer (forward_tail_two_halves) can
// hide AR(wo A) only under wo B and AR(w2 B) only under the next w_qkv A; here every exchange has a third to a half of the other
// micro-batch's layer to hide under (measured with the 1-rank exchange stand-in: profiles/r05_prefill_overlap_emulated_exchange.txt).
// Same kernels and per-row arithmetic as the unsplit forward; dense layers only.
static int forward_layers_two_microbatches(tm_engine* e, int M, int nseq, int max_q_len, int max_k_len, int kflat_stride, float scale_log2)
{
    const tm_model_config& m  = e->cfg.model;
    hipStream_t            st = e->stream;
    struct Part {
        int        r0, rows, s0, nseq;
        const int* cu_q;  // counted from the part's first row
    };
    const Part   part[2] = {{0, e->mb.rows_a, 0, e->mb.seqs_a, e->d_cu_q},
                            {e->mb.rows_a, M - e->mb.rows_a, e->mb.seqs_a, nseq - e->mb.seqs_a, e->d_cu_q_b}};
    const int    H = e->hidden, kq = e->q_heads * e->D;
    const float  eps = m.rms_eps;
    hipEvent_t   done[2] = {nullptr, nullptr};
    for (int li = 0; li < m.layers; ++li) {
        Layer& L = e->layers[li];
        TM_REQUIRE(!L.is_moe, "internal: micro-batch schedule on a MoE layer");
        for (int h = 0; h < 2; ++h) {
            const Part&  P   = part[h];
            const size_t off = (size_t)P.r0 * H;
            if (li > 0) {  // this micro-batch's w2 partial sums of the previous layer are summed by now (or the stream waits here)
                TM_TRY(pipe_wait(e, done[h]));
                TM_PROF(P_RES_NORM, TM_TRY(launch_residual_rmsnorm(e->d_x + off, e->d_resid + off, e->d_tmp + off, nullptr, 0, nullptr, L.attn_norm,
                                                                   eps, P.rows, H, st)));
            }
            half_t* const qkv = e->d_qkv + (size_t)P.r0 * e->qkv_n;
            TM_PROF(P_GEMM_QKV, TM_TRY(linear_plain(e, L.qkv, e->d_x + off, H, qkv, e->qkv_n, P.rows, false)));
            KvCacheView cv = cache_view(e, li);
            cv.block_ptrs += (size_t)P.s0 * e->max_blocks_per_seq;  // cu_block_nums[b] = b * max_blocks_per_seq: the part's sequence 0
            TM_PROF(P_KV_STORE, TM_TRY(launch_kv_rope_store(qkv, e->q_heads, P.cu_q, e->d_k_len + P.s0, P.nseq, P.rows, e->d_rope, e->rope_max_pos,
                                                           cv, st)));
            TM_PROF(P_KV_STORE, TM_TRY(launch_flatten_kv(e->d_kflat, e->d_vflat, 1, e->d_cu_koff + P.s0, e->d_k_len + P.s0, P.nseq, max_k_len,
                                                        kflat_stride, cv, st)));
            PrefillAttnParams p{};
            p.q          = qkv;
            p.q_stride   = e->qkv_n;
            p.out        = e->d_attn + (size_t)P.r0 * kq;
            p.k          = e->d_kflat;
            p.vt         = e->d_vflat;
            p.k_stride   = kflat_stride;
            p.cu_q_len   = P.cu_q;
            p.cu_k_off   = e->d_cu_koff + P.s0;   // offsets into the flat K / V^T scratch of the WHOLE forward
            p.k_len      = e->d_k_len + P.s0;
            p.batch      = P.nseq;
            p.max_q_len  = max_q_len;
            p.q_heads    = e->q_heads;
            p.kv_heads   = e->kv_heads;
            p.scale_log2 = scale_log2;
            TM_PROF(P_ATTN, TM_TRY(launch_prefill_attention(p, st)));
            TM_PROF(P_GEMM_O, TM_TRY(linear_plain(e, L.wo, e->d_attn + (size_t)P.r0 * kq, kq, e->d_tmp + off, H, P.rows, false)));
            TM_TRY(allreduce_rows_side(e, P.r0, P.rows, &done[h]));
        }
        for (int h = 0; h < 2; ++h) {
            const Part&  P   = part[h];
            const size_t off = (size_t)P.r0 * H;
            TM_TRY(pipe_wait(e, done[h]));
            TM_PROF(P_RES_NORM, TM_TRY(launch_residual_rmsnorm(e->d_x + off, e->d_resid + off, e->d_tmp + off, nullptr, 0, nullptr, L.ffn_norm, eps,
                                                               P.rows, H, st)));
            TM_PROF(P_GEMM_GATE_UP, TM_TRY(linear_plain(e, L.w13, e->d_x + off, H, e->d_act + (size_t)P.r0 * e->inter, e->inter, P.rows, true)));
            TM_PROF(P_GEMM_DOWN, TM_TRY(linear_plain(e, L.w2, e->d_act + (size_t)P.r0 * e->inter, e->inter, e->d_tmp + off, H, P.rows, false)));
            TM_TRY(allreduce_rows_side(e, P.r0, P.rows, &done[h]));
        }
    }
    for (int h = 0; h < 2; ++h) {
        const size_t off = (size_t)part[h].r0 * H;
        TM_TRY(pipe_wait(e, done[h]));
        TM_PROF(P_RES_NORM, TM_TRY(launch_residual_rmsnorm(e->d_x + off, e->d_resid + off, e->d_tmp + off, nullptr, 0, nullptr, e->final_norm, eps,
                                                           part[h].rows, H, st)));
    }
    return 0;
}

// ---- RMSNorm folded into the decode GEMMs (tp = 1, dense u4 layers, M <= 64; NormFold in tm_kernels.h) ----------------------------
// the tiling of a folded launch: the measured / heuristic pick when its kernel carries the folded epilogue, else the heuristic's
static void fold_tiling(tm_engine* e, const LinearWeight& w, int M, int* shape, int* splits, bool producer)
{
    dec32_pick(w, M, shape, splits);
    if (!dec32_fold_shape_m(*shape, M, producer)) {
        dec32_pick_ex(w, M, shape, splits, false);
    }
    if (!dec32_fold_shape_m(*shape, M, producer)) {
        // M <= 64: the 128-column tile over the whole k range; above (batch 128): the same tile on 32-row blocks for a producer, the 128-row
        // tile for a consumer (every weight unit read once)
        *shape  = M <= 64 ? 0 : (producer ? 7 : 4);
        *splits = 1;
    }
    if (gemm_workspace_bytes(M, w.N, *splits) > e->gemm_ws_bytes) {
        *splits = 1;
    }
}

static bool fold_ok(const tm_engine* e, const Layer& L, int M)
{
    return e->fold_norm != 0 && !L.is_moe && M <= e->fold_max_rows && dec32_supported(L.qkv.w, M) && dec32_supported(L.wo.w, M) && dec32_supported(L.w13.w, M)
           && dec32_supported(L.w2.w, M);
}

// consumer: y = (x . W) * inv[m] (x = r . g of the producing GEMM); ss_tiles == 0: x is already normalised (plain GEMM)
static int linear_fold_consume(tm_engine* e, LinearSlots& l, const half_t* x, int ldx, half_t* y, int ldy, int M, bool gated, int ss_tiles,
                               bool slabs_ok, int* slabs)
{
    int shape, splits;
    fold_tiling(e, l.w, M, &shape, &splits, false);
    const bool mrg = dec32_is_merge_shape(shape);  // split-K merged in the launch: no slab leaves it, whoever consumes
    // a split-K consumer nobody takes slabs from (w1w3; w_qkv without the fused attention prologue): the slices' slabs already carry the row
    // factor (it is applied to the accumulators before ANY epilogue, and a sum of scaled slices is the scaled sum), so the plain reduce launch
    // finishes them -- what the 128-row tile needs at batch 128, where one slice per column tile leaves half the CUs idle (round 6)
    const bool reduce_after = !slabs_ok && !mrg && splits > 1 && M > 64;
    if (!slabs_ok && !mrg && !reduce_after) {
        splits = 1;
    }
    NormFold nf{};
    nf.ss_in    = ss_tiles > 0 ? e->d_ss : nullptr;
    nf.ss_tiles = ss_tiles;
    nf.inv_h    = 1.0f / (float)e->hidden;
    nf.eps      = e->cfg.model.rms_eps;
    nf.tickets  = e->d_tickets;
    int nslab   = 1;
    TM_TRY(launch_linear_dec32(l.w, x, ldx, y, ldy, M, gated, shape, splits, e->d_gemm_ws, &nslab, e->stream, (ss_tiles > 0 || mrg) ? &nf : nullptr));
    if (reduce_after && nslab > 1) {
        TM_TRY(launch_splitk_reduce(y, ldy, e->d_gemm_ws, nslab, M, l.w.N, gated, e->stream));
        nslab = 1;
    }
    if (slabs) {
        *slabs = nslab;
    }
    return 0;
}

// producer: d_resid += x . W ; d_x = d_resid . norm_w (not normalised) ; d_ss = per-tile sums of squares -> *ss_tiles
static int linear_fold_produce(tm_engine* e, LinearSlots& l, const half_t* x, int ldx, int M, const half_t* norm_w, int* ss_tiles)
{
    int shape, splits;
    fold_tiling(e, l.w, M, &shape, &splits, true);
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
int forward(tm_engine* e, const int* d_ids, int M, int nseq, bool decode, int max_q_len, int max_k_len,
                   int kflat_stride, int slot0, const MixedDecode* md)
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
    // tp > 1 prefill through RCCL: the row-wise part of every dense layer as two row halves, all-reduces on the side stream
    e->pipe_events_used = 0;
    const bool  pipe     = !decode && !md && prefill_pipe_ok(e, M);
    const int   Ma       = pipe ? (M / 2 + 63) / 64 * 64 : 0;
    e->pipe_forwards += pipe ? 1 : 0;
    bool two_mb = pipe && e->mb.seqs_a > 0 && e->mb.seqs_a < nseq && e->mb.rows_a > 0 && e->mb.rows_a < M;
    for (int li = 0; two_mb && li < m.layers; ++li) {
        two_mb = !e->layers[li].is_moe;
    }
    if (two_mb) {
        e->pipe_mb_forwards += 1;
        TM_TRY(forward_layers_two_microbatches(e, M, nseq, max_q_len, max_k_len, kflat_stride, scale_log2));
    }
    bool        qkv_done = false;  // the previous layer's two-halves tail already projected this layer's QKV
    int         ss_tiles   = 0;  // > 0: d_x holds r . g of a folded producer, d_ss its sums of squares (the next GEMM applies the row factor)
    for (int li = 0; li < (two_mb ? 0 : m.layers); ++li) {
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
            if (!qkv_done) {
                TM_PROF(P_GEMM_QKV, TM_TRY(linear_plain(e, L.qkv, e->d_x, e->hidden, e->d_qkv, e->qkv_n, M, false)));
            }
            qkv_done = false;
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
        if (pipe && !L.is_moe) {
            Layer* const nxt = li + 1 < m.layers ? &e->layers[li + 1] : nullptr;
            TM_TRY(forward_tail_two_halves(e, L, M, Ma, next_norm, nxt));
            qkv_done = nxt != nullptr;
            continue;
        }
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
            // static batch with tm_engine_set_logprobs: the kept candidates' logprobs of this step join the record of (slot, step)
            // continuous batching with tm_engine_request_logprobs: the record of (slot, this step), handed to the host behind the step
            const bool           lpr_cb = e->sched && e->cb_logprobs_on;
            const bool           lpr    = (e->logprobs_on && !e->sched) || lpr_cb;
            const SampleLogprobs lp = lpr_cb ? SampleLogprobs{e->d_cb_lp_vals, e->d_cb_lp_idx, e->d_cb_lp_num, e->d_cb_lp_sel, kMaxLogProb, nullptr, 0, 1, slot, 1}
                                             : SampleLogprobs{e->d_lpr_vals, e->d_lpr_idx, e->d_lpr_num, e->d_lpr_sel, e->logprobs_n, e->d_step, 1, e->max_new,
                                                              slot, e->max_new};
            TM_PROF(P_SAMPLE, TM_TRY(launch_sample(ids, lpr ? e->d_kept + slot : nullptr, lg, n, V, V, e->d_temp + slot, e->d_topk + slot,
                                                   e->d_topp + slot, e->d_minp + slot, e->d_u + slot, e->d_sample_ws, st, lpr ? &lp : nullptr)));
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

}  // namespace tmk

extern "C" {

// sampling state: device arrays for all slots (allocated on first use), upload of `n` slots starting at slot0
int sampling_upload(tm_engine* e, const tm_sampling* p, int slot0, int n)
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
extern const tm_logits_param kNoLogitsParam = {1.f, 0, 0, {0}, 0, {0}};

int logits_param_check(const tm_logits_param& p)
{
    TM_REQUIRE(p.repetition_penalty > 0.f, "repetition_penalty must be > 0");
    TM_REQUIRE(p.min_new_tokens >= 0, "min_new_tokens must be >= 0");
    TM_REQUIRE(p.n_bad_ids >= 0 && p.n_bad_ids <= TM_MAX_BAD_IDS, "0 <= n_bad_ids <= TM_MAX_BAD_IDS");
    TM_REQUIRE(p.n_stop_ids >= 0 && p.n_stop_ids <= TM_MAX_STOP_IDS, "0 <= n_stop_ids <= TM_MAX_STOP_IDS");
    return 0;
}

int logits_upload(tm_engine* e, const tm_logits_param* p, const int* prompt_len, const int* eos, int slot0, int n)
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
void setup_decode(tm_engine* e, int batch)
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
int prefill_slots(tm_engine* e, const int* const* seq_ids, const int* host_lens, int batch, int slot0, float* ttft_ms,
                         const MixedStep* mix)
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
        // tensor-parallel forward through RCCL: the sequence boundary nearest to half the rows makes two micro-batches that leapfrog
        // through the layers (forward_layers_two_microbatches); none with both sides >= pipe_min_rows / 2 -> the row-half schedule
        e->mb = {};
        std::vector<int> cu_q_b;  // lives until the synchronisation at the end of this iteration, like the vectors above
        static const bool mb_on = !getenv("TM_PIPE_MICROBATCH") || atoi(getenv("TM_PIPE_MICROBATCH")) != 0;  // A/B switch: 0 = row halves only
        if (mb_on && !merge && nseq >= 2 && prefill_pipe_ok(e, tokens)) {
            const int best = prefill_microbatch_split(cu_q.data(), nseq, e->pipe_min_rows);  // scheduler.h
            if (best > 0) {
                cu_q_b.assign(cu_q.begin() + best, cu_q.end());
                for (int& v : cu_q_b) {
                    v -= cu_q[best];
                }
                TM_HIP_CHECK(hipMemcpyAsync(e->d_cu_q_b, cu_q_b.data(), cu_q_b.size() * 4, hipMemcpyHostToDevice, e->stream));
                e->mb.seqs_a = best;
                e->mb.rows_a = cu_q[best];
            }
        }
        // shift the block tables so that slot 0 of this iteration is sequence b0
        uint64_t* saved_ptrs = e->d_block_ptrs;
        e->d_block_ptrs += (size_t)(slot0 + b0) * e->max_blocks_per_seq;
        const MixedDecode md{nd, merge ? mix->k_len : nullptr, merge ? mix->block_ptrs : nullptr, merge ? mix->cu_q : nullptr,
                             merge ? mix->active : nullptr};
        const int rc    = forward(e, e->d_prefill_ids, nd + tokens, nseq, false, max_q, max_k, e->kflat_stride, slot0 + b0,
                                  merge ? &md : nullptr);
        e->d_block_ptrs = saved_ptrs;
        e->mb           = {};
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
    e->logprobs_next = 0;
    e->logprobs_on   = false;
    e->cb_logprobs_on = false;
    e->cb_lp_used     = 0;
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
    e->logprobs_on = false;
    if (e->logprobs_next > 0) {
        // the logprobs come out of the sampling kernels: greedy rows are their top_k = 1 rows (kept = 1: the token with logprob 0,
        // what the reference reports for top_k = 1 as well)
        if (!e->sampling_on) {
            std::vector<tm_sampling> greedy(batch, tm_sampling{1.f, 1, 1.f, 0.f, 0});
            TM_TRY(sampling_upload(e, greedy.data(), 0, batch));
            e->sampling_on = true;
        }
        const size_t records = (size_t)batch * max_new_tokens, entries = records * e->logprobs_next;
        if (records > e->lpr_records || entries > e->lpr_entries || !e->d_kept) {
            TM_HIP_CHECK(hipStreamSynchronize(e->stream));
            (void)hipFree(e->d_lpr_vals), (void)hipFree(e->d_lpr_idx), (void)hipFree(e->d_lpr_num), (void)hipFree(e->d_lpr_sel);
            e->d_lpr_vals = nullptr, e->d_lpr_idx = nullptr, e->d_lpr_num = nullptr, e->d_lpr_sel = nullptr;
            e->lpr_records = e->lpr_entries = 0;
            TM_TRY(dmalloc(&e->d_lpr_vals, entries));
            TM_TRY(dmalloc(&e->d_lpr_idx, entries));
            TM_TRY(dmalloc(&e->d_lpr_num, records));
            TM_TRY(dmalloc(&e->d_lpr_sel, records));
            if (!e->d_kept) {
                TM_TRY(dmalloc(&e->d_kept, (size_t)e->cfg.max_batch_size));
            }
            e->lpr_records = records;
            e->lpr_entries = entries;
        }
        TM_HIP_CHECK(hipMemsetAsync(e->d_lpr_num, 0, records * 4, e->stream));
        e->logprobs_n  = e->logprobs_next;
        e->logprobs_on = true;
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
                     || e->graph_logits != e->logits_on || e->graph_logprobs != e->logprobs_on || e->logprobs_on)) {
        // (with logprobs on the record pointers and the cap are captured kernel arguments: always re-capture)
        (void)hipGraphExecDestroy(e->graph);
        e->graph = nullptr;
    }
    return 0;
}

// hipGraph capture of one decode step.  Steps that contain RCCL calls (tp > 1) are captured too -- a TP = 8 step is
// ~290 launches + 65 collectives, far too many for eager launches -- but defensively: if the capture or the instantiation
// fails (RCCL build without graph support, ...) the engine falls back to eager steps for good instead of failing.
bool graph_enabled(const tm_engine* e)
{
    if (!e->cfg.use_graph) {
        return false;
    }
    if (!e->use_comm) {
        return true;
    }
    return e->graph_comm && !e->graph_comm_failed;  // TM_GRAPH_COMM=0 (read at create) keeps collectives out of graphs
}

int capture_step(tm_engine* e, int (*step)(tm_engine*), hipGraphExec_t* exec)
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
        e->graph_logprobs = e->logprobs_on;
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

int tm_engine_set_logprobs(tm_engine* e, int n)
{
    TM_REQUIRE(e, "null pointer");
    TM_REQUIRE(e->batch == 0 && !e->sched, "set the logprobs count before tm_engine_prefill");
    TM_REQUIRE(n >= 0 && n <= kMaxLogProb, "0 <= logprobs <= TM_MAX_LOGPROBS");
    TM_REQUIRE(n == 0 || !e->use_comm || e->comm || e->p2p_ready,
               "logprobs with tp > 1 gather the logits like stochastic sampling: tm_engine_comm_init or the native communicator first");
    e->logprobs_next = n;
    return 0;
}

int tm_engine_fetch_logprobs(tm_engine* e, float* host_vals, int* host_idx, int* host_num, float* host_sel)
{
    TM_REQUIRE(e && host_vals && host_idx && host_num && host_sel, "null pointer");
    TM_REQUIRE(e->batch > 0 && !e->sched && e->logprobs_on, "no admitted static batch with logprobs (tm_engine_set_logprobs before the prefill)");
    TM_HIP_CHECK(hipStreamSynchronize(e->stream));
    const size_t records = (size_t)e->batch * e->max_new;
    TM_HIP_CHECK(hipMemcpy(host_vals, e->d_lpr_vals, records * e->logprobs_n * 4, hipMemcpyDeviceToHost));
    TM_HIP_CHECK(hipMemcpy(host_idx, e->d_lpr_idx, records * e->logprobs_n * 4, hipMemcpyDeviceToHost));
    TM_HIP_CHECK(hipMemcpy(host_num, e->d_lpr_num, records * 4, hipMemcpyDeviceToHost));
    TM_HIP_CHECK(hipMemcpy(host_sel, e->d_lpr_sel, records * 4, hipMemcpyDeviceToHost));
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

}  // extern "C"
