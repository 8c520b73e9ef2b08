// Internal launcher declarations (host side of every HIP kernel in this library).
#pragma once
#include <stdio.h>
#include "tm_common.h"
#include <vector>

namespace tmk {

// ---- paged KV cache geometry (reference: kernels/attention/block.h:126-219) ----------------
struct KvLayout {
    int kv_heads;
    int head_dim;   // 128
    int block_len;  // 64
    int bits;       // 16 | 8 | 4
    __host__ __device__ int token_data_size() const { return bits * head_dim / 8; }
    __host__ __device__ int token_param_size() const { return bits < 16 ? 4 : 0; }
    __host__ __device__ int head_data_size() const { return block_len * token_data_size(); }
    __host__ __device__ int head_param_size() const { return block_len * token_param_size(); }
    __host__ __device__ int layer_size() const
    {
        return kv_heads * 2 * head_data_size() + kv_heads * 2 * head_param_size();
    }
    __host__ __device__ int k_data(int head, int ti) const { return head * 2 * head_data_size() + ti * token_data_size(); }
    __host__ __device__ int v_data(int head, int ti) const { return k_data(head, ti) + head_data_size(); }
    __host__ __device__ int k_param(int head, int ti) const
    {
        return kv_heads * 2 * head_data_size() + head * 2 * head_param_size() + ti * token_param_size();
    }
    __host__ __device__ int v_param(int head, int ti) const { return k_param(head, ti) + head_param_size(); }
};

struct KvCacheView {
    const uint64_t* block_ptrs;     // device array of block base addresses (char**)
    const int*      cu_block_nums;  // [B+1] prefix offsets into block_ptrs
    int             block_stride = 0;  // > 0: the table is rectangular, sequence b starts at b * block_stride (cu_block_nums[b] says the
                                       // same): the MFMA decode kernel then fetches a sequence's block pointers with ONE load that
                                       // depends on nothing but kernel arguments (64 pointers, one per lane) instead of two dependent
                                       // hops (prefix offset -> pointer) in front of every cache block
    int64_t         layer_offset;   // bytes: layer * layout.layer_size()
    KvLayout        layout;
};

// ---- norm.hip ---------------------------------------------------------------------------
int launch_rmsnorm(half_t* y, const half_t* x, const half_t* w, float eps, int M, int H, hipStream_t st);
int launch_residual_rmsnorm(half_t* y, half_t* resid, const half_t* hidden, const float* partial, int splits,
                            const half_t* bias, const half_t* w, float eps, int M, int H, hipStream_t st);

// ---- kv_cache.hip -----------------------------------------------------------------------
// RoPE(q,k) in fp16 from a (cos,sin) table, quantise K/V of the new tokens and scatter them into the
// paged cache.  q is rotated in place inside the qkv buffer.
int launch_kv_rope_store(half_t* qkv, int q_heads, const int* cu_q_len, const int* k_len, int batch, int total_tokens,
                         const half2_t* cos_sin, int max_pos, KvCacheView cache, hipStream_t st);
// Gather + dequantise (two-rounding "flatten" form) the whole context of every sequence into linear
// scratch: K [kv_heads][k_stride][D]; V either [kv_heads][k_stride][D] or transposed [kv_heads][D][k_stride].
int launch_flatten_kv(half_t* k_out, half_t* v_out, int transpose_v, const int* cu_k_off, const int* k_len, int batch,
                      int max_k_len, int k_stride, KvCacheView cache, hipStream_t st);

// ---- attention_decode.hip ---------------------------------------------------------------
struct DecodeAttnParams {
    const half_t* q;         // [B][q_stride] roped queries (head h at h*D)
    int           q_stride;  // elements between consecutive tokens
    half_t*       out;       // [B][q_heads*D]
    const int*    k_len;     // [B] context length including the new token
    int           batch;
    int           q_heads;
    float         scale_log2;  // softmax_scale * log2(e)
    int           splits;
    float*        partial_o;   // [B][q_heads][splits][D]
    float*        partial_ml;  // [B][q_heads][splits][2]
    KvCacheView   cache;
    // Fused prologue (int8 MFMA kernel only): q/k/v of the new token are taken straight from the qkv GEMM output --
    // `qkv_splits` fp32 split-K slabs [qkv_splits][B][qkv_n] (or the fp16 [B][qkv_n] result when qkv_splits == 0) --
    // rotated, and K/V are quantised + stored by the attention kernel itself (the reference decode kernel does the
    // same: attention_universal.h:168-330).  Replaces splitk_reduce + kv_rope_store on the decode path.
    const float*   qkv_slabs = nullptr;
    const half_t*  qkv_f16   = nullptr;
    int            qkv_splits = 0;
    int            qkv_n      = 0;
    const half2_t* cos_sin    = nullptr;
    int            max_pos    = 0;
    uint64_t*      dbg        = nullptr;  // optional per-workgroup timing stamps (tm_debug_set_gemm_trace)
};
int launch_decode_attention(const DecodeAttnParams& p, hipStream_t st);
int launch_decode_attention_i8_mfma(const DecodeAttnParams& p, hipStream_t st);  // attention_decode_mfma.hip
size_t decode_attention_workspace_bytes(int batch, int q_heads, int head_dim, int splits);

// ---- attention_prefill.hip --------------------------------------------------------------
struct PrefillAttnParams {
    const half_t* q;         // [T][q_stride] roped
    int           q_stride;
    half_t*       out;       // [T][q_heads*D]
    const half_t* k;         // [kv_heads][k_stride][D]
    const half_t* vt;        // [kv_heads][D][k_stride]
    int           k_stride;
    const int*    cu_q_len;  // [B+1]
    const int*    cu_k_off;  // [B+1] start of every sequence in k / vt (multiple of 64)
    const int*    k_len;     // [B]
    int           batch;
    int           max_q_len;
    int           q_heads;
    int           kv_heads;
    float         scale_log2;
};
int launch_prefill_attention(const PrefillAttnParams& p, hipStream_t st);

// ---- gemm_w4a16.hip ---------------------------------------------------------------------
struct LinearWeight {
    int       K = 0, N = 0, group = 128;
    int       type = 0;          // 0 = u4 (AWQ), 1 = f16 dense, 2 = fp8 e4m3 with 128x128 block scales
    int       role = 0;          // key of the measured dispatch table next to (K, N, M): 0 = any (operator-level handles, imported
                                 // tables without a role column), 1 w_qkv, 2 wo, 3 w1w3, 4 w2 -- two roles of equal (K, N) are timed
                                 // with DIFFERENT consumers by the tuner and keep separate winners
    void*     packed = nullptr;  // fragment-ordered weights (nullptr for a u4 linear prepared p32_only)
    uint32_t* sz     = nullptr;  // fragment-ordered (s, -z*s) half2 pairs (u4), (s, 0) (fp8)
    size_t    packed_bytes = 0, sz_bytes = 0;
    // u4 only, N % 32 == 0: the decode kernel's layout (gemm_decode.hip, "P32": 2176-byte units of 32 columns x 128 k
    // with their (s, -z*s) pairs); `packed` / `sz` stay the layout of the M > 64 kernels
    void*     packed32 = nullptr;
    size_t    packed32_bytes = 0;
    // fp8 only, N % 32 == 0: the fp8 x fp8 kernel's layout (gemm_fp8.hip, "P8": 4224-byte units of 32 columns x 128 k with
    // their even / odd column block scales)
    void*     packed8 = nullptr;
    size_t    packed8_bytes = 0;
    // u4 only, optional (linear_weight_build_f16_image; the engine builds it for dense linears unless TM_PREFILL_F16_IMAGE=0): the fp16
    // [N][K] image of the dequantised weights -- bit for bit the operand the fused tiles build on chip -- that prefill-sized forwards
    // contract on the matrix pipe without any dequantisation work in the loop (gemm_prefill_f16.hip, shape kShapeF16).  2 bytes per
    // parameter of HBM next to the 0.53 the decode path streams: 14 GB for an 8 B model on a 288 GB device.
    half_t*   image16 = nullptr;
};
struct GemmConfig {
    int nt;      // n-tiles (16 cols) per wave: 1,2,4
    int splits;  // split-K
    int waves;   // waves per workgroup: 4 or 8
    int kphases; // split-K inside the workgroup: 1 or 2 (8 waves only)
    int kstage;  // max k-blocks per barrier (0 = auto: 4)
    int d32_shape = -1;  // >= 0: a P32 kernel (gemm_decode.hip / gemm_decode_lc.hip / gemm_prefill.hip) with this workgroup shape
    unsigned* tickets = nullptr;  // arrival counters (zero between launches) of the kShapeMerge shapes: >= ceil(N / 64) * ceil(M / 32) words
};
// (shape 10 was "dequantise + the vendor library's fp16 GEMM" in round 3: removed -- no vendor GEMM on any path of this library)
constexpr int kShapePre256     = 12;   // gemm_prefill.hip: 256 x 256 tiles, weights dequantised once per workgroup tile through LDS (M > 64)
constexpr int kShapeF16        = 13;   // gemm_prefill_f16.hip: 256 x 256 tiles over the resident fp16 image, both operands by LDS-DMA (M > 64; LinearWeight::image16)
constexpr int kShapeLC         = 11;   // gemm_decode_lc.hip: 8 consumer + 4 loader waves, 128 columns x M <= 64 rows per workgroup
// kShapeMerge + s (s = 0..3, 6..9; round 6): the decode tile s whose split-K slices are merged INSIDE the launch by the last-arriving
// slice of each column tile, for the fp16 / gated-SiLU epilogues (the folded-norm producers, epilogue 3, always merge that way): a
// gated w1w3 can then split K across CUs (256-column tiles, half the activation bytes per CU) without slabs leaving the launch.
// Needs arrival counters: NormFold::tickets (engine) / the tail of the workspace (tm_linear_forward).
constexpr int kShapeMerge      = 16;
// 256 columns x 2 k-phases x 16 waves on 2-k-block stages (gemm_dec32_kernel<MH, 8, 2, 2, 2>; round 6): the 256-column decode tile that does not
// spill (shape 1 = the same wave layout on 4-k-block stages with a 4-deep ring: 60 B / lane of scratch at the 128-register cap)
constexpr int kShapeWide2      = 10;
inline bool dec32_is_merge_shape(int shape)
{
    const int b = shape - kShapeMerge;
    return (b >= 0 && b <= 3) || (b >= 6 && b <= 10);
}
// Load-time repack (reference: LinearWeight::prepare, models/linear_weight.cc:101-324)
// p32_only: build ONLY the P32 image (gemm_decode.hip) -- for linears that every M dispatches to those kernels
// (dec32_serves_every_m): the 16-column image of gemm_kernel would never be read.
int    linear_weight_prepare_u4(LinearWeight& w, const int32_t* qweight /*[K][N/8]*/, const half_t* scales,
                                const half_t* zeros, hipStream_t st, bool p32_only = false);
bool   dec32_serves_every_m(int K, int N);  // dense u4 linear, N % 32 == 0, K % 128 == 0, TM_GEMM_D32 / _PREFILL on
int    linear_weight_prepare_f16(LinearWeight& w, const half_t* weight /*[K][N]*/, hipStream_t st);
int    linear_weight_prepare_fp8(LinearWeight& w, const uint8_t* weight /*[K][N] e4m3*/, const float* block_scales /*[K/128][ceil(N/128)]*/,
                                 bool gated_scales /* w1w3: scale row = [w1 blocks | w3 blocks] for interleaved columns */, hipStream_t st);
void   linear_weight_free(LinearWeight& w);
int    linear_weight_build_f16_image(LinearWeight& w, hipStream_t st);  // u4 linear with its P32 image -> w.image16 ([N][K] fp16)
int    launch_dequant_p32_f16(half_t* out_nk /*[N][K]*/, const LinearWeight& w, hipStream_t st);  // gemm_decode.hip: the operand as an fp16 image
size_t gemm_workspace_bytes(int M, int N, int splits);
int    launch_splitk_reduce(half_t* y, int ldy, const float* partial, int splits, int M, int N, bool gated, hipStream_t st);
GemmConfig gemm_pick_config(const LinearWeight& w, int M);  // a P32 kernel when it applies, else ...
GemmConfig gemm_pick_config_general(const LinearWeight& w, int M);  // ... the tiling of gemm_kernel (gemm_w4a16.hip)
// y[M][N (or N/2 if gated)] = x[M][K] . W ; if cfg.splits > 1 fp32 slabs land in `workspace` and, unless
// `defer_reduce`, a reduce kernel writes y.
int launch_linear(const LinearWeight& w, const half_t* x, int ldx, half_t* y, int ldy, int M, bool gated_silu,
                  GemmConfig cfg, float* workspace, bool defer_reduce, int* slabs, hipStream_t st);

// ---- gemm_decode.hip: W4A16 decode GEMM (M <= 64), 32x32x16 MFMA, 16 waves per CU ------------------------------
size_t p32_bytes(int K, int N);
int    launch_repack_p32(void* out, const int32_t* qweight, const half_t* scales, const half_t* zeros, int K, int N, hipStream_t st);
bool   dec32_supported(const LinearWeight& w, int M);
void   dec32_pick(const LinearWeight& w, int M, int* shape_out, int* splits_out);  // measured table first, then the heuristic
void   dec32_pick_ex(const LinearWeight& w, int M, int* shape_out, int* splits_out, bool use_table);
// measured dispatch table of the decode linears: key (K, N, M <= 64) -> (shape, splits)
void   dec32_table_set(int K, int N, int M, int shape, int splits, int role = 0);
bool   dec32_table_get(int K, int N, int M, int* shape, int* splits, int role = 0);  // the role's entry, else the role-0 entry
void   dec32_table_clear();
int    dec32_table_export(const char* path);
int    dec32_table_import(const char* path);
int    dec32_candidates(const LinearWeight& w, int M, int (*out)[2], int cap);
int    dec32_m_bucket(int M);  // table key of a forward with M rows: M itself up to 256, then 512, 1024, ... 8192
// Measured dispatch of everything that is NOT a P32 kernel (reference: gemm.cu:92-224 caches every problem the
// warm-up of turbomind.cc:363-487 meets, not only the dense u4 ones).  One table, key (kind, role, K, N, M bucket):
//   kind kGenDense + type   : gemm_kernel of a dense linear (fp16 lm_head, e4m3 weight-only, u4 with N % 32 != 0) -> {nt, splits, waves, kphases}
//   kind kGenGrouped + type : row-tile height of the grouped expert GEMMs (u4: 16 / 32 / 64 rows, fp8 MFMA: 32 / 64) -> {rows, 0, 0, 0}
// Text form (same file as the P32 table): `G kind role K N M a b c d`.  role 5 = lm_head; grouped entries use role 0.
constexpr int kGenDense   = 16;
constexpr int kGenGrouped = 32;
void   gen_table_set(int kind, int role, int K, int N, int M, const int v[4]);
bool   gen_table_get(int kind, int role, int K, int N, int M, int v[4]);  // the role's entry, else the role-0 entry
void   gen_table_erase(int kind, int role, int K, int N, int M);
void   gen_table_clear();                   // dec32_table_clear() calls it: both tables travel and reset together
void   gen_grouped_rows_override(int rows); // tuner only: the grouped GEMMs launched by THIS thread use `rows`-row tiles (0: off)
int    gen_table_export_lines(FILE* f);      // appends the G lines; returns their number
bool   gen_table_import_line(const char* line);  // a `G ...` line; false = not valid (ignored)
int    gen_dense_candidates(const LinearWeight& w, int M, size_t workspace_bytes, GemmConfig* out, int cap);
int    gen_grouped_candidates(const LinearWeight& proto, int m_cap, int* rows_out, int cap);
// RMSNorm folded into the neighbouring decode GEMMs (round 5; gemm_decode_common.h has the arithmetic).  One struct, two roles:
//   producer (resid != nullptr): the row-parallel GEMM (wo / w2) adds its fp16 output to the residual stream in its own epilogue
//     (split-K: the last-arriving slice of a column tile sums the slices' slabs in slice order), writes y = h(f32(r) * f32(g)) and
//     ss_out[tile][m] = partial sum of f32(r)^2 over the tile's columns; tiles_out = column tiles written (set by the launcher)
//   consumer (ss_in != nullptr): the GEMM fed with that y multiplies its fp32 accumulators by 1 / sqrt(sum_tiles ss / H + eps)
// Replaces: invokeResidualBiasRMSNorm behind wo / w2 (rms_norm.cu:286-362, unified_decoder.cc:226,278,328) -- the residual stream is
// bit-identical to the unfused sequence; the normalised activations differ by one fp16 rounding (applied after the contraction).
struct NormFold {
    const float*  ss_in    = nullptr;
    int           ss_tiles = 0;
    float         inv_h = 0.f, eps = 0.f;
    half_t*       resid   = nullptr;
    const half_t* norm_w  = nullptr;
    float*        ss_out  = nullptr;
    unsigned*     tickets = nullptr;
    int           tiles_out = 0;
};
constexpr int kFoldMaxRows = 128;  // rows of a forward whose RMSNorms fold into the decode GEMMs (round 6: BASELINE config 3 runs batch 128)
bool   dec32_fold_shape_m(int shape, int M, bool producer);  // can (shape, M) run as a folded producer / consumer?
bool   dec32_fold_shape(int shape);  // shapes whose kernel carries the folded-norm epilogue / prologue (0..3, 6..9)
int    launch_linear_dec32(const LinearWeight& w, const half_t* x, int ldx, half_t* y, int ldy, int M, bool gated_silu, int shape,
                           int splits, float* workspace, int* slabs_out, hipStream_t st, NormFold* nf = nullptr);

// ---- gemm_fp8.hip: fp8 x fp8 linear on v_mfma_f32_32x32x16_fp8_fp8 (activations quantised per row and 128 channels) ---
size_t p8_bytes(int K, int N);
int    launch_repack_p8(void* out, const uint8_t* weight, const float* block_scales, int K, int N, bool gated, hipStream_t st);
bool   fp8_mfma_supported(const LinearWeight& w);
size_t fp8_act_workspace_bytes(int rows, int K);
int    launch_quant_fp8_rows(uint8_t* xq, float* sx, const half_t* x, int ldx, int M, int K, int ldsx, hipStream_t st);
int    launch_linear_fp8(const LinearWeight& w, const uint8_t* xq, const float* sx, int ldsx, half_t* y, int ldy, int M, bool gated_silu,
                         int splits, float* workspace, int* slabs_out, hipStream_t st);
int    launch_linear_fp8_grouped(const LinearWeight& proto, const void* d_groups, int E, const uint8_t* xq, const float* sx, int ldsx,
                                 int x_rows, half_t* y, int ldy, int m_cap, int m_hint, bool gated_silu, const int* seg,
                                 const int* row_idx, hipStream_t st);

// ---- comm_p2p.hip: fused one-shot all-reduce + residual + RMSNorm over peer-mapped buffers (TM_COMM=native) ------------
int launch_p2p_allreduce_norm(half_t* const* data, uint32_t* const* flags, int tp, int me, uint32_t* state, size_t tile,
                              const half_t* partial, half_t* y, half_t* resid, const half_t* weight, float eps, int M, int H,
                              hipStream_t st);
// two-shot form (reduce-scatter, norm on the owned row slice, all-gather by push) for messages of MBs; region = fp16 elements of
// one in2 / out2 region; the residual stream is updated for the rank's own slice only
int launch_p2p_allreduce_norm_2shot(half_t* const* in2, half_t* const* out2, uint32_t* const* flags, int tp, int me, uint32_t* state,
                                    size_t region, const half_t* partial, half_t* y, half_t* resid, const half_t* weight, float eps, int M,
                                    int H, hipStream_t st);
int launch_p2p_allreduce_norm_rows(half_t* const* rdata, uint32_t* const* rflags, int rows_cap, int tp, int me, uint32_t* state, size_t tile,
                                   const half_t* partial, half_t* y, half_t* resid, const half_t* weight, float eps, int M, int H, hipStream_t st);
void p2p_timeout_cap_ms(long ms);  // > 0: cap the peer-wait bound (bring-up self-test); 0: TM_P2P_TIMEOUT_MS again
int p2p_allreduce_capacity(int threads, bool one_vec);  // token rows per launch: workgroups resident at once on this device
// dst_stride_words: 32-bit words between the destinations of consecutive ranks (0 = words: dst is [tp][words])
int launch_p2p_allgather(half_t* const* data, uint32_t* const* flags, int tp, int me, uint32_t* state, size_t tile, const void* src,
                         void* dst, int words, hipStream_t st, size_t dst_stride_words = 0);

// sampling.hip: temperature / top-k / top-p / min-p sampling without a sort (see the file header)
size_t sample_workspace_bytes(int batch);
// logprobs of the kept candidates next to the draw (sample_logprobs_kernel): record r = (*step) * step_stride + (row0 + b) * row_stride
// of vals / idx ([records][cap]), num / sel ([records]); step == nullptr: 0; records of steps >= max_steps are dropped
constexpr int kMaxLogProb = 1024;  // = TM_MAX_LOGPROBS (src/turbomind/utils/constant.h:7)
struct SampleLogprobs {
    float*     vals;
    int*       idx;
    int*       num;
    float*     sel;
    int        cap;
    const int* step;
    int        step_stride, row_stride, row0, max_steps;
};
int    launch_sample(int* out_ids, int* kept_out, const half_t* logits, int batch, int V, int ld, const float* temperature,
                     const int* top_k, const float* top_p, const float* min_p, const float* uniform, void* workspace,
                     hipStream_t st, const SampleLogprobs* lp = nullptr);
int    launch_sample_uniform(float* u, const uint64_t* seeds, const int* counters, int batch, hipStream_t st);
float  philox_uniform_host(uint64_t seed, uint32_t ctr);

// logits processors (sampling.hip): persistent per-slot "token seen" bitmask, repetition penalty, banned ids,
// min-length ban of the end ids.  ban = [batch][kMaxBadIds], end = [batch][kMaxEndIds], padded with -1.
constexpr int kMaxBadIds = 32;  // = TM_MAX_BAD_IDS
constexpr int kMaxEndIds = 9;   // eos id + TM_MAX_STOP_IDS
// active (decode rows only, cu_q == nullptr): rows with active[row] == 0 are skipped
int launch_seen_update(uint32_t* seen, int words, const int* ids, const int* cu_q, int nseq, int n_tokens, int vocab,
                       hipStream_t st, const int* active = nullptr);
int launch_logits_process(half_t* logits, int batch, int V, int ld, int vocab_offset, const uint32_t* seen, int words,
                          const float* rep, const int* ban, const int* end, const int* k_len, const int* min_len,
                          hipStream_t st);

struct MoeBlock {
    int                       hidden = 0, inter = 0, experts = 0, top_k = 0;
    bool                      norm_topk    = true;
    float                     routed_scale = 1.f;
    half_t*                   gate         = nullptr;  // device fp16 [hidden][experts]
    std::vector<LinearWeight> w13, w2;                 // per expert: gated (gate_j, up_j)-interleaved [H][2I], [I][H]
    void *                    groups13 = nullptr, *groups2 = nullptr;
    void *                    groups13_p8 = nullptr, *groups2_p8 = nullptr;  // fp8 experts: P8 unit pointers (gemm_fp8.hip)
};
size_t moe_workspace_bytes(const MoeBlock& m, int tokens);
int    moe_prepare(MoeBlock& m, hipStream_t st);
int    moe_forward(const MoeBlock& m, half_t* out, int ldo, const half_t* x, int ldx, int tokens, void* workspace, int* topk_ids_out,
                   float* topk_w_out, hipStream_t st);
void   moe_free(MoeBlock& m);
// mixture of experts (moe.hip, grouped GEMM in gemm_w4a16.hip)
int moe_build_groups(void** d_groups, const LinearWeight* experts, int E, hipStream_t st);
int launch_linear_grouped(const LinearWeight& proto, const void* d_groups, int E, const half_t* x, int ldx, int x_rows,
                          half_t* y, int ldy, int m_cap, int m_hint, bool gated_silu, const int* seg, const int* row_idx,
                          hipStream_t st);
int launch_moe_gate(int* topk_ids, float* topk_w, float* logits_out, const half_t* x, int ldx, const half_t* wg, int T, int H,
                    int E, int k, bool norm_topk, float routed_scale, hipStream_t st);
int launch_moe_route(int* offsets, int* f2n, int* en2f, const int* topk_ids, int T, int E, int k, hipStream_t st);
int launch_moe_combine(half_t* out, int ldo, const half_t* y, int ldy, const float* topk_w, const int* en2f, int T, int H, int k,
                       hipStream_t st);

extern uint64_t* g_gemm_dbg;  // gemm_w4a16.hip: optional per-workgroup timing stamps (tm_debug_set_gemm_trace)
// The trace buffer holds kTraceMaxWorkgroups x 8 stamps; a launch with more workgroups is not traced.  (The only unbounded
// device write the library had: a trace buffer sized for one kernel left set while a larger grid ran -- the probable origin of
// the one-off "memory access fault" of a round-2 diagnostic script; tests/test_gpu_fullsize.py::..._with_canaries is clean.)
constexpr size_t kTraceMaxWorkgroups = 8192;
// Arena mode (tm_debug_trace_arena, round 5): every traced launch gets ITS OWN region of a caller-sized buffer, handed out in
// launch order, and a host-side record (tag, grid, offset) -- so that the launches of a captured hipGraph keep distinct regions
// and one replay leaves the stamps of every kernel of the step (tools/fixed_cost_table.py).  Allocation happens on the host at
// launch / capture time; launches that no longer fit are not traced.
struct TraceRecord {
    char   tag[24];
    int    gx, gy, gz;
    size_t offset_wgs;
};
uint64_t* trace_arena_alloc(size_t workgroups, const char* tag, int gx, int gy, int gz);  // nullptr: no arena / full
bool      trace_arena_active();
void      trace_arena_set(uint64_t* base, size_t cap_wgs);
size_t    trace_arena_records(char* out, size_t cap);
inline uint64_t* gemm_trace_for(size_t workgroups, const char* tag = "", int gx = 0, int gy = 0, int gz = 0)
{
    if (trace_arena_active()) {
        return trace_arena_alloc(workgroups, tag, gx, gy, gz);
    }
    return workgroups <= kTraceMaxWorkgroups ? g_gemm_dbg : nullptr;
}

// ---- misc.hip ---------------------------------------------------------------------------
int launch_embedding(half_t* out, const half_t* table, const int* ids, int T, int H, int vocab, hipStream_t st);
int launch_fill_uniform_f16(half_t* out, size_t n, float amp, uint32_t seed, hipStream_t st);  // tuner stand-in activations
int launch_argmax(int* out_ids, half_t* out_val, const half_t* logits, int B, int V, int ld, int id_offset,
                  hipStream_t st);
int launch_gather_rows(half_t* out, const half_t* in, const int* rows, int n, int H, hipStream_t st);
int launch_silu_mul(half_t* out, const half_t* gate_up, int M, int inter, hipStream_t st);
int launch_quantize_groupwise_u4(int32_t* qweight, half_t* scales, half_t* zeros, half_t* dequant, const half_t* w,
                                 int K, int N, int group, hipStream_t st);

}  // namespace tmk
