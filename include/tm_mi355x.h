/*
 * tm_mi355x.h -- C ABI of libtm_mi355x.so: the MI355X-native (gfx950, hand-written HIP) replacement for the
 * TurboMind quantized decode hot path of InternLM/lmdeploy.
 *
 * Every entry point is `extern "C"`, takes plain pointers / sizes (device pointers unless noted "host"),
 * never throws and never aborts on the request path: it returns a tm_status code, and tm_last_error()
 * describes the last failure on the calling thread.  Streams are hipStream_t passed as void*.
 *
 * What each group replaces in the reference (paths relative to the lmdeploy checkout, v0.16.0):
 *   engine  : pybind11 module `_turbomind` -- TurboMind.create / create_context / create_root / process_weight /
 *             create_engine / ModelRequest.forward  (src/turbomind/python/bind.cpp:743-793,938-999;
 *             src/turbomind/turbomind.cc:159-361), weight slots = Module/Param protocol
 *             (lmdeploy/turbomind/builders/_base.py:72-99).
 *   linear  : `_tm.LinearWeight` + `_tm.LlamaLinear.forward_dense` + `tm.QuantizeGroupwise`
 *             (src/turbomind/python/linear_bind.cpp:111-137,265-294), the operator-level surface the
 *             reference's tests/turbomind/linear fixtures drive.
 *   kernels : invokeRMSNorm / invokeResidualBiasRMSNorm (kernels/norm/rms_norm.cu), invokeProcessKV_v2_ /
 *             invokeFlattenKV_v2_ (kernels/attention/kv_cache_utils_v2.h:36-112), dispatchDecoding /
 *             dispatchAttention (kernels/attention/decoding.cu, attention.cu), embedding + greedy sampling.
 */
#ifndef TM_MI355X_H
#define TM_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Request status codes == src/turbomind/engine/request.h:120-131 */
enum tm_status {
    TM_OK        = 0,
    TM_INVALID   = 1,
    TM_CONFLICT  = 2,
    TM_FAIL      = 5,
    TM_TOO_LONG  = 6,
    TM_FINISH    = 7,
    TM_CANCEL    = 8,
    TM_NO_QUEUE  = 10,
    TM_OOM       = 11
};

typedef void* tm_stream_t; /* hipStream_t; NULL = default stream */

int         tm_version(void);
const char* tm_last_error(void);
/* number of visible gfx950 devices (0 without a GPU); never fails */
int         tm_device_count(void);

/* ----------------------------------------------------------------------------------------------
 * Paged KV cache view (kernels/attention/block.h:126-219, block_iterator.h:9-100)
 * --------------------------------------------------------------------------------------------*/
typedef struct tm_kv_cache {
    const uint64_t* block_ptrs;    /* device: base address of every cache block (char**)           */
    const int*      cu_block_nums; /* device [batch+1]: first block of each sequence in block_ptrs */
    int64_t         layer_offset;  /* bytes: layer * tm_kv_layer_size()                            */
    int             kv_heads;      /* local kv heads                                               */
    int             head_dim;      /* 128                                                          */
    int             block_len;     /* 64 tokens (lmdeploy/messages.py:323)                         */
    int             bits;          /* quant_policy: 16 (none) | 8 | 4 (lmdeploy/messages.py:20-27) */
} tm_kv_cache;

/* bytes of one layer inside a block: kv_heads*2*block_len*(head_dim*bits/8) + kv_heads*2*block_len*4*[bits<16] */
int64_t tm_kv_layer_size(int kv_heads, int head_dim, int block_len, int bits);

/* ----------------------------------------------------------------------------------------------
 * Operator level
 * --------------------------------------------------------------------------------------------*/
/* y = rmsnorm(x) * w          x,y fp16 [M,H], w fp16 [H]                     (rms_norm.cu:21-138) */
int tm_rmsnorm(void* y, const void* x, const void* w, float eps, int M, int H, tm_stream_t st);
/* r <- r + hidden (+ bias), y <- rmsnorm(r) * w ; residual stream accumulates in fp16 (rms_norm.cu:286-362).
 * If `partial` != NULL, hidden is given as `splits` fp32 split-K slabs [splits][M][H] (hidden must be NULL). */
int tm_residual_rmsnorm(void* y, void* resid, const void* hidden, const float* partial, int splits,
                        const void* bias, const void* w, float eps, int M, int H, tm_stream_t st);

/* (cos,sin) fp16 table [max_pos][rope_dim/2][2] on the HOST (rotary_embedding.h:11-52, attention_weight.cc:37-93).
 * rope_type: 0 default, 1 linear, 2 llama3.  The caller uploads it. */
int tm_rope_table(void* host_out, int max_pos, int rope_dim, float base, int rope_type, float factor,
                  float low_freq_factor, float high_freq_factor, int original_max_position);

/* RoPE(q,k) + quantise-and-store K/V of the new tokens into the paged cache (ProcessKV_v2 / decode prologue).
 * qkv fp16 [total_tokens][(q_heads + 2 kv_heads)*128]; q is rotated in place.
 * cu_q_len [batch+1], k_len [batch] = context length of each sequence AFTER adding its new tokens (device).
 * cos_sin may be NULL (no rotation). */
int tm_kv_rope_store(void* qkv, int q_heads, const int* cu_q_len, const int* k_len, int batch, int total_tokens,
                     const void* cos_sin, int max_pos, const tm_kv_cache* cache, tm_stream_t st);
/* FlattenKV_v2: dequantise the whole context into linear fp16 scratch.  k_out [kv_heads][k_stride][128];
 * v_out the same, or transposed [kv_heads][128][k_stride] when transpose_v != 0.  Sequence b starts at
 * cu_k_off[b] (device; must be 64-aligned when transposing). */
int tm_flatten_kv(void* k_out, void* v_out, int transpose_v, const int* cu_k_off, const int* k_len, int batch,
                  int max_k_len, int k_stride, const tm_kv_cache* cache, tm_stream_t st);

/* Paged flash-decode, one query token per sequence (dispatchDecoding).  q fp16 [batch][q_stride] already
 * rotated (head h at h*128), out fp16 [batch][q_heads*128].  splits >= 1; workspace of
 * tm_decode_attention_workspace() bytes needed when splits > 1.  softmax_scale <= 0 -> 1/sqrt(128).
 * Every k_len[b] >= 1 and the first block-table entry of every sequence is a mapped block (the engine parks free batch slots on a
 * dummy block with k_len = 1); a k_len = 0 entry reads that first block and produces no meaningful row. */
size_t tm_decode_attention_workspace(int batch, int q_heads, int splits);
int    tm_decode_attention(void* out, const void* q, int q_stride, const int* k_len, int batch, int q_heads,
                           float softmax_scale, int splits, void* workspace, const tm_kv_cache* cache,
                           tm_stream_t st);
/* Decode attention with the reference decode kernel's fused prologue (attention_universal.h:168-330: the decode
 * kernel itself applies RoPE to q/k and quantises + stores the new K/V).  int8 / int4 KV.  Input is the raw QKV
 * projection of the new token of every sequence: qkv_splits == 0 -> `qkv` is fp16 [batch][qkv_n]; qkv_splits >= 1
 * -> `qkv` is fp32 split-K slabs [qkv_splits][batch][qkv_n] which are summed in order and rounded to fp16 first.
 * Row layout [Q heads | K heads | V heads] x 128.  k_len INCLUDES the new token.  Cache bytes/params written are
 * bit-identical to tm_kv_rope_store; `out` equals tm_decode_attention on that cache. */
int tm_decode_attention_fused(void* out, const void* qkv, int qkv_splits, int qkv_n, const void* cos_sin, int max_pos,
                              const int* k_len, int batch, int q_heads, float softmax_scale, int splits,
                              void* workspace, const tm_kv_cache* cache, tm_stream_t st);
/* Causal prefill attention over flattened KV (dispatchAttention).  vt is the transposed V of tm_flatten_kv. */
int tm_prefill_attention(void* out, const void* q, int q_stride, const void* k, const void* vt, int k_stride,
                         const int* cu_q_len, const int* cu_k_off, const int* k_len, int batch, int max_q_len,
                         int q_heads, int kv_heads, float softmax_scale, tm_stream_t st);

int tm_embedding(void* out, const void* table, const int* ids, int tokens, int hidden, int vocab, tm_stream_t st);
/* greedy top-1 on fp32-cast logits (generation/sampling.cc:92-183); out_val (fp16 [batch]) may be NULL */
int tm_argmax(int* out_ids, void* out_val, const void* logits, int batch, int vocab, int ld, tm_stream_t st);
/* Stochastic sampling (generation/sampling.cc:92-183, logits_processor.cc:105-112, kernels/sampling_*.cu): per row
 * logits / temperature -> keep the top_k most probable (top_k <= 0: all) -> softmax -> keep the shortest prefix whose
 * cumulative probability exceeds top_p (>= 1: all) -> drop p < min_p * p_max -> renormalise -> first candidate whose
 * inclusive prefix sum exceeds uniform[b] (in [0,1)).  Candidates are ordered by descending probability, ties by
 * ascending token id.  Per-row device arrays temperature / top_k / top_p / min_p may be NULL (1, all, 1, 0).
 * kept_out (device int[batch], may be NULL) receives the number of surviving candidates.  workspace:
 * tm_sample_workspace(batch) bytes, ZERO on first use (every call leaves it zero again). */
size_t tm_sample_workspace(int batch);
int    tm_sample(int* out_ids, int* kept_out, const void* logits, int batch, int vocab, int ld, const float* temperature,
                 const int* top_k, const float* top_p, const float* min_p, const float* uniform, void* workspace,
                 tm_stream_t st);
/* tm_sample + the logprobs of the kept candidates (the `sampled_logprobs / sampled_indexes / sampled_nums` outputs of
 * invokeSampling, kernels/sampling_kernels.cu:67-90, which generation/sampling.cc:173-178 arms when a request asks for
 * output_logprobs): per row the first min(kept, cap) candidates in sampling order -- idx [batch][cap] token ids, vals [batch][cap]
 * logf of their renormalised probability (-inf for zero-probability candidates), num [batch] their count -- and sel [batch], the
 * drawn token's own logprob.  cap = TM_MAX_LOGPROBS reproduces the reference's layout: a drawn token beyond the first 1024
 * candidates replaces entry 1023.  Smaller caps keep the first cap candidates untouched (the drawn token's logprob is in sel).
 * All arrays are device memory; kept_out is required. */
#define TM_MAX_LOGPROBS 1024
int    tm_sample_logprobs(int* out_ids, int* kept_out, float* vals, int* idx, int* num, float* sel, int cap, const void* logits,
                          int batch, int vocab, int ld, const float* temperature, const int* top_k, const float* top_p,
                          const float* min_p, const float* uniform, void* workspace, tm_stream_t st);
/* the engine's uniform draw: Philox4x32-10, key = request seed, counter = context length -> [0, 1) (host function) */
float  tm_philox_uniform(uint64_t seed, uint32_t counter);

/* Logits processors ahead of arg-max / sampling: repetition penalty -> bad ids -> min-length ban of the end ids
 * (LogitsProcessor::Forward, generation/logits_processor.cc:66-115; kernels/sampling_penalty_kernels.cu:137-215,
 * kernels/ban_bad_words.cu:51-95 single-token case).  In place on fp16 logits [batch][ld] (columns vocab_offset ..
 * vocab_offset + vocab of the global vocabulary; ld and vocab_offset multiples of 8):
 *   seen     device u32 [batch][seen_words]: bit id%32 of word id/32 set = token id occurs in the sequence (prompt +
 *            generated); maintained with tm_seen_update (the reference rebuilds it from token_ids_ptrs every step)
 *   rep      device float [batch]  repetition penalty (1 = off): seen ids get l < 0 ? l*p : l/p
 *   ban      device int [batch][TM_MAX_BAD_IDS]  ids that get -65504 (-1 = unused; ids <= 0 are skipped, as upstream)
 *   end      device int [batch][1 + TM_MAX_STOP_IDS]  eos / stop ids, banned while k_len + 1 < min_len (ids <= 0 skipped)
 *   k_len    device int [batch]  context length of this forward; min_len = prompt length + min_new_tokens
 * The fp32 result is rounded to fp16 once (the reference keeps fp32 logits from here on).
 * tm_seen_update: OR the tokens ids[0..n_tokens) into the masks; cu_q NULL: token t belongs to row t (decode),
 * else rows are the packed prefill sequences cu_q[r] .. cu_q[r+1]. */
#define TM_MAX_BAD_IDS 32
#define TM_MAX_STOP_IDS 8
int tm_seen_update(void* seen, int seen_words, const int* ids, const int* cu_q, int nseq, int n_tokens, int vocab,
                   tm_stream_t st);
int tm_logits_process(void* logits, int batch, int vocab, int ld, int vocab_offset, const void* seen, int seen_words,
                      const float* rep, const int* ban, const int* end, const int* k_len, const int* min_len,
                      tm_stream_t st);
/* unfused activation on [M][2*inter] laid out [gate | up]            (kernels/activation.cu:27-130) */
int tm_silu_mul(void* out, const void* gate_up, int M, int inter, tm_stream_t st);

/* ---- Linear (mirrors _tm.LinearWeight / LlamaLinear.forward_dense / QuantizeGroupwise) -------- */
typedef struct tm_linear tm_linear;
enum { TM_WEIGHT_U4 = 0, TM_WEIGHT_F16 = 1, TM_WEIGHT_FP8 = 2 };
int tm_linear_create(tm_linear** out, int in_features, int out_features, int weight_type, int group_size);
/* Boundary ("TM") layout, device pointers: qweight int32 [K][N/8] (nibble j of word c = column 8c+j,
 * lmdeploy/turbomind/weight_format.py:63-74), scales / zeros fp16 [K/g][N].  For TM_WEIGHT_F16: weight fp16
 * [K][N], scales = zeros = NULL.  For TM_WEIGHT_FP8 (lmdeploy/turbomind/weight_format.py:349-393): weight = e4m3 bytes
 * [K][N], scales = fp32 128x128 block scales [K/128][ceil(N/128)] (`weight_scale_inv`, transposed like the weight),
 * zeros = NULL; w = fp16(e4m3) * fp16(scale), one rounding (the pre-sm90 weight-only path, linear_weight.cc:138-150).
 * Performs LinearWeight::prepare (repack to MFMA-fragment order, fuse (s,-z*s)). */
int tm_linear_prepare(tm_linear* w, const void* weight, const void* scales, const void* zeros, tm_stream_t st);
/* y fp16 [M][N] (or [M][N/2] when gated_silu: columns interleaved (gate_j, up_j), epilogue.h:159-176).
 * nt / splits / waves: 0 = heuristic.  waves per workgroup: 4 or 8; waves | 0x100 additionally splits K two ways
 * inside the workgroup (8 waves).  workspace >= tm_linear_workspace() bytes when split-K may be used. */
size_t tm_linear_workspace(const tm_linear* w, int M);
int    tm_linear_forward(const tm_linear* w, const void* x, int ldx, void* y, int ldy, int M, int gated_silu,
                         int nt, int splits, int waves, void* workspace, tm_stream_t st);
/* tm_linear_dequant_f16: the fp16 image [N][K] of the weights, bit for bit the operand the W4A16 kernels build on chip
 * (w = h(fma(h(q), s, h(-z*s))), kernels/gemm/transform.h:34-74) -- what the reference's own tests call quant_vs_dequant
 * (tests/turbomind/linear/fixture.py:404-507).  (Round 3 also exported tm_f16_library_available / tm_linear_build_f16_image:
 * prefill-sized forwards through a vendor fp16 GEMM on that image -- removed in round 4, every GEMM of this library is its own.) */
int    tm_linear_dequant_f16(const tm_linear* w, void* out_nk, tm_stream_t st);
/* Row-parallel linear closed by residual + RMSNorm -- wo / w2 of a decoder layer at the decode batch: LlamaLinear::Forward
 * followed by invokeResidualBiasRMSNorm (src/turbomind/models/llama/unified_decoder.cc:149,226; kernels/norm/rms_norm.cu:286-362):
 *   resid += fp16(x . W);  y = RMSNorm(resid) * norm_w.
 * The GEMM (fp32 split-K slabs when it splits), then the reduce-norm kernel.
 * shape: decode tile 0..3 / 6..9 / 11, -1 = dispatch; splits: 0 = dispatch.  workspace >= tm_linear_workspace(w, M) + M * N * 2 bytes.
 * (Round 3's in-launch consumer behind a `fused` flag was removed in round 4; its two dead parameters left the signature in round 5.) */
int    tm_linear_residual_norm(const tm_linear* w, const void* x, int ldx, void* y, void* resid, const void* norm_w, float eps, int M,
                               int shape, int splits, void* workspace, tm_stream_t st);
/* The same pair of operators with the RMSNorm FOLDED into the two GEMMs around it (round 5; the engine's decode step at tp = 1):
 *   RMSNorm(r) . W2 = inv[m] * sum_k (r[m,k] g[k]) W2[k,n],   inv[m] = 1 / sqrt(sum_k r[m,k]^2 / H + eps)
 * tm_linear_fold_produce -- the row-parallel linear (wo / w2): resid += fp16(x . W) in the GEMM's own epilogue (bit-identical to
 *   tm_linear_residual_norm's residual stream, split-K included: the last-arriving slice of a column tile sums the slices' fp32 slabs in
 *   slice order), xg = fp16(f32(resid) * f32(norm_w)) (saturating), ss[tile][m] = sum of f32(resid)^2 over the tile's columns;
 *   *ss_tiles = tiles written (<= N / 64).  ss holds (N / 64) * M floats; workspace >= tm_linear_fold_workspace(w, M).
 * tm_linear_fold_consume -- the linear fed with xg: y = [gated SiLU](inv[m] * (xg . W)); norm_h = H of the folded norm.
 * Replaces invokeResidualBiasRMSNorm + the following LlamaLinear::Forward (unified_decoder.cc:226,278,328; rms_norm.cu:286-362;
 * epilogue.h:159-176): two launches instead of three; the normalised activations carry ONE fp16 rounding (r * g) instead of two and
 * the row factor is applied to fp32 accumulators -- tests/test_gpu_ops.py::test_w4a16_folded_norm states the bound next to the
 * unfused sequence's.  shape: decode tile 0..3 / 6..9, -1 = dispatch; splits: 0 = dispatch. */
size_t tm_linear_fold_workspace(const tm_linear* w, int M);
int    tm_linear_fold_produce(const tm_linear* w, const void* x, int ldx, void* xg, void* resid, const void* norm_w, float* ss,
                              int* ss_tiles, int M, int shape, int splits, void* workspace, tm_stream_t st);
int    tm_linear_fold_consume(const tm_linear* w, const void* xg, int ldx, void* y, int ldy, int M, int gated_silu, const float* ss,
                              int ss_tiles, int norm_h, float eps, int shape, int splits, void* workspace, tm_stream_t st);
/* FP8 x FP8 linear on the fp8 matrix cores -- the reference's path for e4m3 weights on fp8 tensor cores:
 * LlamaLinear::Forward quantises the activations per row and 128-channel group (QuantizeSymm,
 * src/turbomind/kernels/quantization.cu:28-125, called at models/llama/LlamaLinear.cu:67-93) and runs the fp8 GEMM with
 * both scale sets.  tm_quant_fp8_rows = QuantizeSymm (codes [M][K] e4m3, scales fp32 [K/128][ldsx]);
 * tm_linear_forward_fp8 = quantise + GEMM (+ split-K reduce); workspace >= tm_linear_fp8_workspace(w, M) bytes. */
/* the fused w1w3 linear of an fp8 checkpoint: (gate_j, up_j)-interleaved e4m3 columns whose block-scale row is
 * [w1's inter/128 blocks | w3's inter/128 blocks] (w1 / w3 are quantised separately; lmdeploy/turbomind/builders/ffn.py:31-34) */
int    tm_linear_prepare_fp8_gated(tm_linear* w, const void* weight, const void* scales, tm_stream_t st);
size_t tm_linear_fp8_workspace(const tm_linear* w, int M);
int    tm_quant_fp8_rows(void* xq, float* sx, const void* x, int ldx, int M, int K, int ldsx, tm_stream_t st);
int    tm_linear_forward_fp8(const tm_linear* w, const void* x, int ldx, void* y, int ldy, int M, int gated_silu, int splits,
                             void* workspace, tm_stream_t st);
int    tm_linear_destroy(tm_linear* w);
/* ---- Mixture-of-experts FFN block (MoeFfnLayer, models/llama/moe_ffn_layer.cc:43-53,133-325; routing
 * kernels/gemm/moe_utils_v2.cu:355-690; grouped linear LlamaLinear.cu:67-127) -------------------------------------
 * out[t] = sum over the top_k experts e of token t of w_e(t) * W2_e( silu(W1_e x_t) * (W3_e x_t) ), with
 * logits = x Wg (fp32), top-k on the logits (ties: lower expert id), w = softmax over the selected experts
 * (norm_topk != 0) or over all experts, times routed_scale.  Expert weights: weight_type TM_WEIGHT_U4 / TM_WEIGHT_FP8,
 * boundary layouts of tm_linear_prepare; w13 = [hidden][2*inter] with (gate_j, up_j) column-interleaved, w2 =
 * [inter][hidden].  FP8 w13: w1 and w3 are block-quantised separately in a checkpoint, so the scale row of w13 is
 * [w1's inter/128 blocks | w3's inter/128 blocks] (inter % 128 == 0) -- also for the engine's *.w1w3.scales slots.  gate: fp16 [hidden][experts].  All pointers device.  topk_ids_out / topk_w_out (device
 * [tokens][top_k], may be NULL) expose the routing for tests. */
typedef struct tm_moe tm_moe;
int    tm_moe_create(tm_moe** out, int hidden, int inter, int experts, int top_k, int weight_type, int norm_topk,
                     float routed_scale);
int    tm_moe_set_gate(tm_moe* m, const void* gate, tm_stream_t st);
int    tm_moe_set_expert(tm_moe* m, int expert, const void* w13_weight, const void* w13_scales, const void* w13_zeros,
                         const void* w2_weight, const void* w2_scales, const void* w2_zeros, tm_stream_t st);
size_t tm_moe_workspace(const tm_moe* m, int tokens);
int    tm_moe_forward(tm_moe* m, void* out, const void* x, int tokens, void* workspace, int* topk_ids_out, float* topk_w_out,
                      tm_stream_t st);
int    tm_moe_destroy(tm_moe* m);

/* IntegralQuantizer<half,4> (kernels/quantization.cu:384-440): w fp16 [K][N] -> qweight int32 [K][N/8],
 * scales/zeros fp16 [K/g][N], dequant fp16 [K][N] (may be NULL).  Groups run along K. */
int tm_quantize_groupwise(void* qweight, void* scales, void* zeros, void* dequant, const void* w, int K, int N,
                          int group_size, tm_stream_t st);

/* ---- native point-to-point communicator (the reference's comm/cuda_ipc backend; RCCL stays the default) ----------------
 * Fused one-shot all-reduce + residual + RMSNorm over peer-mapped segments: replaces AllreduceResidualBiasRMSnorm
 * (src/turbomind/comm/cuda_ipc/fused_allreduce.cu:406-500, exchange as in allreduce.cu:249-343) and, for the lm_head's
 * (value, index) candidates, AllGather (comm/cuda_ipc/allgather.cu).  Every rank owns a symmetric segment of
 * tm_p2p_segment_bytes(rows, H) bytes -- [256 B of flags | tile 0 | tile 1], a tile = rows x H fp16 -- that its peers
 * map: tm_p2p_segment_create allocates it (zeroed) and returns a 64-byte IPC handle, tm_p2p_segment_open maps a peer's
 * handle (another PROCESS: one process per GPU), tm_p2p_segment_close unmaps (opened = 1) or frees (opened = 0).
 * segs[r] = rank r's segment as seen from this rank (segs[me] = the local one); state = 4 zeroed local words (the rank's call
 * counter and tickets).  Every rank must issue the same sequence of tm_p2p_* calls.
 * tm_p2p_allreduce_norm: y = RMSNorm(resid += fp16(sum_r partial_r)), M <= rows; rank-ordered fp32 sum, so every rank
 * holds the same bits.  tm_p2p_allgather: dst[r][0..words) = rank r's src (32-bit words, words * 4 <= tile bytes).
 * A rank that waits longer than ~1 s for a peer gives up, writes the call number into state[3] and returns garbage. */
size_t tm_p2p_segment_bytes(int rows, int H);
int tm_p2p_segment_create(size_t bytes, void** dev_ptr, void* handle64);
int tm_p2p_segment_open(const void* handle64, void** dev_ptr);
int tm_p2p_segment_close(void* dev_ptr, int opened);
int tm_p2p_allreduce_norm(void* const* segs, int tp, int me, void* state, int rows, const void* partial, void* y, void* resid,
                          const void* weight, float eps, int M, int H, tm_stream_t st);
int tm_p2p_allgather(void* const* segs, int tp, int me, void* state, int rows, int H, const void* src, void* dst, int words,
                     tm_stream_t st);
/* Two-shot form for large messages -- AllreduceResidualBiasRMSnorm_Simple_Push / _Pull and the two-shot all-reduce of the reference
 * (src/turbomind/comm/cuda_ipc/fused_allreduce.cu:25-405, allreduce.cu:22-248): reduce-scatter (rank r sums rows
 * [r * slice, (r + 1) * slice), slice = ceil(M / tp), of all ranks' partials in rank order), residual + RMSNorm on the owned slice,
 * all-gather of the normed rows by pushing them into every peer's segment: (tp - 1) / tp of one message read and written per rank
 * instead of (tp - 1) messages read.  y: all M normed rows (the bits of the one-shot form); resid: updated for the rank's OWN
 * slice only (the reference shards the residual stream the same way).  Segments of tm_p2p_segment_bytes2(rows, rows2, H) bytes =
 * [256 B flags | one-shot tile 0 | tile 1 (rows x H each) | in2 | out2 (rows2 x H each)], M <= rows2; shares `state` and the call
 * sequence with the one-shot calls (it advances the epoch by two). */
size_t tm_p2p_segment_bytes2(int rows, int rows2, int H);
/* One-shot form whose workgroups do not wait for each other (round 6): one workgroup owns a token row end to end -- its own call counter
 * (state[4 + row]), write-through system-scope stores of its partial row, ONE flag word per (sender, row) in every peer's segment, system-
 * scope loads of the peers' rows -- no tickets, no shared epoch, no residency requirement, no L2 write-back / invalidate on the path; same
 * arithmetic and bits as tm_p2p_allreduce_norm.  Segments of tm_p2p_segment_bytes_rows(rows, rows2, H) bytes = the two-shot layout followed
 * by [row tile 0 | row tile 1 (rows x H each) | flags [8][rows]]; state = 4 + rows zeroed local words; M <= rows.  Its calls are
 * independent of the shared call sequence of the other tm_p2p_* entry points (own tiles, own flags).  This is what the engine's decode-sized
 * exchanges run (fused_allreduce.cu:406-500 called at unified_decoder.cc:278-285,328-335). */
size_t tm_p2p_segment_bytes_rows(int rows, int rows2, int H);
int tm_p2p_allreduce_norm_rows(void* const* segs, int tp, int me, void* state, int rows, int rows2, const void* partial, void* y, void* resid,
                               const void* weight, float eps, int M, int H, tm_stream_t st);
int tm_p2p_allreduce_norm_2shot(void* const* segs, int tp, int me, void* state, int rows, int rows2, const void* partial, void* y,
                                void* resid, const void* weight, float eps, int M, int H, tm_stream_t st);

/* debug / measurement: device buffer of 8192 x 8 uint64 (512 KB) that receives s_memrealtime stamps (100 MHz: kernel
 * entry, loop entry, loop exit, exit, ...) of every GEMM / decode-attention workgroup launched afterwards; launches of more
 * than 8192 workgroups are not traced; NULL switches it off. */
int tm_debug_set_gemm_trace(void* dev_buf);
/* Arena form of the same trace (round 5, tools/fixed_cost_table.py): every traced launch (decode GEMMs, decode attention, the
 * norm kernels) is given ITS OWN region of `dev_buf` (capacity_workgroups x 8 uint64), in launch order -- so the launches of a
 * captured hipGraph keep distinct regions and one replay leaves the stamps of every kernel of the decode step.  Slots per
 * workgroup: 0 entry, 5 prologue loads issued, 1 activations landed (first stage barrier), 6 wave 0's first weight unit landed,
 * 2 loop exit, 7 k-phase merge barrier, 3 stores drained, 4 (XCC id << 32 | HW_ID); attention: 1 prologue done, 2 / 6 / 7
 * waves 0 / 1 / 3 done, 5 all waves done.  NULL switches it off and forgets the records.
 * tm_debug_trace_records: text, one line per traced launch `index tag grid.x grid.y grid.z offset_in_workgroups`; returns the
 * bytes the text needs.  Host-side bookkeeping is not thread-safe against concurrent LAUNCHES from other engines. */
int     tm_debug_trace_arena(void* dev_buf, int64_t capacity_workgroups);
int64_t tm_debug_trace_records(char* host_out, int64_t cap);
/* Host-only: the (workgroup shape, split-K) the decode GEMM dispatch picks for a W4A16 linear of K x N at M rows --
 * use_table bit 0: the measured table first (tm_engine_tune_gemm / tm_gemm_import), then the heuristic; clear: heuristic only;
 * bits 8 .. 11: the linear's role -- 0 any, 1 w_qkv, 2 wo, 3 w1w3, 4 w2 (the table is keyed (role, K, N, M): the tuner times each
 * role with its own consumer kernel; an entry of role 0, e.g. an imported line without a role column, serves every role).
 * Shapes: 0..3 decode tiles (M <= 64), 4 / 5 the 128-row tiles (M > 64), 6..9 shapes 3, 0, 2, 1 on 32-row blocks
 * (gemm_decode.hip), 11 the loader / consumer decode kernel (gemm_decode_lc.hip, M <= 64), 12 the 256 x 256 prefill tile with
 * the weights dequantised once per workgroup tile through LDS (gemm_prefill.hip, M > 64).  (10 was a vendor-library path: gone.) */
int tm_debug_pick_tiling(int K, int N, int M, int use_table, int* shape, int* splits);
/* Host-only: the tiling of the general kernel (gemm_w4a16.hip: the fp16 lm_head, e4m3 weight-only linears, u4 with N % 32 != 0)
 * for a K x N linear of `weight_type` (TM_WEIGHT_*) and `role` (0 any, 1 w_qkv, 2 wo, 3 w1w3, 4 w2, 5 lm_head) at M rows:
 * config4 = {tiles per wave, split-K, waves, k-phases} -- the measured entry (tm_engine_tune_gemm / tm_gemm_import, `G` lines)
 * first, then the heuristic.  Reference: the same DispatchCache serves every GEMM of the model (gemm.cu:92-224). */
int tm_debug_pick_general(int weight_type, int role, int K, int N, int M, int* config4);
/* Host-only: the measured row-tile height of the grouped expert GEMMs (u4: 16 / 32 / 64, e4m3 on the fp8 matrix cores: 32 / 64)
 * for `tokens` rows per forward; *rows = 0: no entry, the launcher's own rule (twice the expected rows per expert) applies. */
int tm_debug_grouped_tile(int weight_type, int K, int N, int tokens, int* rows);
/* Host-only: every (shape, splits) pair the start-up tuner (tm_engine_tune_gemm; gemm::Gemm::Run's dispatch candidates,
 * src/turbomind/kernels/gemm/gemm.cu:92-224) may pick for a K x N linear at M rows; *count = how many exist (<= 128), the first
 * min(cap, *count) are written.  The full-size parity tests sweep exactly this list. */
int tm_debug_tiling_candidates(int K, int N, int M, int* shapes, int* splits, int cap, int* count);
/* Operator-level calls that follow treat tm_kv_cache::block_ptrs as a rectangular table: sequence b starts at b * stride
 * (cu_block_nums must say the same); 0 = ragged (default).  The engine's own table is rectangular and always takes this
 * path: the decode kernel then needs no dependent pointer loads (test hook for that path). */
int tm_debug_set_block_stride(int stride);

/* ----------------------------------------------------------------------------------------------
 * Engine level (static batcher around LanguageModel::Forward)
 * --------------------------------------------------------------------------------------------*/
typedef struct tm_model_config {
    int   hidden, layers, q_heads, kv_heads, head_dim, inter, vocab;
    float rms_eps;
    float rope_base;
    int   rope_type; /* 0 default, 1 linear, 2 llama3 */
    float rope_factor, rope_low_freq_factor, rope_high_freq_factor;
    int   rope_original_max_position;
    int   group_size;   /* 128 */
    int   weight_type;  /* TM_WEIGHT_U4 (AWQ), TM_WEIGHT_F16 or TM_WEIGHT_FP8 for the decoder linears; lm_head is fp16 */
    /* mixture of experts (0 experts = dense FFN): every layer's FFN is a router + `moe_experts` expert FFNs of width
     * `inter` (sharded over tp like the dense FFN), `moe_top_k` per token (Mixtral: 8 / 2, norm_topk = 1) */
    int   moe_experts, moe_top_k, moe_norm_topk;
    float moe_routed_scale;
} tm_model_config;

typedef struct tm_engine_config {
    tm_model_config model;
    int   tp;                 /* tensor-parallel world size (one process per GPU)            */
    int   rank;               /* this process' rank                                          */
    int   device;             /* HIP device ordinal                                          */
    int   max_batch_size;     /* TurbomindEngineConfig.max_batch_size                        */
    int   session_len;        /* max context per sequence                                    */
    int   quant_policy;       /* 0 | 4 | 8  (lmdeploy/messages.py:20-27)                     */
    int   cache_block_seq_len;/* 64                                                          */
    float cache_max_entry_count; /* fraction of free HBM for KV blocks if cache_blocks == 0 */
    int   cache_blocks;       /* explicit number of KV blocks (0 = derive)                   */
    int   max_prefill_token_num; /* tokens per prefill iteration (messages.py:334: 8192)    */
    int   decode_splits;      /* 0 = heuristic                                               */
    int   use_graph;          /* capture the decode step in a hipGraph                       */
} tm_engine_config;

typedef struct tm_engine tm_engine;

int tm_engine_create(tm_engine** out, const tm_engine_config* cfg);
int tm_engine_destroy(tm_engine* e);
/* RCCL bootstrap for tp > 1: rank 0 calls tm_comm_unique_id (128 bytes, host), broadcasts it out of band
 * (torch.distributed / TCPStore), then every rank calls tm_engine_comm_init. */
int tm_comm_unique_id(void* host_out128);
int tm_engine_comm_init(tm_engine* e, const void* host_id128);
/* native communicator (opt-in; see tm_p2p_* below): export = allocate this rank's segment, return its IPC handle; import =
 * map the tp ranks' handles (rank order) and route the row-parallel all-reduces of forwards with <= rows tokens through it */
int tm_engine_comm_drop_rccl(tm_engine* e);   /* continue on the native communicator alone (every rank must call it) */
int tm_engine_comm_native_export(tm_engine* e, int rows, void* handle64);              /* after tm_engine_comm_init */
int tm_engine_comm_native_import(tm_engine* e, const void* handles, int count);        /* count = tp handles, rank order */
/* Collective bring-up check of the native communicator (every rank, right after the import): two fused launches over a known pattern;
 * *ok = 0 -> every rank calls tm_engine_comm_native_drop and the decode-sized exchanges stay on RCCL + the residual-norm launch.  The
 * hop between devices (system-scope visibility over xGMI) is what no single-GPU test exercises; the default tensor-parallel arrangement
 * -- native fused all-reduce + residual + RMSNorm for decode-sized forwards (the reference's AllreduceResidualBiasRMSnorm,
 * src/turbomind/comm/cuda_ipc/fused_allreduce.cu:406-500, called at unified_decoder.cc:278-285,328-335), RCCL for prefill-sized ones --
 * is taken only when it passed on every rank. */
int tm_engine_comm_native_selftest(tm_engine* e, int* ok);
int tm_engine_comm_native_drop(tm_engine* e);

/* Weight hand-off: named Param slots, already TP-sharded / QKV-fused / w1w3-interleaved by the loader
 * (lmdeploy/turbomind/builders/_base.py:72-113).  Names: "layers.{i}.attention.w_qkv.{qweight,scales,zeros}",
 * "layers.{i}.attention.wo.*", "layers.{i}.feed_forward.w1w3.*", "layers.{i}.feed_forward.w2.*"
 * (".weight" instead for fp16 linears), "layers.{i}.attention_norm.weight", "layers.{i}.ffn_norm.weight",
 * "tok_embeddings.weight", "norm.weight", "output.weight".
 * tm_engine_weight_bytes: expected byte size of a slot (0 + TM_INVALID for unknown names).
 * tm_engine_weight_copy : host -> device staging copy; byte size must match exactly. */
int64_t tm_engine_weight_bytes(tm_engine* e, const char* name);
int     tm_engine_weight_copy(tm_engine* e, const char* name, const void* host_src, int64_t bytes);
/* Device-side synthetic init (random weights with the real shapes): N(0,1)*0.1/sqrt(K) fp16
 * master -> group-128 asymmetric u4; norms 1+0.02N; embeddings 0.02N.  For benchmarks without checkpoints. */
int     tm_engine_init_synthetic(tm_engine* e, uint64_t seed);
/* ModelRoot::prepare(): repack every linear, free staging */
int     tm_engine_process_weights(tm_engine* e);
/* allocate KV pool + activation buffers, build RoPE table (TurboMind::CreateEngine) */
int     tm_engine_start(tm_engine* e);

/* Static batch.  Admit `batch` sequences (host token ids, concatenated; host lens), run chunked prefill and
 * produce the first token of every sequence.  Each sequence reserves ceil((len + max_new_tokens)/64) blocks. */
int tm_engine_prefill(tm_engine* e, const int* host_ids, const int* host_lens, int batch, int max_new_tokens);
/* run `steps` decode iterations for the admitted batch (every sequence generates one token per step,
 * ignore_eos semantics).  Asynchronous on the engine stream; tm_engine_sync() waits. */
int tm_engine_decode(tm_engine* e, int steps);
int tm_engine_sync(tm_engine* e);
/* per-sequence time-to-first-token of the last tm_engine_prefill, milliseconds since the call started (host [batch]) */
int tm_engine_prefill_times(tm_engine* e, float* host_ms);
/* Run `steps` EAGER decode steps with HIP events around every kernel category on the engine stream and return the
 * mean milliseconds per step per category (host float[TM_PROF_NUM]) and launches per step (host int[TM_PROF_NUM],
 * may be NULL).  Categories: */
enum {
    TM_PROF_EMBED = 0, TM_PROF_GEMM_QKV, TM_PROF_KV_STORE, TM_PROF_ATTN, TM_PROF_GEMM_O, TM_PROF_RES_NORM,
    TM_PROF_GEMM_GATE_UP, TM_PROF_GEMM_DOWN, TM_PROF_LM_HEAD, TM_PROF_SAMPLE, TM_PROF_ALLREDUCE, TM_PROF_NUM
};
int tm_engine_profile_decode(tm_engine* e, int steps, float* host_ms_per_step, int* host_launches_per_step);
/* copy generated tokens so far to host: out [batch][max_new_tokens] int32 (row-major), n_generated (host) */
int tm_engine_fetch(tm_engine* e, int* host_out, int* n_generated);
/* last step's logits of the local vocab shard, fp16 [batch][vocab/tp] -> host (debug / parity tests) */
int tm_engine_fetch_logits(tm_engine* e, void* host_out);
/* parity tests: intermediate state of the last forward -> host.  what = 0: the residual stream, fp16 [a rows][hidden]
 * (UnifiedDecoder's residual buffer, models/llama/unified_decoder.cc:226-330); what = 1: the bytes of KV block b of
 * static-batch sequence a, all layers (block layout = kernels/attention/block.h:126-219); what = 2: int64 count of
 * continuous-batching steps whose decode rows shared a forward with an admission's prefill.  `bytes` must match exactly. */
int tm_engine_debug_read(tm_engine* e, int what, int a, int b, void* host_out, int64_t bytes);
/* Per-sequence sampling parameters (GenerationConfig: temperature, top_k, top_p, min_p, random_seed;
 * lmdeploy/messages.py:35-205).  top_k == 1 is greedy.  The uniform draw of a step is Philox(seed, context length). */
typedef struct tm_sampling {
    float    temperature; /* > 0 */
    int      top_k;       /* <= 0: no top-k filter */
    float    top_p;       /* >= 1: no top-p filter */
    float    min_p;       /* 0: off */
    uint64_t seed;
} tm_sampling;
/* Static batch: sampling parameters of the NEXT tm_engine_prefill (host array [batch], copied); NULL / never called =
 * greedy arg-max.  Cleared by tm_engine_release.  TP > 1 (RCCL communicator): the vocabulary shards of the logits
 * are all-gathered and every rank draws the same token from the full row (models/language_model.cc:304-333 gathers the
 * logits as well) -- over RCCL, or through the native P2P segments when no RCCL communicator exists. */
int tm_engine_set_sampling(tm_engine* e, const tm_sampling* host_params, int batch);
/* Static batch: the NEXT tm_engine_prefill records, for every generated token of every sequence, the first n (1 .. TM_MAX_LOGPROBS;
 * 0 = off) kept candidates with their logprobs and the drawn token's own logprob -- GenerationConfig.logprobs, i.e. the request's
 * output_logprobs of generation/sampling.cc:173-178,229-279 seen through lmdeploy/turbomind/turbomind.py:472-503.  The draw then runs
 * through the sampling kernels (greedy sequences as top_k = 1 rows: one kept candidate with logprob 0, as upstream).  Cleared by
 * tm_engine_release.  tm_engine_fetch_logprobs copies the records to the host: vals / idx [batch][max_new_tokens][n] (entries beyond
 * num are undefined), num / sel [batch][max_new_tokens] (columns beyond the generated steps: num = 0). */
int tm_engine_set_logprobs(tm_engine* e, int n);
/* Continuous batching: the same records for ONE queued request (call right after its submit, before the next scheduler step): every
 * token the request generates from then on carries its first n kept candidates.  tm_engine_poll_logprobs copies the records of the
 * first min(max_tokens, generated) tokens: vals / idx [tokens][n] (entries beyond num[t]: 0 / -1), num / sel [tokens]; *n_tokens =
 * tokens recorded, *n_per_token = n (0: the request did not ask).  The arrays may be NULL to query the counts.  While any request with
 * logprobs is in the session the decode step does not ride on prefill forwards (no mixed steps). */
int tm_engine_request_logprobs(tm_engine* e, int64_t req_id, int n);
int tm_engine_poll_logprobs(tm_engine* e, int64_t req_id, float* host_vals, int* host_idx, int* host_num, float* host_sel, int max_tokens,
                            int* n_tokens, int* n_per_token);
int tm_engine_fetch_logprobs(tm_engine* e, float* host_vals, int* host_idx, int* host_num, float* host_sel);
/* Per-sequence logits processors (GenerationConfig: repetition_penalty, min_new_tokens, bad_token_ids,
 * stop_token_ids; applied in the reference's order, see tm_logits_process).  stop ids end a sequence of the
 * continuous-batching path like its eos id and are banned with it until min_new_tokens tokens exist; the static
 * batch has no engine-side stop (the caller cuts), there they only matter for min_new_tokens. */
typedef struct tm_logits_param {
    float repetition_penalty; /* > 0; 1 = off */
    int   min_new_tokens;     /* 0 = off */
    int   n_bad_ids;
    int   bad_ids[TM_MAX_BAD_IDS];
    int   n_stop_ids;
    int   stop_ids[TM_MAX_STOP_IDS];
} tm_logits_param;
/* Static batch: parameters of the NEXT tm_engine_prefill (host array [batch], copied); NULL = none.  Cleared by
 * tm_engine_release.  Works with tp > 1 (each rank processes its vocabulary shard). */
int tm_engine_set_logits_params(tm_engine* e, const tm_logits_param* host_params, int batch);

/* release the batch (blocks return to the pool); also ends a continuous-batching session */
int tm_engine_release(tm_engine* e);
/* Measured GEMM dispatch (the reference's warm-up tuning, turbomind.cc:363-487 / kernels/gemm/gemm.cu:92-224): time every
 * (workgroup shape, split-K) candidate of the W4A16 kernels for the model's w_qkv / wo / w1w3 / w2 at M rows -- M <= 256: a decode
 * batch, keyed by the exact M; M = 512, 1024, 2048, 4096, 8192 (<= max_prefill_token_num): the size class of prefill forwards
 * with up to M tokens -- as a hipGraph over the engine's own layer weights, each GEMM followed by the kernel that consumes it,
 * and remember the winner per (K, N, M).  After tm_engine_start, before the first batch.  export_path (may be NULL): write the table as text lines
 * "K N M shape splits".  Environment equivalents read by tm_engine_start: TM_GEMM_TUNE=1 (M = max_batch_size),
 * TM_GEMM_EXPORT=<file>, TM_GEMM_IMPORT=<file>; TM_GEMM_TUNE_VERBOSE=1 prints every measurement to stderr. */
int tm_engine_tune_gemm(tm_engine* e, int M, const char* export_path);
int tm_gemm_import(const char* path);
/* write the process's measured dispatch tables (P32 lines + `G` lines) without an engine: what rank 0 hands to the other ranks */
int tm_gemm_export(const char* path);


/* ----------------------------------------------------------------------------------------------
 * Continuous batching.  Replaces ModelRequest.forward / cancel + the engine thread's
 * schedule-forward-update loop (bind.cpp:743-793, engine/engine.cc:434-470,770-870, model_request.cc:36-135):
 * requests are queued, admitted in arrival order when a batch slot and enough KV blocks (prompt + max_new_tokens,
 * 64-token blocks) are free, prefilled (chunked, <= max_prefill_token_num prompt tokens per step) and then decoded
 * together with everything else that is running; a sequence ends on eos_id (eos_id < 0: ignore_eos), at
 * max_new_tokens or on cancel, and its slot / blocks are reused by the next admission.  Greedy sampling.
 * The first tm_engine_submit switches the engine into this mode (no static batch may be admitted);
 * tm_engine_release leaves it.  The caller drives the loop: one tm_engine_step = admissions + ONE decode step.
 * submit: 0 queued | TM_INVALID (empty prompt, max_new_tokens < 1) | TM_TOO_LONG (> session_len) |
 *         TM_OOM (can never fit the block pool).  poll: status 0 = waiting / running, TM_FINISH, TM_CANCEL;
 *         copies min(cap, n_tokens) generated tokens.  Unknown ids: TM_INVALID. */
int tm_engine_submit(tm_engine* e, const int* host_ids, int n, int max_new_tokens, int eos_id, int64_t* req_id);
/* like tm_engine_submit with sampling parameters (NULL = greedy) */
int tm_engine_submit_ex(tm_engine* e, const int* host_ids, int n, int max_new_tokens, int eos_id, const tm_sampling* sampling,
                         int64_t* req_id);
/* ... and logits-processor parameters (NULL = none) */
int tm_engine_submit_gen(tm_engine* e, const int* host_ids, int n, int max_new_tokens, int eos_id, const tm_sampling* sampling,
                         const tm_logits_param* logits_param, int64_t* req_id);
int tm_engine_step(tm_engine* e, int* n_active, int* n_waiting);
/* tm_engine_step_many: up to max_steps iterations of tm_engine_step in one call (stops when nothing runs and nothing waits); *steps_done =
 * iterations executed.  What a tensor-parallel rank group mirrors to its ranks instead of single steps -- the per-rank engine loops of
 * src/turbomind/engine/engine.cc:770-870 exchange admissions, not steps -- and the cheaper host loop at tp = 1 (one poll per burst). */
int tm_engine_step_many(tm_engine* e, int max_steps, int* steps_done, int* n_active, int* n_waiting);
int tm_engine_poll(tm_engine* e, int64_t req_id, int* status, int* host_tokens, int cap, int* n_tokens);
int tm_engine_cancel(tm_engine* e, int64_t req_id);
/* drop the record of a FINISHED request (status != 0) once its tokens were read: a long-lived serving session would
 * otherwise keep every prompt / output until tm_engine_release.  TM_INVALID: unknown id or still queued / running. */
int tm_engine_forget(tm_engine* e, int64_t req_id);

/* Engine thread (the reference's Engine::Impl::InternalThreadEntry, engine/engine.cc:770-870, + the Gateway signal
 * thread that runs the Python callback, bind.cpp:942-952).  After tm_engine_serve_start an engine-owned thread runs
 * the schedule -> forward -> update loop whenever a request is queued or running, and sleeps otherwise;
 * submit / submit_ex / poll / cancel / wait may then be called from ANY thread (they are serialised against the loop by
 * the engine's mutex; without the loop they are serialised against each other the same way).  tm_engine_step returns
 * TM_CONFLICT while the loop runs.  `on_update` (may be NULL) is called ON THE ENGINE THREAD, outside the engine lock,
 * after every scheduler step, once per token produced in that step (a newly admitted request gets its first token
 * from the prefill and its second from the decode pass of the same step), with the request's status at that token
 * (0 = streaming, TM_FINISH) and its number of generated tokens so far; it is never called again for a request
 * after a non-zero status (the reference's callback contract, turbomind.py:808-812).  It may call poll / submit / cancel.
 * A device error inside the loop ends every unfinished request with TM_FAIL, stops the loop and is reported by
 * tm_engine_serve_stop (return code + tm_last_error).  serve_stop joins the thread; requests that are still queued or
 * running stay where they are (a later serve_start or tm_engine_step continues them). */
typedef void (*tm_request_cb)(void* user, int64_t req_id, int status, int n_tokens);
int tm_engine_serve_start(tm_engine* e, tm_request_cb on_update, void* user);
int tm_engine_serve_stop(tm_engine* e);
/* block until request req_id has more than `have_tokens` generated tokens or a non-zero status, or until timeout_ms
 * elapsed (timeout_ms < 0: no timeout); returns the poll result.  Needs the engine thread (TM_INVALID otherwise). */
int tm_engine_wait(tm_engine* e, int64_t req_id, int have_tokens, int timeout_ms, int* status, int* n_tokens);

/* The scheduler by itself (host-only bookkeeping: queue, slots, block accounting) -- what the engine embeds,
 * exported so that its policy is testable without a GPU (engine/scheduler.cc:1018-1078 at the level used here). */
/* Host-only, exported for the CPU tests like the scheduler below: where a tensor-parallel prefill forward of nseq sequences (row offsets
 * cu_q[0 .. nseq], cu_q[0] = 0) splits into the two micro-batches whose all-reduces overlap the other one's kernels (DESIGN.md 6;
 * no reference counterpart: the reference hides the exchange in a fused kernel, comm/cuda_ipc/fused_allreduce.cu:406-500).
 * *seqs_a = sequences of the first micro-batch, *rows_a = its rows; 0 / 0: no usable boundary (fewer than two sequences, or the smaller side
 * below min_rows / 2 rows) -- the engine then splits the row-wise part of every layer by rows instead. */
int tm_prefill_split(const int* cu_q, int nseq, int min_rows, int* seqs_a, int* rows_a);
typedef struct tm_sched tm_sched;
int tm_sched_create(tm_sched** out, int max_batch, int num_blocks, int session_len);
int tm_sched_destroy(tm_sched* s);
int tm_sched_submit(tm_sched* s, const int* ids, int n, int max_new_tokens, int eos_id, int64_t* req_id);
/* admit waiting requests for one step: writes up to cap (request id, slot) pairs */
int tm_sched_admit(tm_sched* s, int token_budget, int64_t* req_ids, int* slots, int cap, int* n_admitted);
/* *ready = 1 if tm_sched_admit would admit at least one request right now (no side effects): the engine asks before it keeps a
 * decode step in flight across scheduler steps (two-phase schedule / forward overlap, turbomind.cc:171) */
int tm_sched_admit_ready(tm_sched* s, int* ready);
int tm_sched_on_token(tm_sched* s, int slot, int token, int* finished);
int tm_sched_cancel(tm_sched* s, int64_t req_id, int* released_slot);
/* status (Request::k*), slot (-1 unless running), tokens generated, blocks held; TM_INVALID for unknown ids */
int tm_sched_query(tm_sched* s, int64_t req_id, int* status, int* slot, int* n_generated, int* n_blocks);
int tm_sched_counts(tm_sched* s, int* n_active, int* n_waiting, int* n_free_blocks);
/* every unfinished request ends with `status` (the engine uses TM_FAIL after a device error) */
int tm_sched_abort_all(tm_sched* s, int status);
/* forget a finished request; TM_INVALID for unknown / unfinished ids */
int tm_sched_forget(tm_sched* s, int64_t req_id);
/* the engine's stream (hipStream_t) so callers can bracket it with their own events */
tm_stream_t tm_engine_stream(tm_engine* e);
/* introspection for benchmarks: bytes of quantised weights + scales + lm_head, KV bytes per token, #blocks */
/* What the tensor-parallel data path of this engine actually runs on (for benchmark records; any pointer may be NULL):
 * backend: 0 = no collectives (tp = 1), bit 0 = RCCL communicator up (comm/nccl/nccl.cu:356-398), bit 1 = native P2P communicator
 * imported (comm/cuda_ipc role); ranks = the communicator's own rank count (ncclCommCount), not the launcher's world size;
 * graph_captured: 1 = decode steps replay a hipGraph (collectives captured inside), 0 = eager launches (never captured yet,
 * use_graph = 0, or the capture with collectives failed and the engine fell back). */
int tm_engine_comm_info(tm_engine* e, int* backend, int* ranks, int* graph_captured);
/* The overlap the north star names ("RCCL all-reduce over xGMI overlapped on a side HIP stream"), as counters a benchmark / test can read
 * (any pointer may be NULL): side_stream = 1 when prefill-sized tensor-parallel forwards run their all-reduces on the engine's side stream
 * under the other half's kernels (TM_COMM_STREAM, default on; needs the RCCL communicator); forwards = forwards that ran that way so far;
 * microbatch_forwards = how many of them split at a sequence boundary into two micro-batches that leapfrog through all layers (the rest
 * split the row-wise part of every layer into two row halves); allreduces = all-reduces enqueued on the side stream so far (4 per dense
 * layer of such a forward).  The reference hides the exchange
 * inside a fused kernel (comm/cuda_ipc/fused_allreduce.cu:406-500, unified_decoder.cc:278,328); decode-sized forwards do the same here
 * (comm_p2p.hip) or call RCCL on the engine stream. */
int tm_engine_comm_overlap_info(tm_engine* e, int* side_stream, int64_t* forwards, int64_t* microbatch_forwards, int64_t* allreduces);
int tm_engine_stats(tm_engine* e, int64_t* weight_bytes, int64_t* kv_bytes_per_token, int64_t* num_blocks,
                    int* decode_splits);

#ifdef __cplusplus
}
#endif
#endif /* TM_MI355X_H */
