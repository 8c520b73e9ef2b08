// C-ABI shims for the operator-level entry points declared in include/tm_mi355x.h.
#include "../../include/tm_mi355x.h"
#include "scheduler.h"
#define TM_TRY_RC(expr)          \
    do {                        \
        const int _rc = (expr); \
        if (_rc) {              \
            return _rc;         \
        }                       \
    } while (0)
#include "tm_common.h"
#include "tm_kernels.h"
#include <string.h>
#include <cmath>
#include <vector>

namespace tmk {

static thread_local std::string g_last_error;

void set_last_error(const std::string& msg)
{
    g_last_error = msg;
}
const char* get_last_error()
{
    return g_last_error.c_str();
}

static int g_dbg_block_stride = 0;

static KvCacheView to_view(const tm_kv_cache* c)
{
    KvCacheView v{};
    v.block_ptrs    = c->block_ptrs;
    v.cu_block_nums = c->cu_block_nums;
    v.block_stride  = g_dbg_block_stride;
    v.layer_offset  = c->layer_offset;
    v.layout        = KvLayout{c->kv_heads, c->head_dim, c->block_len, c->bits};
    return v;
}

// Host-side (cos, sin) table.  Deterministic recipe shared with the oracle: freq and the angle are fp32
// products, exp2 / sin / cos are evaluated in double on those fp32 values and rounded fp32 -> fp16.
int build_rope_table(half_t* out, int max_pos, int dim, float base, int type, float factor, float low, float high,
                     int orig_max_pos)
{
    TM_REQUIRE(dim > 0 && dim % 2 == 0, "rope dim");
    TM_REQUIRE(type >= 0 && type <= 2, "rope_type in {0 default, 1 linear, 2 llama3}");
    const float        scale_factor = (float)(-std::log2((double)base) / dim);
    std::vector<float> inv(dim / 2);
    for (int i = 0; i < dim; i += 2) {
        const float prod = (float)i * scale_factor;
        const float freq = (float)std::exp2((double)prod);
        float       f    = freq;
        if (type == 1) {
            f = (float)(1.0 / factor) * freq;
        }
        else if (type == 2) {
            const double inv_diff   = 1.0 / ((double)high - (double)low);
            const float  alpha      = (float)((double)orig_max_pos / (2.0 * M_PI) * inv_diff);
            const float  beta       = (float)((double)low * inv_diff);
            const float  inv_factor = (float)(1.0 / factor);
            float        smooth     = alpha * freq - beta;
            smooth                  = smooth < 0.f ? 0.f : (smooth > 1.f ? 1.f : smooth);
            f                       = (1.f - smooth) * freq * inv_factor + smooth * freq;
        }
        inv[i / 2] = f;
    }
    for (int t = 0; t < max_pos; ++t) {
        for (int i = 0; i < dim / 2; ++i) {
            const float ang                        = (float)t * inv[i];
            out[((size_t)t * (dim / 2) + i) * 2]   = (half_t)(float)std::cos((double)ang);
            out[((size_t)t * (dim / 2) + i) * 2 + 1] = (half_t)(float)std::sin((double)ang);
        }
    }
    return 0;
}

}  // namespace tmk

using namespace tmk;

struct tm_linear {
    LinearWeight w;
};

extern "C" {

int tm_version(void)
{
    return 100;  // 0.1.0
}

const char* tm_last_error(void)
{
    return get_last_error();
}

int tm_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

int64_t tm_kv_layer_size(int kv_heads, int head_dim, int block_len, int bits)
{
    return KvLayout{kv_heads, head_dim, block_len, bits}.layer_size();
}

int tm_rmsnorm(void* y, const void* x, const void* w, float eps, int M, int H, tm_stream_t st)
{
    TM_REQUIRE(y && x && w, "null pointer");
    return launch_rmsnorm((half_t*)y, (const half_t*)x, (const half_t*)w, eps, M, H, (hipStream_t)st);
}

int tm_residual_rmsnorm(void* y, void* resid, const void* hidden, const float* partial, int splits, const void* bias,
                        const void* w, float eps, int M, int H, tm_stream_t st)
{
    TM_REQUIRE(y && resid && w, "null pointer");
    return launch_residual_rmsnorm((half_t*)y, (half_t*)resid, (const half_t*)hidden, partial, splits,
                                   (const half_t*)bias, (const half_t*)w, eps, M, H, (hipStream_t)st);
}

int tm_rope_table(void* host_out, int max_pos, int rope_dim, float base, int rope_type, float factor,
                  float low_freq_factor, float high_freq_factor, int original_max_position)
{
    TM_REQUIRE(host_out, "null pointer");
    return build_rope_table((half_t*)host_out, max_pos, rope_dim, base, rope_type, factor, low_freq_factor,
                            high_freq_factor, original_max_position);
}

int tm_kv_rope_store(void* qkv, int q_heads, const int* cu_q_len, const int* k_len, int batch, int total_tokens,
                     const void* cos_sin, int max_pos, const tm_kv_cache* cache, tm_stream_t st)
{
    TM_REQUIRE(qkv && cu_q_len && k_len && cache, "null pointer");
    return launch_kv_rope_store((half_t*)qkv, q_heads, cu_q_len, k_len, batch, total_tokens, (const half2_t*)cos_sin,
                                max_pos, to_view(cache), (hipStream_t)st);
}

int tm_flatten_kv(void* k_out, void* v_out, int transpose_v, const int* cu_k_off, const int* k_len, int batch,
                  int max_k_len, int k_stride, const tm_kv_cache* cache, tm_stream_t st)
{
    TM_REQUIRE(k_out && v_out && cu_k_off && k_len && cache, "null pointer");
    return launch_flatten_kv((half_t*)k_out, (half_t*)v_out, transpose_v, cu_k_off, k_len, batch, max_k_len, k_stride,
                             to_view(cache), (hipStream_t)st);
}

size_t tm_decode_attention_workspace(int batch, int q_heads, int splits)
{
    return decode_attention_workspace_bytes(batch, q_heads, 128, splits);
}

int tm_decode_attention(void* out, const void* q, int q_stride, const int* k_len, int batch, int q_heads,
                        float softmax_scale, int splits, void* workspace, const tm_kv_cache* cache, tm_stream_t st)
{
    TM_REQUIRE(out && q && k_len && cache, "null pointer");
    DecodeAttnParams p{};
    p.q          = (const half_t*)q;
    p.q_stride   = q_stride;
    p.out        = (half_t*)out;
    p.k_len      = k_len;
    p.batch      = batch;
    p.q_heads    = q_heads;
    const float s = softmax_scale > 0.f ? softmax_scale : 1.0f / std::sqrt(128.0f);
    p.scale_log2 = s * 1.4426950408889634f;
    p.splits     = splits < 1 ? 1 : splits;
    p.partial_o  = (float*)workspace;
    p.partial_ml = workspace ? (float*)workspace + (size_t)batch * q_heads * p.splits * 128 : nullptr;
    p.cache      = to_view(cache);
    return launch_decode_attention(p, (hipStream_t)st);
}

int tm_decode_attention_fused(void* out, const void* qkv, int qkv_splits, int qkv_n, const void* cos_sin, int max_pos,
                              const int* k_len, int batch, int q_heads, float softmax_scale, int splits,
                              void* workspace, const tm_kv_cache* cache, tm_stream_t st)
{
    TM_REQUIRE(out && qkv && k_len && cache, "null pointer");
    TM_REQUIRE(cache->bits == 8 || cache->bits == 4, "fused decode prologue: int8 / int4 KV only");
    TM_REQUIRE(qkv_n == (q_heads + 2 * cache->kv_heads) * 128, "qkv_n != (q_heads + 2 kv_heads) * 128");
    DecodeAttnParams p{};
    p.out        = (half_t*)out;
    p.k_len      = k_len;
    p.batch      = batch;
    p.q_heads    = q_heads;
    const float s = softmax_scale > 0.f ? softmax_scale : 1.0f / std::sqrt(128.0f);
    p.scale_log2 = s * 1.4426950408889634f;
    p.splits     = splits < 1 ? 1 : splits;
    p.partial_o  = (float*)workspace;
    p.partial_ml = workspace ? (float*)workspace + (size_t)batch * q_heads * p.splits * 128 : nullptr;
    p.cache      = to_view(cache);
    p.qkv_slabs  = qkv_splits > 0 ? (const float*)qkv : nullptr;
    p.qkv_f16    = qkv_splits > 0 ? nullptr : (const half_t*)qkv;
    p.qkv_splits = qkv_splits;
    p.qkv_n      = qkv_n;
    p.cos_sin    = (const half2_t*)cos_sin;
    p.max_pos    = cos_sin ? max_pos : 1 << 30;
    return launch_decode_attention(p, (hipStream_t)st);
}

int tm_prefill_attention(void* out, const void* q, int q_stride, const void* k, const void* vt, int k_stride,
                         const int* cu_q_len, const int* cu_k_off, const int* k_len, int batch, int max_q_len,
                         int q_heads, int kv_heads, float softmax_scale, tm_stream_t st)
{
    TM_REQUIRE(out && q && k && vt && cu_q_len && cu_k_off && k_len, "null pointer");
    PrefillAttnParams p{};
    p.q          = (const half_t*)q;
    p.q_stride   = q_stride;
    p.out        = (half_t*)out;
    p.k          = (const half_t*)k;
    p.vt         = (const half_t*)vt;
    p.k_stride   = k_stride;
    p.cu_q_len   = cu_q_len;
    p.cu_k_off   = cu_k_off;
    p.k_len      = k_len;
    p.batch      = batch;
    p.max_q_len  = max_q_len;
    p.q_heads    = q_heads;
    p.kv_heads   = kv_heads;
    const float s = softmax_scale > 0.f ? softmax_scale : 1.0f / std::sqrt(128.0f);
    p.scale_log2 = s * 1.4426950408889634f;
    return launch_prefill_attention(p, (hipStream_t)st);
}

int tm_embedding(void* out, const void* table, const int* ids, int tokens, int hidden, int vocab, tm_stream_t st)
{
    TM_REQUIRE(out && table && ids, "null pointer");
    return launch_embedding((half_t*)out, (const half_t*)table, ids, tokens, hidden, vocab, (hipStream_t)st);
}

int tm_argmax(int* out_ids, void* out_val, const void* logits, int batch, int vocab, int ld, tm_stream_t st)
{
    TM_REQUIRE(out_ids && logits, "null pointer");
    return launch_argmax(out_ids, (half_t*)out_val, (const half_t*)logits, batch, vocab, ld, 0, (hipStream_t)st);
}

int tm_silu_mul(void* out, const void* gate_up, int M, int inter, tm_stream_t st)
{
    TM_REQUIRE(out && gate_up, "null pointer");
    return launch_silu_mul((half_t*)out, (const half_t*)gate_up, M, inter, (hipStream_t)st);
}

int tm_linear_create(tm_linear** out, int in_features, int out_features, int weight_type, int group_size)
{
    TM_REQUIRE(out, "null pointer");
    TM_REQUIRE(weight_type == TM_WEIGHT_U4 || weight_type == TM_WEIGHT_F16 || weight_type == TM_WEIGHT_FP8, "weight_type");
    TM_REQUIRE(in_features > 0 && out_features > 0, "shape");
    auto* l     = new tm_linear();
    l->w.K      = in_features;
    l->w.N      = out_features;
    l->w.group  = group_size;
    l->w.type   = weight_type;
    *out        = l;
    return 0;
}

int tm_linear_prepare(tm_linear* w, const void* weight, const void* scales, const void* zeros, tm_stream_t st)
{
    TM_REQUIRE(w && weight, "null pointer");
    if (w->w.type == TM_WEIGHT_U4) {
        TM_REQUIRE(scales && zeros, "u4 weights need scales and zeros");
        return linear_weight_prepare_u4(w->w, (const int32_t*)weight, (const half_t*)scales, (const half_t*)zeros,
                                        (hipStream_t)st);
    }
    if (w->w.type == TM_WEIGHT_FP8) {
        TM_REQUIRE(scales, "fp8 weights need their 128x128 block scales");
        return linear_weight_prepare_fp8(w->w, (const uint8_t*)weight, (const float*)scales, false, (hipStream_t)st);
    }
    return linear_weight_prepare_f16(w->w, (const half_t*)weight, (hipStream_t)st);
}

size_t tm_linear_workspace(const tm_linear* w, int M)
{
    if (!w) {
        return 0;
    }
    return gemm_workspace_bytes(M, w->w.N, 16) + 8192;  // + the arrival counters of the in-launch merged split-K tiles (kShapeMerge)
}

int tm_linear_dequant_f16(const tm_linear* w, void* out_nk, tm_stream_t st)
{
    TM_REQUIRE(w && out_nk, "null pointer");
    return launch_dequant_p32_f16((half_t*)out_nk, w->w, (hipStream_t)st);
}


int tm_linear_forward(const tm_linear* w, const void* x, int ldx, void* y, int ldy, int M, int gated_silu, int nt,
                      int splits, int waves, void* workspace, tm_stream_t st)
{
    TM_REQUIRE(w && x && y, "null pointer");
    GemmConfig cfg = gemm_pick_config(w->w, M);  // (workspace: tm_linear_workspace(w, M) bytes)
    if (nt > 0) {
        cfg.nt = nt;
    }
    if (splits > 0) {
        cfg.splits = splits;
    }
    if (nt > 0 || (waves > 0 && !(waves & 0x200))) {
        cfg.d32_shape = -1;  // an explicit tiling of the general kernel
        if (splits <= 0) {
            cfg.splits = gemm_pick_config_general(w->w, M).splits;
        }
    }
    if (waves > 0 && (waves & 0x200) && (waves & 0xff) == kShapeF16 && !w->w.image16) {
        // operator level: the fp16 image is built on first use (the engine builds it at load)
        TM_TRY_RC(linear_weight_build_f16_image(const_cast<tm_linear*>(w)->w, (hipStream_t)st));
    }
    if (waves > 0 && (waves & 0x200)) {
        // 0x200 + shape: the decode kernel (gemm_decode.hip) with an explicit workgroup shape; M <= 64, u4, N % 32 == 0
        TM_REQUIRE(dec32_supported(w->w, M), "decode kernel: u4 weights, N % 32 == 0");
        cfg.d32_shape = waves & 0xff;
        TM_REQUIRE((cfg.d32_shape >= 6 && cfg.d32_shape <= 9) || (cfg.d32_shape <= 5 && (cfg.d32_shape >= 4) == (M > 64))
                       || (cfg.d32_shape == kShapeWide2 && M <= 64)
                       || (cfg.d32_shape == kShapeLC && M <= 64) || ((cfg.d32_shape == kShapePre256 || cfg.d32_shape == kShapeF16) && M > 64)
                       || (dec32_is_merge_shape(cfg.d32_shape) && M <= 64),
                   "P32 kernel shape 0..3 / 11 (M <= 64), 4 / 5 / 12 (M > 64), 6..9 (32-row blocks, any M) or 16 + (0..3 | 6..9) "
                   "(M <= 64: split-K merged inside the launch)");
        waves = 0;
    }
    if (dec32_is_merge_shape(cfg.d32_shape) && workspace && cfg.splits > 1) {
        // arrival counters of the in-launch merge: the 8 KB behind the 16 slabs (tm_linear_workspace)
        cfg.tickets = (unsigned*)((char*)workspace + gemm_workspace_bytes(M, w->w.N, 16));
        TM_HIP_CHECK(hipMemsetAsync(cfg.tickets, 0, 8192, (hipStream_t)st));
    }
    if (waves > 0) {
        // waves per workgroup (4 | 8); + 0x100 = split K two ways INSIDE the workgroup (8 waves only)
        TM_REQUIRE((waves & 0xff) == 4 || (waves & 0xff) == 8 || (waves & 0xff) == 16,
                   "waves in {4, 8, 16} (+0x100: two k-phases)");
        cfg.waves   = waves & 0xff;
        cfg.kphases = (waves & 0x100) ? 2 : 1;
    }
    TM_REQUIRE(cfg.splits <= 16, "splits <= 16");
    if (!workspace) {
        cfg.splits = 1;
    }
    return launch_linear(w->w, (const half_t*)x, ldx, (half_t*)y, ldy, M, gated_silu != 0, cfg, (float*)workspace, false,
                         nullptr, (hipStream_t)st);
}

int tm_linear_residual_norm(const tm_linear* w, const void* x, int ldx, void* y, void* resid, const void* norm_w, float eps, int M,
                            int shape, int splits, void* workspace, tm_stream_t st)
{
    TM_REQUIRE(w && x && y && resid && norm_w && workspace, "null pointer");
    const int N = w->w.N;
    GemmConfig cfg = gemm_pick_config(w->w, M);
    if (shape >= 0) {
        TM_REQUIRE(dec32_supported(w->w, M) && (shape <= 3 || (shape >= 6 && shape <= 10) || shape == kShapeLC)
                       && (M <= 64 || (M <= kFoldMaxRows && shape >= 6 && shape <= 9)),
                   "decode tile 0..3 / 6..10 / 11 at M <= 64; the 32-row-block tiles 6..9 up to 128 rows");
        cfg.d32_shape = shape;
    }
    if (splits > 0) {
        cfg.splits = splits;
    }
    TM_REQUIRE(cfg.splits <= 16, "splits <= 16");
    // slabs (or, with one slice, the fp16 product parked at the end of the workspace) + the reduce-norm kernel
    int     slabs = 1;
    half_t* tmp   = (half_t*)((char*)workspace + gemm_workspace_bytes(M, N, 16));
    TM_TRY_RC(launch_linear(w->w, (const half_t*)x, ldx, tmp, N, M, false, cfg, (float*)workspace, cfg.splits > 1, &slabs, (hipStream_t)st));
    return launch_residual_rmsnorm((half_t*)y, (half_t*)resid, slabs > 1 ? nullptr : tmp, slabs > 1 ? (const float*)workspace : nullptr,
                                   slabs, nullptr, (const half_t*)norm_w, eps, M, N, (hipStream_t)st);
}

size_t tm_linear_fold_workspace(const tm_linear* w, int M)
{
    return w ? gemm_workspace_bytes(M, w->w.N, 16) + 4096 : 0;
}

static int fold_tile_pick(const tm_linear* w, int M, int* shape, int* splits, bool producer)
{
    int sh = *shape, sp = *splits;
    if (sh < 0) {
        dec32_pick(w->w, M, &sh, &sp);
        if (!dec32_fold_shape_m(sh, M, producer)) {
            dec32_pick_ex(w->w, M, &sh, &sp, false);
        }
        if (!dec32_fold_shape_m(sh, M, producer)) {
            sh = M <= 64 ? 0 : (producer ? 7 : 4);
            sp = 1;
        }
        if (*splits > 0) {
            sp = *splits;
        }
    }
    else if (sp <= 0) {
        sp = 1;
    }
    TM_REQUIRE(dec32_fold_shape_m(sh, M, producer) && sp >= 1 && sp <= 16,
               "folded RMSNorm: decode tile 0..3 / 6..10 (M <= 64), 6..9 or -- consumer only -- 4 (64 < M <= 128); 1 <= splits <= 16");
    *shape  = sh;
    *splits = sp;
    return 0;
}

int tm_linear_fold_produce(const tm_linear* w, const void* x, int ldx, void* xg, void* resid, const void* norm_w, float* ss,
                           int* ss_tiles, int M, int shape, int splits, void* workspace, tm_stream_t st)
{
    TM_REQUIRE(w && x && xg && resid && norm_w && ss && ss_tiles && workspace, "null pointer");
    TM_REQUIRE(dec32_supported(w->w, M) && M <= kFoldMaxRows && w->w.N % 64 == 0, "folded RMSNorm: u4 decode linear, M <= 128, N % 64 == 0");
    TM_TRY_RC(fold_tile_pick(w, M, &shape, &splits, true));
    unsigned* tickets = (unsigned*)((char*)workspace + gemm_workspace_bytes(M, w->w.N, 16));
    TM_HIP_CHECK(hipMemsetAsync(tickets, 0, 4096, (hipStream_t)st));
    TM_REQUIRE((size_t)(w->w.N / 64) * ((M + 31) / 32) * sizeof(unsigned) <= 4096, "folded RMSNorm: N / 64 column tiles x row blocks <= 1024");
    NormFold nf{};
    nf.resid   = (half_t*)resid;
    nf.norm_w  = (const half_t*)norm_w;
    nf.ss_out  = ss;
    nf.tickets = tickets;
    TM_TRY_RC(launch_linear_dec32(w->w, (const half_t*)x, ldx, (half_t*)xg, w->w.N, M, false, shape, splits, (float*)workspace, nullptr,
                                  (hipStream_t)st, &nf));
    *ss_tiles = nf.tiles_out;
    return 0;
}

int tm_linear_fold_consume(const tm_linear* w, const void* xg, int ldx, void* y, int ldy, int M, int gated_silu, const float* ss,
                           int ss_tiles, int norm_h, float eps, int shape, int splits, void* workspace, tm_stream_t st)
{
    TM_REQUIRE(w && xg && y && ss, "null pointer");
    TM_REQUIRE(dec32_supported(w->w, M) && M <= kFoldMaxRows && ss_tiles >= 1 && norm_h >= 1, "folded RMSNorm: u4 decode linear, M <= 128");
    TM_TRY_RC(fold_tile_pick(w, M, &shape, &splits, false));
    if (!workspace) {
        splits = 1;
    }
    NormFold nf{};
    nf.ss_in    = ss;
    nf.ss_tiles = ss_tiles;
    nf.inv_h    = 1.0f / (float)norm_h;
    nf.eps      = eps;
    int nslab   = 1;
    TM_TRY_RC(launch_linear_dec32(w->w, (const half_t*)xg, ldx, (half_t*)y, ldy, M, gated_silu != 0, shape, splits, (float*)workspace, &nslab,
                                  (hipStream_t)st, &nf));
    if (nslab > 1) {  // the slabs carry the row factor already (the engine hands them to the attention prologue as they are)
        return launch_splitk_reduce((half_t*)y, ldy, (const float*)workspace, nslab, M, w->w.N, gated_silu != 0, (hipStream_t)st);
    }
    return 0;
}

int tm_linear_prepare_fp8_gated(tm_linear* w, const void* weight, const void* scales, tm_stream_t st)
{
    TM_REQUIRE(w && weight && scales, "null pointer");
    TM_REQUIRE(w->w.type == TM_WEIGHT_FP8, "tm_linear_prepare_fp8_gated: fp8 weights only");
    return linear_weight_prepare_fp8(w->w, (const uint8_t*)weight, (const float*)scales, true, (hipStream_t)st);
}

size_t tm_linear_fp8_workspace(const tm_linear* w, int M)
{
    if (!w) {
        return 0;
    }
    return (fp8_act_workspace_bytes(M, w->w.K) + 255) / 256 * 256 + gemm_workspace_bytes(M, w->w.N, 16);
}

int tm_quant_fp8_rows(void* xq, float* sx, const void* x, int ldx, int M, int K, int ldsx, tm_stream_t st)
{
    TM_REQUIRE(xq && sx && x, "null pointer");
    return launch_quant_fp8_rows((uint8_t*)xq, sx, (const half_t*)x, ldx, M, K, ldsx, (hipStream_t)st);
}

int tm_linear_forward_fp8(const tm_linear* w, const void* x, int ldx, void* y, int ldy, int M, int gated_silu, int splits,
                          void* workspace, tm_stream_t st)
{
    TM_REQUIRE(w && x && y && workspace, "null pointer");
    TM_REQUIRE(fp8_mfma_supported(w->w), "fp8 x fp8 linear: e4m3 weights, N % 32 == 0 (TM_FP8_MFMA=0 disables the path)");
    TM_REQUIRE(splits >= 0 && splits <= 16, "0 <= splits <= 16");
    const int K    = w->w.K;
    uint8_t*  xq   = (uint8_t*)workspace;
    const int ldsx = (M + 3) / 4 * 4;
    float*    sx   = (float*)(xq + (size_t)M * K);
    float*    slab = (float*)((char*)workspace + (fp8_act_workspace_bytes(M, K) + 255) / 256 * 256);
    int       rc   = launch_quant_fp8_rows(xq, sx, (const half_t*)x, ldx, M, K, ldsx, (hipStream_t)st);
    if (rc) {
        return rc;
    }
    int nslab = 1;
    rc        = launch_linear_fp8(w->w, xq, sx, ldsx, (half_t*)y, ldy, M, gated_silu != 0, splits, slab, &nslab, (hipStream_t)st);
    if (rc) {
        return rc;
    }
    if (nslab > 1) {
        return launch_splitk_reduce((half_t*)y, ldy, slab, nslab, M, w->w.N, gated_silu != 0, (hipStream_t)st);
    }
    return 0;
}

int tm_linear_destroy(tm_linear* w)
{
    if (w) {
        linear_weight_free(w->w);
        delete w;
    }
    return 0;
}

/* debug: device buffer of [workgroups][4] uint64 receiving s_memrealtime stamps of every GEMM workgroup (NULL = off) */
size_t tm_sample_workspace(int batch)
{
    return sample_workspace_bytes(batch);
}

int tm_sample(int* out_ids, int* kept_out, const void* logits, int batch, int vocab, int ld, const float* temperature,
              const int* top_k, const float* top_p, const float* min_p, const float* uniform, void* workspace,
              tm_stream_t st)
{
    return launch_sample(out_ids, kept_out, (const half_t*)logits, batch, vocab, ld, temperature, top_k, top_p, min_p,
                         uniform, workspace, (hipStream_t)st);
}

int tm_sample_logprobs(int* out_ids, int* kept_out, float* vals, int* idx, int* num, float* sel, int cap, const void* logits,
                       int batch, int vocab, int ld, const float* temperature, const int* top_k, const float* top_p,
                       const float* min_p, const float* uniform, void* workspace, tm_stream_t st)
{
    const SampleLogprobs lp{vals, idx, num, sel, cap, nullptr, 0, 1, 0, 1};
    return launch_sample(out_ids, kept_out, (const half_t*)logits, batch, vocab, ld, temperature, top_k, top_p, min_p,
                         uniform, workspace, (hipStream_t)st, &lp);
}

float tm_philox_uniform(uint64_t seed, uint32_t counter)
{
    return philox_uniform_host(seed, counter);
}

int tm_seen_update(void* seen, int seen_words, const int* ids, const int* cu_q, int nseq, int n_tokens, int vocab,
                   tm_stream_t st)
{
    TM_REQUIRE(seen && ids && seen_words >= 1 && nseq >= 1 && (int64_t)seen_words * 32 >= vocab, "seen_update: arguments");
    return launch_seen_update((uint32_t*)seen, seen_words, ids, cu_q, nseq, n_tokens, vocab, (hipStream_t)st);
}

int tm_logits_process(void* logits, int batch, int vocab, int ld, int vocab_offset, const void* seen, int seen_words,
                      const float* rep, const int* ban, const int* end, const int* k_len, const int* min_len, tm_stream_t st)
{
    TM_REQUIRE(logits && seen && rep && ban && end && k_len && min_len, "null pointer");
    static_assert(kMaxBadIds == TM_MAX_BAD_IDS && kMaxEndIds == 1 + TM_MAX_STOP_IDS, "header constants");
    return launch_logits_process((half_t*)logits, batch, vocab, ld, vocab_offset, (const uint32_t*)seen, seen_words, rep,
                                 ban, end, k_len, min_len, (hipStream_t)st);
}

// ---- MoE FFN block ------------------------------------------------------------------------------------------------
struct tm_moe {
    tmk::MoeBlock m;
    int           type = 0;
    bool          prepared = false;
};

int tm_moe_create(tm_moe** out, int hidden, int inter, int experts, int top_k, int weight_type, int norm_topk, float routed_scale)
{
    TM_REQUIRE(out, "null pointer");
    TM_REQUIRE(weight_type == TM_WEIGHT_U4 || weight_type == TM_WEIGHT_FP8, "moe experts: u4 or fp8 weights");
    TM_REQUIRE(hidden % 128 == 0 && inter % 128 == 0 && experts >= 1 && experts <= 64 && top_k >= 1 && top_k <= 8 && top_k <= experts,
               "moe geometry");
    auto* o         = new tm_moe();
    o->m.hidden     = hidden;
    o->m.inter      = inter;
    o->m.experts    = experts;
    o->m.top_k      = top_k;
    o->m.norm_topk  = norm_topk != 0;
    o->m.routed_scale = routed_scale;
    o->m.w13.resize(experts);
    o->m.w2.resize(experts);
    for (int e = 0; e < experts; ++e) {
        o->m.w13[e].K = hidden, o->m.w13[e].N = 2 * inter, o->m.w13[e].type = weight_type;
        o->m.w2[e].K = inter, o->m.w2[e].N = hidden, o->m.w2[e].type = weight_type;
    }
    o->type = weight_type;
    *out    = o;
    return 0;
}

int tm_moe_set_gate(tm_moe* m, const void* gate, tm_stream_t st)
{
    TM_REQUIRE(m && gate, "null pointer");
    const size_t bytes = (size_t)m->m.hidden * m->m.experts * 2;
    if (!m->m.gate) {
        TM_HIP_CHECK(hipMalloc((void**)&m->m.gate, bytes));
    }
    TM_HIP_CHECK(hipMemcpyAsync(m->m.gate, gate, bytes, hipMemcpyDeviceToDevice, (hipStream_t)st));
    return 0;
}

int tm_moe_set_expert(tm_moe* m, int expert, const void* w13_weight, const void* w13_scales, const void* w13_zeros,
                      const void* w2_weight, const void* w2_scales, const void* w2_zeros, tm_stream_t st)
{
    TM_REQUIRE(m && w13_weight && w13_scales && w2_weight && w2_scales, "null pointer");
    TM_REQUIRE(expert >= 0 && expert < m->m.experts, "expert index");
    m->prepared = false;
    if (m->type == TM_WEIGHT_U4) {
        TM_REQUIRE(w13_zeros && w2_zeros, "u4 experts need zeros");
        TM_TRY_RC(linear_weight_prepare_u4(m->m.w13[expert], (const int32_t*)w13_weight, (const half_t*)w13_scales,
                                           (const half_t*)w13_zeros, (hipStream_t)st));
        return linear_weight_prepare_u4(m->m.w2[expert], (const int32_t*)w2_weight, (const half_t*)w2_scales,
                                        (const half_t*)w2_zeros, (hipStream_t)st);
    }
    TM_TRY_RC(linear_weight_prepare_fp8(m->m.w13[expert], (const uint8_t*)w13_weight, (const float*)w13_scales, true, (hipStream_t)st));
    return linear_weight_prepare_fp8(m->m.w2[expert], (const uint8_t*)w2_weight, (const float*)w2_scales, false, (hipStream_t)st);
}

size_t tm_moe_workspace(const tm_moe* m, int tokens)
{
    return m ? moe_workspace_bytes(m->m, tokens) : 0;
}

int tm_moe_forward(tm_moe* m, void* out, const void* x, int tokens, void* workspace, int* topk_ids_out, float* topk_w_out,
                   tm_stream_t st)
{
    TM_REQUIRE(m && out && x && workspace, "null pointer");
    if (!m->prepared) {
        TM_TRY_RC(moe_prepare(m->m, (hipStream_t)st));
        m->prepared = true;
    }
    return moe_forward(m->m, (half_t*)out, m->m.hidden, (const half_t*)x, m->m.hidden, tokens, workspace, topk_ids_out, topk_w_out,
                       (hipStream_t)st);
}

int tm_moe_destroy(tm_moe* m)
{
    if (m) {
        moe_free(m->m);
        delete m;
    }
    return 0;
}

// ---- host-only scheduler hooks (scheduler.h) ----------------------------------------------------------------------
struct tm_sched {
    tmk::BatchScheduler impl;
    tm_sched(int b, int n, int s): impl(b, n, s) {}
};

int tm_prefill_split(const int* cu_q, int nseq, int min_rows, int* seqs_a, int* rows_a)
{
    TM_REQUIRE(cu_q && seqs_a && rows_a, "null pointer");
    TM_REQUIRE(nseq >= 1 && cu_q[0] == 0, "cu_q: nseq + 1 ascending row offsets from 0");
    for (int s = 0; s < nseq; ++s) {
        TM_REQUIRE(cu_q[s + 1] > cu_q[s], "cu_q: nseq + 1 ascending row offsets from 0");
    }
    *seqs_a = tmk::prefill_microbatch_split(cu_q, nseq, min_rows);
    *rows_a = *seqs_a > 0 ? cu_q[*seqs_a] : 0;
    return 0;
}

int tm_sched_create(tm_sched** out, int max_batch, int num_blocks, int session_len)
{
    TM_REQUIRE(out && max_batch >= 1 && num_blocks >= 1 && session_len >= 2, "bad scheduler geometry");
    *out = new tm_sched(max_batch, num_blocks, session_len);
    return 0;
}

int tm_sched_destroy(tm_sched* s)
{
    delete s;
    return 0;
}

int tm_sched_submit(tm_sched* s, const int* ids, int n, int max_new_tokens, int eos_id, int64_t* req_id)
{
    TM_REQUIRE(s && req_id, "null pointer");
    return s->impl.submit(ids, n, max_new_tokens, eos_id, req_id);
}

int tm_sched_admit(tm_sched* s, int token_budget, int64_t* req_ids, int* slots, int cap, int* n_admitted)
{
    TM_REQUIRE(s && req_ids && slots && n_admitted, "null pointer");
    const auto a = s->impl.admit(token_budget);
    TM_REQUIRE((int)a.size() <= cap, "output arrays too small");
    for (size_t i = 0; i < a.size(); ++i) {
        req_ids[i] = a[i].id;
        slots[i]   = a[i].slot;
    }
    *n_admitted = (int)a.size();
    return 0;
}

int tm_sched_on_token(tm_sched* s, int slot, int token, int* finished)
{
    TM_REQUIRE(s && finished, "null pointer");
    *finished = s->impl.on_token(slot, token) ? 1 : 0;
    return 0;
}

int tm_sched_cancel(tm_sched* s, int64_t req_id, int* released_slot)
{
    TM_REQUIRE(s, "null pointer");
    return s->impl.cancel(req_id, released_slot);
}

int tm_sched_query(tm_sched* s, int64_t req_id, int* status, int* slot, int* n_generated, int* n_blocks)
{
    TM_REQUIRE(s, "null pointer");
    const tmk::SchedRequest* r = s->impl.find(req_id);
    if (!r) {
        return TM_INVALID;
    }
    if (status) *status = r->status;
    if (slot) *slot = r->slot;
    if (n_generated) *n_generated = (int)r->out.size();
    if (n_blocks) *n_blocks = (int)r->blocks.size();
    return 0;
}

int tm_sched_admit_ready(tm_sched* s, int* ready)
{
    TM_REQUIRE(s && ready, "null pointer");
    *ready = s->impl.admit_ready() ? 1 : 0;
    return 0;
}

int tm_sched_counts(tm_sched* s, int* n_active, int* n_waiting, int* n_free_blocks)
{
    TM_REQUIRE(s, "null pointer");
    if (n_active) *n_active = s->impl.n_active();
    if (n_waiting) *n_waiting = s->impl.n_waiting();
    if (n_free_blocks) *n_free_blocks = s->impl.n_free_blocks();
    return 0;
}

int tm_sched_forget(tm_sched* s, int64_t req_id)
{
    TM_REQUIRE(s, "null pointer");
    if (!s->impl.erase(req_id)) {
        set_last_error("unknown or unfinished request id");
        return TM_INVALID;
    }
    return 0;
}

int tm_sched_abort_all(tm_sched* s, int status)
{
    TM_REQUIRE(s, "null pointer");
    TM_REQUIRE(status != 0, "abort status must be non-zero");
    s->impl.abort_all(status);
    return 0;
}

/* ---- native P2P communicator pieces (comm_p2p.hip) --------------------------------------------------------------------- */
int tm_p2p_segment_create(size_t bytes, void** dev_ptr, void* handle64)
{
    TM_REQUIRE(dev_ptr && handle64 && bytes > 0, "null pointer");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
    void* p = nullptr;
    // Flags and tiles are written and polled MID-KERNEL by other devices: that needs fine-grained (device-coherent) memory.
    // On coarse-grained hipMalloc memory a flag another device wrote is not guaranteed to become visible before the kernel
    // ends, so there is no silent fall-back: the caller stays on RCCL instead (TM_P2P_COARSE_OK=1 overrides, for experiments).
    if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained) != hipSuccess) {
        (void)hipGetLastError();
        const char* ok = getenv("TM_P2P_COARSE_OK");
        TM_REQUIRE(ok && atoi(ok), "native communicator: fine-grained device memory is unavailable (hipExtMallocWithFlags failed)");
        fprintf(stderr, "[tm] native communicator on COARSE-grained memory (TM_P2P_COARSE_OK=1): cross-device visibility is not guaranteed\n");
        TM_HIP_CHECK(hipMalloc(&p, bytes));
    }
    TM_HIP_CHECK(hipMemset(p, 0, bytes));
    TM_HIP_CHECK(hipDeviceSynchronize());
    hipIpcMemHandle_t h;
    TM_HIP_CHECK(hipIpcGetMemHandle(&h, p));
    memcpy(handle64, &h, sizeof(h));
    *dev_ptr = p;
    return 0;
}

int tm_p2p_segment_open(const void* handle64, void** dev_ptr)
{
    TM_REQUIRE(dev_ptr && handle64, "null pointer");
    hipIpcMemHandle_t h;
    memcpy(&h, handle64, sizeof(h));
    TM_HIP_CHECK(hipIpcOpenMemHandle(dev_ptr, h, hipIpcMemLazyEnablePeerAccess));
    return 0;
}

int tm_p2p_segment_close(void* dev_ptr, int opened)
{
    if (!dev_ptr) {
        return 0;
    }
    if (opened) {
        TM_HIP_CHECK(hipIpcCloseMemHandle(dev_ptr));
    }
    else {
        TM_HIP_CHECK(hipFree(dev_ptr));
    }
    return 0;
}

size_t tm_p2p_segment_bytes(int rows, int H)
{
    return 256 + 2 * (size_t)rows * (size_t)H * sizeof(half_t);
}

static void p2p_tables(void* const* segs, int tp, half_t** data, uint32_t** flags)
{
    for (int r = 0; r < tp && r < 8; ++r) {
        flags[r] = (uint32_t*)segs[r];
        data[r]  = (half_t*)((char*)segs[r] + 256);
    }
}

int tm_p2p_allreduce_norm(void* const* segs, int tp, int me, void* state, int rows, const void* partial, void* y, void* resid,
                          const void* weight, float eps, int M, int H, tm_stream_t st)
{
    TM_REQUIRE(segs && state && partial && y && resid && weight, "null pointer");
    TM_REQUIRE(tp >= 1 && tp <= 8, "p2p: 1 <= tp <= 8");
    half_t*   data[8];
    uint32_t* flags[8];
    p2p_tables(segs, tp, data, flags);
    return launch_p2p_allreduce_norm(data, flags, tp, me, (uint32_t*)state, (size_t)rows * H, (const half_t*)partial, (half_t*)y,
                                     (half_t*)resid, (const half_t*)weight, eps, M, H, (hipStream_t)st);
}

size_t tm_p2p_segment_bytes2(int rows, int rows2, int H)
{
    return 256 + 2 * ((size_t)rows + (size_t)rows2) * (size_t)H * sizeof(half_t);
}

int tm_p2p_allreduce_norm_2shot(void* const* segs, int tp, int me, void* state, int rows, int rows2, const void* partial, void* y,
                                void* resid, const void* weight, float eps, int M, int H, tm_stream_t st)
{
    TM_REQUIRE(segs && state && partial && y && resid && weight, "null pointer");
    TM_REQUIRE(tp >= 1 && tp <= 8 && rows >= 0 && rows2 >= 1, "p2p: 1 <= tp <= 8, rows2 >= 1");
    half_t*   data[8];
    uint32_t* flags[8];
    half_t *  in2[8], *out2[8];
    p2p_tables(segs, tp, data, flags);
    for (int r = 0; r < tp; ++r) {  // [flags | tile 0 | tile 1 | in2 | out2]
        in2[r]  = data[r] + 2 * (size_t)rows * H;
        out2[r] = in2[r] + (size_t)rows2 * H;
    }
    return launch_p2p_allreduce_norm_2shot(in2, out2, flags, tp, me, (uint32_t*)state, (size_t)rows2 * H, (const half_t*)partial, (half_t*)y,
                                           (half_t*)resid, (const half_t*)weight, eps, M, H, (hipStream_t)st);
}

size_t tm_p2p_segment_bytes_rows(int rows, int rows2, int H)
{
    return tm_p2p_segment_bytes2(rows, rows2, H) + 2 * (size_t)rows * (size_t)H * sizeof(half_t) + 8 * (size_t)rows * sizeof(uint32_t);
}

int tm_p2p_allreduce_norm_rows(void* const* segs, int tp, int me, void* state, int rows, int rows2, const void* partial, void* y, void* resid,
                               const void* weight, float eps, int M, int H, tm_stream_t st)
{
    TM_REQUIRE(segs && state && partial && y && resid && weight, "null pointer");
    TM_REQUIRE(tp >= 1 && tp <= 8 && rows >= 1 && rows2 >= 0, "p2p: 1 <= tp <= 8");
    half_t*   rdata[8];
    uint32_t* rflags[8];
    for (int r = 0; r < tp; ++r) {  // [flags | tile 0 | tile 1 | in2 | out2 | row tile 0 | row tile 1 | row flags [8][rows]]
        rdata[r]  = (half_t*)((char*)segs[r] + tm_p2p_segment_bytes2(rows, rows2, H));
        rflags[r] = (uint32_t*)(rdata[r] + 2 * (size_t)rows * H);
    }
    return launch_p2p_allreduce_norm_rows(rdata, rflags, rows, tp, me, (uint32_t*)state, (size_t)rows * H, (const half_t*)partial, (half_t*)y,
                                          (half_t*)resid, (const half_t*)weight, eps, M, H, (hipStream_t)st);
}

int tm_p2p_allgather(void* const* segs, int tp, int me, void* state, int rows, int H, const void* src, void* dst, int words,
                     tm_stream_t st)
{
    TM_REQUIRE(segs && state && src && dst, "null pointer");
    TM_REQUIRE(tp >= 1 && tp <= 8, "p2p: 1 <= tp <= 8");
    half_t*   data[8];
    uint32_t* flags[8];
    p2p_tables(segs, tp, data, flags);
    return launch_p2p_allgather(data, flags, tp, me, (uint32_t*)state, (size_t)rows * H, src, dst, words, (hipStream_t)st);
}

int tm_debug_pick_tiling(int K, int N, int M, int use_table, int* shape, int* splits)
{
    TM_REQUIRE(shape && splits && K > 0 && N > 0 && M > 0, "arguments");
    TM_REQUIRE(K % 128 == 0 && N % 32 == 0, "the decode kernels take K % 128 == 0, N % 32 == 0");
    LinearWeight w{};
    w.K    = K;
    w.N    = N;
    w.role = (use_table >> 8) & 0xf;  // 0 any, 1 w_qkv, 2 wo, 3 w1w3, 4 w2: the measured table is keyed (role, K, N, M)
    TM_REQUIRE(w.role <= 4, "role 0 .. 4 in bits 8 .. 11 of use_table");
    dec32_pick_ex(w, M, shape, splits, (use_table & 1) != 0);
    return 0;
}

int tm_debug_pick_general(int weight_type, int role, int K, int N, int M, int* config4)
{
    TM_REQUIRE(config4 && K > 0 && N > 0 && M > 0 && weight_type >= 0 && weight_type <= 2 && role >= 0 && role <= 7, "arguments");
    TM_REQUIRE(K % 128 == 0 && N % 16 == 0, "gemm_kernel takes K % 128 == 0, N % 16 == 0");
    LinearWeight w{};
    w.K    = K;
    w.N    = N;
    w.type = weight_type;
    w.role = role;
    const GemmConfig c = gemm_pick_config_general(w, M);
    config4[0] = c.nt, config4[1] = c.splits, config4[2] = c.waves, config4[3] = c.kphases;
    return 0;
}

int tm_debug_grouped_tile(int weight_type, int K, int N, int tokens, int* rows)
{
    TM_REQUIRE(rows && K > 0 && N > 0 && tokens > 0 && (weight_type == 0 || weight_type == 2), "arguments");
    int tv[4];
    *rows = gen_table_get(kGenGrouped + weight_type, 0, K, N, dec32_m_bucket(tokens), tv) ? tv[0] : 0;
    return 0;
}

int tm_debug_tiling_candidates(int K, int N, int M, int* shapes, int* splits, int cap, int* count)
{
    TM_REQUIRE(shapes && splits && count && cap >= 0 && K > 0 && N > 0 && M > 0, "arguments");
    TM_REQUIRE(K % 128 == 0 && N % 32 == 0, "the decode kernels take K % 128 == 0, N % 32 == 0");
    LinearWeight w{};
    w.K = K;
    w.N = N;
    int       cand[128][2];
    const int n = dec32_candidates(w, M, cand, 128);
    for (int i = 0; i < n && i < cap; ++i) {
        shapes[i] = cand[i][0];
        splits[i] = cand[i][1];
    }
    *count = n;
    return 0;
}

int tm_debug_set_block_stride(int stride)
{
    TM_REQUIRE(stride >= 0, "stride >= 0");
    g_dbg_block_stride = stride;
    return 0;
}

int tm_debug_set_gemm_trace(void* dev_buf)
{
    tmk::g_gemm_dbg = (uint64_t*)dev_buf;
    return 0;
}

int tm_debug_trace_arena(void* dev_buf, int64_t capacity_workgroups)
{
    TM_REQUIRE(!dev_buf || capacity_workgroups > 0, "trace arena: capacity in workgroups (8 x 8 bytes each)");
    tmk::trace_arena_set((uint64_t*)dev_buf, dev_buf ? (size_t)capacity_workgroups : 0);
    return 0;
}

int64_t tm_debug_trace_records(char* host_out, int64_t cap)
{
    return (int64_t)tmk::trace_arena_records(host_out, cap > 0 ? (size_t)cap : 0);
}

int tm_quantize_groupwise(void* qweight, void* scales, void* zeros, void* dequant, const void* w, int K, int N,
                          int group_size, tm_stream_t st)
{
    TM_REQUIRE(qweight && scales && zeros && w, "null pointer");
    return launch_quantize_groupwise_u4((int32_t*)qweight, (half_t*)scales, (half_t*)zeros, (half_t*)dequant,
                                        (const half_t*)w, K, N, group_size, (hipStream_t)st);
}

}  // extern "C"
