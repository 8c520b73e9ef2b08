// Paged KV cache: fused RoPE + online int8/int4 quantise-and-store, and dequantising flatten.
//
// Replaces: invokeProcessKV_v2_ / invokeFlattenKV_v2_ (src/turbomind/kernels/attention/kv_cache_utils_v2.cu
//           :18-210,340-468), the decode kernel's store prologue (attention_universal.h:273-330),
//           the quantiser (quantization.h:316-366,428-489) and FastRoPE (rotary_embedding.h:54-194).
//
// Integer contract (bit exact): per (token, kv head), K (after RoPE) and V separately
//   mn = min x, mx = max x; scale = h((f32(mx)-f32(mn)) * (1/(2^b-1))); zero = mn;
//   inv = h(1/f32(scale)); q = sat_u8(rne(h(h(x-zero)*inv))); b==4: min(q,15), nibbles [0,2,4,6,1,3,5,7].
// Byte layout of a block: kernels/attention/block.h:126-219 (KvLayout in tm_kernels.h).
//
// Work mapping: 16 lanes own one 128-wide head row (8 halves = 16 B each, coalesced 256 B per row);
// min/max are wave-level DPP reductions inside the 16-lane row -- no LDS.  HBM-bound byte work.
#include "tm_common.h"
#include "tm_kernels.h"

namespace tmk {

__device__ __forceinline__ int find_seq(const int* cu, int batch, int token)
{
    // largest b with cu[b] <= token   (cu has batch+1 entries, cu[0] = 0)
    int lo = 0, hi = batch;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (cu[mid] <= token) {
            lo = mid;
        }
        else {
            hi = mid;
        }
    }
    return lo;
}

__device__ __forceinline__ float row16_max(float v)
{
    v = fmaxf(v, dpp_f32<DPP_XOR1>(v));
    v = fmaxf(v, dpp_f32<DPP_XOR2>(v));
    v = fmaxf(v, dpp_f32<DPP_HMIRR>(v));
    v = fmaxf(v, dpp_f32<DPP_ROR8>(v));
    return v;
}

template<int BITS>
__global__ __launch_bounds__(256) void kv_rope_store_kernel(half_t* __restrict__ qkv,
                                                            int q_heads,
                                                            const int* __restrict__ cu_q_len,
                                                            const int* __restrict__ k_len,
                                                            int batch,
                                                            int total_tokens,
                                                            const half2_t* __restrict__ cos_sin,
                                                            int         max_pos,
                                                            KvCacheView cache)
{
    constexpr int  D       = 128;
    const KvLayout L       = cache.layout;
    const int      lane16  = threadIdx.x & 15;
    const int      token   = blockIdx.x * 16 + (threadIdx.x >> 4);
    const int      head    = blockIdx.y;  // [0,Hq): q   [Hq,Hq+Hkv): k   [Hq+Hkv, Hq+2Hkv): v
    const int      kv_heads = L.kv_heads;
    if (token >= total_tokens) {
        return;  // whole 16-lane rows exit together; DPP rows are 16 lanes
    }
    const int stride = (q_heads + 2 * kv_heads) * D;
    half_t*   src    = qkv + (size_t)token * stride + (size_t)head * D + lane16 * 8;
    half8_t   x      = *(const half8_t*)src;

    const int b       = find_seq(cu_q_len, batch, token);
    const int q_len   = cu_q_len[b + 1] - cu_q_len[b];
    const int history = k_len[b] - q_len;
    const int pos     = history + (token - cu_q_len[b]);

    const bool is_q = head < q_heads;
    const bool is_k = !is_q && head < q_heads + kv_heads;

    if ((is_q || is_k) && cos_sin != nullptr) {
        // interleaved pairs (x[2i], x[2i+1]); c,s already cast to fp16; fp16 mul/sub/add, no fma
        const int     p  = pos < max_pos ? pos : max_pos - 1;
        const half8_t cs = *(const half8_t*)(cos_sin + (size_t)p * (D / 2) + lane16 * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const half_t c  = cs[2 * i];
            const half_t s  = cs[2 * i + 1];
            const half_t x0 = x[2 * i];
            const half_t x1 = x[2 * i + 1];
            const half_t a0 = c * x0;
            const half_t a1 = s * x1;
            const half_t b0 = c * x1;
            const half_t b1 = s * x0;
            x[2 * i]        = a0 - a1;
            x[2 * i + 1]    = b0 + b1;
        }
    }
    if (is_q) {
        *(half8_t*)src = x;
        return;
    }

    const int kv_head = is_k ? head - q_heads : head - q_heads - kv_heads;
    const int blk     = pos / L.block_len;
    const int ti      = pos - blk * L.block_len;
    char*     block   = (char*)cache.block_ptrs[cache.cu_block_nums[b] + blk] + cache.layer_offset;
    const int doff    = is_k ? L.k_data(kv_head, ti) : L.v_data(kv_head, ti);

    if constexpr (BITS == 16) {
        *(half8_t*)(block + doff + lane16 * 16) = x;
    }
    else {
        float mx = -INFINITY, mnn = -INFINITY;  // mnn tracks max(-x)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float f = (float)x[e];
            mx            = fmaxf(mx, f);
            mnn           = fmaxf(mnn, -f);
        }
        mx             = row16_max(mx);
        mnn            = row16_max(mnn);
        const float mn = -mnn;
        // fp16 min/max are exact in f32
        const float  inv_q_max = 1.0f / (float)((1 << BITS) - 1);
        const half_t scale     = (half_t)((mx - mn) * inv_q_max);
        const half_t zero      = (half_t)mn;
        const half_t inv       = (half_t)(1.0f / (float)scale);
        uint32_t     q[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const half_t d = x[e] - zero;
            const half_t y = d * inv;
            float        r = __builtin_rintf((float)y);  // RNE
            r              = (r == r) ? r : 0.0f;        // cvt.rni.sat of NaN -> 0
            r              = fminf(fmaxf(r, 0.0f), BITS == 8 ? 255.0f : 15.0f);
            q[e]           = (uint32_t)r;
        }
        if constexpr (BITS == 8) {
            u32x2 w;
            w[0] = q[0] | (q[1] << 8) | (q[2] << 16) | (q[3] << 24);
            w[1] = q[4] | (q[5] << 8) | (q[6] << 16) | (q[7] << 24);
            *(u32x2*)(block + doff + lane16 * 8) = w;
        }
        else {
            // nibble i of the word = element [0,2,4,6,1,3,5,7][i]   (quantization.h:459-471)
            const uint32_t w = q[0] | (q[2] << 4) | (q[4] << 8) | (q[6] << 12) | (q[1] << 16) | (q[3] << 20)
                               | (q[5] << 24) | (q[7] << 28);
            *(uint32_t*)(block + doff + lane16 * 4) = w;
        }
        if (lane16 == 0) {
            const int poff = is_k ? L.k_param(kv_head, ti) : L.v_param(kv_head, ti);
            half2_t   pr   = {scale, zero};
            *(half2_t*)(block + poff) = pr;
        }
    }
}

int launch_kv_rope_store(half_t*        qkv,
                         int            q_heads,
                         const int*     cu_q_len,
                         const int*     k_len,
                         int            batch,
                         int            total_tokens,
                         const half2_t* cos_sin,
                         int            max_pos,
                         KvCacheView    cache,
                         hipStream_t    st)
{
    TM_REQUIRE(cache.layout.head_dim == 128, "head_dim must be 128");
    TM_REQUIRE(cache.layout.bits == 16 || cache.layout.bits == 8 || cache.layout.bits == 4, "kv bits in {16,8,4}");
    if (total_tokens == 0) {
        return 0;
    }
    dim3 grid((total_tokens + 15) / 16, q_heads + 2 * cache.layout.kv_heads);
    switch (cache.layout.bits) {
        case 16:
            kv_rope_store_kernel<16><<<grid, 256, 0, st>>>(qkv, q_heads, cu_q_len, k_len, batch, total_tokens, cos_sin, max_pos, cache);
            break;
        case 8:
            kv_rope_store_kernel<8><<<grid, 256, 0, st>>>(qkv, q_heads, cu_q_len, k_len, batch, total_tokens, cos_sin, max_pos, cache);
            break;
        default:
            kv_rope_store_kernel<4><<<grid, 256, 0, st>>>(qkv, q_heads, cu_q_len, k_len, batch, total_tokens, cos_sin, max_pos, cache);
    }
    TM_HIP_CHECK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Flatten: one workgroup = 64 cached tokens (one block) of one (sequence, kv head).
//   K -> k_out[head][cu_k_off[b] + t][D];  V -> v_out[head][...][D] or transposed v_out[head][d][cu_k_off[b]+t]
// Dequant (two roundings, quantization.h:565-573,694-703): h(h(h(q)*scale) + zero).
// ------------------------------------------------------------------------------------------------
template<int BITS>
__device__ __forceinline__ half8_t load_dequant_flatten(const char* data, const char* param, int lane16)
{
    // returns dims [8*lane16, 8*lane16+8) of one token
    if constexpr (BITS == 16) {
        return *(const half8_t*)(data + lane16 * 16);
    }
    else {
        const half2_t pr = *(const half2_t*)param;
        const half_t  s = pr[0], z = pr[1];
        half8_t       o;
        if constexpr (BITS == 8) {
            const u32x2 w = *(const u32x2*)(data + lane16 * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const uint32_t qv = (w[e >> 2] >> (8 * (e & 3))) & 0xffu;
                const half_t   t  = (half_t)(float)qv * s;
                o[e]              = t + z;
            }
        }
        else {
            const uint32_t w = *(const uint32_t*)(data + lane16 * 4);
            constexpr int  nib_of[8] = {0, 4, 1, 5, 2, 6, 3, 7};  // element e sits in nibble nib_of[e]
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const uint32_t qv = (w >> (4 * nib_of[e])) & 0xfu;
                const half_t   t  = (half_t)(float)qv * s;
                o[e]              = t + z;
            }
        }
        return o;
    }
}

template<int BITS, bool VT>
__global__ __launch_bounds__(256) void flatten_kv_kernel(half_t* __restrict__ k_out,
                                                         half_t* __restrict__ v_out,
                                                         const int* __restrict__ cu_k_off,
                                                         const int* __restrict__ k_len,
                                                         int         k_stride,
                                                         KvCacheView cache)
{
    constexpr int  D  = 128;
    const KvLayout L  = cache.layout;
    const int      b  = blockIdx.z;
    const int      hd = blockIdx.y;
    const int      t0 = blockIdx.x * 64;
    const int      n  = k_len[b];
    if (t0 >= n) {
        return;
    }
    __shared__ half_t vt_smem[VT ? 64 * (D + 8) : 1];

    const int    lane16 = threadIdx.x & 15;
    const int    row    = threadIdx.x >> 4;  // 16 token rows per pass
    const char*  block  = (const char*)cache.block_ptrs[cache.cu_block_nums[b] + blockIdx.x] + cache.layer_offset;
    const size_t obase  = (size_t)hd * k_stride + cu_k_off[b];

#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int  ti    = pass * 16 + row;
        const bool valid = t0 + ti < n;
        half8_t    kk = {}, vv = {};
        if (valid) {
            kk = load_dequant_flatten<BITS>(block + L.k_data(hd, ti), block + L.k_param(hd, ti), lane16);
            vv = load_dequant_flatten<BITS>(block + L.v_data(hd, ti), block + L.v_param(hd, ti), lane16);
            *(half8_t*)(k_out + (obase + t0 + ti) * D + lane16 * 8) = kk;
            if constexpr (!VT) {
                *(half8_t*)(v_out + (obase + t0 + ti) * D + lane16 * 8) = vv;
            }
        }
        if constexpr (VT) {
            *(half8_t*)(vt_smem + ti * (D + 8) + lane16 * 8) = vv;  // zeros for invalid tokens
        }
    }
    if constexpr (VT) {
        __syncthreads();
        // thread -> (d = tid & 127, token octets); 64 tokens x 128 d = 1024 16-B vectors / 256 thr = 4 each
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int vec = threadIdx.x + it * 256;
            const int d   = vec & 127;
            const int to  = vec >> 7;  // 0..7 token octet
            half8_t   o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                o[e] = vt_smem[(to * 8 + e) * (D + 8) + d];
            }
            // transposed layout: [head][d][k_stride]; padded tail tokens (>= n) are written as zeros,
            // the per-sequence region is 64-aligned by contract so this never spills into a neighbour
            *(half8_t*)(v_out + ((size_t)hd * D + d) * k_stride + cu_k_off[b] + t0 + to * 8) = o;
        }
    }
}

int launch_flatten_kv(half_t*     k_out,
                      half_t*     v_out,
                      int         transpose_v,
                      const int*  cu_k_off,
                      const int*  k_len,
                      int         batch,
                      int         max_k_len,
                      int         k_stride,
                      KvCacheView cache,
                      hipStream_t st)
{
    TM_REQUIRE(cache.layout.head_dim == 128, "head_dim must be 128");
    TM_REQUIRE(cache.layout.block_len == 64, "block_len must be 64");
    TM_REQUIRE(!transpose_v || k_stride % 64 == 0, "transposed V needs a 64-aligned k_stride");
    if (batch == 0 || max_k_len == 0) {
        return 0;
    }
    dim3 grid((max_k_len + 63) / 64, cache.layout.kv_heads, batch);
#define TM_FLATTEN(B_, T_) flatten_kv_kernel<B_, T_><<<grid, 256, 0, st>>>(k_out, v_out, cu_k_off, k_len, k_stride, cache)
    const int bits = cache.layout.bits;
    if (transpose_v) {
        if (bits == 16) TM_FLATTEN(16, true);
        else if (bits == 8) TM_FLATTEN(8, true);
        else TM_FLATTEN(4, true);
    }
    else {
        if (bits == 16) TM_FLATTEN(16, false);
        else if (bits == 8) TM_FLATTEN(8, false);
        else TM_FLATTEN(4, false);
    }
#undef TM_FLATTEN
    TM_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace tmk
