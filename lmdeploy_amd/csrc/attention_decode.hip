// Split-K paged flash-decode for GQA with in-register int8/int4 dequantisation (gfx950).
//
// Replaces: dispatchDecoding -> attention_kernel<...DecodingCtaMap...> (src/turbomind/kernels/attention/
//           decoding.cu:12-39, decoding_template.h:13-88, attention_universal.h:355-553, impl_81616.h:313-349,
//           510-591) and the split-K merge invokeReduceV3 (reduce.cu:13-226,229-298).
//
// Arithmetic (per sequence b, kv head g, tiles of 64 cached tokens = one cache block):
//   k^ = h(fma(h(q_k), scale, zero))  (fp16, single rounding)    S = f32 sum_d k^[d]*q[d]
//   m' = max(m, max S); O *= exp2((m-m')c); L = L*exp2((m-m')c) + sum exp2(S c - m' c)
//   P = h(exp2(S c - m' c)); O += sum f32(P) f32(v^);  out = h(O / L)  or split partials (O, m, L)
//   merge: m* = max m_i; w_i = exp2((m_i - m*)c); out = h(sum w_i O_i / sum w_i L_i)
//
// MI355X mapping.  This path is HBM-bound integer/byte streaming (132 KB/token of context for Llama-3-8B
// int8), so it is built around coalesced 16-byte loads, not around MFMA:
//   * one workgroup = one (kv head, sequence, split); its 4 waves take alternate 64-token cache blocks and keep
//     independent online-softmax state (no barrier in the main loop), merged once through LDS;
//   * one load instruction covers TPI whole token rows (lane -> (token, 16-B chunk)): every 128-B line is
//     fetched exactly once; the q.k dot products are finished with DPP butterflies inside the CPT lanes that
//     share a token (wavefront-level reduction, no LDS), the P.V products accumulate per lane in fp32 and are
//     reduced across lanes once per workgroup;
//   * grid = (kv_heads * head_chunks, batch, splits) -> 512..2048 workgroups for the metric shape.
#include "tm_common.h"
#include "tm_kernels.h"
#include <stdlib.h>

namespace tmk {

template<int BITS>
struct KvTraits;
template<>
struct KvTraits<8> {
    static constexpr int LOAD_BYTES = 16;  // per lane per instruction
    static constexpr int E          = 16;  // head dims per lane
    static constexpr int TILE       = 32;  // tokens per online-softmax step (register budget: 2+ waves/SIMD)
};
template<>
struct KvTraits<16> {
    static constexpr int LOAD_BYTES = 16;
    static constexpr int E          = 8;
    static constexpr int TILE       = 32;  // 256-B rows: keep the register tile at 8 loads
};
template<>
struct KvTraits<4> {
    static constexpr int LOAD_BYTES = 8;
    static constexpr int E          = 16;
    static constexpr int TILE       = 32;
};

// Dequantise one lane-load into E/2 half2 pairs (decode form: single-rounding fma).
template<int BITS>
__device__ __forceinline__ void dequant_chunk(const uint32_t (&raw)[KvTraits<BITS>::LOAD_BYTES / 4],
                                              half2_t s2,
                                              half2_t z2,
                                              half2_t (&out)[KvTraits<BITS>::E / 2])
{
    const half2_t k1024 = {(half_t)1024.0f, (half_t)1024.0f};
    if constexpr (BITS == 16) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            out[i] = bit_cast<half2_t>(raw[i]);
        }
    }
    else if constexpr (BITS == 8) {
        // bytes b0..b3 of a dword are dims 4i..4i+3: 0x64bb is fp16(1024 + b)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t lo = __builtin_amdgcn_perm(0x64646464u, raw[i], 0x04010400u);
            const uint32_t hi = __builtin_amdgcn_perm(0x64646464u, raw[i], 0x04030402u);
            out[2 * i]        = h2_fma(bit_cast<half2_t>(lo) - k1024, s2, z2);
            out[2 * i + 1]    = h2_fma(bit_cast<half2_t>(hi) - k1024, s2, z2);
        }
    }
    else {
        // nibble i (i<4) = element 2i, nibble 4+i = element 2i+1 of each group of 8
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const uint32_t t = ((raw[i] >> (4 * p)) & 0x000f000fu) | 0x64006400u;
                out[4 * i + p]   = h2_fma(bit_cast<half2_t>(t) - k1024, s2, z2);
            }
        }
    }
}

template<int BITS, int HPW>
__global__ __launch_bounds__(256, 2) void decode_attention_kernel(DecodeAttnParams p, int head_chunks)
{
    using Tr                 = KvTraits<BITS>;
    constexpr int D          = 128;
    constexpr int E          = Tr::E;
    constexpr int CPT        = D / E;       // lanes sharing a token (8 or 16)
    constexpr int TILE       = Tr::TILE;
    constexpr int TPI        = 64 / CPT;    // tokens per load instruction (8 or 4)
    constexpr int IPT        = TILE / TPI;  // load instructions per tile (8)
    constexpr int LW         = Tr::LOAD_BYTES / 4;
    constexpr int TOKB       = BITS * D / 8;  // bytes per token row
    const KvLayout L         = p.cache.layout;

    const int kv_head = blockIdx.x / head_chunks;
    const int chunk   = blockIdx.x - kv_head * head_chunks;
    const int b       = blockIdx.y;
    const int split   = blockIdx.z;
    const int group   = p.q_heads / L.kv_heads;
    const int head0   = kv_head * group + chunk * HPW;  // first query head handled here

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c    = lane % CPT;  // chunk of the head dim
    const int tg   = lane / CPT;  // token inside a load instruction

    const int ctx        = p.k_len[b];
    const int tiles      = (ctx + TILE - 1) / TILE;
    const int per_split  = (tiles + p.splits - 1) / p.splits;
    const int tile_begin = split * per_split;
    const int tile_end   = min(tile_begin + per_split, tiles);

    // ---- queries: q[h][E dims of chunk c] as half2 pairs, kept in registers ------------------
    half2_t qv[HPW][E / 2];
#pragma unroll
    for (int h = 0; h < HPW; ++h) {
        const half_t* qp = p.q + (size_t)b * p.q_stride + (size_t)(head0 + h) * D + c * E;
#pragma unroll
        for (int i = 0; i < E / 8; ++i) {
            const half8_t t = *(const half8_t*)(qp + i * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                qv[h][i * 4 + e] = half2_t{t[2 * e], t[2 * e + 1]};
            }
        }
    }

    float m[HPW], lsum[HPW], O[HPW][E];
#pragma unroll
    for (int h = 0; h < HPW; ++h) {
        m[h]    = -INFINITY;
        lsum[h] = 0.f;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            O[h][e] = 0.f;
        }
    }
    const float sc = p.scale_log2;

    const uint64_t* blocks = p.cache.block_ptrs + p.cache.cu_block_nums[b];

    // newest -> oldest, waves interleaved
    for (int tile = tile_end - 1 - wave; tile >= tile_begin; tile -= 4) {
        const int   tok0   = tile * TILE;
        const int   toff   = tok0 & 63;
        const char* base   = (const char*)blocks[tok0 >> 6] + p.cache.layer_offset;
        const char* kdata  = base + L.k_data(kv_head, toff);
        const char* vdata  = base + L.v_data(kv_head, toff);
        const int   ntok   = min(TILE, ctx - tok0);  // >= 1

        // per-token (scale, zero): lane = token
        uint32_t kpar = 0, vpar = 0;
        if constexpr (BITS != 16) {
            kpar = *(const uint32_t*)(base + L.k_param(kv_head, toff + (lane & (TILE - 1))));
            vpar = *(const uint32_t*)(base + L.v_param(kv_head, toff + (lane & (TILE - 1))));
            if (lane >= ntok) {
                kpar = 0;  // scale = zero = 0 -> dequantised value 0, never NaN
                vpar = 0;
            }
        }

        // ---- K: S[r][h] -------------------------------------------------------------------
        uint32_t kraw[IPT][LW];
#pragma unroll
        for (int r = 0; r < IPT; ++r) {
            const char* ptr = kdata + (size_t)(r * TPI + tg) * TOKB + c * Tr::LOAD_BYTES;
            if constexpr (LW == 4) {
                const u32x4 t = __builtin_nontemporal_load((const u32x4*)ptr);  // read once per launch: streaming policy
                kraw[r][0] = t[0], kraw[r][1] = t[1], kraw[r][2] = t[2], kraw[r][3] = t[3];
            }
            else {
                const u32x2 t = __builtin_nontemporal_load((const u32x2*)ptr);
                kraw[r][0] = t[0], kraw[r][1] = t[1];
            }
        }
        float S[IPT][HPW];
#pragma unroll
        for (int r = 0; r < IPT; ++r) {
            half2_t s2 = {}, z2 = {};
            if constexpr (BITS != 16) {
                const uint32_t pr = (uint32_t)__shfl((int)kpar, r * TPI + tg);
                const half2_t  pp = bit_cast<half2_t>(pr);
                s2                = half2_t{pp[0], pp[0]};
                z2                = half2_t{pp[1], pp[1]};
            }
            half2_t kd[E / 2];
            dequant_chunk<BITS>(kraw[r], s2, z2, kd);
            const bool valid = r * TPI + tg < ntok;
#pragma unroll
            for (int h = 0; h < HPW; ++h) {
                float acc = 0.f;
#pragma unroll
                for (int i = 0; i < E / 2; ++i) {
                    acc = __builtin_amdgcn_fdot2(kd[i], qv[h][i], acc, false);
                }
                acc     = group_sum<CPT>(acc);
                S[r][h] = valid ? acc : -INFINITY;
            }
        }

        // ---- V loads (issued before the softmax math so they overlap it) -------------------
        uint32_t vraw[IPT][LW];
#pragma unroll
        for (int r = 0; r < IPT; ++r) {
            const char* ptr = vdata + (size_t)(r * TPI + tg) * TOKB + c * Tr::LOAD_BYTES;
            if constexpr (LW == 4) {
                const u32x4 t = __builtin_nontemporal_load((const u32x4*)ptr);  // read once per launch: streaming policy
                vraw[r][0] = t[0], vraw[r][1] = t[1], vraw[r][2] = t[2], vraw[r][3] = t[3];
            }
            else {
                const u32x2 t = __builtin_nontemporal_load((const u32x2*)ptr);
                vraw[r][0] = t[0], vraw[r][1] = t[1];
            }
        }

        // ---- online softmax ---------------------------------------------------------------
        float alpha[HPW];
        bool  rescale = false;
#pragma unroll
        for (int h = 0; h < HPW; ++h) {
            float tm_ = S[0][h];
#pragma unroll
            for (int r = 1; r < IPT; ++r) {
                tm_ = fmaxf(tm_, S[r][h]);
            }
            tm_              = upper_max<CPT>(tm_);  // S is already uniform inside a token's CPT lanes
            const float mnew = fmaxf(m[h], tm_);
            alpha[h]         = (m[h] == -INFINITY) ? 0.f : fast_exp2((m[h] - mnew) * sc);
            rescale |= (mnew != m[h]);
            m[h] = mnew;
        }
        if (__builtin_amdgcn_readfirstlane((int)rescale)) {  // m is wave-uniform -> uniform branch; alpha==1 otherwise
#pragma unroll
            for (int h = 0; h < HPW; ++h) {
                lsum[h] *= alpha[h];
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    O[h][e] *= alpha[h];
                }
            }
        }

        // ---- P.V ----------------------------------------------------------------------------
#pragma unroll
        for (int r = 0; r < IPT; ++r) {
            half2_t s2 = {}, z2 = {};
            if constexpr (BITS != 16) {
                const uint32_t pr = (uint32_t)__shfl((int)vpar, r * TPI + tg);
                const half2_t  pp = bit_cast<half2_t>(pr);
                s2                = half2_t{pp[0], pp[0]};
                z2                = half2_t{pp[1], pp[1]};
            }
            half2_t vd[E / 2];
            dequant_chunk<BITS>(vraw[r], s2, z2, vd);
            const bool valid = r * TPI + tg < ntok;
            if constexpr (BITS == 16) {
                if (!valid) {  // raw fp16 garbage past the context must not reach the accumulators
#pragma unroll
                    for (int i = 0; i < E / 2; ++i) {
                        vd[i] = half2_t{};
                    }
                }
            }
#pragma unroll
            for (int h = 0; h < HPW; ++h) {
                const float  pf = fast_exp2(S[r][h] * sc - m[h] * sc);  // exp2(-inf) = 0 for masked tokens
                const half_t ph = (half_t)pf;
                // every one of the CPT lanes of a token adds pf: only lane c==0 may count it
                lsum[h] += (c == 0) ? pf : 0.f;
#pragma unroll
                for (int i = 0; i < E / 2; ++i) {
                    O[h][2 * i]     = __builtin_fmaf((float)ph, (float)vd[i][0], O[h][2 * i]);
                    O[h][2 * i + 1] = __builtin_fmaf((float)ph, (float)vd[i][1], O[h][2 * i + 1]);
                }
            }
        }
    }

    // ---- reduce over the token lanes of the wave, then over the 4 waves through LDS -------------
    __shared__ float sm_o[4][HPW][D];
    __shared__ float sm_ml[4][HPW][2];
#pragma unroll
    for (int h = 0; h < HPW; ++h) {
        lsum[h] = group_sum<64>(lsum[h]);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            O[h][e] = upper_sum<CPT>(O[h][e]);
        }
        if (tg == 0) {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                sm_o[wave][h][c * E + e] = O[h][e];
            }
        }
        if (lane == 0) {
            sm_ml[wave][h][0] = m[h];
            sm_ml[wave][h][1] = lsum[h];
        }
    }
    __syncthreads();

    for (int idx = threadIdx.x; idx < HPW * D; idx += 256) {
        const int h = idx / D;
        const int d = idx - h * D;
        float     ms = -INFINITY;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            ms = fmaxf(ms, sm_ml[w][h][0]);
        }
        float o = 0.f, l = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float mw = sm_ml[w][h][0];
            const float wt = (mw == -INFINITY) ? 0.f : fast_exp2((mw - ms) * sc);
            o += wt * sm_o[w][h][d];
            l += wt * sm_ml[w][h][1];
        }
        const int hq = head0 + h;
        if (p.splits == 1) {
            p.out[(size_t)b * p.q_heads * D + (size_t)hq * D + d] = (half_t)(o / l);
        }
        else {
            const size_t slot                 = ((size_t)b * p.q_heads + hq) * p.splits + split;
            p.partial_o[slot * D + d]         = o;
            if (d == 0) {
                p.partial_ml[slot * 2]     = ms;
                p.partial_ml[slot * 2 + 1] = l;
            }
        }
    }
}

// split-K merge: one 128-thread workgroup per (sequence, query head)
__global__ __launch_bounds__(128) void decode_reduce_kernel(DecodeAttnParams p)
{
    constexpr int D    = 128;
    const int     hq   = blockIdx.x;
    const int     b    = blockIdx.y;
    const int     d    = threadIdx.x;
    const size_t  slot = ((size_t)b * p.q_heads + hq) * p.splits;
    float         ms   = -INFINITY;
    for (int s = 0; s < p.splits; ++s) {
        ms = fmaxf(ms, p.partial_ml[(slot + s) * 2]);
    }
    float o = 0.f, l = 0.f;
    for (int s = 0; s < p.splits; ++s) {
        const float mw = p.partial_ml[(slot + s) * 2];
        const float wt = (mw == -INFINITY) ? 0.f : fast_exp2((mw - ms) * p.scale_log2);
        o += wt * p.partial_o[(slot + s) * D + d];
        l += wt * p.partial_ml[(slot + s) * 2 + 1];
    }
    p.out[(size_t)b * p.q_heads * D + (size_t)hq * D + d] = (half_t)(o / l);
}

size_t decode_attention_workspace_bytes(int batch, int q_heads, int head_dim, int splits)
{
    return (size_t)batch * q_heads * splits * (head_dim + 2) * sizeof(float);
}

template<int BITS>
static int launch_bits(const DecodeAttnParams& p, hipStream_t st)
{
    const int group = p.q_heads / p.cache.layout.kv_heads;
    int       hpw   = 1;
    for (int cand = 4; cand >= 1; --cand) {
        if (group % cand == 0) {
            hpw = cand;
            break;
        }
    }
    const int chunks = group / hpw;
    dim3      grid(p.cache.layout.kv_heads * chunks, p.batch, p.splits);
    switch (hpw) {
        case 4:
            decode_attention_kernel<BITS, 4><<<grid, 256, 0, st>>>(p, chunks);
            break;
        case 3:
            decode_attention_kernel<BITS, 3><<<grid, 256, 0, st>>>(p, chunks);
            break;
        case 2:
            decode_attention_kernel<BITS, 2><<<grid, 256, 0, st>>>(p, chunks);
            break;
        default:
            decode_attention_kernel<BITS, 1><<<grid, 256, 0, st>>>(p, chunks);
    }
    TM_HIP_CHECK(hipGetLastError());
    if (p.splits > 1) {
        decode_reduce_kernel<<<dim3(p.q_heads, p.batch), 128, 0, st>>>(p);
        TM_HIP_CHECK(hipGetLastError());
    }
    return 0;
}

int launch_decode_attention(const DecodeAttnParams& p, hipStream_t st)
{
    const KvLayout& L = p.cache.layout;
    TM_REQUIRE(L.head_dim == 128 && L.block_len == 64, "decode attention: head_dim 128, block_len 64");
    TM_REQUIRE(p.q_heads % L.kv_heads == 0, "q_heads % kv_heads");
    TM_REQUIRE(p.splits >= 1 && p.splits <= 128, "1 <= splits <= 128 (kMaxKVSplits)");
    TM_REQUIRE(p.splits == 1 || (p.partial_o && p.partial_ml), "split-K needs a workspace");
    if (p.batch == 0) {
        return 0;
    }
    const bool fused = p.qkv_slabs || p.qkv_f16;
    TM_REQUIRE(!fused || L.bits == 8 || L.bits == 4, "fused decode prologue: int8 / int4 KV only");
    switch (L.bits) {
        case 16:
            return launch_bits<16>(p, st);
        case 8: {
            // int8 KV (the headline configuration) runs on the matrix cores; TM_ATTN_VALU=1 keeps the VALU kernel
            // reachable for A/B measurements
            static const bool valu = getenv("TM_ATTN_VALU") && atoi(getenv("TM_ATTN_VALU")) != 0;
            if (valu && !fused) {
                return launch_bits<8>(p, st);
            }
            DecodeAttnParams pt = p;
            int rc = launch_decode_attention_i8_mfma(pt, st);
            if (rc) {
                return rc;
            }
            if (p.splits > 1) {
                decode_reduce_kernel<<<dim3(p.q_heads, p.batch), 128, 0, st>>>(p);
                TM_HIP_CHECK(hipGetLastError());
            }
            return 0;
        }
        case 4: {
            // int4 KV: the same MFMA kernel, nibbles expanded to bytes on the way into LDS (TM_ATTN_VALU=1: VALU kernel)
            static const bool valu4 = getenv("TM_ATTN_VALU") && atoi(getenv("TM_ATTN_VALU")) != 0;
            if (valu4 && !fused) {
                return launch_bits<4>(p, st);
            }
            DecodeAttnParams pt = p;
            int rc = launch_decode_attention_i8_mfma(pt, st);
            if (rc) {
                return rc;
            }
            if (p.splits > 1) {
                decode_reduce_kernel<<<dim3(p.q_heads, p.batch), 128, 0, st>>>(p);
                TM_HIP_CHECK(hipGetLastError());
            }
            return 0;
        }
    }
    TM_REQUIRE(false, "kv bits in {16,8,4}");
}

}  // namespace tmk
