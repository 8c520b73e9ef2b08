// Causal flash-attention for prefill on MFMA (gfx950), over the linear fp16 scratch produced by
// flatten_kv (K [kv_heads][k_stride][D], V transposed [kv_heads][D][k_stride]).
//
// Replaces: dispatchAttention -> attention_kernel<...AttentionCtaMap...> (src/turbomind/kernels/attention/
//           attention.cu:12-28, attention_template.h:13-85, kernel/attention_sm80_128.cu:21-37).
// Like the reference, prefill attention sees the round-tripped (quantised -> dequantised) KV.
//
// Layout trick (wave64, v_mfma_f32_16x16x32_f16): the scores are computed TRANSPOSED, S^T = K Q^T, so the
// accumulator of a lane (col = query row, rows = 4 consecutive keys) is, after exp2 and a cast to fp16,
// directly the B operand of the second contraction O^T = V^T P^T -- no LDS, no cross-lane shuffle for P.
// The k-slot <-> key mapping of that second MFMA is (g, e) -> key 4g+e (e<4) / 16+4g+(e-4) (e>=4), which is
// why V is kept transposed: a lane fetches its 8 keys of one head-dim row with two 8-byte loads.
// One wave = 16 query rows of one head; softmax state is per lane (m, l of its query column).
#include "tm_common.h"
#include "tm_kernels.h"

namespace tmk {

__global__ __launch_bounds__(256) void prefill_attention_kernel(PrefillAttnParams p)
{
    constexpr int D = 128;
    const int b     = blockIdx.z;
    const int hq    = blockIdx.y;
    const int lane  = threadIdx.x & 63;
    const int wave  = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i16   = lane & 15;
    const int g     = lane >> 4;

    const int q_beg = p.cu_q_len[b];
    const int qlen  = p.cu_q_len[b + 1] - q_beg;
    const int klen  = p.k_len[b];
    const int hist  = klen - qlen;
    const int q0    = blockIdx.x * 64 + wave * 16;
    if (q0 >= qlen) {
        return;
    }
    const int group = p.q_heads / p.kv_heads;
    const int kvh   = hq / group;

    // Q^T fragments (B operand): lane (j = query row, g) holds Q[q0+j][32*dd + 8g .. +8)
    const int     qrow = min(q0 + i16, qlen - 1);
    const half_t* qptr = p.q + (size_t)(q_beg + qrow) * p.q_stride + (size_t)hq * D;
    half8_t       qf[4];
#pragma unroll
    for (int dd = 0; dd < 4; ++dd) {
        qf[dd] = *(const half8_t*)(qptr + dd * 32 + g * 8);
    }

    const half_t* kbase = p.k + ((size_t)kvh * p.k_stride + p.cu_k_off[b]) * D;
    const half_t* vbase = p.vt + (size_t)kvh * D * p.k_stride + p.cu_k_off[b];

    floatx4 O[8];
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) {
        O[dt] = floatx4{0.f, 0.f, 0.f, 0.f};
    }
    float       m    = -INFINITY;
    float       l    = 0.f;
    const float sc   = p.scale_log2;
    const int   qpos = hist + q0 + i16;                 // absolute position of this lane's query row
    const int   kend = min(klen, hist + q0 + 16);       // keys visible to the last row of this wave

    for (int ks = 0; ks < kend; ks += 32) {
        // ---- S^T = K Q^T for keys [ks, ks+32) ------------------------------------------------
        floatx4 sa = {0.f, 0.f, 0.f, 0.f}, sb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dd = 0; dd < 4; ++dd) {
            const half8_t ka = *(const half8_t*)(kbase + (size_t)(ks + i16) * D + dd * 32 + g * 8);
            const half8_t kb = *(const half8_t*)(kbase + (size_t)(ks + 16 + i16) * D + dd * 32 + g * 8);
            sa               = __builtin_amdgcn_mfma_f32_16x16x32_f16(ka, qf[dd], sa, 0, 0, 0);
            sb               = __builtin_amdgcn_mfma_f32_16x16x32_f16(kb, qf[dd], sb, 0, 0, 0);
        }
        // lane holds S[key = ks + 4g + r][q] in sa[r] and S[key = ks + 16 + 4g + r][q] in sb[r]
        float s[8];
        float tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ta = ks + 4 * g + r;
            const int tb = ta + 16;
            s[r]         = (ta < klen && ta <= qpos) ? sa[r] : -INFINITY;
            s[4 + r]     = (tb < klen && tb <= qpos) ? sb[r] : -INFINITY;
            tmax         = fmaxf(tmax, fmaxf(s[r], s[4 + r]));
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
        const float mnew  = fmaxf(m, tmax);
        const float alpha = (m == -INFINITY) ? 0.f : fast_exp2((m - mnew) * sc);
        m                 = mnew;
        half8_t pf;
        float   psum = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            // mnew == -inf only for rows past qlen on their very first step: keep them NaN-free
            const float pv = (mnew == -INFINITY) ? 0.f : fast_exp2(s[e] * sc - mnew * sc);
            psum += pv;
            pf[e] = (half_t)pv;
        }
        l = l * alpha + psum;
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) {
            O[dt] *= alpha;
        }
        // ---- O^T += V^T P^T ------------------------------------------------------------------
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) {
            const half_t* vp = vbase + (size_t)(dt * 16 + i16) * p.k_stride + ks + 4 * g;
            const half4_t v0 = *(const half4_t*)vp;
            const half4_t v1 = *(const half4_t*)(vp + 16);
            const half8_t vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            O[dt]            = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf, O[dt], 0, 0, 0);
        }
    }

    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    if (q0 + i16 < qlen) {
        half_t*     optr = p.out + (size_t)(q_beg + q0 + i16) * p.q_heads * D + (size_t)hq * D;
        const float inv  = 1.0f / l;
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) {
            half4_t o = {(half_t)(O[dt][0] * inv), (half_t)(O[dt][1] * inv), (half_t)(O[dt][2] * inv),
                         (half_t)(O[dt][3] * inv)};
            *(half4_t*)(optr + dt * 16 + g * 4) = o;
        }
    }
}

int launch_prefill_attention(const PrefillAttnParams& p, hipStream_t st)
{
    TM_REQUIRE(p.q_heads % p.kv_heads == 0, "q_heads % kv_heads");
    TM_REQUIRE(p.k_stride % 64 == 0, "k_stride must be a multiple of 64");
    if (p.batch == 0 || p.max_q_len == 0) {
        return 0;
    }
    dim3 grid((p.max_q_len + 63) / 64, p.q_heads, p.batch);
    prefill_attention_kernel<<<grid, 256, 0, st>>>(p);
    TM_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace tmk
