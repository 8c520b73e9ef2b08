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
// One wave = 16 query rows of G query heads that share a kv head (GQA): every K / V fragment fetched from the
// L2-resident scratch feeds G MFMAs, which divides the load-instruction and L2 traffic per flop by G (the kernel is
// bound by exactly that: one head per wave measured 50 TFLOP/s).  Softmax state is per lane and head.
// Key order inside a 32-key step is permuted so that a lane's 8 probabilities belong to 8 CONSECUTIVE keys:
// score tile A row 4g+r <-> key 8g+r, tile B row 4g+r <-> key 8g+4+r, hence one 16-byte V^T load per fragment.
#include "tm_common.h"
#include "tm_kernels.h"

namespace tmk {

template<int G>
__global__ __launch_bounds__(256) void prefill_attention_kernel(PrefillAttnParams p)
{
    constexpr int D = 128;
    const int b     = blockIdx.z;
    const int hq0   = blockIdx.y * G;
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
    const int kvh   = hq0 / group;

    // Q^T fragments (B operand): lane (j = query row, g) holds Q[q0+j][32*dd + 8g .. +8) of head hq0 + h
    const int     qrow = min(q0 + i16, qlen - 1);
    const half_t* qptr = p.q + (size_t)(q_beg + qrow) * p.q_stride + (size_t)hq0 * D;
    half8_t       qf[G][4];
#pragma unroll
    for (int h = 0; h < G; ++h) {
#pragma unroll
        for (int dd = 0; dd < 4; ++dd) {
            qf[h][dd] = *(const half8_t*)(qptr + h * D + dd * 32 + g * 8);
        }
    }

    const half_t* kbase = p.k + ((size_t)kvh * p.k_stride + p.cu_k_off[b]) * D;
    const half_t* vbase = p.vt + (size_t)kvh * D * p.k_stride + p.cu_k_off[b];

    floatx4 O[G][8];
    float   m[G], l[G];
#pragma unroll
    for (int h = 0; h < G; ++h) {
        m[h] = -INFINITY;
        l[h] = 0.f;
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) {
            O[h][dt] = floatx4{0.f, 0.f, 0.f, 0.f};
        }
    }
    const float sc   = p.scale_log2;
    const int   qpos = hist + q0 + i16;            // absolute position of this lane's query row
    const int   kend = min(klen, hist + q0 + 16);  // keys visible to the last row of this wave
    const int   krow = 8 * (i16 >> 2) + (i16 & 3); // key (within the 32-key step) of score-tile-A row i16; tile B: +4

    // Software pipeline (one wave per SIMD, nothing else hides the L2 latency): the K fragments of step ks+32 are
    // fetched right after the score MFMAs of step ks have consumed the current ones, the V^T fragments of step ks are
    // fetched before its score MFMAs -- both fly behind ~64 MFMAs + the softmax.
    half8_t ka[4], kb[4];
#pragma unroll
    for (int dd = 0; dd < 4; ++dd) {
        ka[dd] = *(const half8_t*)(kbase + (size_t)krow * D + dd * 32 + g * 8);
        kb[dd] = *(const half8_t*)(kbase + (size_t)(krow + 4) * D + dd * 32 + g * 8);
    }
    for (int ks = 0; ks < kend; ks += 32) {
        half8_t vf[8];
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) {
            vf[dt] = *(const half8_t*)(vbase + (size_t)(dt * 16 + i16) * p.k_stride + ks + 8 * g);
        }
        // the whole 32-key step lies at or below the diagonal for every row of the wave and inside the context:
        // no masking needed (wave-uniform; true for all but the last one or two steps)
        const bool full = ks + 32 <= klen && ks + 31 <= hist + q0;
        half8_t    pf[G];
        floatx4    sa_[G], sb_[G];
#pragma unroll
        for (int h = 0; h < G; ++h) {
            floatx4 sa = {0.f, 0.f, 0.f, 0.f}, sb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int dd = 0; dd < 4; ++dd) {
                sa = __builtin_amdgcn_mfma_f32_16x16x32_f16(ka[dd], qf[h][dd], sa, 0, 0, 0);
                sb = __builtin_amdgcn_mfma_f32_16x16x32_f16(kb[dd], qf[h][dd], sb, 0, 0, 0);
            }
            sa_[h] = sa;
            sb_[h] = sb;
        }
        {
            const int kn = ks + 32 < kend ? ks + 32 : ks;  // wave-uniform; the last step re-reads its own rows (in bounds)
#pragma unroll
            for (int dd = 0; dd < 4; ++dd) {
                ka[dd] = *(const half8_t*)(kbase + (size_t)(kn + krow) * D + dd * 32 + g * 8);
                kb[dd] = *(const half8_t*)(kbase + (size_t)(kn + krow + 4) * D + dd * 32 + g * 8);
            }
        }
#pragma unroll
        for (int h = 0; h < G; ++h) {
            const floatx4 sa = sa_[h], sb = sb_[h];
            // lane holds S[key = ks + 8g + r][q] in sa[r] and S[key = ks + 8g + 4 + r][q] in sb[r]
            float s[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s[r]     = sa[r];
                s[4 + r] = sb[r];
            }
            if (!full) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int t = ks + 8 * g + e;
                    s[e]        = (t < klen && t <= qpos) ? s[e] : -INFINITY;
                }
            }
            float tmax = s[0];
#pragma unroll
            for (int e = 1; e < 8; ++e) {
                tmax = fmaxf(tmax, s[e]);
            }
            tmax = fmaxf(tmax, __shfl_xor(tmax, 16));
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
            const float mnew  = fmaxf(m[h], tmax);
            const float alpha = (m[h] == -INFINITY) ? 0.f : fast_exp2((m[h] - mnew) * sc);
            float       psum  = 0.f;
            const float ms    = mnew * sc;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                // mnew == -inf only for rows past qlen on their very first step: keep them NaN-free
                const float pv = (mnew == -INFINITY) ? 0.f : fast_exp2(__builtin_fmaf(s[e], sc, -ms));
                psum += pv;
                pf[h][e] = (half_t)pv;
            }
            l[h] = l[h] * alpha + psum;
            if (__builtin_amdgcn_readfirstlane((int)__any(mnew != m[h]))) {  // wave-uniform: did any row's max move?
#pragma unroll
                for (int dt = 0; dt < 8; ++dt) {
                    O[h][dt] *= alpha;
                }
            }
            m[h] = mnew;
        }
        // ---- O^T += V^T P^T: one 16-byte V^T fragment (8 consecutive keys of one head-dim row) feeds G MFMAs ----
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) {
#pragma unroll
            for (int h = 0; h < G; ++h) {
                O[h][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf[dt], pf[h], O[h][dt], 0, 0, 0);
            }
        }
    }

    if (q0 + i16 < qlen) {
#pragma unroll
        for (int h = 0; h < G; ++h) {
            float lh = l[h];
            lh += __shfl_xor(lh, 16);
            lh += __shfl_xor(lh, 32);
            half_t*     optr = p.out + (size_t)(q_beg + q0 + i16) * p.q_heads * D + (size_t)(hq0 + h) * D;
            const float inv  = 1.0f / lh;
#pragma unroll
            for (int dt = 0; dt < 8; ++dt) {
                half4_t o = {(half_t)(O[h][dt][0] * inv), (half_t)(O[h][dt][1] * inv), (half_t)(O[h][dt][2] * inv),
                             (half_t)(O[h][dt][3] * inv)};
                *(half4_t*)(optr + dt * 16 + g * 4) = o;
            }
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
    const int group = p.q_heads / p.kv_heads;
    const int G     = group % 4 == 0 ? 4 : (group % 2 == 0 ? 2 : 1);  // query heads per wave (share one kv head)
    dim3      grid((p.max_q_len + 63) / 64, p.q_heads / G, p.batch);
    if (G == 4) {
        prefill_attention_kernel<4><<<grid, 256, 0, st>>>(p);
    }
    else if (G == 2) {
        prefill_attention_kernel<2><<<grid, 256, 0, st>>>(p);
    }
    else {
        prefill_attention_kernel<1><<<grid, 256, 0, st>>>(p);
    }
    TM_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace tmk
