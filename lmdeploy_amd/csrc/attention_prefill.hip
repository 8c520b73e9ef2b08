// Causal flash-attention for prefill on MFMA (gfx950), over the linear fp16 scratch produced by
// flatten_kv (K [kv_heads][k_stride][D], V transposed [kv_heads][D][k_stride]).
//
// Replaces: dispatchAttention -> attention_kernel<...AttentionCtaMap...> (src/turbomind/kernels/attention/
//           attention.cu:12-28, attention_template.h:13-85, kernel/attention_sm80_128.cu:21-37).
// Like the reference, prefill attention sees the round-tripped (quantised -> dequantised) KV.
//
// Layout trick (wave64, v_mfma_f32_16x16x32_f16): the scores are computed TRANSPOSED, S^T = K Q^T, so the
// accumulator of a lane (col = query row, rows = 4 consecutive keys) is, after exp2 and a cast to fp16,
// directly the B operand of the second contraction O^T = V^T P^T -- no LDS round trip, no cross-lane shuffle for P.
// The k-slot <-> key mapping of that second MFMA is (g, e) -> key 4g+e (e<4) / 16+4g+(e-4) (e>=4), which is
// why V is kept transposed: a lane fetches its 8 keys of one head-dim row with one 16-byte read.
// One wave = 16 query rows of G query heads that share a kv head (GQA): every K / V fragment feeds G MFMAs.
// Softmax state is per lane and head.  Key order inside a 32-key step is permuted so that a lane's 8 probabilities
// belong to 8 CONSECUTIVE keys: score tile A row 4g+r <-> key 8g+r, tile B row 4g+r <-> key 8g+4+r.
//
// K / V^T tiles go through LDS (round 3).  The four waves of a workgroup (64 query rows) need the SAME keys; rounds 1-2 read the
// fragments straight from L2 per wave (16 KB per wave and 32-key step, 4 x redundant inside a workgroup).  Now a stage = 64 keys
// (K 16 KB + V^T 16 KB) is brought in ONCE per workgroup by LDS-DMA (buffer_load ... lds: no staging registers -- accumulators
// and Q fragments already take ~200), double buffered, one barrier per stage; the XOR swizzle that makes the 16-byte fragment
// reads bank-conflict free sits on the DMA's source addresses (the DMA writes lane L to slot L of its 1 KB piece).
// Measured (8 x 1024 prompt tokens of Llama-3-8B per launch, rocprofv3 kernel trace): 282 -> 252 us; 235 us = 0.29 PFLOP/s with
// the conflict-free swizzle (ksw) and the leaner softmax below (profiles/r03_pmc_prefill_attention.txt).  What
// bounds it now is the softmax's VALU issue, not memory: per 32-key step and wave 64 MFMAs (1024 matrix-pipe cycles) stand
// against ~500 VALU slots (32 v_exp_f32 at quarter rate alone are 512 cycles), and the two waves of a SIMD share one VALU.
// Prefill attention is 8 % of a prefill chunk (the GEMMs are 82 %), so this is where the kernel was left.
#include "tm_common.h"
#include "tm_kernels.h"

namespace tmk {

// Swizzle term of K-tile row r (16-byte slot = chunk ^ ksw(r)).  A ds_read_b128 is served in four groups of 16 lanes --
// {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... (MI355X_MICROARCH.md, LDS) -- i.e. two neighbouring g with eight score-tile
// rows each; row i16 of a score tile is key 8 (i16 >> 2) + (i16 & 3), so the eight rows of either half differ in
// (bit 3, bits 1..0) of the key row: those three bits, shifted to slot bits 3..1, leave slot bit 0 to g -- 16 distinct slots
// per group.  (XOR with the plain row number, the first version, was a 2-way conflict on every K fragment read.)
__device__ __forceinline__ int ksw(int r)
{
    return (r & 8) | ((r & 3) << 1);
}

// max over the lane pairs l ^ 16 and l ^ 32 with gfx950's row / half swaps: two VALU ops each instead of a ds_bpermute round
// trip through the LDS crossbar in the middle of the softmax's dependency chain.  permlane16_swap(a, a) returns
// ([row0, row0, row2, row2], [row1, row1, row3, row3]), permlane32_swap(a, a) ([lo, lo], [hi, hi]).
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float max_xor16(float v)
{
    const unsigned b = __builtin_bit_cast(unsigned, v);
    const u32x2_t  r = __builtin_amdgcn_permlane16_swap(b, b, false, false);
    return fmaxf(__builtin_bit_cast(float, (unsigned)r[0]), __builtin_bit_cast(float, (unsigned)r[1]));
}
__device__ __forceinline__ float max_xor32(float v)
{
    const unsigned b = __builtin_bit_cast(unsigned, v);
    const u32x2_t  r = __builtin_amdgcn_permlane32_swap(b, b, false, false);
    return fmaxf(__builtin_bit_cast(float, (unsigned)r[0]), __builtin_bit_cast(float, (unsigned)r[1]));
}
__device__ __forceinline__ float sum_xor16_xor32(float v)
{
    unsigned       b = __builtin_bit_cast(unsigned, v);
    const u32x2_t  r = __builtin_amdgcn_permlane16_swap(b, b, false, false);
    b                = __builtin_bit_cast(unsigned, __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]));
    const u32x2_t  q = __builtin_amdgcn_permlane32_swap(b, b, false, false);
    return __builtin_bit_cast(float, (unsigned)q[0]) + __builtin_bit_cast(float, (unsigned)q[1]);
}

// Running maxima start at a large negative FINITE value: every difference / product in the online softmax then stays finite
// or is a clean -inf (masked score -> exp2(-inf) = 0), and the "-inf so far" selects around each exp disappear.
constexpr float kMinScore = -1.0e30f;

template<int G>
__global__ __launch_bounds__(256, 2) void prefill_attention_kernel(PrefillAttnParams p)
{
    constexpr int D     = 128;
    constexpr int KS    = 64;           // keys per stage
    constexpr int KTILE = KS * D * 2;   // bytes of a K tile (and of a V^T tile)
    constexpr int STG   = 2 * KTILE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
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
    // causal: query block bx walks bx + 1 stages (+ history).  The dispatcher hands out workgroups in blockIdx order, so the
    // LONGEST blocks of every (head group, sequence) go first and the short ones fill the tail (longest-processing-time order)
    const int bx = (int)gridDim.x - 1 - (int)blockIdx.x;
    if (bx * 64 >= qlen) {
        return;  // workgroup-uniform
    }
    const int  q0     = bx * 64 + wave * 16;
    const bool active = q0 < qlen;  // wave-uniform; an inactive wave still stages tiles and meets the barriers
    const int  group  = p.q_heads / p.kv_heads;
    const int  kvh    = hq0 / group;

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

    // the sequence's region of the scratch is 64-key aligned, its V^T tail is zero-filled (flatten_kv_kernel): a 64-key stage
    // never leaves the region; K rows past klen may hold stale data, their scores are masked below
    const half_t* kbase    = p.k + ((size_t)kvh * p.k_stride + p.cu_k_off[b]) * D;
    const half_t* vbase    = p.vt + (size_t)kvh * D * p.k_stride + p.cu_k_off[b];
    const int     klen_pad = (klen + KS - 1) / KS * KS;
    const auto    rs_k     = __builtin_amdgcn_make_buffer_rsrc((void*)kbase, 0, klen_pad * D * 2, 0x00020000);
    const auto    rs_v     = __builtin_amdgcn_make_buffer_rsrc((void*)vbase, 0, (int)(((size_t)(D - 1) * p.k_stride + klen_pad) * 2), 0x00020000);
    // DMA pieces of 1 KB: K piece pc = key rows 4pc .. 4pc+3 (256 B each, 16 chunks), V^T piece pc = head-dim rows 8pc .. 8pc+7
    // (128 B each, 8 chunks); wave w moves pieces 4r + w, r = 0..3.  Lane L lands in slot L: it fetches the chunk that belongs there.
    // piece 4r + w: K rows 16r + 4w + (L >> 4), V^T rows 32r + 8w + (L >> 3) -- the swizzle terms (row & 15, d & 7) do not depend on
    // r (ksw looks at bits 3 and 1..0 only), so one per-lane offset each serves all four pieces and r moves into the scalar offset
    const int krow0 = 4 * wave + (lane >> 4);
    const int kdo   = krow0 * 256 + (((lane & 15) ^ ksw(krow0)) << 4);
    const int vrow0 = 8 * wave + (lane >> 3);
    const int vdo   = vrow0 * p.k_stride * 2 + (((lane & 7) ^ (vrow0 & 7)) << 4);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)smem);
    auto stage_in = [&](int ks, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            unsigned       keep;
            const unsigned dk = lds0 + buf * STG + (4 * r + wave) * 1024;
            const unsigned dv = dk + KTILE;
            const int      sk = ks * D * 2 + r * 16 * 256, sv = ks * 2 + r * 32 * p.k_stride * 2;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                         "buffer_load_dwordx4 %1, %2, %4 offen lds\n\t"
                         "s_mov_b32 m0, %5\n\ts_nop 0\n\t"
                         "buffer_load_dwordx4 %6, %7, %8 offen lds\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep)
                         : "v"(kdo), "s"(rs_k), "s"(dk), "s"(sk), "s"(dv), "v"(vdo), "s"(rs_v), "s"(sv)
                         : "memory");
        }
    };

    floatx4 O[G][8];
    float   m[G], l[G];
#pragma unroll
    for (int h = 0; h < G; ++h) {
        m[h] = kMinScore;
        l[h] = 0.f;
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) {
            O[h][dt] = floatx4{0.f, 0.f, 0.f, 0.f};
        }
    }
    const float sc      = p.scale_log2;
    const int   qpos    = hist + q0 + i16;                              // absolute position of this lane's query row
    const int   kend_w  = active ? min(klen, hist + q0 + 16) : 0;       // keys visible to the last row of this wave
    const int   kend_wg = min(klen, hist + bx * 64 + 64);               // ... of this workgroup
    const int   krow    = 8 * (i16 >> 2) + (i16 & 3);                   // key (within the 32-key step) of score-tile-A row i16; tile B: +4

    stage_in(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int buf = 0;
    for (int ks0 = 0; ks0 < kend_wg; ks0 += KS, buf ^= 1) {
        if (ks0 + KS < kend_wg) {
            stage_in(ks0 + KS, buf ^ 1);  // every wave is past the barrier behind its reads of that buffer
        }
        const char* kt = smem + buf * STG;
        const char* vt = kt + KTILE;
#pragma nounroll
        for (int sub = 0; sub < 2; ++sub) {
            const int ks = ks0 + 32 * sub;
            if (ks >= kend_w) {  // wave-uniform: causal end of this wave (or an inactive wave)
                continue;
            }
            half8_t ka[4], kb[4];
            {
                const int ra = 32 * sub + krow, rb = ra + 4;
#pragma unroll
                for (int dd = 0; dd < 4; ++dd) {
                    ka[dd] = *(const half8_t*)(kt + ra * 256 + (((4 * dd + g) ^ ksw(ra)) << 4));
                    kb[dd] = *(const half8_t*)(kt + rb * 256 + (((4 * dd + g) ^ ksw(rb)) << 4));
                }
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
            // one head at a time, its rescale decided right there: the branch per head keeps the four chains from being
            // interleaved -- measured: ONE basic block for all heads (scheduler free to overlap the chains) needs 45 spilled
            // registers and runs 284 us against 239..261 us this way
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
                tmax              = max_xor32(max_xor16(tmax));
                const float mnew  = fmaxf(m[h], tmax);  // finite (kMinScore at least)
                const float alpha = fast_exp2((m[h] - mnew) * sc);
                float       psum  = 0.f;
                const float ms    = mnew * sc;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float pv = fast_exp2(__builtin_fmaf(s[e], sc, -ms));  // masked score: exp2(-inf) = 0
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
            // (fragments fetched here, not above the softmax: 32 more live registers there would spill at G = 4)
            const char* vrow = vt + i16 * 128 + (((4 * sub + g) ^ (i16 & 7)) << 4);  // row dt*16 + i16: same swizzle term for every dt
#pragma unroll
            for (int dt = 0; dt < 8; ++dt) {
                const half8_t vf = *(const half8_t*)(vrow + dt * 16 * 128);
#pragma unroll
                for (int h = 0; h < G; ++h) {
                    O[h][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf[h], O[h][dt], 0, 0, 0);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of the next stage have landed ...
        __syncthreads();                                  // ... everybody's have, and everybody is done reading this one
    }

    if (active && q0 + i16 < qlen) {
#pragma unroll
        for (int h = 0; h < G; ++h) {
            const float lh = sum_xor16_xor32(l[h]);
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
    const int G     = group % 4 == 0 ? 4 : (group % 2 == 0 ? 2 : 1);  // query heads per wave (share one kv head); G = 4 spills ~25
                                                                       // registers at two workgroups per CU -- measured equal to G = 2
    dim3      grid((p.max_q_len + 63) / 64, p.q_heads / G, p.batch);
    constexpr int lds = 2 * 2 * 64 * 128 * 2;  // two stages of (K tile + V^T tile)
    const void* const k = G == 4 ? (const void*)prefill_attention_kernel<4> :
                          G == 2 ? (const void*)prefill_attention_kernel<2> : (const void*)prefill_attention_kernel<1>;
    if (const int rc = ensure_dynamic_lds(k, lds)) {
        return rc;
    }
    if (G == 4) {
        prefill_attention_kernel<4><<<grid, 256, lds, st>>>(p);
    }
    else if (G == 2) {
        prefill_attention_kernel<2><<<grid, 256, lds, st>>>(p);
    }
    else {
        prefill_attention_kernel<1><<<grid, 256, lds, st>>>(p);
    }
    TM_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace tmk
