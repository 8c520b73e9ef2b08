// One token row of (split-K reduce +) residual + RMSNorm: the body of rmsnorm_kernel (norm.hip).
//
// Arithmetic (src/turbomind/kernels/norm/rms_norm.cu:286-362, rms_norm_utils.cuh:6-15): r = h(r + hcur) [then h(r + bias)];
//   inv = rsqrtf(sum f32(r)^2 / H + eps);  y = h( h(f32(r) * inv) * w ).
#pragma once
#include "tm_common.h"

namespace tmk {

constexpr int kNormMaxThreads = 512;
constexpr int kNormMaxVec     = 2;  // 512 thr * 2 vec * 8 halves = 8192 columns max

// threads: one 16-byte vector per thread up to 512 threads, then two
__host__ __device__ inline void norm_geometry(int H, int* threads, int* nv)
{
    const int nvec = H / 8;
    int       t    = (nvec + 63) / 64 * 64;
    t              = t > kNormMaxThreads ? kNormMaxThreads : t;
    *threads       = t;
    *nv            = (nvec + t - 1) / t;
}

// sum over the first `nthreads` threads of the workgroup (a multiple of 64); EVERY thread of the workgroup calls it
__device__ __forceinline__ float norm_block_sum(float v, float* smem, int tid, int nthreads)
{
    v = group_sum<64>(v);
    const int wave  = tid >> 6;
    const int waves = nthreads >> 6;
    if ((tid & 63) == 0 && wave < waves) {
        smem[wave] = v;
    }
    __syncthreads();
    float t = 0.f;
    for (int w = 0; w < waves; ++w) {  // fixed order: deterministic
        t += smem[w];
    }
    return t;
}

// MODE 0: y = rmsnorm(x)
// MODE 1: r += h ; y = rmsnorm(r)         (h fp16)
// MODE 2: r += h(sum_s partial[s]) ; ...  (h given as S fp32 split-K slabs [S][M][H])
// Row `row`, executed by threads tid < nthreads of the workgroup (all threads call; `red` = 8 floats of LDS).  The row is a
// pure latency chain, so EVERY load a thread needs -- residual, hidden or the first four slabs, and the norm weight -- is
// issued before anything is consumed.  Threads past the row end load clamped (valid) addresses and skip the stores.
// Two halves: norm_row_load (what does not depend on the producing GEMM: residual, norm weight), then norm_row_finish
// (slabs -> sum -> residual add -> norm -> stores).
template<int NV>
struct NormRowRegs {
    half8_t wv[NV], r[NV], hc[NV], bv[NV];
    size_t  off[NV];
    bool    ok[NV];
};

template<int MODE, bool HAS_BIAS, int NV>
__device__ __forceinline__ void norm_row_load(NormRowRegs<NV>& g, const half_t* __restrict__ resid, const half_t* __restrict__ hidden,
                                              const half_t* __restrict__ bias, const half_t* __restrict__ weight, int H, int row, int tid,
                                              int nthreads)
{
    const int nvec = H / 8;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int vi = tid + i * nthreads;
        g.ok[i]      = tid < nthreads && vi < nvec;
        const int vc = g.ok[i] ? vi : nvec - 1;
        g.off[i]     = (size_t)row * H + (size_t)vc * 8;
        g.wv[i]      = *(const half8_t*)(weight + (size_t)vc * 8);
        g.r[i]       = *(const half8_t*)((MODE == 0 ? hidden : resid) + g.off[i]);
        if constexpr (MODE == 1) {
            g.hc[i] = *(const half8_t*)(hidden + g.off[i]);
        }
        if constexpr (HAS_BIAS) {
            g.bv[i] = *(const half8_t*)(bias + (size_t)vc * 8);
        }
    }
}

template<int MODE, bool HAS_BIAS, int NV>
__device__ __forceinline__ void norm_row_finish(NormRowRegs<NV>& g, half_t* __restrict__ y, half_t* __restrict__ resid,
                                                const float* __restrict__ partial, int splits, float eps, int M, int H, int tid,
                                                int nthreads, float* red)
{
    const size_t slab = (size_t)M * H;
    floatx4      a[NV][4][2];
    if constexpr (MODE == 2) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const size_t p0 = g.off[i] + (size_t)min(u, splits - 1) * slab;
                a[i][u][0]      = *(const floatx4*)(partial + p0);
                a[i][u][1]      = *(const floatx4*)(partial + p0 + 4);
            }
        }
    }

    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if constexpr (MODE == 1) {
            g.r[i] = g.r[i] + g.hc[i];  // fp16 add, one rounding per element
        }
        if constexpr (MODE == 2) {
            float acc[8] = {};
#pragma unroll
            for (int u = 0; u < 4; ++u) {  // slabs are summed in order (deterministic)
                if (u < splits) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc[e] += a[i][u][0][e];
                        acc[4 + e] += a[i][u][1][e];
                    }
                }
            }
            for (int s = 4; s < splits; ++s) {
                const size_t  p0 = g.off[i] + (size_t)s * slab;
                const floatx4 a0 = *(const floatx4*)(partial + p0);
                const floatx4 a1 = *(const floatx4*)(partial + p0 + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[e] += a0[e];
                    acc[4 + e] += a1[e];
                }
            }
            half8_t hcur;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                hcur[e] = (half_t)acc[e];  // the GEMM's fp16 output rounding
            }
            g.r[i] = g.r[i] + hcur;
        }
        if constexpr (HAS_BIAS) {
            g.r[i] = g.r[i] + g.bv[i];
        }
        if (g.ok[i]) {
            if constexpr (MODE != 0) {
                *(half8_t*)(resid + g.off[i]) = g.r[i];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float f = (float)g.r[i][e];
                ss            = __builtin_fmaf(f, f, ss);
            }
        }
    }

    ss              = norm_block_sum(ss, red, tid, nthreads);
    const float inv = 1.0f / __builtin_sqrtf(ss / (float)H + eps);

#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if (g.ok[i]) {
            half8_t o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const half_t n = (half_t)((float)g.r[i][e] * inv);  // cast to T first ...
                o[e]           = n * g.wv[i][e];                    // ... then multiply by w in T
            }
            *(half8_t*)(y + g.off[i]) = o;
        }
    }
}

template<int MODE, bool HAS_BIAS, int NV>
__device__ __forceinline__ void norm_row(half_t* __restrict__ y, half_t* __restrict__ resid, const half_t* __restrict__ hidden,
                                         const float* __restrict__ partial, int splits, const half_t* __restrict__ bias,
                                         const half_t* __restrict__ weight, float eps, int M, int H, int row, int tid, int nthreads,
                                         float* red)
{
    NormRowRegs<NV> g;
    norm_row_load<MODE, HAS_BIAS, NV>(g, resid, hidden, bias, weight, H, row, tid, nthreads);
    norm_row_finish<MODE, HAS_BIAS, NV>(g, y, resid, partial, splits, eps, M, H, tid, nthreads, red);
}

}  // namespace tmk
