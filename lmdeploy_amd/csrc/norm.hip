// RMSNorm / fused (split-K reduce +) residual + RMSNorm for gfx950.
//
// Replaces: src/turbomind/kernels/norm/rms_norm.cu:21-138 (invokeRMSNorm) and
//           :286-362,423-466 (invokeResidualBiasRMSNorm).
// Arithmetic (rms_norm_utils.cuh:6-15): r = h(r + hcur) [then h(r + bias)];
//   inv = rsqrtf(sum f32(r)^2 / H + eps);  y = h( h(f32(r) * inv) * w ).
// One 256-thread workgroup per token row; 16-byte vector accesses; the row lives in registers
// between the two passes.  HBM-bound (tiny): the point is fusing the split-K reduce of the
// preceding row-parallel GEMM so that no extra launch / HBM round trip is spent on it.
#include "tm_common.h"
#include "tm_kernels.h"

namespace tmk {

constexpr int kNormThreads = 256;
constexpr int kNormMaxVec  = 4;  // 256 thr * 4 vec * 8 halves = 8192 columns max

__device__ __forceinline__ float block_sum(float v, float* smem)
{
    v = group_sum<64>(v);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        smem[wave] = v;
    }
    __syncthreads();
    float t = smem[0] + smem[1] + smem[2] + smem[3];
    __syncthreads();
    return t;
}

// MODE 0: y = rmsnorm(x)
// MODE 1: r += h ; y = rmsnorm(r)         (h fp16)
// MODE 2: r += h(sum_s partial[s]) ; ...  (h given as S fp32 split-K slabs [S][M][H])
template<int MODE, bool HAS_BIAS>
__global__ __launch_bounds__(kNormThreads) void rmsnorm_kernel(half_t* __restrict__ y,
                                                               half_t* __restrict__ resid,
                                                               const half_t* __restrict__ hidden,
                                                               const float* __restrict__ partial,
                                                               int           splits,
                                                               const half_t* __restrict__ bias,
                                                               const half_t* __restrict__ weight,
                                                               float eps,
                                                               int   M,
                                                               int   H)
{
    __shared__ float red[4];
    const int        row  = blockIdx.x;
    const int        nvec = H / 8;
    half8_t          v[kNormMaxVec];
    float            ss = 0.f;

#pragma unroll
    for (int i = 0; i < kNormMaxVec; ++i) {
        const int vi = threadIdx.x + i * kNormThreads;
        if (vi < nvec) {
            const size_t off = (size_t)row * H + (size_t)vi * 8;
            half8_t      r   = *(const half8_t*)((MODE == 0 ? hidden : resid) + off);
            if constexpr (MODE == 1) {
                half8_t hcur = *(const half8_t*)(hidden + off);
                r            = r + hcur;  // fp16 add, one rounding per element
            }
            if constexpr (MODE == 2) {
                float        acc[8] = {};
                const float* p0     = partial + (size_t)row * H + (size_t)vi * 8;
                const size_t slab   = (size_t)M * H;
                int          s      = 0;
                // slabs are summed in order (deterministic); loads of 4 slabs are issued together
                for (; s + 4 <= splits; s += 4) {
                    floatx4 a[4][2];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        a[u][0] = *(const floatx4*)(p0 + (s + u) * slab);
                        a[u][1] = *(const floatx4*)(p0 + (s + u) * slab + 4);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            acc[e] += a[u][0][e];
                            acc[4 + e] += a[u][1][e];
                        }
                    }
                }
                for (; s < splits; ++s) {
                    const floatx4 a0 = *(const floatx4*)(p0 + s * slab);
                    const floatx4 a1 = *(const floatx4*)(p0 + s * slab + 4);
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
                r = r + hcur;
            }
            if constexpr (HAS_BIAS) {
                r = r + *(const half8_t*)(bias + (size_t)vi * 8);
            }
            if constexpr (MODE != 0) {
                *(half8_t*)(resid + off) = r;
            }
            v[i] = r;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float f = (float)r[e];
                ss            = __builtin_fmaf(f, f, ss);
            }
        }
    }

    ss              = block_sum(ss, red);
    const float inv = 1.0f / __builtin_sqrtf(ss / (float)H + eps);

#pragma unroll
    for (int i = 0; i < kNormMaxVec; ++i) {
        const int vi = threadIdx.x + i * kNormThreads;
        if (vi < nvec) {
            const half8_t w = *(const half8_t*)(weight + (size_t)vi * 8);
            half8_t       o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const half_t n = (half_t)((float)v[i][e] * inv);  // cast to T first ...
                o[e]           = n * w[e];                        // ... then multiply by w in T
            }
            *(half8_t*)(y + (size_t)row * H + (size_t)vi * 8) = o;
        }
    }
}

int launch_rmsnorm(half_t* y, const half_t* x, const half_t* w, float eps, int M, int H, hipStream_t st)
{
    TM_REQUIRE(H % 8 == 0 && H <= kNormThreads * kNormMaxVec * 8, "rmsnorm: H must be a multiple of 8 and <= 8192");
    if (M == 0) {
        return 0;
    }
    rmsnorm_kernel<0, false><<<M, kNormThreads, 0, st>>>(y, nullptr, x, nullptr, 0, nullptr, w, eps, M, H);
    TM_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_residual_rmsnorm(half_t*       y,
                            half_t*       resid,
                            const half_t* hidden,
                            const float*  partial,
                            int           splits,
                            const half_t* bias,
                            const half_t* w,
                            float         eps,
                            int           M,
                            int           H,
                            hipStream_t   st)
{
    TM_REQUIRE(H % 8 == 0 && H <= kNormThreads * kNormMaxVec * 8, "rmsnorm: H must be a multiple of 8 and <= 8192");
    TM_REQUIRE((hidden != nullptr) != (partial != nullptr), "exactly one of hidden / partial");
    if (M == 0) {
        return 0;
    }
    if (partial) {
        if (bias) {
            rmsnorm_kernel<2, true><<<M, kNormThreads, 0, st>>>(y, resid, nullptr, partial, splits, bias, w, eps, M, H);
        }
        else {
            rmsnorm_kernel<2, false><<<M, kNormThreads, 0, st>>>(y, resid, nullptr, partial, splits, nullptr, w, eps, M, H);
        }
    }
    else {
        if (bias) {
            rmsnorm_kernel<1, true><<<M, kNormThreads, 0, st>>>(y, resid, hidden, nullptr, 0, bias, w, eps, M, H);
        }
        else {
            rmsnorm_kernel<1, false><<<M, kNormThreads, 0, st>>>(y, resid, hidden, nullptr, 0, nullptr, w, eps, M, H);
        }
    }
    TM_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace tmk
