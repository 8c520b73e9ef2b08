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

constexpr int kNormMaxThreads = 512;
constexpr int kNormMaxVec     = 2;  // 512 thr * 2 vec * 8 halves = 8192 columns max

__device__ __forceinline__ float block_sum(float v, float* smem)
{
    v = group_sum<64>(v);
    const int wave  = threadIdx.x >> 6;
    const int waves = blockDim.x >> 6;
    if ((threadIdx.x & 63) == 0) {
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
// The kernel is a pure latency chain (64 rows at decode), so EVERY load a thread needs -- residual, hidden or the
// first four slabs, and the norm weight -- is issued before anything is consumed: one memory round trip, one
// reduction, one store.  Threads past the row end load clamped (valid) addresses and skip the stores.
template<int MODE, bool HAS_BIAS, int NV>
__global__ __launch_bounds__(kNormMaxThreads) void rmsnorm_kernel(half_t* __restrict__ y,
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
    __shared__ float red[8];
    const int        row  = blockIdx.x;
    const int        nvec = H / 8;
    const size_t     slab = (size_t)M * H;

    half8_t wv[NV], r[NV], hc[NV], bv[NV];
    floatx4 a[NV][4][2];
    size_t  off[NV];
    bool    ok[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int vi = threadIdx.x + i * blockDim.x;
        ok[i]        = vi < nvec;
        const int vc = ok[i] ? vi : nvec - 1;
        off[i]       = (size_t)row * H + (size_t)vc * 8;
        wv[i]        = *(const half8_t*)(weight + (size_t)vc * 8);
        r[i]         = *(const half8_t*)((MODE == 0 ? hidden : resid) + off[i]);
        if constexpr (MODE == 1) {
            hc[i] = *(const half8_t*)(hidden + off[i]);
        }
        if constexpr (MODE == 2) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float* p0 = partial + off[i] + (size_t)min(u, splits - 1) * slab;
                a[i][u][0]      = *(const floatx4*)p0;
                a[i][u][1]      = *(const floatx4*)(p0 + 4);
            }
        }
        if constexpr (HAS_BIAS) {
            bv[i] = *(const half8_t*)(bias + (size_t)vc * 8);
        }
    }

    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if constexpr (MODE == 1) {
            r[i] = r[i] + hc[i];  // fp16 add, one rounding per element
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
                const float*  p0 = partial + off[i] + (size_t)s * slab;
                const floatx4 a0 = *(const floatx4*)p0;
                const floatx4 a1 = *(const floatx4*)(p0 + 4);
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
            r[i] = r[i] + hcur;
        }
        if constexpr (HAS_BIAS) {
            r[i] = r[i] + bv[i];
        }
        if (ok[i]) {
            if constexpr (MODE != 0) {
                *(half8_t*)(resid + off[i]) = r[i];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float f = (float)r[i][e];
                ss            = __builtin_fmaf(f, f, ss);
            }
        }
    }

    ss              = block_sum(ss, red);
    const float inv = 1.0f / __builtin_sqrtf(ss / (float)H + eps);

#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if (ok[i]) {
            half8_t o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const half_t n = (half_t)((float)r[i][e] * inv);  // cast to T first ...
                o[e]           = n * wv[i][e];                    // ... then multiply by w in T
            }
            *(half8_t*)(y + off[i]) = o;
        }
    }
}

// threads: one 16-byte vector per thread up to 512 threads, then two
static void norm_geometry(int H, int* threads, int* nv)
{
    const int nvec = H / 8;
    int       t    = (nvec + 63) / 64 * 64;
    t              = t > kNormMaxThreads ? kNormMaxThreads : t;
    *threads       = t;
    *nv            = (nvec + t - 1) / t;
}

int launch_rmsnorm(half_t* y, const half_t* x, const half_t* w, float eps, int M, int H, hipStream_t st)
{
    TM_REQUIRE(H % 8 == 0 && H <= kNormMaxThreads * kNormMaxVec * 8, "rmsnorm: H must be a multiple of 8 and <= 8192");
    if (M == 0) {
        return 0;
    }
    int threads, nv;
    norm_geometry(H, &threads, &nv);
    if (nv == 1) {
        rmsnorm_kernel<0, false, 1><<<M, threads, 0, st>>>(y, nullptr, x, nullptr, 0, nullptr, w, eps, M, H);
    }
    else {
        rmsnorm_kernel<0, false, 2><<<M, threads, 0, st>>>(y, nullptr, x, nullptr, 0, nullptr, w, eps, M, H);
    }
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
    TM_REQUIRE(H % 8 == 0 && H <= kNormMaxThreads * kNormMaxVec * 8, "rmsnorm: H must be a multiple of 8 and <= 8192");
    TM_REQUIRE((hidden != nullptr) != (partial != nullptr), "exactly one of hidden / partial");
    TM_REQUIRE(!partial || splits >= 1, "splits >= 1");
    if (M == 0) {
        return 0;
    }
    int threads, nv;
    norm_geometry(H, &threads, &nv);
#define TM_NORM_LAUNCH(MODE, BIAS)                                                                                    \
    if (nv == 1) {                                                                                                    \
        rmsnorm_kernel<MODE, BIAS, 1><<<M, threads, 0, st>>>(y, resid, hidden, partial, splits, bias, w, eps, M, H);  \
    }                                                                                                                 \
    else {                                                                                                            \
        rmsnorm_kernel<MODE, BIAS, 2><<<M, threads, 0, st>>>(y, resid, hidden, partial, splits, bias, w, eps, M, H);  \
    }
    if (partial) {
        if (bias) {
            TM_NORM_LAUNCH(2, true)
        }
        else {
            TM_NORM_LAUNCH(2, false)
        }
    }
    else {
        if (bias) {
            TM_NORM_LAUNCH(1, true)
        }
        else {
            TM_NORM_LAUNCH(1, false)
        }
    }
#undef TM_NORM_LAUNCH
    TM_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace tmk
