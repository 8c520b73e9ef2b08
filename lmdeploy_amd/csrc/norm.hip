// RMSNorm / fused (split-K reduce +) residual + RMSNorm for gfx950.
//
// Replaces: src/turbomind/kernels/norm/rms_norm.cu:21-138 (invokeRMSNorm) and
//           :286-362,423-466 (invokeResidualBiasRMSNorm).
// Arithmetic (rms_norm_utils.cuh:6-15): r = h(r + hcur) [then h(r + bias)];
//   inv = rsqrtf(sum f32(r)^2 / H + eps);  y = h( h(f32(r) * inv) * w ).
// One workgroup per token row; 16-byte vector accesses; the row lives in registers between the two
// passes (norm_row.h).  Latency-bound (tiny): the point is fusing the split-K reduce of the
// preceding row-parallel GEMM so that no extra launch / HBM round trip is spent on it.
#include "tm_common.h"
#include "tm_kernels.h"
#include "norm_row.h"

namespace tmk {

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
                                                                  int   H,
                                                                  uint64_t* dbg)
{
    __shared__ float red[8];
    if (dbg && threadIdx.x == 0) {
        dbg[blockIdx.x * 8 + 0] = __builtin_amdgcn_s_memrealtime();
        dbg[blockIdx.x * 8 + 4] = ((uint64_t)__builtin_amdgcn_s_getreg(20 | (31 << 11)) << 32) | (uint32_t)__builtin_amdgcn_s_getreg(4 | (31 << 11));
    }
    norm_row<MODE, HAS_BIAS, NV>(y, resid, hidden, partial, splits, bias, weight, eps, M, H, blockIdx.x, threadIdx.x, blockDim.x, red);
    if (dbg && threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        dbg[blockIdx.x * 8 + 3] = __builtin_amdgcn_s_memrealtime();
    }
}

int launch_rmsnorm(half_t* y, const half_t* x, const half_t* w, float eps, int M, int H, hipStream_t st)
{
    TM_REQUIRE(H % 8 == 0 && H <= kNormMaxThreads * kNormMaxVec * 8, "rmsnorm: H must be a multiple of 8 and <= 8192");
    if (M == 0) {
        return 0;
    }
    int threads, nv;
    norm_geometry(H, &threads, &nv);
    uint64_t* const dbg = gemm_trace_for((size_t)M, "norm", M, 1, 1);
    if (nv == 1) {
        rmsnorm_kernel<0, false, 1><<<M, threads, 0, st>>>(y, nullptr, x, nullptr, 0, nullptr, w, eps, M, H, dbg);
    }
    else {
        rmsnorm_kernel<0, false, 2><<<M, threads, 0, st>>>(y, nullptr, x, nullptr, 0, nullptr, w, eps, M, H, dbg);
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
    uint64_t* const dbg = gemm_trace_for((size_t)M, partial ? "reduce_norm" : "resid_norm", M, 1, 1);
#define TM_NORM_LAUNCH(MODE, BIAS)                                                                                    \
    if (nv == 1) {                                                                                                    \
        rmsnorm_kernel<MODE, BIAS, 1><<<M, threads, 0, st>>>(y, resid, hidden, partial, splits, bias, w, eps, M, H, dbg);  \
    }                                                                                                                 \
    else {                                                                                                            \
        rmsnorm_kernel<MODE, BIAS, 2><<<M, threads, 0, st>>>(y, resid, hidden, partial, splits, bias, w, eps, M, H, dbg);  \
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
