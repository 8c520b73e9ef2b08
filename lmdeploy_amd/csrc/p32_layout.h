// Layout "P32" of a dense u4 (AWQ) linear, shared by the kernels that read it (gemm_decode.hip: the W4A16 GEMMs;
// gemm_decode_lc.hip, gemm_prefill.hip; tm_linear_dequant_f16: the operand as an fp16 image).  Built once by repack_p32_kernel at load
// (LinearWeight::prepare's role, models/linear_weight.cc:101-324):
//   unit (kb, cg) at byte ((kb * N/32) + cg) * 2176:
//     [0, 2048)    dword d = (p*64 + lane)*4 + jj  (p = 0..1, jj = 0..3): j = 4p + jj is the 16-k step of the k-block,
//                  lane l holds column 32cg + (l & 31), k = 128kb + 16j + 8(l >> 5) + e, e = 0..7 in nibble order
//                  [k0,k2,k4,k6,k1,k3,k5,k7] -- exactly the A operand of v_mfma_f32_32x32x16_f16 (A[i = l&31][k = 8(l>>5)+e]);
//     [2048, 2176) 32 x (s, -z*s) half2 pairs of the group's columns.
#pragma once
#include "tm_common.h"

namespace tmk {

constexpr int kP32Unit = 2176;

__device__ __forceinline__ half8_t dequant8_p32(uint32_t w, half2_t s2, half2_t z2, uint32_t m1024, uint32_t m64)
{
    // same arithmetic as dequant8 of gemm_w4a16.hip (quantization.h:503-524 magic numbers, exact subtract, one fma)
    const half2_t  k1024 = {(half_t)1024.0f, (half_t)1024.0f};
    const half2_t  k64   = {(half_t)64.0f, (half_t)64.0f};
    const uint32_t hi    = w >> 8;
    half2_t        p0    = bit_cast<half2_t>((w & 0x000f000fu) | m1024) - k1024;
    half2_t        p1    = bit_cast<half2_t>((w & 0x00f000f0u) | m64) - k64;
    half2_t        p2    = bit_cast<half2_t>((hi & 0x000f000fu) | m1024) - k1024;
    half2_t        p3    = bit_cast<half2_t>((hi & 0x00f000f0u) | m64) - k64;
    p0                   = h2_fma(p0, s2, z2);
    p1                   = h2_fma(p1, s2, z2);
    p2                   = h2_fma(p2, s2, z2);
    p3                   = h2_fma(p3, s2, z2);
    return half8_t{p0[0], p0[1], p1[0], p1[1], p2[0], p2[1], p3[0], p3[1]};
}

}  // namespace tmk
