// Common device/host helpers for the MI355X-native TurboMind hot path (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

namespace tmk {

typedef _Float16 half_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float    floatx4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int kWave = 64;

// ---- error plumbing (C-ABI never throws) -----------------------------------------------
void        set_last_error(const std::string& msg);
const char* get_last_error();
// Raise a kernel's dynamic-LDS limit on the CURRENT device, once per (kernel, device): hipFuncAttributeMaxDynamicSharedMemorySize
// is a per-device function attribute.  Thread-safe (engines of several devices / threads launch concurrently).
int         ensure_dynamic_lds(const void* kernel, int bytes);

#define TM_HIP_CHECK(expr)                                                                         \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) {                                                                    \
            ::tmk::set_last_error(std::string(#expr) + ": " + hipGetErrorString(_e) + " @" + __FILE__ + ":" \
                                 + std::to_string(__LINE__));                                      \
            return 5; /* TM_FAIL */                                                                \
        }                                                                                          \
    } while (0)

#define TM_REQUIRE(cond, msg)                                                                      \
    do {                                                                                           \
        if (!(cond)) {                                                                             \
            ::tmk::set_last_error(std::string("invalid argument: ") + (msg) + " (" #cond ")");      \
            return 1; /* TM_INVALID */                                                             \
        }                                                                                          \
    } while (0)

// ---- tiny device helpers ---------------------------------------------------------------
template<class To, class From>
__device__ __forceinline__ To bit_cast(const From& f)
{
    return __builtin_bit_cast(To, f);
}

__device__ __forceinline__ half2_t h2_fma(half2_t a, half2_t b, half2_t c)
{
    return __builtin_elementwise_fma(a, b, c);  // v_pk_fma_f16, single rounding
}

// DPP lane exchange inside a row of 16 lanes (no LDS).  ctrl: quad_perm / row_* codes.
template<int CTRL>
__device__ __forceinline__ float dpp_f32(float v)
{
    return bit_cast<float>(__builtin_amdgcn_update_dpp(0, bit_cast<int>(v), CTRL, 0xF, 0xF, false));
}
template<int CTRL>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, false);
}

constexpr int DPP_XOR1  = 0xB1;   // quad_perm [1,0,3,2]
constexpr int DPP_XOR2  = 0x4E;   // quad_perm [2,3,0,1]
constexpr int DPP_HMIRR = 0x141;  // row_half_mirror (lane i <-> 7-i in each 8)
constexpr int DPP_ROR8  = 0x128;  // row_ror:8      (lane i <- i^8 in each 16)

// butterfly all-reduce (sum / max) over groups of N consecutive lanes, N in {8,16,64}
template<int N>
__device__ __forceinline__ float group_sum(float v)
{
    v += dpp_f32<DPP_XOR1>(v);
    v += dpp_f32<DPP_XOR2>(v);
    v += dpp_f32<DPP_HMIRR>(v);
    if constexpr (N >= 16) {
        v += dpp_f32<DPP_ROR8>(v);
    }
    if constexpr (N >= 64) {
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
    }
    return v;
}
template<int N>
__device__ __forceinline__ float group_max(float v)
{
    v = fmaxf(v, dpp_f32<DPP_XOR1>(v));
    v = fmaxf(v, dpp_f32<DPP_XOR2>(v));
    v = fmaxf(v, dpp_f32<DPP_HMIRR>(v));
    if constexpr (N >= 16) {
        v = fmaxf(v, dpp_f32<DPP_ROR8>(v));
    }
    if constexpr (N >= 64) {
        v = fmaxf(v, __shfl_xor(v, 16));
        v = fmaxf(v, __shfl_xor(v, 32));
    }
    return v;
}

// reduce across the lanes that differ in bits >= log2(LOW) (i.e. keep the low LOW lanes distinct)
template<int LOW>
__device__ __forceinline__ float upper_sum(float v)
{
    if constexpr (LOW <= 8) {
        v += dpp_f32<DPP_ROR8>(v);
    }
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}
template<int LOW>
__device__ __forceinline__ float upper_max(float v)
{
    if constexpr (LOW <= 8) {
        v = fmaxf(v, dpp_f32<DPP_ROR8>(v));
    }
    v = fmaxf(v, __shfl_xor(v, 16));
    v = fmaxf(v, __shfl_xor(v, 32));
    return v;
}

__device__ __forceinline__ float fast_exp2(float x)
{
    return __builtin_amdgcn_exp2f(x);  // v_exp_f32 (no denormal range handling needed here)
}

}  // namespace tmk
