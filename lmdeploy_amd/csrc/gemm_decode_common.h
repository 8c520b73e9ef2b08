// Shared by the W4A16 decode GEMM kernels (gemm_decode.hip: the 16-wave stage kernel and the 128-row prefill tiles;
// gemm_decode_lc.hip: the loader / consumer kernel): launch parameters, write-through stores, compile-time loops.
#pragma once
#include "tm_common.h"
#include "tm_kernels.h"
#include "p32_layout.h"
#include <type_traits>

namespace tmk {

typedef float floatx16 __attribute__((ext_vector_type(16)));

struct Dec32Params {
    const half_t* x;
    int           ldx;
    const void*   wp;  // P32 units
    half_t*       y;
    int           ldy;
    float*        partial;  // [splits][M][N] fp32 slabs (epilogue 2)
    int           M, N, K, KB, ncg;
    int           kb_per_split;
    int           epilogue;  // 0: fp16   1: gated SiLU fp16 (N/2 columns)   2: fp32 slab of split blockIdx.y
    int           wt;        // bit 0: split-K slabs, bit 1: fp16 outputs leave through write-through (sc1) stores: they drain to memory
                             // while the other workgroups still stream instead of sitting dirty in L2 until the end-of-kernel
                             // release writes them back (the kernel boundary then waits for MBs of fp32 slabs)
    uint64_t*     dbg;       // optional [workgroups][8] s_memrealtime stamps (tm_debug_set_gemm_trace): start, loop, epilogue, end,
                             // hw id, -, -, after the k-phase reduction barrier
};

__device__ __forceinline__ void store_wt(floatx4* dst, floatx4 v, int mode = 1)
{
    // mode (TM_D32_WT >> 4, experiment arms): 0/1 sc1, 2 sc0 sc1, 3 nt, 4 nt sc0 sc1
    if (mode <= 1) {
        asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(v) : "memory");
    }
    else if (mode == 2) {
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(v) : "memory");
    }
    else if (mode == 3) {
        asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(dst), "v"(v) : "memory");
    }
    else {
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(dst), "v"(v) : "memory");
    }
}
__device__ __forceinline__ void store_wt(half4_t* dst, half4_t v)
{
    asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(dst), "v"(v) : "memory");
}
__device__ __forceinline__ void store_wt(half2_t* dst, half2_t v)
{
    asm volatile("global_store_dword %0, %1, off sc1" ::"v"(dst), "v"(v) : "memory");
}

template<int N, class F, int I = 0>
__device__ __forceinline__ void static_for(F&& f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<N, F, I + 1>(static_cast<F&&>(f));
    }
}

// gemm_decode_lc.hip: the loader / consumer decode kernel (shape kShapeLC), grid = (ceil(ncg / 4), splits)
int launch_dec_lc(const Dec32Params& p, dim3 grid, hipStream_t st);
// gemm_prefill.hip: 256 x 256 prefill tiles, weights dequantised once per workgroup tile through LDS (shape kShapePre256),
// grid = (ceil(ncg / 8), splits, ceil(M / 256))
int launch_pre256(const Dec32Params& p, dim3 grid, hipStream_t st);

}  // namespace tmk
