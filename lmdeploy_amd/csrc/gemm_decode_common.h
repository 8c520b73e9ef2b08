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
    // ---- RMSNorm folded into the neighbouring decode GEMMs (round 5; gemm_dec32_kernel only) ------------------------------
    // y = RMSNorm(r) . W = inv[m] * sum_k (r[m,k] g[k]) W[k,n]: the GEMM that PRODUCES r (epilogue 3) adds its fp16 output to
    // the residual stream (bit-exact: r = h(r + h(acc)), rms_norm.cu:286-362), writes xg = h(f32(r) * f32(g)) as the next
    // GEMM's activations and, per column tile, the partial row sums of f32(r)^2; the GEMM that CONSUMES xg multiplies its fp32
    // accumulators by inv[m] = 1 / sqrt(sum_tiles ss / H + eps) before its epilogue (fp16 / gated SiLU / slabs).
    half_t*       resid;     // epilogue 3: [M][N] residual stream, updated in place
    const half_t* norm_w;    // epilogue 3: [N] weight g of the RMSNorm that follows
    float*        ss_out;    // epilogue 3: [gridDim.x][M] per-tile sums of squares of the updated residual rows
    unsigned*     tickets;   // epilogue 3 with split-K: one arrival counter per (column tile, row block); the last arriver of a tile
                             // sums the slices' slabs in slice order and runs the epilogue; it leaves the counter at 0
    const float*  ss_in;     // consumer: [ss_tiles][M] partial sums written by the producing GEMM (nullptr: x is already normalised)
    int           ss_tiles;
    float         ss_inv_h;  // 1 / H of the norm
    float         ss_eps;
    // ---- split-K merged INSIDE the launch for the fp16 / gated-SiLU epilogues (round 6; shapes kShapeMerge + 0..3 / 6..9): the slices
    // park their fp32 tiles write-through and take a ticket exactly as epilogue 3 does; the last arriver of a column tile sums the
    // slices in slice order from zero (the bits of splitk_reduce_kernel) and runs the epilogue.  No slab leaves the launch, no reduce
    // launch follows: what a 256-column tile with a 2-way cross-CU k-split needs for the gated w1w3.
    int           merge;     // 1: epilogues 0 / 1 with gridDim.y > 1 merge in the launch (tickets != nullptr)
};
constexpr int kDec32NormLds = 32 * 64 * 4 + 512;  // consumer scratch behind the stage buffers / reduction image: [parts][rows] sums + inv[rows]

// 16 bytes another workgroup of THIS launch stored write-through (store_wt, sc1): agent-scope relaxed loads (global_load ... sc1)
// bypass this CU's vector L1; valid after the producer drained its stores and bumped an agent-scope counter that this workgroup
// read (MI355X_MICROARCH.md, inter-workgroup visibility: "sc1 loads may replace the acquire only when the producer stored sc1")
__device__ __forceinline__ floatx4 load_agent(const float* src)
{
    typedef unsigned long long u64;
    const u64 lo = __hip_atomic_load((const u64*)src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const u64 hi = __hip_atomic_load((const u64*)src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return floatx4{__builtin_bit_cast(float, (uint32_t)lo), __builtin_bit_cast(float, (uint32_t)(lo >> 32)),
                   __builtin_bit_cast(float, (uint32_t)hi), __builtin_bit_cast(float, (uint32_t)(hi >> 32))};
}

// The 16-byte asm stores END WITH `s_nop 1`: hipcc neither counts nor pads an asm statement, and a VALU write to the store's data
// registers within two wait states of a > 64-bit store corrupts what the store reads (cdna_hip_programming.md 5.7 item 1).  Round 6 found
// it the hard way: the merged split-K park loop put `v_or_b32 v6, ...` one instruction behind `global_store_dwordx4 v[2:3], v[6:9]` --
// dword 0 of lanes 12..15 of every 16-lane row arrived in the slab as 0.0 on some launches (tools/r06_dbg_merge.py).
__device__ __forceinline__ void store_wt(floatx4* dst, floatx4 v, int mode = 1)
{
    // mode (TM_D32_WT >> 4, experiment arms): 0/1 sc1, 2 sc0 sc1, 3 nt, 4 nt sc0 sc1
    if (mode <= 1) {
        asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
    }
    else if (mode == 2) {
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
    }
    else if (mode == 3) {
        asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
    }
    else {
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
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
// gemm_prefill_f16.hip: 256 x 256 tiles over the resident fp16 image (p.wp = LinearWeight::image16), both operands by LDS-DMA (shape kShapeF16),
// grid = (ceil(N / 256), 1, ceil(M / 256))
int launch_f16_256(const Dec32Params& p, dim3 grid, hipStream_t st);

}  // namespace tmk
