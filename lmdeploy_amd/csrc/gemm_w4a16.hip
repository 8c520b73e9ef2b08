// W4A16 (AWQ, group 128) weight-only GEMM as an MFMA contraction for gfx950, plus the fp16 dense
// variant used by lm_head.
//
// Replaces: LlamaLinear::Forward -> gemm::Gemm::Run (src/turbomind/models/llama/LlamaLinear.cu:140-216,
//           kernels/gemm/gemm.cu:257-344), the u4 dequant transform (kernels/gemm/transform.h:34-74),
//           scale/zero fusion (cast.cu:134-165), the gated-SiLU epilogue (epilogue.h:159-176,428-446),
//           the load-time repack LinearWeight::prepare (models/linear_weight.cc:101-324).
//
// Arithmetic: w[k,n] = h(fma(h(q[k,n]), s, h(-z*s)))  (single rounding; q exact via the 0x6400 magic number
// followed by an exact fp16 subtract) ; y = h(sum_k f32(x)*f32(w)) with fp32 accumulation on the MFMA ;
// gated: out[m,j] = h(silu_f32(acc[m,2j]) * acc[m,2j+1]).
//
// MI355X design (decode, M <= 64: HBM-bound weight streaming at the roofline ridge):
//   * Weights are repacked ONCE at load into MFMA-fragment order.  The op is computed transposed,
//     Y^T = W^T X^T, so the weight is the MFMA "A" operand of v_mfma_f32_16x16x32_f16: lane l holds
//     n = 16*nt + (l&15) and eight consecutive k.  One lane-dword = 8 u4 = one MFMA operand; one 16-B lane
//     load = 4 MFMA k-steps = 128 k = exactly one quantisation group => ONE (s, -z*s) pair per 16-B load,
//     and a wave-load is a fully coalesced 1 KiB.  Nibbles are stored [k0,k2,k4,k6,k1,k3,k5,k7] so that
//     (w >> 4p) & 0x000f000f yields the packed pair (k_2p, k_2p+1) directly.
//   * Each dequantised operand (4 VGPRs) is reused by MT MFMAs (MT = M/16 row tiles), activations come
//     from LDS (XOR-swizzled 16-B chunks, conflict-free ds_read_b128) and are reused by NT column tiles.
//   * A wave streams NT contiguous K-panels; weight loads for the next K-chunk are in flight while the
//     current one is contracted (register double buffer), activations are staged global->reg->LDS.
//   * Split-K over grid.y with fp32 slabs; the slab reduce is fused into the consumer (residual+RMSNorm,
//     norm.hip) or done by splitk_reduce_kernel.  Grid is sized for >= 256 workgroups.
#include "tm_common.h"
#include "tm_kernels.h"
#include <stdlib.h>
#include <array>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>
#include <stdio.h>

namespace tmk {

// ------------------------------------------------------------------------------------------------
// load-time repack
// ------------------------------------------------------------------------------------------------
__global__ void repack_u4_kernel(uint32_t* __restrict__ out, const int32_t* __restrict__ qw, int K, int N)
{
    // one thread per output dword: idx = ((kb*NTILES + nt)*64 + lane)*4 + j   (k-block major: all column tiles of
    // one k-block are contiguous, so the whole grid walks HBM as ONE sequential stream -- panel-major layouts
    // (one 32 KiB stream per wave) thrash DRAM pages across ~2000 concurrent streams)
    const size_t idx   = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)K * N / 8;
    if (idx >= total) {
        return;
    }
    const int    NTILES = N / 16;
    const int    j    = idx & 3;
    const int    lane = (idx >> 2) & 63;
    const size_t tile = idx >> 8;
    const int    nt   = tile % NTILES;
    const int    kb   = tile / NTILES;
    const int    n    = nt * 16 + (lane & 15);
    const int    k0   = kb * 128 + j * 32 + (lane >> 4) * 8;
    uint32_t     w    = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const uint32_t word = (uint32_t)qw[(size_t)(k0 + e) * (N / 8) + (n >> 3)];
        const uint32_t q    = (word >> (4 * (n & 7))) & 15u;
        const int      nib  = (e & 1) ? 4 + (e >> 1) : (e >> 1);
        w |= q << (4 * nib);
    }
    out[idx] = w;
}

__global__ void repack_sz_kernel(uint32_t* __restrict__ out,
                                 const half_t* __restrict__ scales,
                                 const half_t* __restrict__ zeros,
                                 int KB,
                                 int N)
{
    // idx = (kb*NTILES + nt)*16 + i
    const size_t idx   = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)KB * N;
    if (idx >= total) {
        return;
    }
    const int    i    = idx & 15;
    const size_t tile = idx >> 4;
    const int    nt   = tile % (N / 16);
    const int    kb   = tile / (N / 16);
    const int    n    = nt * 16 + i;
    const half_t s    = scales[(size_t)kb * N + n];
    const half_t z    = zeros[(size_t)kb * N + n];
    const half_t zs   = (-z) * s;  // one fp16 rounding (cast.cu:151-156)
    half2_t      pr   = {s, zs};
    out[idx]          = bit_cast<uint32_t>(pr);
}

__global__ void repack_f16_kernel(half_t* __restrict__ out, const half_t* __restrict__ w, int K, int N)
{
    // one thread per 8 halves: idx = ((kb*NTILES + nt)*4 + v)*64 + lane, k-quarter kq = 4*kb + v
    const size_t idx   = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)K * N / 8;
    if (idx >= total) {
        return;
    }
    const int    lane = idx & 63;
    const int    v    = (idx >> 6) & 3;
    const size_t tile = idx >> 8;
    const int    nt   = tile % (N / 16);
    const int    kq   = (int)(tile / (N / 16)) * 4 + v;
    const int    n    = nt * 16 + (lane & 15);
    const int    k0   = kq * 32 + (lane >> 4) * 8;
    half8_t      o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        o[e] = w[(size_t)(k0 + e) * N + n];
    }
    *(half8_t*)(out + idx * 8) = o;
}

// fp8 (e4m3) weights [K][N] -> tiles [kb][nt][v = 0..1][64 lanes][16 B]: lane l holds column 16 nt + (l & 15) and, in
// u32 (j & 1) * 2 .. + 1 of vector v = j >> 1, the 8 bytes k = 128 kb + 32 j + 8 (l >> 4) + 0..7 of 32-k step j
__global__ void repack_fp8_kernel(uint8_t* __restrict__ out, const uint8_t* __restrict__ w, int K, int N)
{
    // one thread per 16 output bytes: idx = ((kb*NTILES + nt)*2 + v)*64 + lane
    const size_t idx   = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)K * N / 16;
    if (idx >= total) {
        return;
    }
    const int    lane = idx & 63;
    const int    v    = (idx >> 6) & 1;
    const size_t tile = idx >> 7;
    const int    nt   = tile % (N / 16);
    const int    kb   = (int)(tile / (N / 16));
    const int    n    = nt * 16 + (lane & 15);
    uint8_t      o[16];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        const int k0 = kb * 128 + (2 * v + jj) * 32 + (lane >> 4) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            o[jj * 8 + e] = w[(size_t)(k0 + e) * N + n];
        }
    }
    *(u32x4*)(out + idx * 16) = *(const u32x4*)o;
}

// 128 x 128 block scales (fp32 [K/128][N/128]) -> per-column group scales in the (s, 0) half2 slots of the u4 path:
// the reference expands each block scale over its 128 output channels and casts it to the activation type
// (BlockscaleToGroupscale, models/linear_weight.cc:138-150); w = h(f16(e4m3) * s) needs no zero point.
// gated = the fused w1w3 linear with (gate_j, up_j)-interleaved columns: w1 and w3 are block-quantised SEPARATELY in a
// checkpoint, so the scale row is [w1's inter/128 blocks | w3's inter/128 blocks] and column n = 2j + which reads block
// which * inter/128 + j/128 (the interleave of ffn.py:31-34 applied to the expanded per-column scales).
__global__ void repack_sz_fp8_kernel(uint32_t* __restrict__ out, const float* __restrict__ block_scales, int KB, int N, int gated)
{
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)KB * N) {
        return;
    }
    const int     kb = (int)(idx / N);
    const int     n  = (int)(idx % N);
    const int     nb = (N + 127) / 128;
    const int     b  = gated ? (n & 1) * (nb / 2) + (n >> 1) / 128 : n / 128;
    const half_t  s  = (half_t)block_scales[(size_t)kb * nb + b];
    const half2_t pr = {s, (half_t)0.f};
    out[idx]         = bit_cast<uint32_t>(pr);  // [kb][nt][16] == [kb][n]
}

void linear_weight_free(LinearWeight& w)
{
    if (w.packed) {
        (void)hipFree(w.packed);
    }
    if (w.sz) {
        (void)hipFree(w.sz);
    }
    if (w.packed32) {
        (void)hipFree(w.packed32);
    }
    if (w.packed8) {
        (void)hipFree(w.packed8);
    }
    if (w.image16) {
        (void)hipFree(w.image16);
    }
    w.image16 = nullptr;
    w.packed8 = nullptr;
    w.packed   = nullptr;
    w.sz       = nullptr;
    w.packed32 = nullptr;
}

int linear_weight_prepare_u4(LinearWeight& w, const int32_t* qweight, const half_t* scales, const half_t* zeros,
                             hipStream_t st, bool p32_only)
{
    TM_REQUIRE(w.group == 128, "AWQ group size must be 128 (lmdeploy/turbomind/converter.py:86-92)");
    TM_REQUIRE(w.K % 128 == 0 && w.N % 16 == 0, "K % 128 == 0 and N % 16 == 0");
    TM_REQUIRE(!p32_only || dec32_serves_every_m(w.K, w.N), "p32_only: the decode kernels must serve every M of this linear");
    w.type         = 0;
    w.packed_bytes = (size_t)w.K * w.N / 2;  // algorithmic size of the codes / (s, -z*s) pairs, whichever image holds them
    w.sz_bytes     = (size_t)(w.K / 128) * w.N * 4;
    if (!p32_only) {
        if (!w.packed) {
            TM_HIP_CHECK(hipMalloc(&w.packed, w.packed_bytes));
            TM_HIP_CHECK(hipMalloc((void**)&w.sz, w.sz_bytes));
        }
        const size_t nd = (size_t)w.K * w.N / 8;
        repack_u4_kernel<<<(nd + 255) / 256, 256, 0, st>>>((uint32_t*)w.packed, qweight, w.K, w.N);
        TM_HIP_CHECK(hipGetLastError());
        const size_t ns = (size_t)(w.K / 128) * w.N;
        repack_sz_kernel<<<(ns + 255) / 256, 256, 0, st>>>(w.sz, scales, zeros, w.K / 128, w.N);
        TM_HIP_CHECK(hipGetLastError());
    }
    if (w.N % 32 == 0) {  // the decode kernel's layout (gemm_decode.hip)
        w.packed32_bytes = p32_bytes(w.K, w.N);
        if (!w.packed32) {
            TM_HIP_CHECK(hipMalloc(&w.packed32, w.packed32_bytes));
        }
        return launch_repack_p32(w.packed32, qweight, scales, zeros, w.K, w.N, st);
    }
    return 0;
}

int linear_weight_build_f16_image(LinearWeight& w, hipStream_t st)
{
    TM_REQUIRE(w.type == 0 && w.packed32 != nullptr && w.N % 32 == 0 && w.K % 128 == 0, "fp16 image: a u4 linear with its P32 image");
    if (!w.image16) {
        TM_HIP_CHECK(hipMalloc((void**)&w.image16, (size_t)w.N * w.K * sizeof(half_t)));
    }
    return launch_dequant_p32_f16(w.image16, w, st);
}

int linear_weight_prepare_fp8(LinearWeight& w, const uint8_t* weight, const float* block_scales, bool gated_scales, hipStream_t st)
{
    TM_REQUIRE(w.K % 128 == 0 && w.N % 16 == 0, "K % 128 == 0 and N % 16 == 0");
    TM_REQUIRE(!gated_scales || w.N % 256 == 0, "fp8 w1w3: inter must be a multiple of the 128-column scale block");
    w.type         = 2;
    w.group        = 128;
    w.packed_bytes = (size_t)w.K * w.N;
    w.sz_bytes     = (size_t)(w.K / 128) * w.N * 4;
    if (!w.packed) {
        TM_HIP_CHECK(hipMalloc(&w.packed, w.packed_bytes));
        TM_HIP_CHECK(hipMalloc((void**)&w.sz, w.sz_bytes));
    }
    const size_t nv = (size_t)w.K * w.N / 16;
    repack_fp8_kernel<<<(nv + 255) / 256, 256, 0, st>>>((uint8_t*)w.packed, weight, w.K, w.N);
    TM_HIP_CHECK(hipGetLastError());
    const size_t ns = (size_t)(w.K / 128) * w.N;
    repack_sz_fp8_kernel<<<(ns + 255) / 256, 256, 0, st>>>(w.sz, block_scales, w.K / 128, w.N, gated_scales ? 1 : 0);
    TM_HIP_CHECK(hipGetLastError());
    if (w.N % 32 == 0) {  // the fp8 x fp8 kernel's layout (gemm_fp8.hip)
        w.packed8_bytes = p8_bytes(w.K, w.N);
        if (!w.packed8) {
            TM_HIP_CHECK(hipMalloc(&w.packed8, w.packed8_bytes));
        }
        return launch_repack_p8(w.packed8, weight, block_scales, w.K, w.N, gated_scales, st);
    }
    return 0;
}

int linear_weight_prepare_f16(LinearWeight& w, const half_t* weight, hipStream_t st)
{
    TM_REQUIRE(w.K % 128 == 0 && w.N % 16 == 0, "K % 128 == 0 and N % 16 == 0");
    w.type         = 1;
    w.packed_bytes = (size_t)w.K * w.N * 2;
    w.sz_bytes     = 0;
    if (!w.packed) {
        TM_HIP_CHECK(hipMalloc(&w.packed, w.packed_bytes));
    }
    const size_t nv = (size_t)w.K * w.N / 8;
    repack_f16_kernel<<<(nv + 255) / 256, 256, 0, st>>>((half_t*)w.packed, weight, w.K, w.N);
    TM_HIP_CHECK(hipGetLastError());
    return 0;
}


// ------------------------------------------------------------------------------------------------
// main kernel
// ------------------------------------------------------------------------------------------------
struct GemmGroup {
    const void* wq;
    const void* sz;
};

struct GemmParams {
    const half_t*   x;
    int             ldx;
    const u32x4*    wq;  // packed weights
    const uint32_t* sz;  // packed (s, -z*s)
    half_t*         y;
    int             ldy;
    float*          partial;  // [splits][M][N]
    int             M, N, K;
    int             KB;            // K / 128
    int             kb_per_split;  // k-blocks (128 k) per grid.y slice
    int             epilogue;      // 0: fp16 store  1: gated silu fp16 store  2: fp32 partial slabs
    int             rotate_k;      // per-workgroup rotation of the K walk (L2 hot-spot avoidance)
    uint64_t*       dbg;           // optional [workgroups][4] s_memrealtime stamps (100 MHz): start, loop, epilogue, end
    // grouped (mixture-of-experts) mode: grid.z = experts x zper row blocks; expert e contracts the flat rows
    // seg[e] .. seg[e+1] (device routing offsets) with ITS weights; x row of flat row f = row_idx ? row_idx[f] : f
    const GemmGroup* groups;       // device [experts]: packed weight / scale pointers (nullptr = plain GEMM)
    const int*       seg;          // device [experts + 1]
    const int*       row_idx;      // device [flat rows] or nullptr
    int              zper;         // row blocks per expert
    int              x_rows;       // rows of x (bounds of the activation buffer descriptor)
};

// m1024 / m64 hold 0x64006400 / 0x54005400 in VGPRs (made opaque by the caller): with the magic in a register
// hipcc selects ONE v_and_or_b32 per pair instead of v_and + v_or (VOP3 takes a single literal on gfx9).
__device__ __forceinline__ half8_t dequant8(uint32_t w, half2_t s2, half2_t z2, uint32_t m1024, uint32_t m64)
{
    // nibble p (p<4) = k_2p, nibble 4+p = k_2p+1.  Bits 0-3 / 16-19 under 0x6400 read 1024+q, bits 4-7 / 20-23 under
    // 0x5400 read 64+q (quantization.h:503-524); both subtractions are exact in fp16.
    const half2_t k1024 = {(half_t)1024.0f, (half_t)1024.0f};
    const half2_t k64   = {(half_t)64.0f, (half_t)64.0f};
    const uint32_t hi   = w >> 8;
    half2_t       p0 = bit_cast<half2_t>((w & 0x000f000fu) | m1024) - k1024;
    half2_t       p1 = bit_cast<half2_t>((w & 0x00f000f0u) | m64) - k64;
    half2_t       p2 = bit_cast<half2_t>((hi & 0x000f000fu) | m1024) - k1024;
    half2_t       p3 = bit_cast<half2_t>((hi & 0x00f000f0u) | m64) - k64;
    p0               = h2_fma(p0, s2, z2);
    p1               = h2_fma(p1, s2, z2);
    p2               = h2_fma(p2, s2, z2);
    p3               = h2_fma(p3, s2, z2);
    return half8_t{p0[0], p0[1], p1[0], p1[1], p2[0], p2[1], p3[0], p3[1]};
}

// One workgroup = WN x WK waves: WN column groups (NT column tiles of 16 each) x WK k-phases, MT row tiles (16 tall),
// over a slice of K.  Per ITERATION the workgroup consumes KS*WK k-blocks (a k-block = 128 k = one quantisation
// group = one 16-B lane load per tile): wave (wn, wk) contracts the k-blocks kb0 + (i*KS + kk)*WK + wk, kk < KS.
// The WK partial sums are added through LDS at the end (split-K INSIDE the workgroup, no fp32 slabs in HBM).
//   * weights: per-wave register ring, PF iterations (= PF*KS k-blocks) deep -- the HBM stream is never waited on
//     for less than PF iterations;
//   * activations: [KS*WK][MB][128] fp16 per iteration through a double-buffered, XOR-swizzled LDS stage shared by
//     all waves (global -> register ring -> LDS one iteration ahead), ONE barrier per iteration.
//   * KS > 1 exists because an iteration costs a fixed ~1200 cycles of exposed latencies (barrier, LDS round trip,
//     waitcnt) that nothing hides while all waves of the CU move in lockstep (measured by ablation: the phases of an
//     iteration are additive): more k-blocks per barrier amortise that chain.
// ABL: ablation bit mask (timing experiments of -DTM_EXPERIMENTS builds only, results are garbage):
//   1 no dequant VALU, 2 no MFMA, 4 no LDS x reads, 8 no x staging (loads + LDS writes), 16 no weight loads in the loop,
//   32 half of the activation loads, 64 half of the activation LDS writes
template<int WT, int MT, int NT, int WN, int WK, int KS, int PF, int ABL = 0, bool GRP = false>
__global__ __launch_bounds__(WN * WK * 64) void gemm_kernel(GemmParams p)
{
    static_assert(PF % 2 == 0, "ring depth must be even (LDS stages alternate)");
    constexpr int WAVES   = WN * WK;
    constexpr int MB      = 16 * MT;
    constexpr int THREADS = WAVES * 64;
    constexpr int ROWB    = 256;                 // one k-block of one row
    constexpr int PHB     = MB * ROWB;           // one k-block of a stage
    constexpr int SUBS    = KS * WK;             // k-blocks per stage
    constexpr int BUFB    = SUBS * PHB;          // one LDS stage
    constexpr int NCHUNK  = SUBS * MB * 16;      // 16-B chunks per stage
    constexpr int XR      = (NCHUNK + THREADS - 1) / THREADS;
    constexpr int WV      = WT == 0 ? 1 : (WT == 2 ? 2 : 4);  // u32x4 per (tile, k-block) per lane: u4 / fp8 / f16
    constexpr bool XFULL  = NCHUNK % THREADS == 0;  // every thread stages exactly XR chunks

    extern __shared__ __attribute__((aligned(16))) char smem[];  // max(2 * BUFB, reduction scratch)

    const int tid  = threadIdx.x;
    const int wgid = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    if (p.dbg && tid == 0) {
        p.dbg[wgid * 8 + 0] = __builtin_amdgcn_s_memrealtime();
        // where this workgroup runs: XCC_ID (reg 20) and HW_ID (reg 4: wave/simd/pipe/cu/sh/se ids)
        p.dbg[wgid * 8 + 4] = ((uint64_t)__builtin_amdgcn_s_getreg(20 | (31 << 11)) << 32) | (uint32_t)__builtin_amdgcn_s_getreg(4 | (31 << 11));
    }
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn   = wave % WN;
    const int wk   = wave / WN;
    const int i16  = lane & 15;
    const int g    = lane >> 4;

    const int ntiles = p.N / 16;
    const int nt0    = (blockIdx.x * WN + wn) * NT;
    // plain GEMM: rows m0 .. of x / y.  Grouped: this workgroup's expert, its row segment and its weights.
    int         m0 = blockIdx.z * MB, row0 = 0, Mloc = p.M;
    const void* wq_base = p.wq;
    const void* sz_base = p.sz;
    if constexpr (GRP) {  // compile-time: the plain kernel keeps its exact instruction stream
        const int e  = blockIdx.z / p.zper;
        m0           = (blockIdx.z - e * p.zper) * MB;
        row0         = p.seg[e];
        Mloc         = min(p.seg[e + 1] - row0, p.M);
        wq_base      = p.groups[e].wq;
        sz_base      = p.groups[e].sz;
        if (m0 >= Mloc) {
            return;  // no rows for this block (whole workgroup, before any barrier)
        }
    }
    const int kb0    = blockIdx.y * p.kb_per_split;
    const int nkb    = min(p.kb_per_split, p.KB - kb0);  // multiple of SUBS (host guarantees)
    const int nit    = nkb / SUBS;

    // Buffer descriptors (SRD) + per-lane 32-bit byte offsets + SCALAR k-block offsets: the address math of every
    // load in the loop is SALU-only (raw pointers cost ~10 VALU per load in 64-bit adds).
    // Tiles past the edge are clamped: loads stay in bounds, stores are skipped.
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)wq_base, 0, (int)((size_t)p.KB * ntiles * 1024 * WV), 0x00020000);
    const auto rs_s = __builtin_amdgcn_make_buffer_rsrc((void*)sz_base, 0, WT != 1 ? p.KB * ntiles * 64 : 0, 0x00020000);
    const auto rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)(((size_t)((GRP ? p.x_rows : p.M) - 1) * p.ldx + p.K) * 2), 0x00020000);
    int woff[NT], soff[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int nt = min(nt0 + t, ntiles - 1);
        woff[t]      = (nt * 64 * WV + lane) * 16;
        soff[t]      = (nt * 16 + i16) * 4;
    }
    const int wstride = ntiles * 1024 * WV;  // bytes per k-block of packed weights
    const int sstride = ntiles * 64;         // bytes per k-block of (s, -z*s) pairs

    floatx4 acc[NT][MT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            acc[t][mt] = floatx4{0.f, 0.f, 0.f, 0.f};
        }
    }

    u32x4    ring[PF][KS][NT][WV];
    uint32_t sring[PF][KS][NT];
    u32x4    xs[PF][XR];

    // x chunk q of a stage: k-block q/(MB*16) of the stage, row (q/16)%MB, 16-B chunk q%16.  Loads are
    // UNCONDITIONAL (clamped): a per-lane "load or zero" select makes hipcc branch around every load and drain
    // vmcnt(0) each time.  Rows past M only feed output rows that are never stored.
    int xoff[XR];
    int xlds[XR];
#pragma unroll
    for (int r = 0; r < XR; ++r) {
        const int q  = tid + THREADS * r;
        const int qc = min(q, NCHUNK - 1);
        const int sb = qc / (MB * 16);
        const int m  = (qc >> 4) % MB;
        const int ci = qc & 15;
        int xrow     = row0 + min(m0 + m, Mloc - 1);
        if (GRP && p.row_idx) {
            xrow = p.row_idx[xrow];
        }
        xoff[r]      = (xrow * p.ldx + ci * 8) * 2 + sb * 256;
        xlds[r]      = q < NCHUNK ? sb * PHB + m * ROWB + ((ci ^ (m & 15)) << 4) : -1;
    }
    uint32_t m1024 = 0x64006400u, m64 = 0x54005400u;
    asm volatile("" : "+v"(m1024), "+v"(m64));  // keep the magic numbers in VGPRs (see dequant8)
    const int last = nit - 1;  // iteration indices are clamped to `last`: the ring tail re-loads harmlessly

#define TM_LOAD_W(slot, i)                                                                                   \
    {                                                                                                        \
        const int kb_ = kb0 + min((i), last) * SUBS + wk;                                                    \
        _Pragma("unroll") for (int kk = 0; kk < KS; ++kk)                                                    \
        {                                                                                                    \
            _Pragma("unroll") for (int t = 0; t < NT; ++t)                                                   \
            {                                                                                                \
                _Pragma("unroll") for (int v = 0; v < WV; ++v)                                               \
                {                                                                                            \
                    ring[slot][kk][t][v] = __builtin_amdgcn_raw_buffer_load_b128(                            \
                        rs_w, woff[t] + v * 1024, (kb_ + kk * WK) * wstride, /*nt*/ 2);                      \
                }                                                                                            \
                if constexpr (WT != 1) {                                                                     \
                    sring[slot][kk][t] =                                                                     \
                        __builtin_amdgcn_raw_buffer_load_b32(rs_s, soff[t], (kb_ + kk * WK) * sstride, 0);   \
                }                                                                                            \
            }                                                                                                \
        }                                                                                                    \
    }
#define TM_LOAD_X(set, i)                                                                                    \
    {                                                                                                        \
        const int kb_ = kb0 + min((i), last) * SUBS;                                                         \
        _Pragma("unroll") for (int r = 0; r < ((ABL & 32) ? (XR + 1) / 2 : XR); ++r)                         \
        {                                                                                                    \
            xs[set][r] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, xoff[r], kb_ * 256, 0);                 \
        }                                                                                                    \
    }
#define TM_STORE_X(set, buf)                                                                                 \
    _Pragma("unroll") for (int r = 0; r < ((ABL & 64) ? (XR + 1) / 2 : XR); ++r)                             \
    {                                                                                                        \
        if (XFULL || xlds[r] >= 0) {                                                                         \
            *(u32x4*)(smem + (buf)*BUFB + xlds[r]) = xs[set][r];                                             \
        }                                                                                                    \
    }

    if (nit > 0) {
        // ---- prologue --------------------------------------------------------------------------
        // Issue order matters: VMEM loads return IN ORDER, so a wait for x(i) also waits for every older
        // load.  x(i) is therefore always issued right before w(i), PF iterations ahead of its use: no
        // wait in the loop ever covers a load younger than PF-1 iterations.
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            TM_LOAD_X(u, u);
            TM_LOAD_W(u, u);
            // pin the issue order: if the scheduler moves slot 0's loads to the end of the prologue, the waitcnt
            // pass needs vmcnt(0) on the loop-entry edge and -- merging edges conservatively -- drains the whole
            // ring at the top of EVERY trip (measured: 61 % of wave time in s_waitcnt)
            __builtin_amdgcn_sched_barrier(0);
        }
        TM_STORE_X(0, 0);
        __syncthreads();
        if (p.dbg && tid == 0) {
            p.dbg[wgid * 8 + 1] = __builtin_amdgcn_s_memrealtime();
        p.dbg[wgid * 8 + 5] = __builtin_amdgcn_s_memtime();  // shader-clock ticks: effective clock of the main loop
        }

        // ---- main loop: branch-free body, statically unrolled over the ring --------------------------
        // Iterations past nit (ring padding) contract zeroed weights, so they add exactly 0.
        // Software pipeline: the dequantised A operand of the next 32-k step (VALU) and its activation fragments (LDS)
        // are produced while the MFMAs of the current step run.
        auto dq = [&](int slot, int kk, int t, int j, bool live) -> half8_t {
            if constexpr (WT == 0) {
                const half2_t pr = bit_cast<half2_t>(live ? sring[slot][kk][t] : 0u);  // (s, -z*s) = 0 -> w = 0
                if constexpr (ABL & 1) {
                    const uint32_t rw = ring[slot][kk][t][0][j] ^ bit_cast<uint32_t>(pr);
                    return bit_cast<half8_t>(u32x4{rw, rw, rw, rw});
                }
                else {
                    return dequant8(ring[slot][kk][t][0][j], half2_t{pr[0], pr[0]}, half2_t{pr[1], pr[1]}, m1024, m64);
                }
            }
            else if constexpr (WT == 2) {
                // e4m3 -> f16 is exact (v_cvt_scalef32_pk_f16_fp8, scale 1); w = h(f16(q) * s), one rounding
                // (kernels/attention/quantization.h:820-846 + the group scale of kernels/gemm/transform.h)
                const half2_t  pr = bit_cast<half2_t>(live ? sring[slot][kk][t] : 0u);
                const half2_t  s2 = {pr[0], pr[0]};
                const u32x4    wv = ring[slot][kk][t][j >> 1];
                const uint32_t w0 = wv[(j & 1) * 2], w1 = wv[(j & 1) * 2 + 1];
                const half2_t  a0 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w0, 1.0f, false) * s2;
                const half2_t  a1 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w0, 1.0f, true) * s2;
                const half2_t  a2 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w1, 1.0f, false) * s2;
                const half2_t  a3 = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8(w1, 1.0f, true) * s2;
                return half8_t{a0[0], a0[1], a1[0], a1[1], a2[0], a2[1], a3[0], a3[1]};
            }
            else {
                const u32x4 wv = ring[slot][kk][t][j];
                return bit_cast<half8_t>(live ? wv : u32x4{0u, 0u, 0u, 0u});
            }
        };
        auto ldx = [&](const char* stage, int kk, int j, int mt) -> half8_t {
            if constexpr (ABL & 4) {
                return bit_cast<half8_t>(ring[0][0][0][0]);
            }
            else {
                return *(const half8_t*)(stage + (kk * WK + wk) * PHB + (mt * 16 + i16) * ROWB + (((j * 4 + g) ^ i16) << 4));
            }
        };
        half8_t wfn[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            wfn[t] = dq(0, 0, t, 0, true);
        }
        for (int base = 0; base < nit; base += PF) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int  i         = base + u;
                const bool live      = i < nit;  // wave-uniform
                const bool live_next = i + 1 < nit;
                const char* stage    = smem + (u & 1) * BUFB;
                half8_t     xf[MT], xfn[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    xf[mt] = ldx(stage, 0, 0, mt);
                }
#pragma unroll
                for (int s = 0; s < KS * 4; ++s) {  // s = kk*4 + j: 32-k steps of this iteration
                    const int kk = s >> 2, j = s & 3;
                    half8_t   wf[NT];
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        wf[t] = wfn[t];
                    }
                    // produce the next step (or step 0 of the next iteration) while the MFMAs below are in flight
                    if (s + 1 < KS * 4) {
#pragma unroll
                        for (int t = 0; t < NT; ++t) {
                            wfn[t] = dq(u, (s + 1) >> 2, t, (s + 1) & 3, live);
                        }
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) {
                            xfn[mt] = ldx(stage, (s + 1) >> 2, (s + 1) & 3, mt);
                        }
                    }
                    else {
                        if constexpr ((ABL & 128) != 0) {
                            // The first fragment of the NEXT iteration reads the ring slot that was refilled at the end of
                            // the PREVIOUS iteration.  Unpinned, the scheduler hoists this VALU to the top of the
                            // iteration -- a few hundred cycles after the loads were issued -- and the waitcnt pass
                            // has to drain vmcnt(0) there (ISA: one full memory round trip per trip of the loop).
                            __builtin_amdgcn_sched_barrier(0);
                        }
#pragma unroll
                        for (int t = 0; t < NT; ++t) {
                            wfn[t] = dq((u + 1) % PF, 0, t, 0, live_next);
                        }
                    }
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) {
                            if constexpr (ABL & 2) {
                                acc[t][mt][0] += (float)wf[t][0] + (float)xf[mt][0];
                                asm volatile("" ::"v"(wf[t]), "v"(xf[mt]));
                            }
                            else {
                                acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[t], xf[mt], acc[t][mt], 0, 0, 0);
                            }
                        }
                    }
                    if (s + 1 < KS * 4) {
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) {
                            xf[mt] = xfn[mt];
                        }
                    }
                    (void)kk;
                    (void)j;
                }
                if constexpr (!(ABL & 8)) {
                    TM_STORE_X((u + 1) % PF, (u + 1) & 1);  // x(i+1), issued PF-1 iterations ago, -> the other LDS stage
                    TM_LOAD_X(u, i + PF);                   // refill slot u: x first, then w (see prologue)
                }
                if constexpr (!(ABL & 16)) {
                    TM_LOAD_W(u, i + PF);
                }
                __syncthreads();
            }
        }
    }
#undef TM_LOAD_W
#undef TM_LOAD_X
#undef TM_STORE_X
    if (p.dbg && tid == 0) {
        p.dbg[wgid * 8 + 2] = __builtin_amdgcn_s_memrealtime();
        p.dbg[wgid * 8 + 6] = __builtin_amdgcn_s_memtime();
    }


    // ---- add the WK k-phase partial sums through LDS (phase 0 keeps its own in registers) ---------------
    if constexpr (WK > 1) {
        floatx4* red = (floatx4*)smem;  // [(wk-1)][wn][t][mt][lane]; the staging buffers are dead after the last barrier
        if (wk > 0) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    red[((((wk - 1) * WN + wn) * NT + t) * MT + mt) * 64 + lane] = acc[t][mt];
                }
            }
        }
        __syncthreads();
        if (wk > 0) {
            return;
        }
#pragma unroll
        for (int ph = 1; ph < WK; ++ph) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    acc[t][mt] += red[((((ph - 1) * WN + wn) * NT + t) * MT + mt) * 64 + lane];
                }
            }
        }
    }

    // ---- epilogue: lane holds y[m = m0+16mt+i16][n = 16(nt0+t) + 4g + r], r = 0..3 ------------------
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (nt0 + t >= ntiles) {
            continue;
        }
        const int n = (nt0 + t) * 16 + g * 4;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            if (m0 + mt * 16 + i16 >= Mloc) {
                continue;
            }
            const int     m = row0 + m0 + mt * 16 + i16;  // flat output row
            const floatx4 a = acc[t][mt];
            if (p.epilogue == 2) {
                *(floatx4*)(p.partial + ((size_t)blockIdx.y * p.M + m) * p.N + n) = a;
            }
            else if (p.epilogue == 1) {
                const float s0 = a[0] / (1.0f + __builtin_expf(-a[0]));
                const float s1 = a[2] / (1.0f + __builtin_expf(-a[2]));
                half2_t     o  = {(half_t)(s0 * a[1]), (half_t)(s1 * a[3])};
                *(half2_t*)(p.y + (size_t)m * p.ldy + (n >> 1)) = o;
            }
            else {
                half4_t o = {(half_t)a[0], (half_t)a[1], (half_t)a[2], (half_t)a[3]};
                *(half4_t*)(p.y + (size_t)m * p.ldy + n) = o;
            }
        }
    }
    if (p.dbg && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        p.dbg[wgid * 8 + 3] = __builtin_amdgcn_s_memrealtime();
    }
}

// y = h(sum_s partial[s]) (optionally through the gated-SiLU epilogue): 4 columns per thread
__global__ __launch_bounds__(256) void splitk_reduce_kernel(half_t* __restrict__ y,
                                                            int ldy,
                                                            const float* __restrict__ partial,
                                                            int splits,
                                                            int M,
                                                            int N,
                                                            int gated)
{
    const size_t idx   = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)M * N / 4;
    if (idx >= total) {
        return;
    }
    const int m = idx / (N / 4);
    const int n = (idx - (size_t)m * (N / 4)) * 4;
    floatx4   a = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < splits; ++s) {
        a += *(const floatx4*)(partial + ((size_t)s * M + m) * N + n);
    }
    if (gated) {
        const float s0 = a[0] / (1.0f + __builtin_expf(-a[0]));
        const float s1 = a[2] / (1.0f + __builtin_expf(-a[2]));
        half2_t     o  = {(half_t)(s0 * a[1]), (half_t)(s1 * a[3])};
        *(half2_t*)(y + (size_t)m * ldy + (n >> 1)) = o;
    }
    else {
        half4_t o = {(half_t)a[0], (half_t)a[1], (half_t)a[2], (half_t)a[3]};
        *(half4_t*)(y + (size_t)m * ldy + n) = o;
    }
}

int launch_splitk_reduce(half_t* y, int ldy, const float* partial, int splits, int M, int N, bool gated, hipStream_t st)
{
    const size_t total = (size_t)M * N / 4;
    splitk_reduce_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(y, ldy, partial, splits, M, N, gated ? 1 : 0);
    TM_HIP_CHECK(hipGetLastError());
    return 0;
}

size_t gemm_workspace_bytes(int M, int N, int splits)
{
    return splits > 1 ? (size_t)splits * M * N * sizeof(float) : 0;
}

uint64_t* g_gemm_dbg = nullptr;  // set through tm_debug_set_gemm_trace (timing experiments)

// trace arena (tm_debug_trace_arena): see tm_kernels.h
static std::mutex               g_arena_mutex;
static uint64_t*                g_arena_base = nullptr;
static size_t                   g_arena_cap = 0, g_arena_used = 0;
static std::vector<TraceRecord> g_arena_recs;

bool trace_arena_active()
{
    return g_arena_base != nullptr;
}

uint64_t* trace_arena_alloc(size_t workgroups, const char* tag, int gx, int gy, int gz)
{
    std::lock_guard<std::mutex> lk(g_arena_mutex);
    if (!g_arena_base || g_arena_used + workgroups > g_arena_cap) {
        return nullptr;
    }
    TraceRecord r{};
    snprintf(r.tag, sizeof r.tag, "%s", tag ? tag : "");
    r.gx = gx, r.gy = gy, r.gz = gz;
    r.offset_wgs = g_arena_used;
    g_arena_recs.push_back(r);
    uint64_t* out = g_arena_base + g_arena_used * 8;
    g_arena_used += workgroups;
    return out;
}

void trace_arena_set(uint64_t* base, size_t cap_wgs)
{
    std::lock_guard<std::mutex> lk(g_arena_mutex);
    g_arena_base = base;
    g_arena_cap  = base ? cap_wgs : 0;
    g_arena_used = 0;
    g_arena_recs.clear();
}

// text: one line per traced launch, `index tag gx gy gz offset_wgs`; returns the bytes needed (incl. the terminator)
size_t trace_arena_records(char* out, size_t cap)
{
    std::lock_guard<std::mutex> lk(g_arena_mutex);
    std::string                 s;
    char                        line[128];
    for (size_t i = 0; i < g_arena_recs.size(); ++i) {
        const TraceRecord& r = g_arena_recs[i];
        snprintf(line, sizeof line, "%zu %s %d %d %d %zu\n", i, r.tag[0] ? r.tag : "-", r.gx, r.gy, r.gz, r.offset_wgs);
        s += line;
    }
    if (out && cap > 0) {
        snprintf(out, cap, "%s", s.c_str());
    }
    return s.size() + 1;
}

static int env_int(const char* name, int dflt)
{
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}

// ---- measured dispatch of the general kernel and of the grouped GEMMs (tm_kernels.h: gen_table_*) ----
static std::map<std::tuple<int, int, int, int, int>, std::array<int, 4>> g_gen_table;  // (kind, role, K, N, M bucket)
static std::mutex                                                      g_gen_mutex;

void gen_table_set(int kind, int role, int K, int N, int M, const int v[4])
{
    std::lock_guard<std::mutex> lk(g_gen_mutex);
    g_gen_table[std::make_tuple(kind, role, K, N, M)] = {v[0], v[1], v[2], v[3]};
}

bool gen_table_get(int kind, int role, int K, int N, int M, int v[4])
{
    std::lock_guard<std::mutex> lk(g_gen_mutex);
    auto                        it = g_gen_table.find(std::make_tuple(kind, role, K, N, M));
    if (it == g_gen_table.end() && role != 0) {
        it = g_gen_table.find(std::make_tuple(kind, 0, K, N, M));
    }
    if (it == g_gen_table.end()) {
        return false;
    }
    for (int i = 0; i < 4; ++i) {
        v[i] = it->second[i];
    }
    return true;
}

// the start-up tuner's candidate row tile of the grouped expert GEMMs: an override of THIS thread only (a transient entry
// in the process-wide table was picked up by other engines / rank threads launching MoE forwards during the timing window)
static thread_local int tl_grouped_rows = 0;
void gen_grouped_rows_override(int rows)
{
    tl_grouped_rows = rows;
}

void gen_table_clear()
{
    std::lock_guard<std::mutex> lk(g_gen_mutex);
    g_gen_table.clear();
}

void gen_table_erase(int kind, int role, int K, int N, int M)
{
    std::lock_guard<std::mutex> lk(g_gen_mutex);
    g_gen_table.erase(std::make_tuple(kind, role, K, N, M));
}

int gen_table_export_lines(FILE* f)
{
    std::lock_guard<std::mutex> lk(g_gen_mutex);
    for (const auto& kv : g_gen_table) {
        fprintf(f, "G %d %d %d %d %d %d %d %d %d\n", std::get<0>(kv.first), std::get<1>(kv.first), std::get<2>(kv.first),
                std::get<3>(kv.first), std::get<4>(kv.first), kv.second[0], kv.second[1], kv.second[2], kv.second[3]);
    }
    return (int)g_gen_table.size();
}

// a config the launchers accept for this kind (they clamp per weight type, so validity here = sane ranges)
static bool gen_entry_valid(int kind, int role, int K, int N, int M, const int v[4])
{
    if (role < 0 || role > 7 || K <= 0 || N <= 0 || M <= 0 || K % 128 != 0 || N % 16 != 0 || M != dec32_m_bucket(M)) {
        return false;
    }
    if (kind >= kGenDense && kind <= kGenDense + 2) {
        return (v[0] == 1 || v[0] == 2 || v[0] == 4) && v[1] >= 1 && v[1] <= 16 && (v[2] == 4 || v[2] == 8 || v[2] == 16)
               && (v[3] == 1 || v[3] == 2);
    }
    if (kind == kGenGrouped) {  // u4 experts: 16 / 32 / 64-row tiles (decode batches only)
        return (v[0] == 16 || v[0] == 32 || v[0] == 64) && M <= 64;
    }
    if (kind == kGenGrouped + 2) {  // e4m3 experts on the fp8 matrix cores: 32 / 64-row tiles
        return v[0] == 32 || v[0] == 64;
    }
    return false;
}

bool gen_table_import_line(const char* line)
{
    int kind, role, K, N, M, v[4];
    if (sscanf(line, "G %d %d %d %d %d %d %d %d %d", &kind, &role, &K, &N, &M, &v[0], &v[1], &v[2], &v[3]) != 9
        || !gen_entry_valid(kind, role, K, N, M, v)) {
        return false;
    }
    gen_table_set(kind, role, K, N, M, v);
    return true;
}

// every tiling of gemm_kernel worth timing for a dense linear that the P32 kernels do not serve.  fp16 (lm_head): 64- or
// 128-column workgroups; e4m3 weight-only: one or two tiles per wave; both with the split-K counts that keep >= 8 k-blocks per
// slice and whose slabs fit the workspace.  (u4 linears of this kernel -- N % 32 != 0 -- keep the heuristic: no model of
// BASELINE.json has one.)
int gen_dense_candidates(const LinearWeight& w, int M, size_t workspace_bytes, GemmConfig* out, int cap)
{
    if ((w.type != 1 && w.type != 2) || M > 256) {
        return 0;
    }
    const int KB = w.K / 128;
    int       n  = 0;
    for (int nt = 1; nt <= 2; ++nt) {
        for (int sp = 1; sp <= 8; sp *= 2) {
            if (sp > 1 && (KB % sp != 0 || KB / sp < 8 || gemm_workspace_bytes(M, w.N, sp) > workspace_bytes)) {
                continue;
            }
            if (n < cap) {
                out[n]         = GemmConfig{};
                out[n].nt      = nt;
                out[n].splits  = sp;
                out[n].waves   = w.type == 1 ? 4 : 8;
                out[n].kphases = 1;
                ++n;
            }
        }
    }
    return n;
}

int gen_grouped_candidates(const LinearWeight& proto, int m_cap, int* rows_out, int cap)
{
    int n = 0;
    if (proto.type == 0 && m_cap <= 64) {
        for (int r : {16, 32, 64}) {
            if (n < cap && (r == 16 || r / 2 < m_cap)) {
                rows_out[n++] = r;
            }
        }
    }
    else if (proto.type == 2) {
        for (int r : {32, 64}) {
            if (n < cap && (r == 32 || m_cap > 32)) {
                rows_out[n++] = r;
            }
        }
    }
    return n;
}

GemmConfig gemm_pick_config(const LinearWeight& w, int M)
{
    if (dec32_supported(w, M)) {  // decode batch: the weight-streaming kernel of gemm_decode.hip
        GemmConfig cfg{};
        dec32_pick_ex(w, M, &cfg.d32_shape, &cfg.splits, true);
        cfg.nt      = 2;
        cfg.waves   = 16;
        cfg.kphases = 1;
        return cfg;
    }
    return gemm_pick_config_general(w, M);
}

GemmConfig gemm_pick_config_general(const LinearWeight& w, int M)
{
    // Heuristic (measured on MI355X with tools/tune_gemm.py): the decode GEMMs are latency /
    // issue bound, so aim at ~256 workgroups of 8 waves.  TM_GEMM_NT / _SPLITS / _WAVES / _KPHASES override.
    GemmConfig cfg{};
    const int  ntiles = w.N / 16;
    const int  KB     = w.K / 128;
    const int  mblk   = (M + 63) / 64;
    int        tv[4];
    if (gen_table_get(kGenDense + w.type, w.role, w.K, w.N, dec32_m_bucket(M), tv)) {  // measured for this problem (tm_engine_tune_gemm)
        cfg.nt      = tv[0];
        cfg.splits  = tv[1];
        cfg.waves   = tv[2];
        cfg.kphases = tv[3];
        // the documented TM_GEMM_* overrides apply to measured entries as well (they silently stopped working once a
        // `G` line existed)
        cfg.nt      = env_int("TM_GEMM_NT", cfg.nt);
        cfg.splits  = env_int("TM_GEMM_SPLITS", cfg.splits);
        cfg.waves   = env_int("TM_GEMM_WAVES", cfg.waves);
        cfg.kphases = env_int("TM_GEMM_KPHASES", cfg.kphases);
        cfg.kstage  = env_int("TM_GEMM_KSTAGE", 0);
        cfg.splits  = cfg.splits > KB ? KB : cfg.splits;
        return cfg;
    }
    cfg.kphases       = 1;
    if (w.type == 1) {
        cfg.nt     = 2;
        cfg.waves  = 4;
        cfg.splits = 1;
    }
    else if (mblk > 1 && M > 256) {  // prefill: plenty of row blocks, maximise weight reuse per workgroup
        static const int pw = env_int("TM_GEMM_PREFILL_WAVES", 8);
        cfg.nt     = pw == 8 ? 2 : 4;
        cfg.waves  = pw == 8 ? 8 : 4;
        cfg.splits = 1;
        // mid-size M (a continuous-batching admission, a short prompt): 128-row x 256-column tiles give fewer than
        // 256 workgroups for the narrow linears (wo / w2 / w_qkv) -- split K until the chip is covered, >= 8
        // k-blocks per slice.  Measured (tools/tune_gemm.py --m 512 | 1024): w2 171.6 -> 71.8 us and wo 56.5 -> ~28 us
        // at M = 512, w2 182.8 -> 123.0 us at M = 1024; M >= 2048 never splits.
        static const int psplit = env_int("TM_GEMM_PREFILL_SPLIT", 1);
        if (psplit && w.type == 0) {
            const int wgs = (ntiles + 15) / 16 * ((M + 127) / 128);
            int       sp  = 1;
            while (wgs * sp * 2 <= 256 && KB / (sp * 2) >= 8 && KB % (sp * 2) == 0) {
                sp *= 2;
            }
            cfg.splits = sp;
        }
    }
    else {
        // split-K only until ~256 workgroups exist and never below 8 k-blocks per slice (slab traffic + reduce)
        // (64 < M <= 256, e.g. decode at batch 128: still weight-streaming bound -- 64-row blocks on grid.z, each block
        // re-reads the weights through L2 / the Infinity Cache, and the workgroup count includes the row blocks)
        cfg.waves = 8;
        cfg.nt    = 1;
        const int col_wgs = (ntiles + cfg.waves * cfg.nt - 1) / (cfg.waves * cfg.nt) * mblk;
        int       splits  = 1;
        static const int min_kb = env_int("TM_GEMM_MIN_KB", 8);
        while (col_wgs * splits * 2 <= 256 && KB / (splits * 2) >= min_kb && splits < 16) {
            splits *= 2;
        }
        cfg.splits = splits;
        // one row block (batch <= 64), K <= 8192: 4 column groups x 2 k-phases per workgroup (64 columns, the two phases
        // take alternate k-blocks and are summed through LDS) -- tools/tune_gemm.py --m 64 finds it 4-7 % ahead of 8 column
        // groups for w_qkv / wo / w1w3 in isolation; in the model (bench.py --steps 128, same box, back to back):
        // wo 11.6 -> 11.0 us, w_qkv 14.0 -> 13.8 us, w1w3 27.8 -> 27.6 us, step 3.733 -> 3.705 ms
        static const int dkp = env_int("TM_GEMM_DECODE_KP", 1);
        if (dkp && mblk == 1 && w.type == 0 && KB <= 64 && KB % 2 == 0) {
            cfg.kphases = 2;
            cfg.nt      = (ntiles + 3) / 4 > 256 ? 2 : 1;
            const int c2 = (ntiles + 4 * cfg.nt - 1) / (4 * cfg.nt);
            int       s2 = 1;
            while (c2 * s2 * 2 <= 256 && KB / (s2 * 2) >= min_kb && (KB / (s2 * 2)) % 2 == 0 && s2 < 16) {
                s2 *= 2;
            }
            cfg.splits = s2;
        }
        // 2+ row blocks (batch 65..256) and a wide N (w1w3): four-wave workgroups of 2 tiles per wave, no split -- >= 448
        // small workgroups, two per CU that run out of phase.  Measured with tools/tune_gemm.py --m 128 | 256 on the
        // Llama-3-8B / InternLM2-20B shapes: 78.3 -> 69.0 us, 54.5 -> 47.6 us (M = 128), 165.6 -> 131.9 us, 111.0 -> 88.5 us
        // (M = 256).  (Splitting K further for the long-K w2 was measured too and is slower.)
        static const int mid = env_int("TM_GEMM_MID_M", 1);
        if (mid && mblk > 1 && w.type == 0 && (ntiles + 7) / 8 * mblk >= 448 && splits == 1) {
            cfg.waves = 4;
            cfg.nt    = 2;
        }
    }
    cfg.nt      = env_int("TM_GEMM_NT", cfg.nt);
    cfg.splits  = env_int("TM_GEMM_SPLITS", cfg.splits);
    cfg.waves   = env_int("TM_GEMM_WAVES", cfg.waves);
    cfg.kphases = env_int("TM_GEMM_KPHASES", cfg.kphases);
    cfg.kstage  = env_int("TM_GEMM_KSTAGE", 0);
    if (cfg.splits > KB) {
        cfg.splits = KB;
    }
    return cfg;
}

template<int WT, int MT, int NT, int WN, int WK, int KS, int PF, int ABL = 0, bool GRP = false>
static int launch_one(const GemmParams& p, dim3 grid, hipStream_t st)
{
    constexpr int stage = 2 * KS * WK * 16 * MT * 256;
    constexpr int red   = (WK - 1) * WN * NT * MT * 1024;
    constexpr int lds_need = stage > red ? stage : red;
    // A grid of <= 256 eight-wave workgroups must land one per CU: the dispatcher otherwise co-locates two of them
    // on one CU while others idle (measured with tm_debug_set_gemm_trace: stragglers with 2x the loop time).
    // Requesting more than half of the 160 KB LDS makes co-residency impossible.
    static const int exclusive = env_int("TM_GEMM_EXCLUSIVE_CU", 1);
    const int lds = (exclusive && WN * WK >= 8 && lds_need < 84 * 1024 && grid.x * grid.y * grid.z <= 256) ? 84 * 1024 : lds_need;
    if (const int rc = ensure_dynamic_lds((const void*)gemm_kernel<WT, MT, NT, WN, WK, KS, PF, ABL, GRP>,
                                          lds_need > 84 * 1024 ? lds_need : 84 * 1024)) {
        return rc;
    }
    gemm_kernel<WT, MT, NT, WN, WK, KS, PF, ABL, GRP><<<grid, WN * WK * 64, lds, st>>>(p);
    TM_HIP_CHECK(hipGetLastError());
    return 0;
}

// u4, 8 waves, 1 tile per wave: the decode work-horse, instantiated for every stage depth
template<int MT, int WN, int WK>
static int launch_u4_nt1(const GemmParams& p, dim3 grid, int ks, hipStream_t st)
{
    if constexpr (WK == 1) {
      if (ks == 4) {
#ifdef TM_EXPERIMENTS  // timing-ablation instantiations (results are garbage by design): not in the shipped library
        if constexpr (MT == 4) {
            switch (env_int("TM_GEMM_ABL", 0)) {  // ablation timing experiments (round 1)
                case 1: return launch_one<0, MT, 1, WN, WK, 4, 2, 1>(p, grid, st);
                case 2: return launch_one<0, MT, 1, WN, WK, 4, 2, 2>(p, grid, st);
                case 4: return launch_one<0, MT, 1, WN, WK, 4, 2, 4>(p, grid, st);
                case 8: return launch_one<0, MT, 1, WN, WK, 4, 2, 8>(p, grid, st);
                case 16: return launch_one<0, MT, 1, WN, WK, 4, 2, 16>(p, grid, st);
                case 7: return launch_one<0, MT, 1, WN, WK, 4, 2, 7>(p, grid, st);
                case 32: return launch_one<0, MT, 1, WN, WK, 4, 2, 32>(p, grid, st);
                case 64: return launch_one<0, MT, 1, WN, WK, 4, 2, 64>(p, grid, st);
                case 96: return launch_one<0, MT, 1, WN, WK, 4, 2, 96>(p, grid, st);
                case 24: return launch_one<0, MT, 1, WN, WK, 4, 2, 24>(p, grid, st);
                case 3: return launch_one<0, MT, 1, WN, WK, 4, 2, 3>(p, grid, st);
                case 12: return launch_one<0, MT, 1, WN, WK, 4, 2, 12>(p, grid, st);
                case 15: return launch_one<0, MT, 1, WN, WK, 4, 2, 15>(p, grid, st);
                case 31: return launch_one<0, MT, 1, WN, WK, 4, 2, 31>(p, grid, st);
                default: break;
            }
        }
#endif
        return launch_one<0, MT, 1, WN, WK, 4, 2>(p, grid, st);
      }
    }
    if (ks >= 2) return launch_one<0, MT, 1, WN, WK, 2, 4>(p, grid, st);
    static const int pf = env_int("TM_GEMM_PF", 8);  // ring depth experiment (4 | 8)
    return (p.kb_per_split / WK >= 16 && pf >= 8) ? launch_one<0, MT, 1, WN, WK, 1, 8>(p, grid, st) :
                                       launch_one<0, MT, 1, WN, WK, 1, 4>(p, grid, st);
}

template<int WT, int MT>
static int launch_mt(const GemmParams& p, dim3 grid, int nt, int waves, int wk, int ks, hipStream_t st)
{
    if constexpr (WT == 1) {
        if (nt == 1) return launch_one<1, MT, 1, 4, 1, 1, 4>(p, grid, st);
        return launch_one<1, MT, 2, 4, 1, 1, 2>(p, grid, st);
    }
    else if constexpr (WT == 2) {  // fp8: 8 waves, one tile per wave (decode) or two (prefill rows)
        if (nt == 1) return launch_one<2, MT, 1, 8, 1, 1, 4>(p, grid, st);
        return launch_one<2, MT, 2, 8, 1, 1, 2>(p, grid, st);
    }
    else {
        if (waves == 16) {  // 8 column groups x 2 k-phases: 4 waves per SIMD, same activation traffic per CU as 8x1
            return p.kb_per_split / 2 >= 16 ? launch_one<0, MT, 1, 8, 2, 1, 8>(p, grid, st) :
                                              launch_one<0, MT, 1, 8, 2, 1, 4>(p, grid, st);
        }
        if (waves == 8 && wk == 2) {
            if (nt == 1) return launch_u4_nt1<MT, 4, 2>(p, grid, ks >= 2 ? 2 : 1, st);
            return ks >= 2 ? launch_one<0, MT, 2, 4, 2, 2, 2>(p, grid, st) : launch_one<0, MT, 2, 4, 2, 1, 4>(p, grid, st);
        }
        if (waves == 8) {
            if (nt == 1) return launch_u4_nt1<MT, 8, 1>(p, grid, ks, st);
            return ks >= 2 ? launch_one<0, MT, 2, 8, 1, 2, 2>(p, grid, st) : launch_one<0, MT, 2, 8, 1, 1, 4>(p, grid, st);
        }
        if (nt == 1) return ks >= 2 ? launch_one<0, MT, 1, 4, 1, 2, 4>(p, grid, st) : launch_one<0, MT, 1, 4, 1, 1, 8>(p, grid, st);
        if (nt == 2) return ks >= 2 ? launch_one<0, MT, 2, 4, 1, 2, 2>(p, grid, st) : launch_one<0, MT, 2, 4, 1, 1, 4>(p, grid, st);
        return launch_one<0, MT, 4, 4, 1, 1, 4>(p, grid, st);
    }
}

static int KB_of(const LinearWeight& w)
{
    return w.K / 128;
}

int launch_linear(const LinearWeight& w,
                  const half_t*       x,
                  int                 ldx,
                  half_t*             y,
                  int                 ldy,
                  int                 M,
                  bool                gated_silu,
                  GemmConfig          cfg,
                  float*              workspace,
                  bool                defer_reduce,
                  int*                slabs,
                  hipStream_t         st)
{
    if (slabs) {
        *slabs = 1;
    }
    TM_REQUIRE(w.packed != nullptr || w.packed32 != nullptr, "linear weight not prepared");
    TM_REQUIRE(ldx % 8 == 0, "x rows must be 16-byte aligned");
    TM_REQUIRE(!gated_silu || w.N % 32 == 0, "gated epilogue needs N % 32 == 0");
    if (M == 0) {
        return 0;
    }
    if (cfg.d32_shape >= 0 && dec32_supported(w, M)) {
        int       nslab = 1;
        const int sp    = workspace ? (cfg.splits < 1 ? 1 : cfg.splits) : 1;
        TM_REQUIRE(!defer_reduce || sp > 1, "defer_reduce only with split-K");
        NormFold  tk{};  // kShapeMerge shapes: the arrival counters ride in a NormFold that neither produces nor consumes
        tk.tickets = cfg.tickets;
        int shape  = cfg.d32_shape;
        if (dec32_is_merge_shape(shape) && (!cfg.tickets || M > 64 || defer_reduce)) {
            shape -= kShapeMerge;  // no counters / the caller wants the slabs: the plain form of the same tile
        }
        const int rc = launch_linear_dec32(w, x, ldx, y, ldy, M, gated_silu, shape, sp, workspace, &nslab, st, cfg.tickets ? &tk : nullptr);
        if (rc) {
            return rc;
        }
        if (nslab > 1 && !defer_reduce) {
            const size_t total = (size_t)M * w.N / 4;
            splitk_reduce_kernel<<<(total + 255) / 256, 256, 0, st>>>(y, ldy, workspace, nslab, M, w.N, gated_silu ? 1 : 0);
            TM_HIP_CHECK(hipGetLastError());
        }
        if (slabs) {
            *slabs = nslab;
        }
        return 0;
    }
    TM_REQUIRE(w.packed != nullptr, "this linear holds only the decode kernels' image (prepared p32_only)");
    int nt    = cfg.nt;
    int waves = cfg.waves == 16 ? 16 : (cfg.waves == 8 ? 8 : 4);
    int wk    = cfg.kphases == 2 || waves == 16 ? 2 : 1;
    if (w.type == 1) {
        waves = 4;
        wk    = 1;
        nt    = nt > 2 ? 2 : nt;
    }
    if (w.type == 2) {  // fp8: 8 waves x (1 | 2) tiles, no in-workgroup k split
        waves = 8;
        wk    = 1;
        nt    = nt > 1 ? 2 : 1;
    }
    if (waves == 4) {
        wk = 1;
    }
    if (waves == 8 && nt > 2) {
        nt = 2;
    }
    if (waves == 16) {
        nt = 1;
        if (KB_of(w) % 2 != 0) {
            waves = 8;
            wk    = 1;
        }
    }
    TM_REQUIRE(nt == 1 || nt == 2 || nt == 4, "nt in {1,2,4}");
    const int KB     = w.K / 128;
    int       mt     = M <= 16 ? 1 : (M <= 32 ? 2 : 4);
    // prefill (many row blocks): 128-row workgroup tiles amortise each dequantised weight fragment over 8 MFMAs
    // (one wave per SIMD, accumulators in AGPRs) -- with 64-row tiles the dequant VALU work equals the MFMA time
    static const int mt_prefill = env_int("TM_GEMM_MT_PREFILL", 8);
    if (w.type == 0 && ((waves == 4 && nt == 4) || (waves == 8 && nt == 2 && wk == 1)) && M >= 128 && mt_prefill == 8) {
        mt = 8;  // (a 256-row tile spills: 256 accumulator registers + fragments exceed the 512-register file)
    }
    int       splits = cfg.splits < 1 ? 1 : cfg.splits;
    if (splits > KB) {
        splits = KB;
    }
    TM_REQUIRE(splits == 1 || workspace != nullptr, "split-K needs a workspace");
    TM_REQUIRE(!defer_reduce || splits > 1, "defer_reduce only with split-K");
    if (KB % wk != 0) {
        wk = 1;
    }

    GemmParams p{};
    p.x        = x;
    p.ldx      = ldx;
    p.wq       = (const u32x4*)w.packed;
    p.sz       = w.sz;
    p.y        = y;
    p.ldy      = ldy;
    p.partial  = workspace;
    p.M        = M;
    p.N        = w.N;
    p.K        = w.K;
    p.KB       = KB;
    p.rotate_k = 0;
    p.dbg      = nullptr;  // set below, once the grid is known
    // k-blocks per grid.y slice: whole iterations of ks*wk k-blocks, for every slice including the last one.
    // ks (k-blocks per barrier) = the largest of {4, 2, 1} (capped by cfg.kstage) that divides the slice.
    // measured (round 1 ablations): more k-blocks per barrier does NOT pay (the loop is issue-bound, not
    // barrier-bound), so the default is 1; TM_GEMM_KSTAGE=2|4 keeps the experiment reachable.
    int ks_cap = cfg.kstage > 0 ? cfg.kstage : 1;
    if (w.type != 0 || (waves == 4 && nt == 4) || waves == 16) {
        ks_cap = 1;
    }
    else if (wk == 2 && ks_cap > 2) {
        ks_cap = 2;
    }
    int ks = 1;
    for (int cand = ks_cap; cand >= 1; cand >>= 1) {
        const int unit = cand * wk;
        int       per  = (KB + splits - 1) / splits;
        per            = (per + unit - 1) / unit * unit;
        if (KB % unit == 0 && per <= KB) {
            ks             = cand;
            p.kb_per_split = per;
            break;
        }
    }
    if (p.kb_per_split == 0) {
        p.kb_per_split = (KB + splits - 1) / splits;
    }
    splits     = (KB + p.kb_per_split - 1) / p.kb_per_split;  // no empty splits
    p.epilogue = splits > 1 ? 2 : (gated_silu ? 1 : 0);

    const int wn     = waves / wk;
    const int ntiles = w.N / 16;
    dim3      grid((ntiles + wn * nt - 1) / (wn * nt), splits, (M + 16 * mt - 1) / (16 * mt));
    p.dbg        = gemm_trace_for((size_t)grid.x * grid.y * grid.z, w.role == 5 ? "lm_head" : "gemm_general", grid.x, grid.y, grid.z);
    int       rc = 0;
    if (w.type == 0 && mt == 8) {
        static const int xpin = env_int("TM_GEMM_XPIN", 1);  // pinned next-iteration dequant (see the main loop): +2..6 % at M = 8192
        static const int pks = env_int("TM_GEMM_PREFILL_KS", 1);  // k-blocks per LDS stage (= per barrier)
        if (waves == 8 && pks == 2 && p.kb_per_split % 2 == 0) {
            rc = launch_one<0, 8, 2, 8, 1, 2, 2, 128>(p, grid, st);
        }
        else {
            rc = waves == 8 ? (xpin ? launch_one<0, 8, 2, 8, 1, 1, 2, 128>(p, grid, st) : launch_one<0, 8, 2, 8, 1, 1, 2>(p, grid, st)) :
                              launch_one<0, 8, 4, 4, 1, 1, 2>(p, grid, st);
        }
    }
    else if (w.type == 0) {
        rc = mt == 1 ? launch_mt<0, 1>(p, grid, nt, waves, wk, ks, st) :
             mt == 2 ? launch_mt<0, 2>(p, grid, nt, waves, wk, ks, st) :
                       launch_mt<0, 4>(p, grid, nt, waves, wk, ks, st);
    }
    else if (w.type == 2) {
        rc = mt == 1 ? launch_mt<2, 1>(p, grid, nt, waves, wk, ks, st) :
             mt == 2 ? launch_mt<2, 2>(p, grid, nt, waves, wk, ks, st) :
                       launch_mt<2, 4>(p, grid, nt, waves, wk, ks, st);
    }
    else {
        rc = mt == 1 ? launch_mt<1, 1>(p, grid, nt, waves, wk, ks, st) :
             mt == 2 ? launch_mt<1, 2>(p, grid, nt, waves, wk, ks, st) :
                       launch_mt<1, 4>(p, grid, nt, waves, wk, ks, st);
    }
    if (rc) {
        return rc;
    }
    if (splits > 1 && !defer_reduce) {
        const size_t total = (size_t)M * w.N / 4;
        splitk_reduce_kernel<<<(total + 255) / 256, 256, 0, st>>>(y, ldy, workspace, splits, M, w.N, gated_silu ? 1 : 0);
        TM_HIP_CHECK(hipGetLastError());
    }
    if (slabs) {
        *slabs = splits;  // number of fp32 slabs written (1 = direct epilogue, nothing in the workspace)
    }
    return 0;
}

// Grouped GEMM over experts (reference: LlamaLinear::Forward with idxs / offsets, models/llama/LlamaLinear.cu:67-127):
// y[f] = x[row_idx ? row_idx[f] : f] . W_e for the flat rows f of expert e.  All experts share shape and format; the
// descriptors (device) are built once by moe_build_groups.  m_cap = upper bound of rows per expert (tokens).
int moe_build_groups(void** d_groups, const LinearWeight* experts, int E, hipStream_t st)
{
    std::vector<GemmGroup> h(E);
    for (int e = 0; e < E; ++e) {
        TM_REQUIRE(experts[e].packed != nullptr, "expert weight not prepared");
        h[e] = GemmGroup{experts[e].packed, experts[e].sz};
    }
    if (!*d_groups) {
        TM_HIP_CHECK(hipMalloc(d_groups, sizeof(GemmGroup) * E));
    }
    TM_HIP_CHECK(hipMemcpyAsync(*d_groups, h.data(), sizeof(GemmGroup) * E, hipMemcpyHostToDevice, st));
    TM_HIP_CHECK(hipStreamSynchronize(st));
    return 0;
}

int launch_linear_grouped(const LinearWeight& proto, const void* d_groups, int E, const half_t* x, int ldx, int x_rows,
                          half_t* y, int ldy, int m_cap, int m_hint, bool gated_silu, const int* seg, const int* row_idx,
                          hipStream_t st)
{
    TM_REQUIRE(proto.type == 0 || proto.type == 2, "grouped GEMM: u4 or fp8 expert weights");
    TM_REQUIRE(ldx % 8 == 0 && (!gated_silu || proto.N % 32 == 0), "grouped GEMM: alignment");
    if (m_cap == 0 || E == 0) {
        return 0;
    }
    GemmParams p{};
    p.x            = x;
    p.ldx          = ldx;
    p.y            = y;
    p.ldy          = ldy;
    p.M            = m_cap;
    p.N            = proto.N;
    p.K            = proto.K;
    p.KB           = proto.K / 128;
    p.kb_per_split = p.KB;
    p.epilogue     = gated_silu ? 1 : 0;
    p.dbg          = nullptr;
    p.groups       = (const GemmGroup*)d_groups;
    p.seg          = seg;
    p.row_idx      = row_idx;
    p.x_rows       = x_rows;
    const int ntiles = proto.N / 16;
    int       rc     = 0;
    if (m_cap <= 64) {
        // decode: an expert sees at most `tokens` rows but typically tokens * top_k / experts (m_hint): the row tile is
        // sized for twice that, the (rare) overflow goes to further row blocks -- a 64-row tile for ~16 real rows would
        // spend 3/4 of the MFMA and LDS work on clamped duplicates
        const int want = std::min(m_cap, std::max(1, 2 * m_hint));
        int       mt   = want <= 16 ? 1 : (want <= 32 ? 2 : 4);
        int       tv[4];
        if (tl_grouped_rows > 0) {
            mt = tl_grouped_rows / 16;  // the tuner's candidate on this thread
        }
        else if (gen_table_get(kGenGrouped + proto.type, 0, proto.K, proto.N, dec32_m_bucket(m_cap), tv)) {
            mt = tv[0] / 16;  // measured (tm_engine_tune_gemm): 16 / 32 / 64-row tiles
        }
        p.zper         = (m_cap + 16 * mt - 1) / (16 * mt);
        dim3 grid((ntiles + 7) / 8, 1, E * p.zper);
        if (proto.type == 0) {
            rc = mt == 1 ? launch_one<0, 1, 1, 8, 1, 1, 4, 0, true>(p, grid, st) : mt == 2 ? launch_one<0, 2, 1, 8, 1, 1, 4, 0, true>(p, grid, st) :
                                                                                    launch_one<0, 4, 1, 8, 1, 1, 4, 0, true>(p, grid, st);
        }
        else {
            rc = mt == 1 ? launch_one<2, 1, 1, 8, 1, 1, 4, 0, true>(p, grid, st) : mt == 2 ? launch_one<2, 2, 1, 8, 1, 1, 4, 0, true>(p, grid, st) :
                                                                                    launch_one<2, 4, 1, 8, 1, 1, 4, 0, true>(p, grid, st);
        }
    }
    else {  // prefill: 64-row blocks x 2 tiles per wave; blocks past an expert's segment exit at once
        p.zper = (m_cap + 63) / 64;
        dim3 grid((ntiles + 15) / 16, 1, E * p.zper);
        rc = proto.type == 0 ? launch_one<0, 4, 2, 8, 1, 1, 4, 0, true>(p, grid, st) : launch_one<2, 4, 2, 8, 1, 1, 2, 0, true>(p, grid, st);
    }
    return rc;
}

}  // namespace tmk
