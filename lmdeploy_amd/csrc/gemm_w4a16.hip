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

void linear_weight_free(LinearWeight& w)
{
    if (w.packed) {
        (void)hipFree(w.packed);
    }
    if (w.sz) {
        (void)hipFree(w.sz);
    }
    w.packed = nullptr;
    w.sz     = nullptr;
}

int linear_weight_prepare_u4(LinearWeight& w, const int32_t* qweight, const half_t* scales, const half_t* zeros,
                             hipStream_t st)
{
    TM_REQUIRE(w.group == 128, "AWQ group size must be 128 (lmdeploy/turbomind/converter.py:86-92)");
    TM_REQUIRE(w.K % 128 == 0 && w.N % 16 == 0, "K % 128 == 0 and N % 16 == 0");
    w.type         = 0;
    w.packed_bytes = (size_t)w.K * w.N / 2;
    w.sz_bytes     = (size_t)(w.K / 128) * w.N * 4;
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

// One workgroup = WAVES waves x NT column tiles (16 wide) x MT row tiles (16 tall) over a slice of K.
// K advances one k-block (128 = one quantisation group = one 16-B lane load per tile) per iteration:
//   * weights: per-wave register ring, PF k-blocks deep (PF KiB per tile in flight per wave) -- the HBM stream
//     is never waited on for less than PF iterations;
//   * activations: [MB][128] fp16 per k-block through a double-buffered, XOR-swizzled LDS tile shared by all
//     waves (global -> registers two iterations ahead -> LDS one iteration ahead), ONE barrier per iteration.
template<int WT, int MT, int NT, int WAVES, int PF>
__global__ __launch_bounds__(WAVES * 64) void gemm_kernel(GemmParams p)
{
    static_assert(PF % 2 == 0, "ring depth must be even (x register sets alternate)");
    constexpr int MB      = 16 * MT;
    constexpr int THREADS = WAVES * 64;
    constexpr int ROWB    = 256;            // one k-block of one row
    constexpr int BUFB    = MB * ROWB;      // one LDS stage
    constexpr int NCHUNK  = MB * 16;        // 16-B chunks per stage
    constexpr int XR      = (NCHUNK + THREADS - 1) / THREADS;
    constexpr int WV      = WT == 0 ? 1 : 4;  // u32x4 per (tile, k-block) per lane
    constexpr bool XFULL  = NCHUNK % THREADS == 0;  // every thread stages exactly XR chunks

    extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 * BUFB

    const int tid  = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16  = lane & 15;
    const int g    = lane >> 4;

    const int ntiles = p.N / 16;
    const int nt0    = (blockIdx.x * WAVES + wave) * NT;
    const int m0     = blockIdx.z * MB;
    const int kb0    = blockIdx.y * p.kb_per_split;
    const int nkb    = min(p.kb_per_split, p.KB - kb0);

    // Buffer descriptors (SRD) + per-lane 32-bit byte offsets + SCALAR k-block offsets: the address math of every
    // load in the loop is SALU-only (raw pointers cost ~10 VALU per load in 64-bit adds -- the loop is VALU-bound).
    // Tiles past the edge are clamped: loads stay in bounds, stores are skipped.
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.wq, 0, (int)((size_t)p.KB * ntiles * 1024 * WV), 0x00020000);
    const auto rs_s = __builtin_amdgcn_make_buffer_rsrc((void*)p.sz, 0, WT == 0 ? p.KB * ntiles * 64 : 0, 0x00020000);
    const auto rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)(((size_t)(p.M - 1) * p.ldx + p.K) * 2), 0x00020000);
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

    u32x4    ring[PF][NT][WV];
    uint32_t sring[PF][NT];
    u32x4    xs[PF][XR];

    // x chunk q of a k-block: row q/16, 16-B chunk q%16.  Loads are UNCONDITIONAL (clamped row / chunk): a
    // per-lane "load or zero" select makes hipcc branch around every load and drain vmcnt(0) each time.
    // Rows past M only feed output rows that are never stored.
    int xoff[XR];
    int xlds[XR];
#pragma unroll
    for (int r = 0; r < XR; ++r) {
        const int q  = tid + THREADS * r;
        const int qc = min(q, NCHUNK - 1);
        const int m  = qc >> 4;
        const int ci = qc & 15;
        xoff[r]      = (min(m0 + m, p.M - 1) * p.ldx + ci * 8) * 2;
        xlds[r]      = q < NCHUNK ? m * ROWB + ((ci ^ (m & 15)) << 4) : -1;
    }
    uint32_t m1024 = 0x64006400u, m64 = 0x54005400u;
    asm volatile("" : "+v"(m1024), "+v"(m64));  // keep the magic numbers in VGPRs (see dequant8)
    const int last = nkb - 1;  // k-block indices are clamped to `last`: the tail re-loads harmlessly

#define TM_LOAD_W(slot, i)                                                                                   \
    {                                                                                                        \
        const int kb_ = kb0 + min((i), last);                                                                \
        _Pragma("unroll") for (int t = 0; t < NT; ++t)                                                       \
        {                                                                                                    \
            _Pragma("unroll") for (int v = 0; v < WV; ++v)                                                   \
            {                                                                                                \
                ring[slot][t][v] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, woff[t] + v * 1024,           \
                                                                         kb_ * wstride, /*nt*/ 2);           \
            }                                                                                                \
            if constexpr (WT == 0) {                                                                         \
                sring[slot][t] = __builtin_amdgcn_raw_buffer_load_b32(rs_s, soff[t], kb_ * sstride, 0);      \
            }                                                                                                \
        }                                                                                                    \
    }
#define TM_LOAD_X(set, i)                                                                                    \
    {                                                                                                        \
        const int kb_ = kb0 + min((i), last);                                                                \
        _Pragma("unroll") for (int r = 0; r < XR; ++r)                                                       \
        {                                                                                                    \
            xs[set][r] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, xoff[r], kb_ * 256, 0);                 \
        }                                                                                                    \
    }
#define TM_STORE_X(set, buf)                                                                                 \
    _Pragma("unroll") for (int r = 0; r < XR; ++r)                                                           \
    {                                                                                                        \
        if (XFULL || xlds[r] >= 0) {                                                                         \
            *(u32x4*)(smem + (buf)*BUFB + xlds[r]) = xs[set][r];                                             \
        }                                                                                                    \
    }

    if (nkb > 0) {
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

        // ---- main loop: branch-free body, statically unrolled over the ring --------------------------
        // Iterations past nkb (ring padding) contract zeroed weights, so they add exactly 0.
        for (int base = 0; base < nkb; base += PF) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int  i    = base + u;
                const bool live = i < nkb;  // wave-uniform
                const char* xb = smem + (u & 1) * BUFB;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    half8_t xf[MT];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        xf[mt] = *(const half8_t*)(xb + (mt * 16 + i16) * ROWB + (((j * 4 + g) ^ i16) << 4));
                    }
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        half8_t wf;
                        if constexpr (WT == 0) {
                            const half2_t pr = bit_cast<half2_t>(live ? sring[u][t] : 0u);  // (s, -z*s) = 0 -> w = 0
                            wf               = dequant8(ring[u][t][0][j], half2_t{pr[0], pr[0]}, half2_t{pr[1], pr[1]}, m1024, m64);
                        }
                        else {
                            const u32x4 wv = ring[u][t][j];
                            wf = bit_cast<half8_t>(live ? wv : u32x4{0u, 0u, 0u, 0u});
                        }
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) {
                            acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, xf[mt], acc[t][mt], 0, 0, 0);
                        }
                    }
                }
                TM_STORE_X((u + 1) % PF, (u + 1) & 1);  // x(i+1), issued PF-1 iterations ago, -> the other LDS stage
                TM_LOAD_X(u, i + PF);                   // refill slot u: x first, then w (see prologue)
                TM_LOAD_W(u, i + PF);
                __syncthreads();
            }
        }
    }
#undef TM_LOAD_W
#undef TM_LOAD_X
#undef TM_STORE_X

    // ---- epilogue: lane holds y[m = m0+16mt+i16][n = 16(nt0+t) + 4g + r], r = 0..3 ------------------
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (nt0 + t >= ntiles) {
            continue;
        }
        const int n = (nt0 + t) * 16 + g * 4;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = m0 + mt * 16 + i16;
            if (m >= p.M) {
                continue;
            }
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

size_t gemm_workspace_bytes(int M, int N, int splits)
{
    return splits > 1 ? (size_t)splits * M * N * sizeof(float) : 0;
}

static int env_int(const char* name, int dflt)
{
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}

GemmConfig gemm_pick_config(const LinearWeight& w, int M)
{
    // Heuristic (measured on MI355X with tools/tune_gemm.py): the decode GEMMs are latency /
    // occupancy bound, so aim at ~256..512 workgroups with the widest column tile that still gets there.
    // TM_GEMM_NT / TM_GEMM_SPLITS / TM_GEMM_WAVES override.
    GemmConfig cfg{};
    const int  ntiles = w.N / 16;
    const int  KB     = w.K / 128;
    const int  mblk   = (M + 63) / 64;
    if (w.type == 1) {
        cfg.nt     = 2;
        cfg.waves  = 4;
        cfg.splits = 1;
    }
    else if (mblk > 1) {  // prefill: plenty of row blocks, maximise weight reuse per workgroup
        cfg.nt     = 4;
        cfg.waves  = 4;
        cfg.splits = 1;
    }
    else {
        // measured (tools/tune_gemm.py, Llama-3-8B decode shapes, M=64): 8 waves x 1 tile per wave wins everywhere;
        // split-K only until ~256 workgroups exist and never below 8 k-blocks per slice (slab traffic + reduce).
        cfg.waves = 8;
        cfg.nt    = 1;
        const int col_wgs = (ntiles + cfg.waves * cfg.nt - 1) / (cfg.waves * cfg.nt);
        int       splits  = 1;
        while (col_wgs * splits * 2 <= 256 && KB / (splits * 2) >= 8 && splits < 16) {
            splits *= 2;
        }
        cfg.splits = splits;
    }
    cfg.nt     = env_int("TM_GEMM_NT", cfg.nt);
    cfg.splits = env_int("TM_GEMM_SPLITS", cfg.splits);
    cfg.waves  = env_int("TM_GEMM_WAVES", cfg.waves);
    if (cfg.splits > KB) {
        cfg.splits = KB;
    }
    return cfg;
}

template<int WT, int MT, int NT, int WAVES, int PF>
static int launch_one(const GemmParams& p, dim3 grid, hipStream_t st)
{
    constexpr int lds = 2 * 16 * MT * 256;
    gemm_kernel<WT, MT, NT, WAVES, PF><<<grid, WAVES * 64, lds, st>>>(p);
    TM_HIP_CHECK(hipGetLastError());
    return 0;
}

template<int WT, int MT>
static int launch_mt(const GemmParams& p, dim3 grid, int nt, int waves, hipStream_t st)
{
    // ring depth: 8 k-blocks when the K slice is long enough to use it, else 4 (padding iterations are wasted work)
    const bool deep = p.kb_per_split >= 16;
    if constexpr (WT == 1) {
        if (nt == 1) return launch_one<1, MT, 1, 4, 4>(p, grid, st);
        return launch_one<1, MT, 2, 4, 2>(p, grid, st);
    }
    else {
        if (waves == 8) {
            if (nt == 1) return deep ? launch_one<0, MT, 1, 8, 8>(p, grid, st) : launch_one<0, MT, 1, 8, 4>(p, grid, st);
            return deep ? launch_one<0, MT, 2, 8, 8>(p, grid, st) : launch_one<0, MT, 2, 8, 4>(p, grid, st);
        }
        if (nt == 1) return deep ? launch_one<0, MT, 1, 4, 8>(p, grid, st) : launch_one<0, MT, 1, 4, 4>(p, grid, st);
        if (nt == 2) return deep ? launch_one<0, MT, 2, 4, 8>(p, grid, st) : launch_one<0, MT, 2, 4, 4>(p, grid, st);
        return launch_one<0, MT, 4, 4, 4>(p, grid, st);
    }
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
    TM_REQUIRE(w.packed != nullptr, "linear weight not prepared");
    TM_REQUIRE(ldx % 8 == 0, "x rows must be 16-byte aligned");
    TM_REQUIRE(!gated_silu || w.N % 32 == 0, "gated epilogue needs N % 32 == 0");
    if (M == 0) {
        return 0;
    }
    int nt    = cfg.nt;
    int waves = cfg.waves == 8 ? 8 : 4;
    if (w.type == 1) {
        waves = 4;
        nt    = nt > 2 ? 2 : nt;
    }
    if (waves == 8 && nt > 2) {
        nt = 2;
    }
    TM_REQUIRE(nt == 1 || nt == 2 || nt == 4, "nt in {1,2,4}");
    const int KB     = w.K / 128;
    const int mt     = M <= 16 ? 1 : (M <= 32 ? 2 : 4);
    int       splits = cfg.splits < 1 ? 1 : cfg.splits;
    if (splits > KB) {
        splits = KB;
    }
    TM_REQUIRE(splits == 1 || workspace != nullptr, "split-K needs a workspace");
    TM_REQUIRE(!defer_reduce || splits > 1, "defer_reduce only with split-K");

    GemmParams p{};
    p.x            = x;
    p.ldx          = ldx;
    p.wq           = (const u32x4*)w.packed;
    p.sz           = w.sz;
    p.y            = y;
    p.ldy          = ldy;
    p.partial      = workspace;
    p.M            = M;
    p.N            = w.N;
    p.K            = w.K;
    p.KB           = KB;
    p.kb_per_split = (KB + splits - 1) / splits;
    splits         = (KB + p.kb_per_split - 1) / p.kb_per_split;  // no empty splits
    p.epilogue     = splits > 1 ? 2 : (gated_silu ? 1 : 0);

    const int ntiles = w.N / 16;
    dim3      grid((ntiles + waves * nt - 1) / (waves * nt), splits, (M + 16 * mt - 1) / (16 * mt));
    int       rc = 0;
    if (w.type == 0) {
        rc = mt == 1 ? launch_mt<0, 1>(p, grid, nt, waves, st) :
             mt == 2 ? launch_mt<0, 2>(p, grid, nt, waves, st) :
                       launch_mt<0, 4>(p, grid, nt, waves, st);
    }
    else {
        rc = mt == 1 ? launch_mt<1, 1>(p, grid, nt, waves, st) :
             mt == 2 ? launch_mt<1, 2>(p, grid, nt, waves, st) :
                       launch_mt<1, 4>(p, grid, nt, waves, st);
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

}  // namespace tmk
