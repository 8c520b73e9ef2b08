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
    // one thread per output dword: idx = ((nt*KB + kb)*64 + lane)*4 + j
    const size_t idx   = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)K * N / 8;
    if (idx >= total) {
        return;
    }
    const int    KB   = K / 128;
    const int    j    = idx & 3;
    const int    lane = (idx >> 2) & 63;
    const size_t tile = idx >> 8;
    const int    kb   = tile % KB;
    const int    nt   = tile / KB;
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
    // idx = (nt*KB + kb)*16 + i
    const size_t idx   = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)KB * N;
    if (idx >= total) {
        return;
    }
    const int    i    = idx & 15;
    const size_t tile = idx >> 4;
    const int    kb   = tile % KB;
    const int    nt   = tile / KB;
    const int    n    = nt * 16 + i;
    const half_t s    = scales[(size_t)kb * N + n];
    const half_t z    = zeros[(size_t)kb * N + n];
    const half_t zs   = (-z) * s;  // one fp16 rounding (cast.cu:151-156)
    half2_t      pr   = {s, zs};
    out[idx]          = bit_cast<uint32_t>(pr);
}

__global__ void repack_f16_kernel(half_t* __restrict__ out, const half_t* __restrict__ w, int K, int N)
{
    // one thread per 8 halves: idx = (nt*KQ + kq)*64 + lane
    const size_t idx   = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)K * N / 8;
    if (idx >= total) {
        return;
    }
    const int    KQ   = K / 32;
    const int    lane = idx & 63;
    const size_t tile = idx >> 6;
    const int    kq   = tile % KQ;
    const int    nt   = tile / KQ;
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
    int             KB;                // K / 128
    int             chunks_per_split;  // in units of KBC k-blocks
    int             total_chunks;
    int             epilogue;  // 0: fp16 store  1: gated silu fp16 store  2: fp32 partial slabs
};

__device__ __forceinline__ half8_t dequant8(uint32_t w, half2_t s2, half2_t z2)
{
    const half2_t k1024 = {(half_t)1024.0f, (half_t)1024.0f};
    half2_t       p0 = bit_cast<half2_t>((w & 0x000f000fu) | 0x64006400u) - k1024;
    half2_t       p1 = bit_cast<half2_t>(((w >> 4) & 0x000f000fu) | 0x64006400u) - k1024;
    half2_t       p2 = bit_cast<half2_t>(((w >> 8) & 0x000f000fu) | 0x64006400u) - k1024;
    half2_t       p3 = bit_cast<half2_t>(((w >> 12) & 0x000f000fu) | 0x64006400u) - k1024;
    p0               = h2_fma(p0, s2, z2);
    p1               = h2_fma(p1, s2, z2);
    p2               = h2_fma(p2, s2, z2);
    p3               = h2_fma(p3, s2, z2);
    return half8_t{p0[0], p0[1], p1[0], p1[1], p2[0], p2[1], p3[0], p3[1]};
}

template<int WT, int MT, int NT, int KBC>
__global__ __launch_bounds__(256) void gemm_kernel(GemmParams p)
{
    constexpr int MB   = 16 * MT;      // rows per workgroup
    constexpr int KCH  = 128 * KBC;    // k per chunk
    constexpr int ROWB = KCH * 2;      // LDS row bytes (multiple of 256)
    constexpr int CPR  = KCH / 8;      // 16-B chunks per row
    constexpr int XR   = (MB * CPR + 255) / 256;  // x staging vectors per thread
    constexpr int WV   = WT == 0 ? 1 : 4;         // u32x4 per (tile, k-block) per lane

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid  = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16  = lane & 15;
    const int g    = lane >> 4;

    const int ntiles = p.N / 16;
    const int nt0    = (blockIdx.x * 4 + wave) * NT;
    const int m0     = blockIdx.z * MB;
    const int c_beg  = blockIdx.y * p.chunks_per_split;
    const int c_end  = min(c_beg + p.chunks_per_split, p.total_chunks);

    // per-tile streams (tiles past the edge are clamped: loads stay in bounds, stores are skipped)
    const u32x4*    wp[NT];
    const uint32_t* sp[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int nt = min(nt0 + t, ntiles - 1);
        wp[t]        = p.wq + ((size_t)nt * p.KB) * 64 * WV + lane;
        sp[t]        = WT == 0 ? p.sz + ((size_t)nt * p.KB) * 16 + i16 : nullptr;
    }

    floatx4 acc[NT][MT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            acc[t][mt] = floatx4{0.f, 0.f, 0.f, 0.f};
        }
    }

    u32x4    wcur[NT][KBC][WV], wnxt[NT][KBC][WV];
    uint32_t scur[NT][KBC], snxt[NT][KBC];
    u32x4    xr[XR];

    auto load_w = [&](int c, u32x4 (&wb)[NT][KBC][WV], uint32_t (&sb)[NT][KBC]) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int kk = 0; kk < KBC; ++kk) {
                const int kb = c * KBC + kk;
#pragma unroll
                for (int v = 0; v < WV; ++v) {
                    wb[t][kk][v] = __builtin_nontemporal_load(wp[t] + ((size_t)kb * WV + v) * 64);
                }
                if constexpr (WT == 0) {
                    sb[t][kk] = sp[t][(size_t)kb * 16];
                }
            }
        }
    };
    auto load_x = [&](int c) {
#pragma unroll
        for (int r = 0; r < XR; ++r) {
            const int q  = tid + 256 * r;
            const int m  = q / CPR;
            const int ci = q % CPR;
            u32x4     v  = {0u, 0u, 0u, 0u};
            if (m < MB && m0 + m < p.M) {
                v = *(const u32x4*)(p.x + (size_t)(m0 + m) * p.ldx + (size_t)c * KCH + ci * 8);
            }
            xr[r] = v;
        }
    };
    auto store_x = [&]() {
#pragma unroll
        for (int r = 0; r < XR; ++r) {
            const int q  = tid + 256 * r;
            const int m  = q / CPR;
            const int ci = q % CPR;
            if (m < MB) {
                *(u32x4*)(smem + m * ROWB + ((ci ^ (m & 15)) << 4)) = xr[r];
            }
        }
    };

    if (c_beg < c_end) {
        load_x(c_beg);
        load_w(c_beg, wcur, scur);
    }

    for (int c = c_beg; c < c_end; ++c) {
        __syncthreads();  // every wave is done reading the previous chunk
        store_x();
        __syncthreads();
        if (c + 1 < c_end) {
            load_x(c + 1);
            load_w(c + 1, wnxt, snxt);
        }

#pragma unroll
        for (int kk = 0; kk < KBC; ++kk) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                half8_t xf[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int ci = kk * 16 + j * 4 + g;
                    xf[mt]       = *(const half8_t*)(smem + (mt * 16 + i16) * ROWB + ((ci ^ i16) << 4));
                }
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    half8_t wf;
                    if constexpr (WT == 0) {
                        const half2_t pr = bit_cast<half2_t>(scur[t][kk]);
                        wf               = dequant8(wcur[t][kk][0][j], half2_t{pr[0], pr[0]}, half2_t{pr[1], pr[1]});
                    }
                    else {
                        wf = bit_cast<half8_t>(wcur[t][kk][j]);
                    }
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, xf[mt], acc[t][mt], 0, 0, 0);
                    }
                }
            }
        }

        if (c + 1 < c_end) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
#pragma unroll
                for (int kk = 0; kk < KBC; ++kk) {
#pragma unroll
                    for (int v = 0; v < WV; ++v) {
                        wcur[t][kk][v] = wnxt[t][kk][v];
                    }
                    scur[t][kk] = snxt[t][kk];
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
    // Heuristic: keep >= ~256 workgroups in flight (one per CU) with as little split-K as possible.
    // TM_GEMM_NT / TM_GEMM_SPLITS override (the tuner in tools/ uses them).
    GemmConfig cfg{};
    const int  ntiles = w.N / 16;
    const int  KB     = w.K / 128;
    const int  mblk   = (M + 63) / 64;
    if (w.type == 1) {
        cfg.nt = 2;
    }
    else {
        cfg.nt = ntiles >= 4 * 4 * 256 / mblk ? 4 : (ntiles >= 2 * 4 * 256 / mblk ? 2 : (M > 64 ? 4 : 2));
    }
    const int col_wgs = (ntiles + 4 * cfg.nt - 1) / (4 * cfg.nt);
    int       splits  = 1;
    const int kbc     = (w.type == 0 && KB % 2 == 0) ? 2 : 1;
    const int chunks  = KB / kbc;
    while (col_wgs * mblk * splits < 256 && splits * 2 <= chunks / 2 && splits < 16) {
        splits *= 2;
    }
    cfg.splits = splits;
    cfg.nt     = env_int("TM_GEMM_NT", cfg.nt);
    cfg.splits = env_int("TM_GEMM_SPLITS", cfg.splits);
    if (cfg.splits > chunks) {
        cfg.splits = chunks;
    }
    return cfg;
}

template<int WT, int MT, int NT, int KBC>
static int launch_one(const GemmParams& p, dim3 grid, hipStream_t st)
{
    constexpr int lds = 16 * MT * 128 * KBC * 2;
    gemm_kernel<WT, MT, NT, KBC><<<grid, 256, lds, st>>>(p);
    TM_HIP_CHECK(hipGetLastError());
    return 0;
}

template<int WT, int MT>
static int launch_mt(const GemmParams& p, dim3 grid, int nt, int kbc, hipStream_t st)
{
    if constexpr (WT == 1) {
        if (nt == 1) return launch_one<1, MT, 1, 1>(p, grid, st);
        return launch_one<1, MT, 2, 1>(p, grid, st);
    }
    else {
        if (kbc == 2) {
            if (nt == 1) return launch_one<0, MT, 1, 2>(p, grid, st);
            if (nt == 2) return launch_one<0, MT, 2, 2>(p, grid, st);
            return launch_one<0, MT, 4, 2>(p, grid, st);
        }
        if (nt == 1) return launch_one<0, MT, 1, 1>(p, grid, st);
        if (nt == 2) return launch_one<0, MT, 2, 1>(p, grid, st);
        return launch_one<0, MT, 4, 1>(p, grid, st);
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
    int nt = cfg.nt;
    if (w.type == 1 && nt > 2) {
        nt = 2;
    }
    TM_REQUIRE(nt == 1 || nt == 2 || nt == 4, "nt in {1,2,4}");
    const int KB  = w.K / 128;
    const int kbc = (w.type == 0 && KB % 2 == 0) ? 2 : 1;
    const int mt  = M <= 16 ? 1 : (M <= 32 ? 2 : 4);
    int       splits = cfg.splits < 1 ? 1 : cfg.splits;
    const int chunks = KB / kbc;
    if (splits > chunks) {
        splits = chunks;
    }
    TM_REQUIRE(splits == 1 || workspace != nullptr, "split-K needs a workspace");
    TM_REQUIRE(!defer_reduce || splits > 1, "defer_reduce only with split-K");

    GemmParams p{};
    p.x                = x;
    p.ldx              = ldx;
    p.wq               = (const u32x4*)w.packed;
    p.sz               = w.sz;
    p.y                = y;
    p.ldy              = ldy;
    p.partial          = workspace;
    p.M                = M;
    p.N                = w.N;
    p.K                = w.K;
    p.KB               = KB;
    p.total_chunks     = chunks;
    p.chunks_per_split = (chunks + splits - 1) / splits;
    splits             = (chunks + p.chunks_per_split - 1) / p.chunks_per_split;  // no empty splits
    p.epilogue         = splits > 1 ? 2 : (gated_silu ? 1 : 0);

    const int ntiles = w.N / 16;
    dim3      grid((ntiles + 4 * nt - 1) / (4 * nt), splits, (M + 16 * mt - 1) / (16 * mt));
    int       rc = 0;
    if (w.type == 0) {
        rc = mt == 1 ? launch_mt<0, 1>(p, grid, nt, kbc, st) :
             mt == 2 ? launch_mt<0, 2>(p, grid, nt, kbc, st) :
                       launch_mt<0, 4>(p, grid, nt, kbc, st);
    }
    else {
        rc = mt == 1 ? launch_mt<1, 1>(p, grid, nt, kbc, st) :
             mt == 2 ? launch_mt<1, 2>(p, grid, nt, kbc, st) :
                       launch_mt<1, 4>(p, grid, nt, kbc, st);
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
