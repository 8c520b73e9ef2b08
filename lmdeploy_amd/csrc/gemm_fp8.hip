// FP8 x FP8 linear on the CDNA4 fp8 matrix cores (v_mfma_f32_32x32x16_fp8_fp8): e4m3 weights with 128 x 128 block scales,
// activations quantised on the fly per row and 128-channel group.  Plain and grouped (mixture-of-experts) form.
//
// Replaces: the fp8 path of LlamaLinear::Forward (src/turbomind/models/llama/LlamaLinear.cu:67-127: QuantizeSymm of the
//           input, then the fp8 GEMM with activation scales U and weight scales V), QuantizeSymm itself
//           (kernels/quantization.cu:28-125), the weight format lmdeploy/turbomind/weight_format.py:349-393, and for
//           experts the grouped launch with idxs / offsets (LlamaLinear.cu:95-127, moe_ffn_layer.cc:133-300).
// Arithmetic (oracle: tm_oracle.fp8_quant_rows / fp8_act_linear_acc):
//   activation: absmax over 128 channels (clamped to 1e-8), scale = absmax / 448, q = e4m3_rne_sat(x * (448 / absmax));
//   y[m, n] = sum_g  sx[g, m] * sw[g, n / 128] * sum_{k in g} xq[m, k] * wq[k, n]        (fp32)
//   the inner sum runs on the matrix core (products of two e4m3 values are exact in fp32), one k-group = 8 MFMAs.
//
// MI355X design -- the structure of gemm_decode.hip with the fp8 data path:
//   * the weights go from HBM straight into the MFMA: no dequantisation VALU at all.  Layout "P8": unit (kb, cg) =
//     32 columns x 128 k = 4096 B in A-operand order (lane l: column 32cg + (l & 31), 16-B vector v holds the 16-k steps
//     2v and 2v+1, 8 bytes = k 16j + 8(l >> 5) .. +8) followed by the two fp32 block scales of its even / odd columns
//     (they differ only for the fused w1w3 linear, whose (gate_j, up_j)-interleaved columns come from separately
//     quantised w1 / w3) -- 4224-byte units, one descriptor, k-block-major like P32;
//   * x codes go global -> LDS by LDS-DMA in stages of S k-blocks (128 B per row and k-block: half the fp16 volume), 16-byte
//     units XOR-swizzled by (row >> 1) & 7 on the source address; B fragments are ds_read_b64;
//   * per k-group the partial tile is scaled into the running fp32 accumulators with one fma per element: the lane's
//     row scale sx (a lane owns ONE row of the D tile) times the even / odd column scale;
//   * CG column groups x WK k-phases per workgroup, partial tiles of the k-phases summed through LDS, whole-row stores
//     (fp16, gated SiLU, fp32 split-K slabs) -- as in gemm_decode.hip;
//   * grouped: grid.z = expert x row block; device-side row segments (`seg`), x rows gathered through `row_idx`, per-expert
//     weight pointers; a block without rows exits before its first barrier.
#include "tm_common.h"
#include "tm_kernels.h"
#include <algorithm>
#include <stdlib.h>
#include <type_traits>

namespace tmk {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int kP8Unit = 4096 + 128;

// ---- activation quantiser ----------------------------------------------------------------------------------------
// 16 lanes = one (row, 128-channel group): 8 halves per lane, absmax by DPP, 8 codes (one 8-byte store) per lane.
__global__ __launch_bounds__(256) void quant_fp8_rows_kernel(uint8_t* __restrict__ xq, float* __restrict__ sx, const half_t* __restrict__ x,
                                                             int ldx, int M, int K, int ldsx)
{
    const int groups = K / 128;
    const int gid    = blockIdx.x * 16 + (threadIdx.x >> 4);  // (row, group) index
    const int l16    = threadIdx.x & 15;
    const bool ok    = gid < M * groups;
    const int gc     = ok ? gid : M * groups - 1;
    const int m      = gc / groups;
    const int g      = gc - m * groups;
    const half8_t v  = *(const half8_t*)(x + (size_t)m * ldx + g * 128 + l16 * 8);
    float amax       = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        amax = fmaxf(amax, fabsf((float)v[e]));
    }
    amax = group_max<16>(amax);
    amax = fmaxf(amax, 1e-8f);
    const float inv = 448.0f / amax;
    float       f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        f[e] = fminf(fmaxf((float)v[e] * inv, -448.0f), 448.0f);  // saturating RNE (the conversion below rounds to nearest even)
    }
    int w0 = 0, w1 = 0;
    w0     = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], w0, false);
    w0     = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], w0, true);
    w1     = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], w1, false);
    w1     = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], w1, true);
    if (ok) {
        *(u32x2*)(xq + (size_t)m * K + g * 128 + l16 * 8) = u32x2{(uint32_t)w0, (uint32_t)w1};
        if (l16 == 0) {
            sx[(size_t)g * ldsx + m] = amax / 448.0f;
        }
    }
}

int launch_quant_fp8_rows(uint8_t* xq, float* sx, const half_t* x, int ldx, int M, int K, int ldsx, hipStream_t st)
{
    TM_REQUIRE(K % 128 == 0 && ldx % 8 == 0 && ldsx >= M, "fp8 row quantiser: K % 128 == 0, 16-byte aligned rows, ldsx >= M");
    if (M == 0) {
        return 0;
    }
    const int n = M * (K / 128);
    quant_fp8_rows_kernel<<<(n + 15) / 16, 256, 0, st>>>(xq, sx, x, ldx, M, K, ldsx);
    TM_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---- load-time repack ---------------------------------------------------------------------------------------------
__global__ void repack_p8_kernel(uint32_t* __restrict__ out, const uint8_t* __restrict__ w, const float* __restrict__ block_scales, int K,
                                 int N, int gated)
{
    const size_t idx   = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // one dword
    const int    ncg   = N / 32;
    const size_t total = (size_t)(K / 128) * ncg * (kP8Unit / 4);
    if (idx >= total) {
        return;
    }
    const size_t unit = idx / (kP8Unit / 4);
    const int    d    = (int)(idx % (kP8Unit / 4));
    const int    cg   = (int)(unit % ncg);
    const int    kb   = (int)(unit / ncg);
    if (d >= 1024) {  // [even column scale, odd column scale, padding]
        const int nb = (N + 127) / 128;
        float     s  = 0.f;
        if (d - 1024 < 2) {
            const int n = cg * 32 + (d - 1024);
            const int b = gated ? (n & 1) * (nb / 2) + (n >> 1) / 128 : n / 128;
            s           = block_scales[(size_t)kb * nb + b];
        }
        out[idx] = bit_cast<uint32_t>(s);
        return;
    }
    // dword d: vector v = d >> 8, lane = (d >> 2) & 63, dword-in-vector dd = d & 3 -> j = 2v + (dd >> 1), bytes 4(dd & 1) .. +4
    const int v    = d >> 8;
    const int lane = (d >> 2) & 63;
    const int dd   = d & 3;
    const int j    = 2 * v + (dd >> 1);
    const int n    = cg * 32 + (lane & 31);
    const int k0   = kb * 128 + 16 * j + 8 * (lane >> 5) + 4 * (dd & 1);
    uint32_t  o    = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        o |= (uint32_t)w[(size_t)(k0 + e) * N + n] << (8 * e);
    }
    out[idx] = o;
}

size_t p8_bytes(int K, int N)
{
    return (size_t)(K / 128) * (N / 32) * kP8Unit;
}

int launch_repack_p8(void* out, const uint8_t* weight, const float* block_scales, int K, int N, bool gated, hipStream_t st)
{
    const size_t total = p8_bytes(K, N) / 4;
    repack_p8_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>((uint32_t*)out, weight, block_scales, K, N, gated ? 1 : 0);
    TM_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---- GEMM -----------------------------------------------------------------------------------------------------------
struct Fp8Params {
    const uint8_t* xq;   // [rows][K] e4m3 codes
    const float*   sx;   // [K/128][ldsx] activation scales
    int            ldsx;
    const void*    wp;   // P8 units (plain) ...
    const void* const* groups;  // ... or device [experts] unit pointers (grouped)
    const int*     seg;         // grouped: device [experts + 1] flat-row offsets
    const int*     row_idx;     // grouped: x / sx row of flat row f (nullptr: f)
    int            zper;        // grouped: row blocks per expert
    half_t*        y;
    int            ldy;
    float*         partial;
    int            M, N, K, KB, ncg;  // M: rows (plain) / flat rows in total (grouped, slab stride)
    int            x_rows;            // rows of xq / sx (descriptor bounds)
    int            kb_per_split;
    int            epilogue;  // 0 fp16, 1 gated SiLU fp16, 2 fp32 slab of split blockIdx.y
};

template<int N, class F, int I = 0>
__device__ __forceinline__ void static_for8(F&& f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for8<N, F, I + 1>(static_cast<F&&>(f));
    }
}

template<int MH, int CG, int WK, int S, bool GRP>
__global__ __launch_bounds__(CG* WK * 64) void gemm_fp8_kernel(Fp8Params p)
{
    constexpr int WAVES = CG * WK;
    constexpr int T     = WAVES * 64;
    constexpr int ROWS  = 32 * MH;
    constexpr int KBB   = ROWS * 128;  // LDS bytes of one k-block of codes
    constexpr int STG   = S * KBB;
    constexpr int BPS   = S / WK;
    constexpr int PF    = 2 * BPS;     // ring = two stages
    constexpr int NPC   = S * ROWS / 8;  // 1-KiB DMA pieces (8 rows x 128 B) per stage
    constexpr int DR    = (NPC + WAVES - 1) / WAVES;
    static_assert(S % WK == 0, "tile parameters");
    (void)T;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid  = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cgl  = wave % CG;
    const int wk   = wave / CG;
    const int l31  = lane & 31;
    const int half = lane >> 5;

    // rows of this workgroup
    int         m0 = blockIdx.z * ROWS, row0 = 0, Mloc = min(ROWS, p.M - m0);
    const void* wbase = p.wp;
    if constexpr (GRP) {
        const int e = blockIdx.z / p.zper;
        m0          = (blockIdx.z - e * p.zper) * ROWS;
        row0        = p.seg[e];
        Mloc        = min(ROWS, p.seg[e + 1] - row0 - m0);
        wbase       = p.groups[e];
        if (Mloc <= 0) {
            return;  // whole workgroup, before any barrier
        }
    }
    const int cg  = blockIdx.x * CG + cgl;
    const int cgc = min(cg, p.ncg - 1);
    const int kb0 = blockIdx.y * p.kb_per_split;
    const int nkb = min(p.kb_per_split, p.KB - kb0);
    const int nst = (nkb + S - 1) / S;

    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)wbase, 0, (int)((size_t)p.KB * p.ncg * kP8Unit), 0x00020000);
    const auto rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.xq, 0, (int)((size_t)p.x_rows * p.K), 0x00020000);
    const auto rs_s = __builtin_amdgcn_make_buffer_rsrc((void*)p.sx, 0, (int)((size_t)p.KB * p.ldsx * 4), 0x00020000);

    // x / sx row of local row m (clamped duplicates past Mloc only feed rows that are never stored)
    auto xrow = [&](int m) {
        const int f = row0 + m0 + min(m, Mloc - 1);
        return (GRP && p.row_idx) ? p.row_idx[f] : f;
    };

    floatx16 acc[MH];
#pragma unroll
    for (int h = 0; h < MH; ++h) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            acc[h][r] = 0.f;
        }
    }
    int sxoff[MH];  // byte offset of this lane's row in one k-group row of sx
#pragma unroll
    for (int h = 0; h < MH; ++h) {
        sxoff[h] = xrow(32 * h + l31) * 4;
    }
    // B fragment of 16-k step j: row (l & 31) [+ 32h], 16-byte unit j XOR-swizzled by (row >> 1) & 7, 8-byte half (l >> 5)
    int coff[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        coff[j] = l31 * 128 + ((j ^ ((l31 >> 1) & 7)) << 4) + half * 8;
    }
    // DMA pieces of a stage: piece pc = r * WAVES + wave = k-block pc / (ROWS / 8), rows 8 (pc % (ROWS / 8)) + (lane >> 3);
    // lane L lands at slot L & 7 of its row and therefore fetches unit (L & 7) ^ ((row >> 1) & 7)
    int            doff[DR];
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)smem);
#pragma unroll
    for (int r = 0; r < DR; ++r) {
        const int pc  = min(r * WAVES + wave, NPC - 1);
        const int kbi = pc / (ROWS / 8);
        const int row = (pc % (ROWS / 8)) * 8 + (lane >> 3);
        const int u   = (lane & 7) ^ ((row >> 1) & 7);
        doff[r]       = xrow(row) * p.K + kbi * 128 + u * 16;
    }
    const int my_dr = NPC % WAVES == 0 ? DR : (wave < NPC - (DR - 1) * WAVES ? DR : DR - 1);  // wave-uniform
#define F8_DMA_X(t, buf)                                                                                          \
    _Pragma("unroll") for (int r = 0; r < DR; ++r)                                                                \
    {                                                                                                             \
        if (r < my_dr) {                                                                                          \
            unsigned       keep_;                                                                                 \
            const unsigned dst_ = lds0 + (buf)*STG + (r * WAVES + wave) * 1024;                                   \
            const int      so_  = (kb0 + (t)*S) * 128;                                                            \
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"                                       \
                         "buffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"                               \
                         : "=&s"(keep_)                                                                           \
                         : "v"(doff[r]), "s"(rs_x), "s"(dst_), "s"(so_)                                           \
                         : "memory");                                                                             \
        }                                                                                                         \
    }

    u32x4 ring[PF][4];
    u32x2 swr[PF];      // (even, odd) column scale of the unit
    float sxr[PF][MH];  // this lane's row scale(s) for the k-group
    const int vw = lane * 16;
#define F8_LOAD_W(slot, b)                                                                                        \
    {                                                                                                             \
        const int kbx_ = kb0 + min((b), nkb - 1);                                                                 \
        const int uo_  = (kbx_ * p.ncg + cgc) * kP8Unit;                                                          \
        _Pragma("unroll") for (int v = 0; v < 4; ++v)                                                             \
        {                                                                                                         \
            ring[slot][v] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, vw + v * 1024, uo_, /*nt*/ 2);            \
        }                                                                                                         \
        swr[slot] = __builtin_amdgcn_raw_buffer_load_b64(rs_w, 4096, uo_, 0);                                     \
        _Pragma("unroll") for (int h = 0; h < MH; ++h)                                                            \
        {                                                                                                         \
            sxr[slot][h] = bit_cast<float>(__builtin_amdgcn_raw_buffer_load_b32(rs_s, sxoff[h], kbx_ * p.ldsx * 4, 0)); \
        }                                                                                                         \
    }
    constexpr int LPB = 4 + 1 + MH;  // loads per ring slot

    if (nst > 0) {
        F8_DMA_X(0, 0);
#pragma unroll
        for (int q = 0; q < PF; ++q) {
            F8_LOAD_W(q, (q / BPS) * S + wk + (q % BPS) * WK);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(%0)" ::"i"(LPB * PF) : "memory");  // everything older than the ring: x(0)
        __syncthreads();

        auto stage = [&](auto U, const int t) __attribute__((always_inline)) {
            constexpr int u   = decltype(U)::value;
            const int     buf = u & 1;
            // make hipcc wait for this stage's ring slots HERE, then start the (invisible to its waitcnt pass) DMA of x(t+1)
#pragma unroll
            for (int i = 0; i < BPS; ++i) {
                asm volatile("" ::"v"(ring[u * BPS + i][0]), "v"(ring[u * BPS + i][3]), "v"(swr[u * BPS + i]), "v"(sxr[u * BPS + i][MH - 1]));
            }
            if (t + 1 < nst) {  // uniform; see gemm_dec32_kernel: a DMA behind the last stage could land on the reduction image
                F8_DMA_X(t + 1, buf ^ 1);
            }
#pragma unroll
            for (int i = 0; i < BPS; ++i) {
                const int  slot = u * BPS + i;
                const int  kbi  = wk + i * WK;
                const bool live = t * S + kbi < nkb;
                const char* xb  = smem + buf * STG + kbi * KBB;
                floatx16    tt[MH];
                static_for8<8>([&](auto J) {
                    constexpr int j  = decltype(J)::value;
                    const u32x4   wv = ring[slot][j >> 1];
                    const long    a  = bit_cast<long>(u32x2{wv[(j & 1) * 2], wv[(j & 1) * 2 + 1]});
#pragma unroll
                    for (int h = 0; h < MH; ++h) {
                        const long b = *(const long*)(xb + h * 4096 + coff[j]);
                        if constexpr (j == 0) {
                            tt[h] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a, b, floatx16{}, 0, 0, 0);
                        }
                        else {
                            tt[h] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a, b, tt[h], 0, 0, 0);
                        }
                    }
                });
                if (live) {  // wave-uniform; a k-block past the slice contributes nothing (its operands are clamped re-reads)
                    const float s_even = bit_cast<float>(swr[slot][0]);
                    const float s_odd  = bit_cast<float>(swr[slot][1]);
#pragma unroll
                    for (int h = 0; h < MH; ++h) {
                        const float se = sxr[slot][h] * s_even, so = sxr[slot][h] * s_odd;
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            acc[h][r] = __builtin_fmaf(tt[h][r], (r & 1) ? so : se, acc[h][r]);
                        }
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < BPS; ++i) {
                F8_LOAD_W(u * BPS + i, (t + 2) * S + wk + i * WK);
            }
            asm volatile("s_waitcnt vmcnt(%0)" ::"i"(LPB * BPS) : "memory");  // my DMA pieces of x(t+1) have landed
            __syncthreads();
        };
        int t0 = 0;
        for (; t0 + 2 <= nst; t0 += 2) {
            static_for8<2>([&](auto U) { stage(U, t0 + decltype(U)::value); });
        }
        if (t0 < nst) {
            stage(std::integral_constant<int, 0>{}, t0);
        }
    }
#undef F8_DMA_X
#undef F8_LOAD_W

    // ---- k-phase reduction + whole-row stores (as gemm_decode.hip) ------------------------------------------------------
    {
        constexpr int C4 = CG * 8;
        floatx4*      red = (floatx4*)smem;
#pragma unroll
        for (int h = 0; h < MH; ++h) {
            const int m = 32 * h + l31;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int c4 = cgl * 8 + 2 * g4 + half;
                red[(wk * ROWS + m) * C4 + (c4 ^ (m & 7))] =
                    floatx4{acc[h][4 * g4], acc[h][4 * g4 + 1], acc[h][4 * g4 + 2], acc[h][4 * g4 + 3]};
            }
        }
        __syncthreads();
        constexpr int NE    = ROWS * C4;
        const int     ncol0 = blockIdx.x * CG * 32;
#pragma unroll
        for (int e0 = 0; e0 < NE; e0 += T) {
            const int e = e0 + tid;
            if (NE % T != 0 && e >= NE) {
                break;
            }
            const int m  = e / C4;
            const int c4 = e % C4;
            floatx4   a  = red[m * C4 + (c4 ^ (m & 7))];
#pragma unroll
            for (int k = 1; k < WK; ++k) {
                a += red[(k * ROWS + m) * C4 + (c4 ^ (m & 7))];
            }
            const int n = ncol0 + c4 * 4;
            if (m >= Mloc || n >= p.N) {
                continue;
            }
            const size_t mg = (size_t)row0 + m0 + m;  // (flat) output row
            if (p.epilogue == 2) {
                *(floatx4*)(p.partial + ((size_t)blockIdx.y * p.M + mg) * p.N + n) = a;
            }
            else if (p.epilogue == 1) {
                const float s0 = a[0] / (1.0f + __builtin_expf(-a[0]));
                const float s1 = a[2] / (1.0f + __builtin_expf(-a[2]));
                half2_t     o  = {(half_t)(s0 * a[1]), (half_t)(s1 * a[3])};
                *(half2_t*)(p.y + mg * p.ldy + (n >> 1)) = o;
            }
            else {
                half4_t o = {(half_t)a[0], (half_t)a[1], (half_t)a[2], (half_t)a[3]};
                *(half4_t*)(p.y + mg * p.ldy + n) = o;
            }
        }
    }
}

template<int MH, int CG, int WK, bool GRP>
static int launch_fp8_one(const Fp8Params& p, dim3 grid, hipStream_t st)
{
    constexpr int S     = 4;
    constexpr int stage = 2 * S * 32 * MH * 128;
    constexpr int red   = WK * 32 * MH * CG * 128;
    constexpr int lds   = stage > red ? stage : red;
    if (const int rc = ensure_dynamic_lds((const void*)gemm_fp8_kernel<MH, CG, WK, S, GRP>, lds)) {
        return rc;
    }
    gemm_fp8_kernel<MH, CG, WK, S, GRP><<<grid, CG * WK * 64, lds, st>>>(p);
    TM_HIP_CHECK(hipGetLastError());
    return 0;
}

bool fp8_mfma_supported(const LinearWeight& w)
{
    const char* ev = getenv("TM_FP8_MFMA");  // 0: keep e4m3 weights on the weight-only (dequantise to f16) path
    const int   on = ev ? atoi(ev) : 1;
    return on && w.type == 2 && w.packed8 != nullptr && w.N % 32 == 0 && w.K % 128 == 0;
}

size_t fp8_act_workspace_bytes(int rows, int K)
{
    // codes [rows][K] + scales [K/128][rows rounded up to 4]
    return (size_t)rows * K + (size_t)(K / 128) * ((rows + 3) / 4 * 4) * sizeof(float) + 256;
}

static int fp8_splits(int col_wgs, int KB, int want)
{
    if (want > 0) {
        return want;
    }
    int splits = 1;
    for (int s = 2; s <= 16; ++s) {
        int per = (KB + s - 1) / s;
        per     = (per + 3) / 4 * 4;
        if ((KB + per - 1) / per == s && col_wgs * s <= 256 && per >= 4) {
            splits = s;
        }
    }
    return splits;
}

// y = fp8 x fp8 linear of PRE-QUANTISED activations (xq, sx from launch_quant_fp8_rows).  splits = 0: heuristic.
int launch_linear_fp8(const LinearWeight& w, const uint8_t* xq, const float* sx, int ldsx, half_t* y, int ldy, int M, bool gated_silu,
                      int splits, float* workspace, int* slabs_out, hipStream_t st)
{
    // (the TM_FP8_MFMA switch is evaluated where the path is CHOSEN -- fp8_mfma_supported at load / dispatch time --, not per launch)
    TM_REQUIRE(w.type == 2 && w.packed8 != nullptr && w.N % 32 == 0 && w.K % 128 == 0, "fp8 MFMA linear: e4m3 weights in P8 layout, N % 32 == 0");
    if (slabs_out) {
        *slabs_out = 1;
    }
    if (M == 0) {
        return 0;
    }
    Fp8Params p{};
    p.xq = xq, p.sx = sx, p.ldsx = ldsx, p.wp = w.packed8, p.y = y, p.ldy = ldy, p.partial = workspace;
    p.M = M, p.N = w.N, p.K = w.K, p.KB = w.K / 128, p.ncg = w.N / 32, p.x_rows = M;
    const int rows_per = M <= 32 ? 32 : 64;
    const int zb       = (M + rows_per - 1) / rows_per;
    const int col_wgs  = (p.ncg + 3) / 4 * zb;
    int       sp       = workspace ? fp8_splits(col_wgs, p.KB, splits) : 1;
    int       per      = (p.KB + sp - 1) / sp;
    per                = std::min((per + 3) / 4 * 4, p.KB);
    sp                 = (p.KB + per - 1) / per;
    p.kb_per_split     = per;
    p.epilogue         = sp > 1 ? 2 : (gated_silu ? 1 : 0);
    dim3 grid((p.ncg + 3) / 4, sp, zb);
    const int rc = M <= 32 ? launch_fp8_one<1, 4, 4, false>(p, grid, st) : launch_fp8_one<2, 4, 2, false>(p, grid, st);
    if (rc) {
        return rc;
    }
    if (slabs_out) {
        *slabs_out = sp;
    }
    return 0;
}

// grouped: flat rows seg[e] .. seg[e+1] of expert e, x / sx row of flat row f = row_idx ? row_idx[f] : f; m_hint = expected rows
// per expert (tile height), m_cap = upper bound
int launch_linear_fp8_grouped(const LinearWeight& proto, const void* d_groups, int E, const uint8_t* xq, const float* sx, int ldsx,
                              int x_rows, half_t* y, int ldy, int m_cap, int m_hint, bool gated_silu, const int* seg,
                              const int* row_idx, hipStream_t st)
{
    TM_REQUIRE(proto.type == 2 && proto.N % 32 == 0 && proto.K % 128 == 0, "grouped fp8 MFMA linear: e4m3 experts, N % 32 == 0");
    if (m_cap == 0 || E == 0) {
        return 0;
    }
    Fp8Params p{};
    p.xq = xq, p.sx = sx, p.ldsx = ldsx, p.groups = (const void* const*)d_groups, p.seg = seg, p.row_idx = row_idx;
    p.y = y, p.ldy = ldy, p.M = m_cap, p.N = proto.N, p.K = proto.K, p.KB = proto.K / 128, p.ncg = proto.N / 32, p.x_rows = x_rows;
    p.kb_per_split = p.KB;
    p.epilogue     = gated_silu ? 1 : 0;
    const int want = std::min(m_cap, std::max(1, 2 * m_hint));
    int       tv[4];
    const bool measured = gen_table_get(kGenGrouped + 2, 0, proto.K, proto.N, dec32_m_bucket(m_cap), tv);  // tm_engine_tune_gemm
    if (measured ? tv[0] == 32 : want <= 32) {
        p.zper = (m_cap + 31) / 32;
        dim3 grid((p.ncg + 3) / 4, 1, E * p.zper);
        return launch_fp8_one<1, 4, 4, true>(p, grid, st);
    }
    p.zper = (m_cap + 63) / 64;
    dim3 grid((p.ncg + 3) / 4, 1, E * p.zper);
    return launch_fp8_one<2, 4, 2, true>(p, grid, st);
}

}  // namespace tmk
