// Round-6 probe (no product code): does the activation stream of the M = 64 decode GEMM -- 512 KB of L2-resident x per workgroup, brought into LDS
// by DMA one stage ahead of the stage being consumed, one barrier per stage -- run faster when MORE stages are in flight?  Both decode GEMM kernels
// (gemm_dec32_kernel, gemm_dec_lc_kernel) hold exactly one 64 KB stage of x in flight per CU; their loops run at 45-54 GB/s per CU, the
// guide quotes ~90 GB/s for a CU's LDS-DMA path.  This probe is the loader / consumer skeleton of gemm_dec_lc_kernel with the stage size and the
// number of stages in flight as parameters:
//   4 loader waves: DMA (buffer_load_dwordx4 ... lds, 1 KB pieces = 4 rows x 256 B) of stage t + DEPTH, wait until stage t + 1 has landed, barrier
//   8 consumer waves: optionally a weight stream from HBM (nt loads, 3 stages deep, 2 x 2176-byte units per 128-k block and wave, as the kernel),
//                     optionally the kernel's matrix work on the stage (ds_read_b128 fragments + 32 MFMA 32x32x16 per k-block), barrier
// grid = 256 workgroups (one per CU: >= 84 KB of LDS each), x = 64 rows x 4096 k fp16 shared by all, weights 278 KB per workgroup from a rotating
// 1 GB arena (no Infinity-Cache reuse between launches).  Prints us per launch and GB/s per CU for every (stage KB, depth, weights, mfma).
//   hipcc --offload-arch=gfx950 -O2 -o xdma_depth_probe xdma_depth_probe.hip && ./xdma_depth_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define CK(x)                                                                                                  \
    do {                                                                                                       \
        hipError_t e_ = (x);                                                                                   \
        if (e_ != hipSuccess) {                                                                                \
            printf("%s -> %s\n", #x, hipGetErrorString(e_));                                                   \
            return 1;                                                                                          \
        }                                                                                                      \
    } while (0)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int kRows = 64, kK = 4096, kKB = kK / 128;   // 32 k-blocks of 128
constexpr int kBlockBytes = kRows * 256;               // 16 KB of x per k-block
constexpr int kUnit = 2176;                            // bytes of one P32 weight unit (32 columns x 128 k)
constexpr int kNLoad = 4, kNCons = 8, kThreads = (kNLoad + kNCons) * 64;

template<int N>
__device__ __forceinline__ void wait_vm()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}
// at most `rem` whole stages (DR pieces each) may stay in flight, rem in [0, R]
template<int R, int DR>
__device__ __forceinline__ void wait_stages(int rem)
{
    if (rem >= R) {
        wait_vm<R * DR>();
    }
    else if constexpr (R > 0) {
        wait_stages<R - 1, DR>(rem);
    }
}

template<int SKB /*k-blocks per stage*/, int DEPTH, int WITH_W, int WITH_MFMA, int RD /*weight ring depth in blocks per wave*/>
__global__ __launch_bounds__(kThreads) void probe(const _Float16* __restrict__ x, const char* __restrict__ w, float* sink, int with_x)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int STG  = SKB * kBlockBytes;
    constexpr int NBUF = DEPTH + 1;
    constexpr int NST  = kKB / SKB;
    constexpr int NPC  = STG / 1024;       // 1 KB pieces per stage
    constexpr int DR   = NPC / kNLoad;     // pieces per loader wave and stage
    static_assert(NPC % kNLoad == 0, "pieces per loader");
    const int      lane = threadIdx.x & 63;
    const int      wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)smem);
    if (wave >= kNCons) {
        const int  lw   = wave - kNCons;
        const auto rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, kRows * kK * 2, 0x00020000);
        int        doff[DR];
#pragma unroll
        for (int r = 0; r < DR; ++r) {
            const int pc  = r * kNLoad + lw;
            const int kbi = pc / (kRows / 4);
            const int row = (pc % (kRows / 4)) * 4 + (lane >> 4);
            const int ch  = (lane & 15) ^ (row & 15);
            doff[r]       = (row * kK + ch * 8) * 2 + kbi * 256;
        }
        auto dma = [&](int t) __attribute__((always_inline)) {
            const int buf = t % NBUF;
#pragma unroll
            for (int r = 0; r < DR; ++r) {
                unsigned       keep_;
                const unsigned dst_ = lds0 + buf * STG + (r * kNLoad + lw) * 1024;
                const int      so_  = t * SKB * 256;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                             "buffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep_)
                             : "v"(doff[r]), "s"(rs_x), "s"(dst_), "s"(so_)
                             : "memory");
            }
        };
        // prologue: stages 0 .. DEPTH-1 in flight; stage 0 landed before the first barrier
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            if (d < NST && with_x) {
                dma(d);
            }
        }
        asm volatile("s_waitcnt vmcnt(%0)" ::"i"((DEPTH - 1) * DR) : "memory");
        __builtin_amdgcn_s_barrier();
        for (int t = 0; t < NST; ++t) {
            // consumers work on stage t; stage t + DEPTH goes into the buffer stage t - 1 left at the last barrier
            if (t + DEPTH < NST && with_x) {
                dma(t + DEPTH);
            }
            wait_stages<DEPTH - 1, DR>(NST - 2 - t);  // stage t + 1 landed: stages t + 2 .. min(t + DEPTH, NST - 1) may stay in flight
            __builtin_amdgcn_s_barrier();
        }
        return;
    }
    // ---- consumers ----
    // SKB = 4: column half cw, k-phase wk of 4 (a block = 2 units x 2 row halves: 4 MFMAs per fragment pair, the kernel's arrangement);
    // SKB = 2: k-phase wave & 1 of 2, column unit wave >> 1 of 4 (a block = 1 unit x 2 row halves: the same MFMA count and weight bytes per wave)
    constexpr int NPH = SKB >= 4 ? 4 : 2;
    const int cw = SKB >= 4 ? (wave & 1) : (wave >> 1), wk = SKB >= 4 ? (wave >> 1) : (wave & 1);
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, 1 << 30, 0x00020000);
    // my weight units: block kb (of 32), columns cw: units (kb, 4 blockIdx + 2 cw + {0, 1}) of a [kb][1024 units] image
    u32x4    ring[RD][4];
    floatx16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            acc[i][r] = 0.f;
        }
    }
    // block j of this wave = its block of stage j (one per stage in both arrangements): kb = NPH * (j % (32 / NPH)) + wk; blocks past the last
    // stage are requested out of the descriptor's range (no memory traffic, zeros) so that every path issues the same loads (static vmcnt counts)
    auto loadw = [&](u32x4(&slot)[4], int j) __attribute__((always_inline)) {
        const int kb = NPH * j + wk;
        const int uo = j < NST ? (kb * 1024 + (blockIdx.x * 4 + (SKB >= 4 ? cw * 2 : cw))) * kUnit : (1 << 30);
        slot[0] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, lane * 16, uo, 2);
        slot[1] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, lane * 16 + 1024, uo, 2);
        if constexpr (SKB >= 4) {
            slot[2] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, lane * 16, uo + kUnit, 2);
            slot[3] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, lane * 16 + 1024, uo + kUnit, 2);
        }
    };
    if (WITH_W) {
#pragma unroll
        for (int r = 0; r < RD; ++r) {
            loadw(ring[r], r);
        }
    }
    __builtin_amdgcn_s_barrier();
    const int l31 = lane & 31, half = lane >> 5;
    static_assert(NST % RD == 0, "ring depth must divide the stage count");
    for (int t0 = 0; t0 < NST; t0 += RD) {
#pragma unroll
        for (int r = 0; r < RD; ++r) {
            const int t   = t0 + r;
            const int buf = t % NBUF;
            const int i   = SKB >= 4 ? wk : wk;  // my k-block inside the stage
            u32x4     a[4] = {};
            if (WITH_W) {
                a[0] = ring[r][0], a[1] = ring[r][1];
                if constexpr (SKB >= 4) {
                    a[2] = ring[r][2], a[3] = ring[r][3];
                }
                loadw(ring[r], t + RD);
            }
            if (WITH_MFMA) {
                const char* xb = smem + buf * STG + i * kBlockBytes;
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    const int   off = l31 * 256 + (((2 * s + half) ^ (l31 & 15)) << 4);
                    const half8 b0  = *(const half8*)(xb + off);
                    const half8 b1  = *(const half8*)(xb + 8192 + off);
                    // the kernel's operand: 8 fp16 built from one dword of codes (here: bit patterns made finite and small)
                    const unsigned w0 = a[s >> 2][s & 3], w1 = a[2 + (s >> 2)][s & 3];
                    const u32x4    q0 = {(w0 & 0x03ff03ffu) | 0x20002000u, ((w0 >> 4) & 0x03ff03ffu) | 0x20002000u, ((w0 >> 8) & 0x03ff03ffu) | 0x20002000u,
                                         ((w0 >> 12) & 0x03ff03ffu) | 0x20002000u};
                    const u32x4    q1 = {(w1 & 0x03ff03ffu) | 0x20002000u, ((w1 >> 4) & 0x03ff03ffu) | 0x20002000u, ((w1 >> 8) & 0x03ff03ffu) | 0x20002000u,
                                         ((w1 >> 12) & 0x03ff03ffu) | 0x20002000u};
                    const half8    a0 = __builtin_bit_cast(half8, q0), a1 = __builtin_bit_cast(half8, q1);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc[1], 0, 0, 0);
                    if constexpr (SKB >= 4) {
                        acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc[2], 0, 0, 0);
                        acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc[3], 0, 0, 0);
                    }
                }
            }
            else if (WITH_W) {
                acc[0][0] += __builtin_bit_cast(float, a[0][0] ^ a[1][1] ^ a[2][2] ^ a[3][3]);
            }
            __builtin_amdgcn_s_barrier();
        }
    }
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            v += acc[i][r];
        }
    }
    if (v == 123.456f) {
        sink[threadIdx.x] = v;
    }
}

template<int SKB, int DEPTH, int WITH_W, int WITH_MFMA, int RD>
static int run(const _Float16* x, const char* w, float* sink, hipStream_t st, int with_x = 1)
{
    constexpr int lds = (DEPTH + 1) * SKB * kBlockBytes;
    const int     dyn = lds < 84 * 1024 ? 84 * 1024 : lds;  // one workgroup per CU in every configuration
    if (dyn > 160 * 1024) {
        return 0;
    }
    auto k = probe<SKB, DEPTH, WITH_W, WITH_MFMA, RD>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, dyn));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const size_t region = (size_t)32 * 1024 * kUnit;  // one [32][1024 units] image = 71 MB
    for (int i = 0; i < 3; ++i) {
        k<<<256, kThreads, dyn, st>>>(x, w + (size_t)(i % 14) * region, sink, with_x);
    }
    CK(hipStreamSynchronize(st));
    const int N = 28;
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < N; ++i) {
        k<<<256, kThreads, dyn, st>>>(x, w + (size_t)(i % 14) * region, sink, with_x);
    }
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us    = ms * 1000.0 / N;
    const double bytes = (with_x ? 512.0 * 1024 : 0.0) + (WITH_W ? 8 * 32.0 * 1024 : 0.0);  // per workgroup: x + 8 consumers x 32 KB of weight loads
    printf("x %d  stage %3d KB  depth %d  weights %d (ring %d blocks)  mfma %d  lds %3d KB : %7.2f us per launch (incl. ~2 us boundary)  %6.1f GB/s per CU  %5.2f TB/s of weights\n",
           with_x, SKB * 16, DEPTH, WITH_W, RD, WITH_MFMA, lds / 1024, us, bytes / us / 1e3, WITH_W ? 256 * 8 * 32.0 * 1024 / us / 1e6 : 0.0);
    fflush(stdout);
    return 0;
}

template<int W, int M>
static int sweep(const _Float16* x, const char* w, float* sink, hipStream_t st)
{
    if (run<4, 1, W, M, 2>(x, w, sink, st)) return 1;   // the kernels' own arrangement: 64 KB stages, one in flight, ring of 2 stages
    if (W) {
        if (run<4, 1, W, M, 1>(x, w, sink, st)) return 1;
        if (run<4, 1, W, M, 4>(x, w, sink, st)) return 1;
        if (run<4, 1, W, M, 8>(x, w, sink, st)) return 1;
    }
    if (run<2, 1, W, M, 4>(x, w, sink, st)) return 1;
    if (run<2, 2, W, M, 4>(x, w, sink, st)) return 1;
    if (run<2, 3, W, M, 4>(x, w, sink, st)) return 1;
    if (W) {
        if (run<2, 3, W, M, 8>(x, w, sink, st)) return 1;
        if (run<2, 3, W, M, 16>(x, w, sink, st)) return 1;
    }
    return 0;
}

int main()
{
    _Float16* x;
    char*     w;
    float*    sink;
    CK(hipMalloc((void**)&x, (size_t)kRows * kK * 2));
    CK(hipMalloc((void**)&w, (size_t)1 << 30));
    CK(hipMalloc((void**)&sink, 4096 * 4));
    std::vector<uint16_t> hx((size_t)kRows * kK);
    for (size_t i = 0; i < hx.size(); ++i) {
        hx[i] = (uint16_t)(0x3000 + (i * 2654435761u >> 20 & 0x7ff));  // finite fp16 values around 0.1 .. 0.25
    }
    CK(hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
    std::vector<uint32_t> hw((size_t)1 << 22);
    for (size_t i = 0; i < hw.size(); ++i) {
        hw[i] = (uint32_t)(i * 2246822519u + 374761393u);
    }
    for (size_t off = 0; off < ((size_t)1 << 30); off += hw.size() * 4) {
        CK(hipMemcpy(w + off, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    }
    hipStream_t st;
    CK(hipStreamCreate(&st));
    printf("# x only\n");
    if (sweep<0, 0>(x, w, sink, st)) return 1;
    printf("# x + weight stream (no matrix work)\n");
    if (sweep<1, 0>(x, w, sink, st)) return 1;
    printf("# x + weight stream + matrix work\n");
    if (sweep<1, 1>(x, w, sink, st)) return 1;
    printf("# weight stream alone (no x DMA)\n");
    if (run<4, 1, 1, 0, 2>(x, w, sink, st, 0)) return 1;
    if (run<4, 1, 1, 0, 8>(x, w, sink, st, 0)) return 1;
    if (run<2, 1, 1, 0, 4>(x, w, sink, st, 0)) return 1;
    printf("# weight stream + matrix work (LDS fragments of a stale stage, no x DMA)\n");
    if (run<4, 1, 1, 1, 2>(x, w, sink, st, 0)) return 1;
    if (run<4, 1, 1, 1, 1>(x, w, sink, st, 0)) return 1;
    printf("# x + matrix work on zero weights (no weight stream)\n");
    if (sweep<0, 1>(x, w, sink, st)) return 1;
    return 0;
}
