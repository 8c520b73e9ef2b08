// W4A16 decode GEMM (M <= 64 rows), loader / consumer form for gfx950.
//
// Replaces (same role and arithmetic as gemm_decode.hip): LlamaLinear::Forward -> gemm::Gemm::Run for the decode batch
//           (src/turbomind/models/llama/LlamaLinear.cu:140-216, kernels/gemm/gemm.cu:257-344), dequant
//           kernels/gemm/transform.h:34-74, gated-SiLU epilogue kernels/gemm/epilogue.h:159-176.
// w = h(fma(h(q), s, h(-z*s))) per element (bit-identical to the reference's operand), fp32 MFMA accumulation, one rounding.
//
// Structure (round 4).  gemm_dec32_kernel lets every wave issue BOTH its weight-ring refills and its share of the activation
// DMA; VMEM returns in order, so the wait that covers a wave's activation pieces also retires every older weight load.  Here
// the two streams live in different waves (different vmcnt queues) and a wave's tile is twice as wide:
//   * 4 LOADER waves bring the activations of a stage (WK k-blocks x ROWS rows) into LDS by DMA (buffer_load ... lds),
//     double buffered, and do nothing else;
//   * 8 CONSUMER waves = 2 column halves x 4 k-phases; a consumer owns 64 columns (two P32 units per k-block) x all rows:
//     its vmcnt queue holds ONLY weights, in consumption order, PFS stages deep, loads the compiler counts;
//   * one activation fragment read from LDS feeds FOUR MFMAs (2 units x 2 row halves): half the LDS read volume per
//     weight byte of the 32-column waves of gemm_dec32_kernel; the dequant VALU work is interleaved into the MFMA stream
//     (as in gemm_pre64_kernel);
//   * one barrier per stage for all 12 waves; the WK k-phase tiles meet in LDS once, all 768 threads sum and store.
// Measured (profiles/r04_gemm_lc_ablations.txt, w1w3 at M = 64, 224 workgroups): the compute side alone (dequant + MFMA +
// fragment reads) runs the loop in 8.1 us at 2.22 GHz -- the matrix-pipe floor; gemm_dec32_kernel needed 12.2 -- and every PAIR
// of {weights, activations, compute} overlaps (10.4 / 11.7 / 9.3 us), but all three together take 14.3 .. 14.9 us at a shader
// clock of 1.66 .. 1.70 GHz: the launch is POWER-limited (the same cycles at 2.2 GHz would be 11 us), and underneath that the
// CU's load path moves 790 KB (278 KB of weights + 512 KB of activations, L2 -> LDS) at ~36 B/clk.  Ring depth 2 / 3 stages,
// loader priority and an activations-first prologue all measured within noise: the depth of the prefetch is not what binds.
// The tuner (tune_decode_gemms) times this shape against the 16-wave tiles per linear; it wins where the matrix side matters.
// Layout P32 (p32_layout.h) unchanged: the same HBM image serves every kernel.
#include "gemm_decode_common.h"
#include <stdlib.h>

namespace tmk {

typedef float floatx2 __attribute__((ext_vector_type(2)));

// experiment (TM_D32_ABL 0x40): the raw codes as fp16 subnormals q * 2^-24, no scale / zero point in the operand
__device__ __forceinline__ half8_t raw8_p32(uint32_t w)
{
    const uint32_t m = 0x000f000fu;
    return bit_cast<half8_t>(u32x4{w & m, (w >> 4) & m, (w >> 8) & m, (w >> 12) & m});
}

template<int MH, int PFS, int ABL = 0>
__global__ __launch_bounds__(768) void gemm_dec_lc_kernel(Dec32Params p)
{
    constexpr int NCW   = 2;  // 32-column groups per consumer wave
    constexpr int CGW   = 2;  // consumer waves side by side
    constexpr int WK    = 4;  // k-phases = k-blocks per stage
    constexpr int NCONS = CGW * WK;
    constexpr int NLOAD = 4;
    constexpr int T     = (NCONS + NLOAD) * 64;
    constexpr int ROWS  = 32 * MH;
    constexpr int KBB   = ROWS * 256;  // LDS bytes of one k-block of x
    constexpr int S     = WK;
    constexpr int STG   = S * KBB;  // one stage
    constexpr int CGT   = CGW * NCW;  // column groups per workgroup
    constexpr int NPC   = S * ROWS / 4;  // 1 KiB DMA pieces (4 rows x 256 B) per stage
    constexpr int DR    = NPC / NLOAD;
    static_assert(NPC % NLOAD == 0, "DMA pieces per loader wave");
    static_assert(WK * ROWS * CGT * 128 <= 2 * STG, "reduction image must fit the stage buffers");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid  = threadIdx.x;
    const int wgid = blockIdx.y * gridDim.x + blockIdx.x;
    if (p.dbg && tid == 0) {
        p.dbg[wgid * 8 + 0] = __builtin_amdgcn_s_memrealtime();
        p.dbg[wgid * 8 + 4] = ((uint64_t)__builtin_amdgcn_s_getreg(20 | (31 << 11)) << 32) | (uint32_t)__builtin_amdgcn_s_getreg(4 | (31 << 11));
    }
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31  = lane & 31;
    const int half = lane >> 5;

    const int kb0 = blockIdx.y * p.kb_per_split;
    const int nkb = min(p.kb_per_split, p.KB - kb0);
    const int nst = (nkb + S - 1) / S;
    const int Mloc = min(ROWS, p.M);

    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)smem);

    floatx16 acc[NCW][MH];
#pragma unroll
    for (int c = 0; c < NCW; ++c) {
#pragma unroll
        for (int h = 0; h < MH; ++h) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[c][h][r] = 0.f;
            }
        }
    }
    const int cw = wave & (CGW - 1);  // consumers: column half, k-phase
    const int wk = wave / CGW;

    if (wave >= NCONS) {
        // ---- loader: the activations of stage t+1 land in the other buffer while the consumers work on stage t ------------
        // A stage image = NPC pieces of 1 KiB (4 rows x 256 B), piece pc = r * NLOAD + lw.  The DMA writes lane L to slot L
        // of the piece, so the XOR swizzle of the image sits on the SOURCE address: lane L fetches 16-byte chunk
        // (L & 15) ^ (row & 15) of row 4 (pc % (ROWS / 4)) + (L >> 4) of k-block pc / (ROWS / 4).  Blocks past the slice
        // re-read whatever follows in the row (the consumers zero their scales); rows past M re-read row M - 1.
        const int  lw   = wave - NCONS;
        const auto rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)(((size_t)(Mloc - 1) * p.ldx + p.K) * 2), 0x00020000);
        int        doff[DR];
#pragma unroll
        for (int r = 0; r < DR; ++r) {
            const int pc  = r * NLOAD + lw;
            const int kbi = pc / (ROWS / 4);
            const int row = (pc % (ROWS / 4)) * 4 + (lane >> 4);
            const int ch  = (lane & 15) ^ (row & 15);
            doff[r]       = (ABL & 0x20) ? (kbi * ROWS + row) * 256 + ch * 16  // experiment: x k-block major ([kb][row][128])
                                          : (min(row, Mloc - 1) * p.ldx + ch * 8) * 2 + kbi * 256;
        }
        // one DMA instruction per piece; M0 = LDS byte address of the piece (saved / restored: the compiler owns M0)
#define LC_DMA_X(t, buf)                                                                                          \
    _Pragma("unroll") for (int r = 0; r < DR; ++r)                                                                \
    {                                                                                                             \
        unsigned       keep_;                                                                                     \
        const unsigned dst_ = lds0 + (buf)*STG + (r * NLOAD + lw) * 1024;                                         \
        const int      so_  = (kb0 + (t)*S) * ((ABL & 0x20) ? ROWS * 256 : 256);                                  \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"                                       \
                     "buffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"                               \
                     : "=&s"(keep_)                                                                               \
                     : "v"(doff[r]), "s"(rs_x), "s"(dst_), "s"(so_)                                               \
                     : "memory");                                                                                 \
    }
        if (nst > 0) {
            if constexpr (!(ABL & 8)) {
                LC_DMA_X(0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            for (int t = 0; t < nst; ++t) {
                if constexpr (!(ABL & 8)) {
                    if (t + 1 < nst) {
                        LC_DMA_X(t + 1, (t + 1) & 1);
                    }
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
        }
#undef LC_DMA_X
    }
    else {
        // ---- consumer ---------------------------------------------------------------------------------------------------
        int cgc[NCW];
#pragma unroll
        for (int c = 0; c < NCW; ++c) {
            cgc[c] = min((int)blockIdx.x * CGT + cw * NCW + c, p.ncg - 1);
        }
        const auto rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, (int)((size_t)p.KB * p.ncg * kP32Unit), 0x00020000);
        const int  vw   = lane * 16;
        const int  vs   = 2048 + l31 * 4;
        u32x4      ring[PFS][NCW][2];
        uint32_t   sring[PFS][NCW];
        // B fragment of 16-k step j: row (l & 31) [+ 32 h], chunk (2j + half) ^ (row & 15) = 2j ^ (half ^ (row & 15))
        const int frow = l31 * 256;
        const int fsw  = (half ^ (l31 & 15)) << 4;
        uint32_t  m1024 = 0x64006400u, m64 = 0x54005400u;
        asm volatile("" : "+v"(m1024), "+v"(m64));  // magic numbers in VGPRs: one v_and_or_b32 per pair

        // k-block b (relative to kb0; clamped past the slice, its scales are then zeroed) of this wave's units -> ring slot
#define LC_LOAD_W(slot, b)                                                                                        \
    _Pragma("unroll") for (int c = 0; c < NCW; ++c)                                                               \
    {                                                                                                             \
        const int uo_      = ((kb0 + min((b), nkb - 1)) * p.ncg + cgc[c]) * kP32Unit;                             \
        ring[slot][c][0]   = __builtin_amdgcn_raw_buffer_load_b128(rs_w, vw, uo_, /*nt*/ 2);                      \
        ring[slot][c][1]   = __builtin_amdgcn_raw_buffer_load_b128(rs_w, vw + 1024, uo_, /*nt*/ 2);               \
        sring[slot][c]     = __builtin_amdgcn_raw_buffer_load_b32(rs_w, vs, uo_, 0);                              \
    }
        if (nst > 0) {
            // issue order = consumption order: the whole ring, oldest stage first (the steady state of the loop)
#pragma unroll
            for (int u = 0; u < PFS; ++u) {
                if constexpr (!(ABL & 16)) {
                    LC_LOAD_W(u, u * S + wk);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_s_barrier();  // stage 0 of x has landed
            if (p.dbg && tid == 0) {
                p.dbg[wgid * 8 + 1] = __builtin_amdgcn_s_memrealtime();
                p.dbg[wgid * 8 + 5] = __builtin_amdgcn_s_memtime();  // shader-clock counter: (d6 - d5) / loop time = the clock
            }
            auto stage = [&](auto U, const int t) __attribute__((always_inline)) {
                constexpr int u    = decltype(U)::value;  // ring slot
                const int     b    = t * S + wk;
                const bool    live = b < nkb;
                half2_t       s2[NCW], z2[NCW];
#pragma unroll
                for (int c = 0; c < NCW; ++c) {
                    const half2_t pr = bit_cast<half2_t>(live ? sring[u][c] : 0u);
                    s2[c]            = half2_t{pr[0], pr[0]};
                    z2[c]            = half2_t{pr[1], pr[1]};
                }
                half8_t        f0[MH], f1[MH];
                const unsigned xa = lds0 + (t & 1) * STG + wk * KBB;
                auto           rd = [&](half8_t(&f)[MH], int j) __attribute__((always_inline)) {
                    const unsigned ad = xa + (unsigned)(frow + ((32 * j) ^ fsw));
#pragma unroll
                    for (int h = 0; h < MH; ++h) {
                        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f[h]) : "v"(ad), "i"(h * 8192));
                    }
                };
                auto wt = [&](half8_t(&f)[MH], auto N) __attribute__((always_inline)) {
                    constexpr int n = decltype(N)::value;
                    if constexpr (MH == 1) {
                        asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f[0]) : "i"(n));
                    }
                    else {
                        asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(f[0]), "+v"(f[1]) : "i"(n));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                };
                if constexpr (!(ABL & 4)) {
                    rd(f0, 0);
                }
                // Two dequantised fragments live (a0: first unit, a1: second).  a1(j) is built behind the MFMAs that use
                // a0(j), a0(j+1) behind those that use a1(j): an MFMA occupies the matrix pipe for 32 cycles but the issue
                // port for 4, so ~6 VALU ops fit behind each one.
                half8_t a0 = (ABL & 0x40) ? raw8_p32(ring[u][0][0][0]) : dequant8_p32(ring[u][0][0][0], s2[0], z2[0], m1024, m64), a1;
                if constexpr (ABL & 0x80) {  // experiment: the accumulators move to this group's scale (one packed multiply per pair)
#pragma unroll
                    for (int c = 0; c < NCW; ++c) {
                        const float   rt = 1.0f + (float)s2[c][0];
                        const floatx2 rr = {rt, rt};
#pragma unroll
                        for (int h = 0; h < MH; ++h) {
#pragma unroll
                            for (int r = 0; r < 8; ++r) {
                                floatx2 v        = {acc[c][h][2 * r], acc[c][h][2 * r + 1]};
                                v                = v * rr;
                                acc[c][h][2 * r] = v[0], acc[c][h][2 * r + 1] = v[1];
                            }
                        }
                    }
                }
                static_for<8>([&](auto J) {
                    constexpr int  j  = decltype(J)::value;
                    half8_t(&cur)[MH] = (j & 1) ? f1 : f0;
                    half8_t(&nxt)[MH] = (j & 1) ? f0 : f1;
                    if constexpr (!(ABL & 4)) {
                        if constexpr (j + 1 < 8) {
                            rd(nxt, j + 1);
                        }
                        wt(cur, std::integral_constant<int, (j + 1 < 8) ? MH : 0>{});
                    }
                    else {
#pragma unroll
                        for (int h = 0; h < MH; ++h) {
                            cur[h] = bit_cast<half8_t>(ring[u][h & 1][1]);
                        }
                    }
                    if constexpr (ABL & 1) {
                        a1 = bit_cast<half8_t>(ring[u][1][j >> 2]);
                    }
                    else if constexpr (ABL & 0x40) {
                        a1 = raw8_p32(ring[u][1][j >> 2][j & 3]);
                    }
                    else {
                        a1 = dequant8_p32(ring[u][1][j >> 2][j & 3], s2[1], z2[1], m1024, m64);
                    }
#pragma unroll
                    for (int h = 0; h < MH; ++h) {
                        if constexpr (ABL & 2) {
                            asm volatile("" ::"v"(a0), "v"(cur[h]));
                        }
                        else {
                            acc[0][h] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, cur[h], acc[0][h], 0, 0, 0);
                        }
                    }
                    if constexpr (j + 1 < 8) {
                        if constexpr (ABL & 1) {
                            a0 = bit_cast<half8_t>(ring[u][0][(j + 1) >> 2]);
                        }
                        else if constexpr (ABL & 0x40) {
                            a0 = raw8_p32(ring[u][0][(j + 1) >> 2][(j + 1) & 3]);
                        }
                        else {
                            a0 = dequant8_p32(ring[u][0][(j + 1) >> 2][(j + 1) & 3], s2[0], z2[0], m1024, m64);
                        }
                    }
#pragma unroll
                    for (int h = 0; h < MH; ++h) {
                        if constexpr (ABL & 2) {
                            asm volatile("" ::"v"(a1), "v"(cur[h]));
                        }
                        else {
                            acc[1][h] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, cur[h], acc[1][h], 0, 0, 0);
                        }
                    }
                    if constexpr (!(ABL & (1 | 2))) {
#pragma unroll
                        for (int g = 0; g < 2 * MH; ++g) {  // 2 MH x (1 MFMA, up to 7 VALU)
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x002, MH == 2 ? 7 : 14, 0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
                // this stage's ring slot refilled for stage t + PFS (unconditional: past the slice the last block again,
                // nobody consumes it -- a branch around a load makes hipcc's waitcnt pass over-wait afterwards)
                if constexpr (!(ABL & 16)) {
                    LC_LOAD_W(u, (t + PFS) * S + wk);
                }
                __builtin_amdgcn_s_barrier();
            };
            // whole unrolled bodies first, WITHOUT an exit inside (see gemm_dec32_kernel); the remainder runs once
            int t0 = 0;
            for (; t0 + PFS <= nst; t0 += PFS) {
                static_for<PFS>([&](auto U) { stage(U, t0 + decltype(U)::value); });
            }
            static_for<PFS>([&](auto U) {
                if (t0 + decltype(U)::value < nst) {  // uniform over the workgroup
                    stage(U, t0 + decltype(U)::value);
                }
            });
        }
#undef LC_LOAD_W
    }
    if (p.dbg && tid == 0) {
        p.dbg[wgid * 8 + 2] = __builtin_amdgcn_s_memrealtime();
        p.dbg[wgid * 8 + 6] = __builtin_amdgcn_s_memtime();
    }

    // ---- the WK k-phase partial tiles meet in LDS: red[wk][row][c4 ^ (row & 7)] (floatx4 units, CGT * 8 per row) ----------
    // consumer lane holds, per unit c, half h and register r: row m = 32h + (l & 31),
    // column 32 (cw NCW + c) + 8 (r >> 2) + 4 (l >> 5) + (r & 3).  (Every x read of the last stage is behind its barrier.)
    {
        constexpr int C4  = CGT * 8;  // floatx4 units per row
        floatx4*      red = (floatx4*)smem;
        if (wave < NCONS) {
#pragma unroll
            for (int c = 0; c < NCW; ++c) {
#pragma unroll
                for (int h = 0; h < MH; ++h) {
                    const int m = 32 * h + l31;
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const int c4 = (cw * NCW + c) * 8 + 2 * g4 + half;
                        red[(wk * ROWS + m) * C4 + (c4 ^ (m & 7))] =
                            floatx4{acc[c][h][4 * g4], acc[c][h][4 * g4 + 1], acc[c][h][4 * g4 + 2], acc[c][h][4 * g4 + 3]};
                    }
                }
            }
        }
        __syncthreads();
        if (p.dbg && tid == 0) {
            p.dbg[wgid * 8 + 7] = __builtin_amdgcn_s_memrealtime();
        }
        constexpr int NE    = ROWS * C4;  // floatx4 elements of the output tile
        const int     ncol0 = blockIdx.x * CGT * 32;
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
            for (int k = 1; k < WK; ++k) {  // fixed order: deterministic
                a += red[(k * ROWS + m) * C4 + (c4 ^ (m & 7))];
            }
            const int n = ncol0 + c4 * 4;
            if (m >= Mloc || n >= p.N) {
                continue;
            }
            if (p.epilogue == 2) {
                floatx4* dst = (floatx4*)(p.partial + ((size_t)blockIdx.y * p.M + m) * p.N + n);
                if (p.wt & 1) {
                    store_wt(dst, a, p.wt >> 4);
                }
                else {
                    *dst = a;
                }
            }
            else if (p.epilogue == 1) {
                const float s0 = a[0] / (1.0f + __builtin_expf(-a[0]));
                const float s1 = a[2] / (1.0f + __builtin_expf(-a[2]));
                half2_t     o  = {(half_t)(s0 * a[1]), (half_t)(s1 * a[3])};
                half2_t*    dst = (half2_t*)(p.y + (size_t)m * p.ldy + (n >> 1));
                if (p.wt & 2) {
                    store_wt(dst, o);
                }
                else {
                    *dst = o;
                }
            }
            else {
                half4_t  o   = {(half_t)a[0], (half_t)a[1], (half_t)a[2], (half_t)a[3]};
                half4_t* dst = (half4_t*)(p.y + (size_t)m * p.ldy + n);
                if (p.wt & 2) {
                    store_wt(dst, o);
                }
                else {
                    *dst = o;
                }
            }
        }
    }
    if (p.dbg && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        p.dbg[wgid * 8 + 3] = __builtin_amdgcn_s_memrealtime();
    }
}

template<int MH, int PFS, int ABL = 0>
static int launch_lc_one(const Dec32Params& p, dim3 grid, hipStream_t st)
{
    constexpr int lds = 2 * 4 * 32 * MH * 256;
    if (const int rc = ensure_dynamic_lds((const void*)gemm_dec_lc_kernel<MH, PFS, ABL>, lds)) {
        return rc;
    }
    gemm_dec_lc_kernel<MH, PFS, ABL><<<grid, 768, lds, st>>>(p);
    TM_HIP_CHECK(hipGetLastError());
    return 0;
}

// shape kShapeLC: 128 columns x all rows x the k slice per workgroup, 8 consumer + 4 loader waves
int launch_dec_lc(const Dec32Params& p, dim3 grid, hipStream_t st)
{
    static const int pfs = []() {
        const char* v = getenv("TM_LC_PFS");
        return v ? atoi(v) : 3;
    }();
#ifdef TM_EXPERIMENTS
    static const int abl = []() {
        const char* v = getenv("TM_D32_ABL");
        return v ? atoi(v) : -1;
    }();
    if (p.M > 32) {
        switch (abl) {
#define LC_CASE(v) case v: return launch_lc_one<2, 3, v>(p, grid, st)
            LC_CASE(0x20); LC_CASE(0x40); LC_CASE(0xc0); LC_CASE(0xe0); LC_CASE(1); LC_CASE(2); LC_CASE(4); LC_CASE(8); LC_CASE(16); LC_CASE(7); LC_CASE(15); LC_CASE(24); LC_CASE(31);
#undef LC_CASE
            default: break;
        }
    }
#endif
    if (p.M <= 32) {
        return pfs == 4 ? launch_lc_one<1, 4>(p, grid, st) : pfs == 2 ? launch_lc_one<1, 2>(p, grid, st) : launch_lc_one<1, 3>(p, grid, st);
    }
    return pfs == 4 ? launch_lc_one<2, 4>(p, grid, st) : pfs == 2 ? launch_lc_one<2, 2>(p, grid, st) : launch_lc_one<2, 3>(p, grid, st);
}

}  // namespace tmk
