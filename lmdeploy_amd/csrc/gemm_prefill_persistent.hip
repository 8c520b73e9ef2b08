// W4A16 prefill GEMM for gfx950: the 128 x 512 tile of gemm_pre64_kernel (gemm_decode.hip) as a PERSISTENT workgroup.
//
// Replaces: the large-M end of gemm::Gemm::Run (src/turbomind/kernels/gemm/gemm.cu:257-344; the reference's sm90 kernels are
//           persistent with a tile scheduler too: kernels/gemm/scheduler.cuh), dequant kernels/gemm/transform.h:34-74, gated-SiLU
//           epilogue kernels/gemm/epilogue.h:159-176.  Arithmetic, operand order and output bits = gemm_pre64_kernel's.
//
// Why (round 4, profiles/r04_prefill_gemm.txt, call 8): at M = 8192 the K = 4096 linears (w_qkv, wo, w1w3: 77 % of the prefill
// flops) spend 104 us in the k loop of a tile and ~15 us around it -- the epilogue's stores, the end of the workgroup, the
// dispatch of the next one, its prologue (descriptor set-up, first loads, first barrier): 12.6 % of the launch with the matrix
// pipe idle (w_qkv: span 357 us = 3 rounds x (104 + 15)).  Here ONE workgroup per CU walks the tiles w = blockIdx.x, + gridDim.x,
// ...: after the k loop of tile w it first ISSUES the prologue loads of tile w + gridDim.x (activation rows and the first weight
// units go to registers that are dead by then), then stores tile w from the accumulators while those loads fly, then carries on.
// No workgroup turnover, the prologue's memory latency sits under the epilogue.
// Tile order = the hardware's dispatch order of the non-persistent grid (x fastest), so the tiles that run together share
// the same weight columns / activation rows in L2 as before.
// grid = min(tiles, 256) workgroups; logical grid (gx, gy, gz) = (ceil(N / 512), splits, ceil(M / 128)) as kernel arguments.
#include "gemm_decode_common.h"
#include <stdlib.h>

namespace tmk {

__global__ __launch_bounds__(512) void gemm_pre64p_kernel(Dec32Params p, int gx, int gy, int gz)
{
    constexpr int MH = 4, CG = 8, NB = 2, T = 512, ROWS = 128;
    constexpr int KBB = ROWS * 256;  // LDS bytes of one k-block of x
    constexpr int XR  = ROWS * 16 / T;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid  = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31  = lane & 31;
    const int half = lane >> 5;

    const int total = gx * gy * gz;
    int       w     = blockIdx.x;  // wave-uniform
    if (w >= total) {
        return;
    }
    // ---- the tile a workgroup is loading / contracting (set by setup()) ----
    int cgc[NB];
    int bx = 0, by = 0, m0 = 0, Mloc = 0, kb0 = 0, nkb = 0;
    int xoff[XR];
    // the activations' buffer descriptor depends on the tile's row block: rebuilt per tile (4 SGPRs)
    auto rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, 0, 0x00020000);
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, (int)((size_t)p.KB * p.ncg * kP32Unit), 0x00020000);
    const int  vw   = lane * 16;
    const int  vs   = 2048 + l31 * 4;

    auto setup = [&](int tile) __attribute__((always_inline)) {
        const int t_x = tile % gx;
        const int t_r = tile / gx;
        const int t_y = t_r % gy;
        const int t_z = t_r / gy;
        bx            = t_x;
        by            = t_y;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            cgc[nb] = min((t_x * NB + nb) * CG + wave, p.ncg - 1);
        }
        kb0  = t_y * p.kb_per_split;
        nkb  = min(p.kb_per_split, p.KB - kb0);
        m0   = t_z * ROWS;
        Mloc = min(ROWS, p.M - m0);
        rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (size_t)m0 * p.ldx), 0, (int)(((size_t)(Mloc - 1) * p.ldx + p.K) * 2),
                                                 0x00020000);
#pragma unroll
        for (int r = 0; r < XR; ++r) {
            xoff[r] = (min((tid >> 4) + 32 * r, Mloc - 1) * p.ldx + (tid & 15) * 8) * 2;
        }
    };

    floatx16 acc[NB][MH];
    u32x4    ring[2][NB][2];
    uint32_t sring[2][NB];
    u32x4    xr[XR];
    // staging: thread -> (row = tid / 16 + 32 r, 16-byte chunk tid % 16); rows 32 apart share the swizzle term, so the LDS
    // address of piece r is xlds0 + r * 8192 (an immediate) and only the global offsets need registers
    const int xlds0 = (tid >> 4) * 256 + (((tid & 15) ^ ((tid >> 4) & 15)) << 4);
    // B fragment of 16-k step j: row (l & 31) [+ 32 h], chunk (2j + half) ^ (row & 15) = 2j ^ (half ^ (row & 15))
    const int frow = l31 * 256;
    const int fsw  = (half ^ (l31 & 15)) << 4;
    uint32_t  m1024 = 0x64006400u, m64 = 0x54005400u;
    asm volatile("" : "+v"(m1024), "+v"(m64));
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)smem);

    // k-block b (relative to kb0; clamped past the slice, its scales are then zeroed) -> ring slot
#define P64_LOAD_W(slot, b)                                                                                       \
    _Pragma("unroll") for (int nb = 0; nb < NB; ++nb)                                                             \
    {                                                                                                             \
        const int uo_       = ((kb0 + min((b), nkb - 1)) * p.ncg + cgc[nb]) * kP32Unit;                           \
        ring[slot][nb][0]   = __builtin_amdgcn_raw_buffer_load_b128(rs_w, vw, uo_, /*nt*/ 2);                     \
        ring[slot][nb][1]   = __builtin_amdgcn_raw_buffer_load_b128(rs_w, vw + 1024, uo_, /*nt*/ 2);              \
        sring[slot][nb]     = __builtin_amdgcn_raw_buffer_load_b32(rs_w, vs, uo_, 0);                             \
    }
#define P64_LOAD_X(b)                                                                                             \
    _Pragma("unroll") for (int r = 0; r < XR; ++r)                                                                \
    {                                                                                                             \
        xr[r] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, xoff[r], (kb0 + min((b), nkb - 1)) * 256, 0);         \
    }
#define P64_STORE_X(buf)                                                                                          \
    _Pragma("unroll") for (int r = 0; r < XR; ++r)                                                                \
    {                                                                                                             \
        *(u32x4*)(smem + (buf)*KBB + xlds0 + r * 8192) = xr[r];                                                   \
    }

    setup(w);
    // first half of a tile's prologue: x(0) and W(0) into registers (issue order = steady-state order, oldest first)
    P64_LOAD_X(0);
    __builtin_amdgcn_sched_barrier(0);
    P64_LOAD_W(0, 0);
    __builtin_amdgcn_sched_barrier(0);

    for (;;) {
        // ---- second half of the prologue: x(0) -> LDS buffer 0 (free: every read of the previous tile retired before its last
        // barrier), x(1) and W(1) into registers ----
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
            for (int h = 0; h < MH; ++h) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    acc[nb][h][r] = 0.f;
                }
            }
        }
        P64_STORE_X(0);
        __builtin_amdgcn_sched_barrier(0);
        P64_LOAD_X(1);
        __builtin_amdgcn_sched_barrier(0);
        P64_LOAD_W(1, 1);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();

        auto stage = [&](auto U, const int t) __attribute__((always_inline)) {
            constexpr int u    = decltype(U)::value;  // ring slot = LDS buffer = parity of t
            const bool    live = t < nkb;
            half2_t       s2[NB], z2[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const half2_t pr = bit_cast<half2_t>(live ? sring[u][nb] : 0u);
                s2[nb]           = half2_t{pr[0], pr[0]};
                z2[nb]           = half2_t{pr[1], pr[1]};
            }
            half8_t        f0[MH], f1[MH];
            const unsigned xa = lds0 + u * KBB;
            auto           rd = [&](half8_t(&f)[MH], int j) __attribute__((always_inline)) {
                const unsigned ad = xa + (unsigned)(frow + ((32 * j) ^ fsw));
#pragma unroll
                for (int h = 0; h < MH; ++h) {
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f[h]) : "v"(ad), "i"(h * 8192));
                }
            };
            auto wt = [&](half8_t(&f)[MH], auto N) __attribute__((always_inline)) {
                constexpr int n = decltype(N)::value;
                asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]) : "i"(n));
                __builtin_amdgcn_sched_barrier(0);
            };
            rd(f0, 0);
            // the dequant of the NEXT fragment is interleaved into the MFMA stream of the current one (see gemm_pre64_kernel)
            half8_t a0 = dequant8_p32(ring[u][0][0][0], s2[0], z2[0], m1024, m64), a1;
            static_for<8>([&](auto J) {
                constexpr int  j  = decltype(J)::value;
                half8_t(&cur)[MH] = (j & 1) ? f1 : f0;
                half8_t(&nxt)[MH] = (j & 1) ? f0 : f1;
                if constexpr (j + 1 < 8) {
                    rd(nxt, j + 1);
                }
                wt(cur, std::integral_constant<int, (j + 1 < 8) ? MH : 0>{});
                a1 = dequant8_p32(ring[u][1][j >> 2][j & 3], s2[1], z2[1], m1024, m64);
#pragma unroll
                for (int h = 0; h < MH; ++h) {
                    acc[0][h] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, cur[h], acc[0][h], 0, 0, 0);
                }
                if constexpr (j + 1 < 8) {
                    a0 = dequant8_p32(ring[u][0][(j + 1) >> 2][(j + 1) & 3], s2[0], z2[0], m1024, m64);
                }
#pragma unroll
                for (int h = 0; h < MH; ++h) {
                    acc[1][h] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, cur[h], acc[1][h], 0, 0, 0);
                }
#pragma unroll
                for (int g = 0; g < 2 * MH; ++g) {  // 8 x (1 MFMA, up to 4 VALU)
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            // x of k-block t+1 (loaded one stage ago) -> the other buffer, x of t+2 into the same registers, this stage's ring
            // slot refilled for t+2; all unconditional (past the slice: the last block again, nobody consumes it)
            P64_STORE_X(u ^ 1);
            P64_LOAD_X(t + 2);
            P64_LOAD_W(u, t + 2);
            __syncthreads();
        };
        int t0 = 0;
        for (; t0 + 2 <= nkb; t0 += 2) {
            static_for<2>([&](auto U) { stage(U, t0 + decltype(U)::value); });
        }
        static_for<2>([&](auto U) {
            if (t0 + decltype(U)::value < nkb) {
                stage(U, t0 + decltype(U)::value);
            }
        });

        // ---- this tile's output coordinates, then the NEXT tile's first loads, then the stores ----
        const int  e_bx = bx, e_by = by, e_m0 = m0, e_Mloc = Mloc;
        const int  wn       = w + (int)gridDim.x;
        const bool has_next = wn < total;  // wave-uniform
        if (has_next) {
            setup(wn);
            P64_LOAD_X(0);
            __builtin_amdgcn_sched_barrier(0);
            P64_LOAD_W(0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // epilogue: straight from the accumulators.  Lane holds, per half h and register r: row m = 32h + (l & 31), column
        // 32 cg + 8 (r >> 2) + 4 (l >> 5) + (r & 3)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int ncol0 = ((e_bx * NB + nb) * CG + wave) * 32;
#pragma unroll
            for (int h = 0; h < MH; ++h) {
                const int m = 32 * h + l31;
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int n = ncol0 + 8 * g4 + 4 * half;
                    if (m >= e_Mloc || n >= p.N) {
                        continue;
                    }
                    const floatx4 a  = {acc[nb][h][4 * g4], acc[nb][h][4 * g4 + 1], acc[nb][h][4 * g4 + 2], acc[nb][h][4 * g4 + 3]};
                    const size_t  mg = (size_t)e_m0 + m;
                    if (p.epilogue == 2) {
                        floatx4* dst = (floatx4*)(p.partial + ((size_t)e_by * p.M + mg) * p.N + n);
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
                        *(half2_t*)(p.y + mg * p.ldy + (n >> 1)) = o;
                    }
                    else {
                        half4_t o = {(half_t)a[0], (half_t)a[1], (half_t)a[2], (half_t)a[3]};
                        *(half4_t*)(p.y + mg * p.ldy + n) = o;
                    }
                }
            }
        }
        if (!has_next) {
            break;
        }
        w = wn;
    }
#undef P64_LOAD_W
#undef P64_LOAD_X
#undef P64_STORE_X
}

// logical grid = (ceil(N / 512), splits, ceil(M / 128)); one workgroup per CU walks it (the caller checks that there are more
// tiles than CUs -- otherwise the plain kernel is the same thing)
int launch_pre64_persistent(const Dec32Params& p, dim3 grid, hipStream_t st)
{
    constexpr int lds = 96 * 1024;  // two 32 KB stages; > 80 KB so that exactly one workgroup (8 waves x 256 registers) owns a CU
    if (const int rc = ensure_dynamic_lds((const void*)gemm_pre64p_kernel, lds)) {
        return rc;
    }
    const long total = (long)grid.x * grid.y * grid.z;
    const int  wgs   = (int)(total < 256 ? total : 256);
    gemm_pre64p_kernel<<<wgs, 512, lds, st>>>(p, (int)grid.x, (int)grid.y, (int)grid.z);
    TM_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace tmk
