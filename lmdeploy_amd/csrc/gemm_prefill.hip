// W4A16 prefill GEMM for gfx950: 256 x 256 output tiles, the u4 weights dequantised ONCE per workgroup tile THROUGH LDS.
//
// Replaces: the large-M end of gemm::Gemm::Run (src/turbomind/kernels/gemm/gemm.cu:257-344, tile family
//           kernels/gemm/kernel/sm80_16816_4.cu:18-61), dequant kernels/gemm/transform.h:34-74, gated-SiLU epilogue
//           kernels/gemm/epilogue.h:159-176.  Same operand as every other W4A16 kernel here: w = h(fma(h(q), s, h(-z*s))).
//
// Why (round 3 / 4 measurements): a compute-bound MFMA kernel on real data is POWER-limited (~1.6 GHz instead of
// 2.4), and in gemm_pre64_kernel every wave dequantises its own 64 columns for 128 rows -- 13 packed VALU ops per 4 MFMAs, each
// of them power the matrix pipe does not get (1.10 .. 1.24 PF/s against 1.42 .. 1.49 of a plain fp16 library GEMM).  Here a
// 64-deep k-slice of 256 columns is dequantised once per workgroup (4 dwords = 52 VALU ops per thread and stage, 1.6 per MFMA)
// into an fp16 LDS image in A-fragment order, the activations of 256 rows arrive by LDS-DMA (swizzle on the source address),
// and the 8 waves (2 row halves x 4 column quarters, 128 x 64 outputs each) read BOTH operands from LDS:
//   stage = 64 k:  x image 256 rows x 128 B (32 KB) + w image 8 column groups x 4 k-steps x 1 KB (32 KB), double buffered;
//   per 16-k step and wave: 2 weight + 4 activation fragments (ds_read_b128) feed 8 v_mfma_f32_32x32x16_f16;
//   the next stage's weights are fetched to registers a stage ahead, dequantised between the MFMAs (register-only VALU) and
//   written to the other buffer at the end of the stage; one barrier per stage.
// grid = (ceil(N / 256), splits, ceil(M / 256)); epilogue straight from the accumulators (fp16, gated SiLU, fp32 slabs).
#include "gemm_decode_common.h"
#include <stdlib.h>

namespace tmk {

// NW = 8: 2 row halves x 4 column quarters of waves (128 x 64 outputs each, two waves per SIMD);
// NW = 4: 2 x 2 waves of 128 x 128 outputs (one wave per SIMD, 256 accumulator registers): 8 fragment reads per 16 MFMAs
// instead of 6 per 8 -- a third less LDS traffic per MFMA.
template<int NW, int ABL = 0>
__global__ __launch_bounds__(NW * 64) void gemm_pre256_kernel(Dec32Params p)
{
    constexpr int BM = 256, BK = 64;
    constexpr int XB  = BM * BK * 2;      // 32 KB: x image of one stage, row-major 128-B rows, 16-B chunks XOR-swizzled
    constexpr int WB  = 8 * 4 * 1024;     // 32 KB: w image of one stage: [column group 8][k-step 4][lane 64][16 B]
    constexpr int STG = XB + WB;
    constexpr int MH = 4, NC = 16 / NW;   // per wave: 4 row blocks of 32, NC column groups of 32
    constexpr int NWC = 8 / NC;           // waves side by side
    constexpr int SG  = 8 / NW;           // column groups a wave dequantises per stage
    constexpr int XP  = 32 / NW;          // DMA pieces of x per wave and stage
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid  = threadIdx.x;
    const int wgid = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    if (p.dbg && tid == 0) {
        p.dbg[wgid * 8 + 0] = __builtin_amdgcn_s_memrealtime();
    }
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31  = lane & 31;
    const int half = lane >> 5;
    const int wr   = wave / NWC;  // row half of the tile (128 rows)
    const int wc   = wave % NWC;  // column part (NC x 32 columns)

    const int kb0  = blockIdx.y * p.kb_per_split;
    const int nkb  = min(p.kb_per_split, p.KB - kb0);
    const int nst  = 2 * nkb;  // stages of 64 k
    const int m0   = blockIdx.z * BM;
    const int Mloc = min(BM, p.M - m0);

    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, (int)((size_t)p.KB * p.ncg * kP32Unit), 0x00020000);
    const auto rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (size_t)m0 * p.ldx), 0,
                                                        (int)(((size_t)(Mloc - 1) * p.ldx + p.K) * 2), 0x00020000);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)smem);

    // ---- staging roles ------------------------------------------------------------------------------------------------------
    // weights: wave w dequantises column groups SG w .. of the tile (32 columns x 64 k = lane's 4 dwords: the k-steps of a half unit)
    int cg_st[SG];
#pragma unroll
    for (int g = 0; g < SG; ++g) {
        cg_st[g] = min((int)blockIdx.x * 8 + wave * SG + g, p.ncg - 1);
    }
    const int vw    = lane * 16;
    const int vs    = 2048 + l31 * 4;
    // activations: 32 DMA pieces of 1 KiB (8 rows x 128 B) per stage, XP per wave: piece pc = XP wave + r, lane L fetches 16-byte
    // chunk (L & 7) ^ ((row >> 1) & 7) of row 8 pc + (L >> 3) -- the image is lane-linear, the swizzle sits on the source
    int xoff[XP];
#pragma unroll
    for (int r = 0; r < XP; ++r) {
        const int row = 8 * (XP * wave + r) + (lane >> 3);
        const int ch  = (lane & 7) ^ ((row >> 1) & 7);
        xoff[r]       = (min(row, Mloc - 1) * p.ldx + ch * 8) * 2;
    }
#define P256_DMA_X(st, buf)                                                                                       \
    _Pragma("unroll") for (int r = 0; r < XP; ++r)                                                                \
    {                                                                                                             \
        unsigned       keep_;                                                                                     \
        const unsigned dst_ = lds0 + (buf)*STG + (XP * wave + r) * 1024;                                          \
        const int      so_  = (kb0 * 128 + (st)*BK) * 2;                                                          \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"                                       \
                     "buffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"                               \
                     : "=&s"(keep_)                                                                               \
                     : "v"(xoff[r]), "s"(rs_x), "s"(dst_), "s"(so_)                                               \
                     : "memory");                                                                                 \
    }
    // half unit of stage st (k-block st / 2, half st & 1) of this wave's staging column group -> registers.  Inline asm: beside
    // the LDS-DMA (invisible to hipcc's waitcnt pass) a compiler-counted load would be waited for with the wrong count
    // (cdna_hip_programming.md 5.7); every load of the loop is retired by the ONE hand-written vmcnt(0) at the end of a stage.
    u32x4    wq[SG], wn[SG];
    uint32_t wsz[SG], wszn[SG];
#define P256_LOAD_W(st, q_, sz_)                                                                                  \
    _Pragma("unroll") for (int g = 0; g < SG; ++g)                                                                \
    {                                                                                                             \
        const int st_ = min((st), nst - 1);                                                                       \
        const int uo_ = ((kb0 + (st_ >> 1)) * p.ncg + cg_st[g]) * kP32Unit + (st_ & 1) * 1024;                    \
        const int vs_ = vs - (st_ & 1) * 1024;                                                                    \
        asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %2, %4, %5 offen\n\tbuffer_load_dword %1, %3, %4, %5 offen"  \
                     : "=&v"(q_[g]), "=&v"(sz_[g])                                                                \
                     : "v"(vw), "v"(vs_), "s"(rs_w), "s"(uo_)                                                     \
                     : "memory");                                                                                 \
    }
    // a wait that names the ring registers it retires (no use of them can be scheduled above it)
#define P256_WAIT(n_, q_, sz_)                                                                                    \
    if constexpr (SG == 1) {                                                                                      \
        asm volatile("s_waitcnt vmcnt(%2)" : "+v"(q_[0]), "+v"(sz_[0]) : "i"(n_) : "memory");                    \
    }                                                                                                             \
    else {                                                                                                        \
        asm volatile("s_waitcnt vmcnt(%4)" : "+v"(q_[0]), "+v"(sz_[0]), "+v"(q_[1]), "+v"(sz_[1]) : "i"(n_) : "memory"); \
    }
    uint32_t m1024 = 0x64006400u, m64 = 0x54005400u;
    asm volatile("" : "+v"(m1024), "+v"(m64));

    // ---- fragment addresses ---------------------------------------------------------------------------------------------------
    // x: lane l reads row 128 wr + 32 h + (l & 31), chunk (2j + half) ^ ((row >> 1) & 7); (row >> 1) & 7 does not depend on h
    const int fx  = (128 * wr + l31) * 128 + ((half ^ ((l31 >> 1) & 1)) << 4);  // + ((2j) ^ (sw & 6)) << 4
    const int fsw = ((l31 >> 1) & 6) << 4;
    // w: column group NC wc + c, k-step j: [cg][j][lane]
    const int fw = XB + (NC * wc) * 4096 + lane * 16;

    floatx16 acc[NC][MH];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
#pragma unroll
        for (int h = 0; h < MH; ++h) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[c][h][r] = 0.f;
            }
        }
    }

    auto dequant_store = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int g = 0; g < SG; ++g) {
            const half2_t pr = bit_cast<half2_t>(wsz[g]);
            const half2_t s2 = {pr[0], pr[0]};
            const half2_t z2 = {pr[1], pr[1]};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const half8_t a = dequant8_p32(wq[g][j], s2, z2, m1024, m64);
                *(half8_t*)(smem + buf * STG + XB + (wave * SG + g) * 4096 + j * 1024 + lane * 16) = a;
            }
        }
    };

    if (nst > 0) {
        // prologue: stage 0 into buffer 0 (x by DMA, w through registers), stage 1's weights into registers
        P256_DMA_X(0, 0);
        P256_LOAD_W(0, wq, wsz);
        P256_LOAD_W(1, wn, wszn);
        P256_WAIT(2 * SG, wq, wsz);  // stage 0 (x pieces + weights); stage 1's weights fly on
        dequant_store(0);
        P256_WAIT(0, wn, wszn);
#pragma unroll
        for (int g = 0; g < SG; ++g) {
            wq[g]  = wn[g];
            wsz[g] = wszn[g];
        }
        __syncthreads();
        if (p.dbg && tid == 0) {
            p.dbg[wgid * 8 + 1] = __builtin_amdgcn_s_memrealtime();
            p.dbg[wgid * 8 + 5] = __builtin_amdgcn_s_memtime();
        }
        // Fragment registers live across the stages: the loop is ROTATED -- the MFMAs of a stage's last 16-k step run AFTER the stage
        // barrier, from registers, behind the first fragment reads of the next stage, so the matrix pipe works through the LDS
        // round trip that follows every barrier (measured before the rotation: 3045 cycles per stage against 2048 of MFMA work).
        half8_t fa[2][NC], fb[2][MH];
        auto    rd = [&](unsigned xa, int q, int j) __attribute__((always_inline)) {
            const unsigned ax = xa + (unsigned)(fx + ((32 * j) ^ fsw));
            const unsigned aw = xa + (unsigned)(fw + j * 1024);
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[q][c]) : "v"(aw), "i"(c * 4096));
            }
#pragma unroll
            for (int h = 0; h < MH; ++h) {
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[q][h]) : "v"(ax), "i"(h * 4096));
            }
        };
        auto wt = [&](auto Q, auto N) __attribute__((always_inline)) {  // retire buffer q's reads: at most n younger LDS operations stay in flight
            constexpr int q = decltype(Q)::value, n = decltype(N)::value;
            if constexpr (NC == 2) {
                asm volatile("s_waitcnt lgkmcnt(%6)"
                             : "+v"(fa[q][0]), "+v"(fa[q][1]), "+v"(fb[q][0]), "+v"(fb[q][1]), "+v"(fb[q][2]), "+v"(fb[q][3])
                             : "i"(n));
            }
            else {
                asm volatile("s_waitcnt lgkmcnt(%8)"
                             : "+v"(fa[q][0]), "+v"(fa[q][1]), "+v"(fa[q][NC - 2]), "+v"(fa[q][NC - 1]), "+v"(fb[q][0]), "+v"(fb[q][1]),
                               "+v"(fb[q][2]), "+v"(fb[q][3])
                             : "i"(n));
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        auto mma = [&](auto Q, auto NVALU) __attribute__((always_inline)) {
            constexpr int q = decltype(Q)::value, nv = decltype(NVALU)::value;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
#pragma unroll
                for (int h = 0; h < MH; ++h) {
                    if constexpr (ABL & 2) {
                        asm volatile("" ::"v"(fa[q][c]), "v"(fb[q][h]));
                    }
                    else {
                        acc[c][h] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[q][c], fb[q][h], acc[c][h], 0, 0, 0);
                    }
                }
            }
            if constexpr (!(ABL & 2) && nv > 0) {
#pragma unroll
                for (int g = 0; g < NC * MH; ++g) {  // NC MH x (1 MFMA, up to nv VALU)
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, nv, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
        using I4 = std::integral_constant<int, 4>;
        using IR = std::integral_constant<int, NC + MH>;
        rd(lds0, 0, 0);
        for (int t = 0; t < nst; ++t) {
            const int      buf = t & 1;
            const unsigned xa  = lds0 + buf * STG;
            // top of stage t: no VMEM in flight.  (wq, wsz) = the weights of stage t+1.  Issue the weights of stage t+2 and the x pieces of
            // stage t+1 (into the other buffer: every read of it retired before the barrier at the end of stage t-1): both have the
            // whole stage to land.
            P256_LOAD_W(t + 2, wn, wszn);
            if (t + 1 < nst) {
                P256_DMA_X(t + 1, buf ^ 1);
            }
            half2_t s2[SG], z2[SG];
#pragma unroll
            for (int g = 0; g < SG; ++g) {
                const half2_t pr = bit_cast<half2_t>(wsz[g]);
                s2[g]            = half2_t{pr[0], pr[0]};
                z2[g]            = half2_t{pr[1], pr[1]};
            }
            half8_t wd[SG][4];
            // 16-k steps 0 .. 2: the next step's fragments are requested before this step's are waited for; the NEXT stage's weights are
            // dequantised (register-only VALU) between the MFMAs: one dword per staged column group in steps 0 and 1, two in step 2
            rd(xa, 1, 1);
            wt(I0{}, IR{});
#pragma unroll
            for (int g = 0; g < SG; ++g) {
                wd[g][0] = dequant8_p32(wq[g][0], s2[g], z2[g], m1024, m64);
            }
            mma(I0{}, I2{});
            rd(xa, 0, 2);
            wt(I1{}, IR{});
#pragma unroll
            for (int g = 0; g < SG; ++g) {
                wd[g][1] = dequant8_p32(wq[g][1], s2[g], z2[g], m1024, m64);
            }
            mma(I1{}, I2{});
            rd(xa, 1, 3);
            wt(I0{}, IR{});
#pragma unroll
            for (int g = 0; g < SG; ++g) {
                wd[g][2] = dequant8_p32(wq[g][2], s2[g], z2[g], m1024, m64);
                wd[g][3] = dequant8_p32(wq[g][3], s2[g], z2[g], m1024, m64);
            }
            mma(I0{}, I4{});
            wt(I1{}, I0{});  // step 3's fragments are in registers: every LDS read of this buffer has retired
            // the dequantised weights of stage t+1 -> the other buffer (unconditional: behind the last stage nobody reads them,
            // and a branch here lets hipcc sink the whole dequant out of the MFMA stream into it)
#pragma unroll
            for (int g = 0; g < SG; ++g) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    *(half8_t*)(smem + (buf ^ 1) * STG + XB + (wave * SG + g) * 4096 + j * 1024 + lane * 16) = wd[g][j];
                }
            }
            P256_WAIT(0, wn, wszn);  // x pieces of stage t+1, weights of stage t+2
#pragma unroll
            for (int g = 0; g < SG; ++g) {
                wq[g]  = wn[g];
                wsz[g] = wszn[g];
            }
            __syncthreads();
            if (t + 1 < nst) {
                rd(lds0 + (buf ^ 1) * STG, 0, 0);  // first fragments of stage t+1 ...
            }
            mma(I1{}, I0{});  // ... behind them, step 3 of stage t from registers
        }
    }
#undef P256_DMA_X
#undef P256_LOAD_W
#undef P256_WAIT
    if (p.dbg && tid == 0) {
        p.dbg[wgid * 8 + 2] = __builtin_amdgcn_s_memrealtime();
        p.dbg[wgid * 8 + 6] = __builtin_amdgcn_s_memtime();
        p.dbg[wgid * 8 + 7] = p.dbg[wgid * 8 + 2];
    }
    // ---- epilogue: straight from the accumulators.  Lane holds, per unit c, row block h and register r: row
    // m = 128 wr + 32 h + (l & 31), column 32 cg + 8 (r >> 2) + 4 (l >> 5) + (r & 3)
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int ncol0 = ((int)blockIdx.x * 8 + NC * wc + c) * 32;
#pragma unroll
        for (int h = 0; h < MH; ++h) {
            const int m = 128 * wr + 32 * h + l31;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int n = ncol0 + 8 * g4 + 4 * half;
                if (m >= Mloc || n >= p.N) {
                    continue;
                }
                const floatx4 a  = {acc[c][h][4 * g4], acc[c][h][4 * g4 + 1], acc[c][h][4 * g4 + 2], acc[c][h][4 * g4 + 3]};
                const size_t  mg = (size_t)m0 + m;
                if (p.epilogue == 2) {
                    floatx4* dst = (floatx4*)(p.partial + ((size_t)blockIdx.y * p.M + mg) * p.N + n);
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
    if (p.dbg && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        p.dbg[wgid * 8 + 3] = __builtin_amdgcn_s_memrealtime();
    }
}

// shape kShapePre256: grid = (ceil(ncg / 8), splits, ceil(M / 256))
int launch_pre256(const Dec32Params& p, dim3 grid, hipStream_t st)
{
    constexpr int lds = 2 * (256 * 64 * 2 + 8 * 4 * 1024);
    static const int nw = [] {
        const char* v = getenv("TM_PRE256_WAVES");
        return v ? atoi(v) : 8;
    }();
    if (nw == 4) {
        if (const int rc = ensure_dynamic_lds((const void*)gemm_pre256_kernel<4>, lds)) {
            return rc;
        }
        gemm_pre256_kernel<4><<<grid, 256, lds, st>>>(p);
        TM_HIP_CHECK(hipGetLastError());
        return 0;
    }
    if (const int rc = ensure_dynamic_lds((const void*)gemm_pre256_kernel<8>, lds)) {
        return rc;
    }
    gemm_pre256_kernel<8><<<grid, 512, lds, st>>>(p);
    TM_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace tmk
