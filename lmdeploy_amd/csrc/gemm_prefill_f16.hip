// Prefill GEMM over the RESIDENT fp16 image of a W4A16 linear (gfx950): 256 x 256 output tiles, both operands by LDS-DMA.
//
// Replaces: the large-M end of gemm::Gemm::Run (src/turbomind/kernels/gemm/gemm.cu:257-344; tile family kernels/gemm/kernel/
//           sm80_16816_4.cu:18-61, multi-stage main loop kernels/gemm/mainloop_sm80_v2.h) for prefill-sized forwards.  The operand is the one
//           every W4A16 kernel here builds on chip, w = h(fma(h(q), s, h(-z*s))) (kernels/gemm/transform.h:34-74) -- materialised ONCE at
//           load by dequant_p32_f16_kernel (bit for bit: tests/test_gpu_fullsize.py::test_w4a16_dequantised_image_full_size), so results
//           are those of the fused tiles to the accumulation order.
//
// Why (rounds 3-6): a compute-bound MFMA kernel on real data is POWER-limited on this chip (~1.6 GHz instead of 2.4).  The fused tiles
// spend part of that budget on the dequantisation -- gemm_pre64_kernel: 14 packed VALU ops per 4 MFMAs in every wave; gemm_pre256_kernel:
// 52 per thread and stage plus the LDS write pass -- and reach 1.10 .. 1.20 PF/s where a plain fp16 x fp16 GEMM reaches 1.4 .. 1.5
// (round 3 measured the vendor library on the same image; cdna_hip_programming.md quotes 1.32 .. 1.47 PF/s on random data for its own
// 256 x 256 LDS-DMA template).  MI355X has 288 GB of HBM: the fp16 image of an 8 B-parameter model is 14 GB.  The decode path keeps
// streaming the 4-bit image (bandwidth); prefill-sized forwards contract the fp16 image (matrix pipe), with NO VALU work in the main loop:
//   stage = 64 k: x image 256 rows x 128 B (32 KB) + w image 256 columns x 128 B (32 KB), both row-major with the 16-byte chunks
//   XOR-swizzled by (row >> 1) & 7 on the SOURCE address of the DMA (the LDS image is lane-linear), double buffered (128 KB);
//   8 waves = 2 row halves x 4 column quarters, 128 x 64 outputs each (8 accumulator tiles of 32 x 32); per 16-k step and wave 2 weight +
//   4 activation fragments (ds_read_b128, conflict-free) feed 8 v_mfma_f32_32x32x16_f16; the loop is rotated so that the last step of a
//   stage runs from registers behind the barrier; the DMA of stage t+1 has the whole of stage t to land.
// grid = (ceil(N / 256), 1, ceil(M / 256)); epilogue straight from the accumulators (fp16, gated SiLU).
#include "gemm_decode_common.h"
#include <stdlib.h>

namespace tmk {

template<int ABL = 0>
__global__ __launch_bounds__(512) void gemm_f16_256_kernel(Dec32Params p)
{
    constexpr int BM = 256, BK = 64;
    constexpr int XB  = BM * BK * 2;  // 32 KB: one operand image of one stage
    constexpr int STG = 2 * XB;
    constexpr int MH = 4, NC = 2;     // per wave: 4 row blocks of 32, 2 column groups of 32
    constexpr int XP = 4;             // DMA pieces (8 rows x 128 B) per operand, wave and stage
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid  = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31  = lane & 31;
    const int half = lane >> 5;
    const int wr   = wave >> 2;  // row half of the tile (128 rows)
    const int wc   = wave & 3;   // column quarter (64 columns)

    const int nst  = p.K / BK;
    // ---- XCD-aware rasterisation (ABL bit 32): workgroup i runs on XCD i % 8 (dispatch order, MI355X_MICROARCH.md), so the 32 workgroups an
    // XCD holds at a time are the linear ids {8 l + j}: give every XCD its own band of row blocks and walk it in 4 x 8 patches (4 row
    // blocks x 8 column tiles), so that what is resident on one L2 together shares its operand slices 8 / 4 ways.  Falls back to the plain
    // map when the tile counts do not divide (any map is correct: a bijection of the grid).
    int tm = blockIdx.z, tn = blockIdx.x;
    if constexpr ((ABL & 32) != 0) {
        const int TM = gridDim.z, TN = gridDim.x;
        if (TM % 32 == 0 || (TM % 8 == 0 && TM / 8 >= 1)) {
            const int wg   = blockIdx.z * TN + blockIdx.x;
            const int xcd  = wg & 7, loc = wg >> 3;
            const int rpb  = TM / 8;                      // row blocks of an XCD's band
            const int pm   = rpb >= 4 ? 4 : rpb;          // patch height
            const int pn   = 32 / pm;                     // patch width (column tiles); the last patch of a band may be narrower
            const int per_band_row = pm * TN;             // workgroups of one patch row (pm row blocks x all column tiles)
            const int br   = loc / per_band_row;          // which group of pm row blocks inside the band
            const int rem  = loc % per_band_row;
            const int pc   = rem / (pm * pn);             // patch column
            const int in   = rem % (pm * pn);
            const int wlast = TN - pc * pn < pn ? TN - pc * pn : pn;   // width of this patch
            tm = xcd * rpb + br * pm + in / wlast;
            tn = pc * pn + in % wlast;
            if (rpb % pm != 0) {  // ragged band: plain map
                tm = blockIdx.z, tn = blockIdx.x;
            }
        }
    }
    const int m0   = tm * BM;
    const int n0   = tn * BM;
    const int Mloc = min(BM, p.M - m0);
    const int Nloc = min(BM, p.N - n0);

    const half_t* const wimg = (const half_t*)p.wp;  // [N][K] fp16
    const auto rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (size_t)m0 * p.ldx), 0, (int)(((size_t)(Mloc - 1) * p.ldx + p.K) * 2), 0x00020000);
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)(wimg + (size_t)n0 * p.K), 0, (int)((size_t)Nloc * p.K * 2), 0x00020000);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)smem);

    // 32 DMA pieces of 1 KiB (8 rows x 128 B) per operand and stage, XP per wave: piece pc = XP wave + r, lane L fetches 16-byte chunk
    // (L & 7) ^ ((row >> 1) & 7) of row 8 pc + (L >> 3)
    int xoff[XP], woff[XP];
#pragma unroll
    for (int r = 0; r < XP; ++r) {
        const int row = 8 * (XP * wave + r) + (lane >> 3);
        const int ch  = (lane & 7) ^ ((row >> 1) & 7);
        xoff[r]       = (min(row, Mloc - 1) * p.ldx + ch * 8) * 2;
        woff[r]       = (min(row, Nloc - 1) * p.K + ch * 8) * 2;
    }
#define F16_DMA(st, buf)                                                                                          \
    _Pragma("unroll") for (int r = 0; r < XP; ++r)                                                                \
    {                                                                                                             \
        unsigned       keep_;                                                                                     \
        const unsigned dst_ = lds0 + (buf)*STG + (XP * wave + r) * 1024;                                          \
        const int      so_  = (st)*BK * 2;                                                                        \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\t"                                       \
                     "buffer_load_dwordx4 %1, %3, %6 offen lds\n\t"                                               \
                     "s_mov_b32 m0, %5\n\ts_nop 0\n\t"                                                            \
                     "buffer_load_dwordx4 %2, %7, %6 offen lds\n\ts_mov_b32 m0, %0"                               \
                     : "=&s"(keep_)                                                                               \
                     : "v"(xoff[r]), "v"(woff[r]), "s"(rs_x), "s"(dst_), "s"(dst_ + XB), "s"(so_), "s"(rs_w)      \
                     : "memory");                                                                                 \
    }

    // L2 prefetch of the stage after next (ABL bit 16): lane L touches 4 bytes of 64-byte sector (L & 15) of this wave's piece L >> 4 -- the
    // DMA of a stage is a burst of 64 KB per CU whose lines come from the Infinity Cache / HBM (a 256-column slice of the fp16 image is
    // shared by ~2 resident workgroups of the XCD): one memory latency per stage, ~1.3 us against 1.1 us of MFMA work (ablations,
    // profiles/r06_prefill_f16_image_*).  Touched a stage earlier, the lines are in the XCD's L2 when the DMA asks for them.
    const int pfr  = 8 * (XP * wave + (lane >> 4)) + ((lane & 15) >> 1);
    const int xpf  = (min(pfr, Mloc - 1) * p.ldx) * 2 + (lane & 1) * 64;
    const int wpf  = (min(pfr, Nloc - 1) * p.K) * 2 + (lane & 1) * 64;
    uint32_t  pfd0 = 0, pfd1 = 0;

    // fragment addresses: lane l reads row base + (l & 31), chunk (2j + half) ^ ((row >> 1) & 7); the swizzle term depends on l only
    const int fsw = ((l31 >> 1) & 6) << 4;
    const int fxb = (128 * wr + l31) * 128 + ((half ^ ((l31 >> 1) & 1)) << 4);       // x: + h * 4096 (32 rows)
    const int fwb = XB + (64 * wc + l31) * 128 + ((half ^ ((l31 >> 1) & 1)) << 4);   // w: + c * 4096 (32 columns)

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

    if (nst > 0) {
        F16_DMA(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        half8_t fa[2][NC] = {}, fb[2][MH] = {};
        bool    rd_done = false;
        auto    rd = [&](unsigned xa, int q, int j) __attribute__((always_inline)) {
            const unsigned sw = (unsigned)((32 * j) ^ fsw);
            const unsigned ax = xa + (unsigned)fxb + sw;
            const unsigned aw = xa + (unsigned)fwb + sw;
            if constexpr ((ABL & 64) != 0) {  // timing only: fragments are read during the first stage, then the registers keep that (real) data
                if (xa != lds0 || rd_done) {
                    asm volatile("" : "+v"(fa[q][0]), "+v"(fa[q][1]), "+v"(fb[q][0]), "+v"(fb[q][1]), "+v"(fb[q][2]), "+v"(fb[q][3]) : "v"(aw), "v"(ax));
                    return;
                }
            }
            if constexpr (ABL & 4) {  // timing only: no fragment reads (the registers keep whatever they hold)
                asm volatile("" : "+v"(fa[q][0]), "+v"(fa[q][1]), "+v"(fb[q][0]), "+v"(fb[q][1]), "+v"(fb[q][2]), "+v"(fb[q][3]) : "v"(aw), "v"(ax));
                return;
            }
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
            asm volatile("s_waitcnt lgkmcnt(%6)"
                         : "+v"(fa[q][0]), "+v"(fa[q][1]), "+v"(fb[q][0]), "+v"(fb[q][1]), "+v"(fb[q][2]), "+v"(fb[q][3])
                         : "i"(n));
            __builtin_amdgcn_sched_barrier(0);
        };
        auto mma = [&](auto Q) __attribute__((always_inline)) {
            constexpr int q = decltype(Q)::value;
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
            __builtin_amdgcn_sched_barrier(0);
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using IR = std::integral_constant<int, NC + MH>;
        rd(lds0, 0, 0);
        for (int t = 0; t < nst; ++t) {
            const int      buf = t & 1;
            const unsigned xa  = lds0 + buf * STG;
            // top of stage t: no VMEM in flight; every read of the other buffer retired before the barrier at the end of stage t-1
            if (t + 1 < nst && !(ABL & 8)) {
                F16_DMA(t + 1, buf ^ 1);
            }
            if constexpr ((ABL & 16) != 0) {
                const int so2 = min(t + 2, nst - 1) * BK * 2;
                asm volatile("buffer_load_dword %0, %2, %4, %6 offen sc1\n\tbuffer_load_dword %1, %3, %5, %6 offen sc1"
                             : "=&v"(pfd0), "=&v"(pfd1)
                             : "v"(xpf), "v"(wpf), "s"(rs_x), "s"(rs_w), "s"(so2)
                             : "memory");
            }
            rd(xa, 1, 1);
            wt(I0{}, IR{});
            mma(I0{});
            rd(xa, 0, 2);
            wt(I1{}, IR{});
            mma(I1{});
            rd(xa, 1, 3);
            wt(I0{}, IR{});
            mma(I0{});
            wt(I1{}, I0{});  // step 3's fragments are in registers: every LDS read of this buffer has retired
            rd_done = true;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the pieces of stage t+1 (issued a stage ago)
            if constexpr ((ABL & 16) != 0) {
                asm volatile("" ::"v"(pfd0), "v"(pfd1));  // the prefetch registers stay reserved until their loads have landed
            }
            __syncthreads();
            if (t + 1 < nst) {
                rd(lds0 + (buf ^ 1) * STG, 0, 0);  // first fragments of stage t+1 ...
            }
            mma(I1{});  // ... behind them, step 3 of stage t from registers
        }
    }
#undef F16_DMA
    // ---- epilogue: straight from the accumulators.  Lane holds, per column group c, row block h and register r: row
    // m = 128 wr + 32 h + (l & 31), column 64 wc + 32 c + 8 (r >> 2) + 4 (l >> 5) + (r & 3)
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int ncol0 = n0 + 64 * wc + 32 * c;
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
                if (p.epilogue == 1) {
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
}

// shape kShapeF16: grid = (ceil(N / 256), 1, ceil(M / 256)); p.wp = the fp16 [N][K] image
int launch_f16_256(const Dec32Params& p, dim3 grid, hipStream_t st)
{
    constexpr int lds = 2 * 2 * 256 * 64 * 2;
    TM_REQUIRE(p.K % 64 == 0 && p.N % 4 == 0, "fp16-image prefill tile: K % 64 == 0, N % 4 == 0");
    static const int abl = [] {
        const char* v = getenv("TM_F16_ABL");
        return v ? atoi(v) : 0;
    }();
#define F16_CASE(v)                                                                            \
    if (abl == v) {                                                                            \
        if (const int rc = ensure_dynamic_lds((const void*)gemm_f16_256_kernel<v>, lds)) {     \
            return rc;                                                                         \
        }                                                                                      \
        gemm_f16_256_kernel<v><<<grid, 512, lds, st>>>(p);                                     \
        TM_HIP_CHECK(hipGetLastError());                                                       \
        return 0;                                                                              \
    }
    F16_CASE(2) F16_CASE(4) F16_CASE(8) F16_CASE(12) F16_CASE(6) F16_CASE(16) F16_CASE(18) F16_CASE(32) F16_CASE(34) F16_CASE(64) F16_CASE(72)
#undef F16_CASE
    if (const int rc = ensure_dynamic_lds((const void*)gemm_f16_256_kernel<0>, lds)) {
        return rc;
    }
    gemm_f16_256_kernel<0><<<grid, 512, lds, st>>>(p);
    TM_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace tmk
