// W4A16 decode GEMM (M <= 64 rows) for gfx950: the weight-streaming kernel of the decode step.
//
// Replaces: LlamaLinear::Forward -> gemm::Gemm::Run for the decode batch (src/turbomind/models/llama/LlamaLinear.cu:140-216,
//           kernels/gemm/gemm.cu:257-344; tile family kernels/gemm/arch/config_sm80_s16816.h:108-136), dequant
//           kernels/gemm/transform.h:34-74, gated-SiLU epilogue kernels/gemm/epilogue.h:159-176.
// Arithmetic is the one of gemm_w4a16.hip: w = h(fma(h(q), s, h(-z*s))), fp32 MFMA accumulation, one rounding to fp16.
//
// Why a second kernel (round-1 measurements): with 8 waves per CU moving in lockstep through one barrier
// per k-block, the per-CU costs of a 16-column x 128-k weight tile ADD UP -- HBM 109 clk (at 10 B/clk/CU), MFMA 64,
// x fragments out of LDS 64, dequant VALU 26..52, x staging -- to ~290 clk, which is what was measured (0.19 of the
// HBM roofline).  This kernel is built so that they overlap instead:
//   * v_mfma_f32_32x32x16_f16, Y^T = W^T X^T: a wave owns 32 columns and all M rows, so one x fragment read from LDS
//     feeds 32 columns instead of 16 -> half the LDS read volume per weight byte (32 clk per tile);
//   * 16 waves per CU (4 per SIMD, <= 128 VGPRs): while one wave's MFMAs occupy a SIMD's matrix pipe the other three
//     dequantise / read LDS / wait for HBM -- in-order issue inside a wave no longer serialises the phases;
//   * the workgroup = CG column groups x WK k-phases: 128 columns per CU keep >= 224 workgroups alive for the wide
//     w1w3 while the 4 k-phases are summed ON CHIP (through LDS, once per kernel) instead of through fp32 slabs;
//   * x goes global -> registers -> LDS in stages of S k-blocks, double buffered, ONE barrier per stage (not per
//     k-block); the loads of stage t+1 are issued at the top of stage t and written after its compute;
//   * weights + (s, -z*s) of one (k-block, 32-column group) are ONE contiguous 2176-byte unit (layout "P32" below):
//     one buffer descriptor, scalar unit offsets, `nt` loads into a per-wave register ring PF k-blocks deep;
//   * epilogue: the WK partial tiles meet in LDS (XOR-swizzled, conflict-free), all threads then sum them in a fixed
//     order and store whole row segments (256..512 B) -- fp16, gated-SiLU fp16, or fp32 split-K slabs.
//
// Layout P32 (built once by repack_p32_kernel at load, LinearWeight::prepare's role, models/linear_weight.cc:101-324):
//   unit (kb, cg) at byte ((kb * N/32) + cg) * 2176:
//     [0, 2048)    dword d = (p*64 + lane)*4 + jj  (p = 0..1, jj = 0..3): j = 4p + jj is the 16-k step of the k-block,
//                  lane l holds column 32cg + (l & 31), k = 128kb + 16j + 8(l >> 5) + e, e = 0..7 in nibble order
//                  [k0,k2,k4,k6,k1,k3,k5,k7] -- exactly the A operand of v_mfma_f32_32x32x16_f16 (A[i = l&31][k = 8(l>>5)+e]);
//     [2048, 2176) 32 x (s, -z*s) half2 pairs of the group's columns.
#include "tm_common.h"
#include "tm_kernels.h"
#include "p32_layout.h"
#include "gemm_decode_common.h"
#include <stdlib.h>
#include <algorithm>
#include <type_traits>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <stdio.h>

namespace tmk {

__global__ void repack_p32_kernel(uint32_t* __restrict__ out, const int32_t* __restrict__ qw, const half_t* __restrict__ scales,
                                  const half_t* __restrict__ zeros, int K, int N)
{
    const size_t idx   = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int    ncg   = N / 32;
    const size_t total = (size_t)(K / 128) * ncg * (kP32Unit / 4);
    if (idx >= total) {
        return;
    }
    const size_t unit = idx / (kP32Unit / 4);
    const int    d    = (int)(idx % (kP32Unit / 4));
    const int    cg   = (int)(unit % ncg);
    const int    kb   = (int)(unit / ncg);
    if (d >= 512) {  // (s, -z*s), one fp16 rounding of the product (cast.cu:151-156)
        const int    n  = cg * 32 + (d - 512);
        const half_t s  = scales[(size_t)kb * N + n];
        const half_t z  = zeros[(size_t)kb * N + n];
        const half_t zs = (-z) * s;
        out[idx]        = bit_cast<uint32_t>(half2_t{s, zs});
        return;
    }
    const int jj   = d & 3;
    const int lane = (d >> 2) & 63;
    const int p    = d >> 8;
    const int j    = 4 * p + jj;
    const int n    = cg * 32 + (lane & 31);
    const int k0   = kb * 128 + 16 * j + 8 * (lane >> 5);
    uint32_t  w    = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const uint32_t word = (uint32_t)qw[(size_t)(k0 + e) * (N / 8) + (n >> 3)];
        const uint32_t q    = (word >> (4 * (n & 7))) & 15u;
        const int      nib  = (e & 1) ? 4 + (e >> 1) : (e >> 1);
        w |= q << (4 * nib);
    }
    out[idx] = w;
}

// The fp16 [N][K] image of a P32 linear (operator level, tm_linear_dequant_f16): bit for bit the operand the GEMM kernels build.
// One wave per P32 unit (32 columns x 128 k): the lane dequantises its two 16-byte pieces exactly as the GEMM's fragment
// pipeline does (8 x dequant8_p32 -> row l & 31, k = 16 j + 8 (l >> 5) + e), the 32 x 128 fp16 tile is transposed through a
// wave-private LDS image and leaves as whole 256-byte row segments of the [N][K] image.
__global__ __launch_bounds__(256) void dequant_p32_f16_kernel(half_t* __restrict__ out, const char* __restrict__ wp, int KB, int ncg)
{
    constexpr int kRow = 256 + 16;  // bytes per image row (+16: rows 4 banks apart -> the 16-byte column writes do not collide)
    __shared__ __attribute__((aligned(16))) char smem[4 * 32 * kRow];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int cg   = blockIdx.x * 4 + wave;
    const int kb   = blockIdx.y;
    if (cg >= ncg) {
        return;
    }
    const char*    unit = wp + ((size_t)kb * ncg + cg) * kP32Unit;
    const u32x4    w0   = *(const u32x4*)(unit + lane * 16);
    const u32x4    w1   = *(const u32x4*)(unit + 1024 + lane * 16);
    const half2_t  pr   = *(const half2_t*)(unit + 2048 + (lane & 31) * 4);
    const half2_t  s2   = {pr[0], pr[0]};
    const half2_t  z2   = {pr[1], pr[1]};
    const uint32_t m1024 = 0x64006400u, m64 = 0x54005400u;
    char*          img   = smem + wave * 32 * kRow;
    char*          mine  = img + (lane & 31) * kRow + (lane >> 5) * 16;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        *(half8_t*)(mine + j * 32) = dequant8_p32(j < 4 ? w0[j & 3] : w1[j & 3], s2, z2, m1024, m64);
    }
    __builtin_amdgcn_wave_barrier();  // wave-private image, in-order LDS pipe: no workgroup barrier
    const size_t K = (size_t)KB * 128;
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int idx = it * 64 + lane;
        const int row = idx >> 4, chunk = idx & 15;
        const u32x4 v = *(const u32x4*)(img + row * kRow + chunk * 16);
        __builtin_nontemporal_store(v, (u32x4*)(out + ((size_t)cg * 32 + row) * K + (size_t)kb * 128 + chunk * 8));
    }
}

int launch_dequant_p32_f16(half_t* out_nk, const LinearWeight& w, hipStream_t st)
{
    TM_REQUIRE(w.type == 0 && w.packed32 != nullptr && w.N % 32 == 0 && w.K % 128 == 0, "fp16 image: a u4 linear with its P32 image");
    const int ncg = w.N / 32, KB = w.K / 128;
    dequant_p32_f16_kernel<<<dim3((ncg + 3) / 4, KB), 256, 0, st>>>(out_nk, (const char*)w.packed32, KB, ncg);
    TM_HIP_CHECK(hipGetLastError());
    return 0;
}

size_t p32_bytes(int K, int N)
{
    return (size_t)(K / 128) * (N / 32) * kP32Unit;
}

int launch_repack_p32(void* out, const int32_t* qweight, const half_t* scales, const half_t* zeros, int K, int N, hipStream_t st)
{
    const size_t total = p32_bytes(K, N) / 4;
    repack_p32_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>((uint32_t*)out, qweight, scales, zeros, K, N);
    TM_HIP_CHECK(hipGetLastError());
    return 0;
}

// MH: 32-row halves of the batch (1: M <= 32, 2: M <= 64).  CG x WK waves.  S k-blocks per LDS stage (S % WK == 0).
// PF: ring depth in k-blocks per wave, a multiple of 2 * S / WK (the unrolled body covers PF / (S / WK) stages, an even
// number, so that ring slots and the LDS buffer parity are compile-time constants).
// ABL high bits select structure variants (results stay correct): 0x100 x staging by LDS-DMA (buffer_load ... lds, no
// VGPR round trip, no ds_write: asynchronous, overlaps the stage's compute), 0x200 scheduling fence that keeps the LDS
// fragment reads of step j+1 ahead of the MFMAs of step j, 0x400 s_setprio around the MFMAs.
// ABL (timing experiments only, results are garbage): 1 no dequant, 2 no MFMA, 4 no LDS fragment reads, 8 no x staging,
// 16 no weight loads inside the loop.
template<int MH, int CG, int WK, int S, int PF, int ABL = 0>
__global__ __launch_bounds__(CG* WK * 64) void gemm_dec32_kernel(Dec32Params p)
{
    constexpr int WAVES = CG * WK;
    constexpr int T     = WAVES * 64;
    constexpr int ROWS  = 32 * MH;
    constexpr int KBB   = ROWS * 256;  // LDS bytes of one k-block of x
    constexpr int STG   = S * KBB;     // one stage
    constexpr int BPS   = S / WK;      // k-blocks per wave per stage
    constexpr int UNR   = PF / BPS;    // stages per unrolled body
    constexpr int NCH   = S * ROWS * 16;
    constexpr int XR    = NCH / T;
    static_assert(S % WK == 0 && PF % BPS == 0 && UNR % 2 == 0 && NCH % T == 0, "tile parameters");
    constexpr int REDB = WK * ROWS * CG * 128;  // reduction image: [wk][row][CG * 32 floats]
    static_assert(REDB <= 2 * STG || REDB <= 160 * 1024, "reduction image must fit");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid  = threadIdx.x;
    const int wgid = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    // (Round 5 measured an XCD-aware re-deal of split-K grids -- one slice's workgroups on 8 / slices XCDs, so that each L2 pulls only its
    // slice's columns of x: FETCH 1.40 -> 1.28 x the weight bytes and the step 1.8 % SLOWER, because the plain mapping below keeps the
    // slices of one column tile on ONE XCD, blockIdx.x % 8, which is what the slab merge wants.  Removed;
    // profiles/r05_fold_modes_and_xcd_placement_ab.txt.)
    const int bx = blockIdx.x, by = blockIdx.y;
    // every kernel argument the prologue needs, fetched by ONE batch of scalar loads at entry: left to itself hipcc sinks the loads
    // of late-used fields behind branches -- three dependent s_load round trips on the way to the first HBM request (seen in the ISA;
    // round 5, profiles/r05_fixed_cost_by_launch.txt: "issue" 0.41 us per launch)
    asm volatile("" ::"s"(p.x), "s"(p.wp), "s"(p.y), "s"(p.partial), "s"(p.ldx), "s"(p.ldy), "s"(p.M), "s"(p.N), "s"(p.K), "s"(p.KB),
                 "s"(p.ncg), "s"(p.kb_per_split), "s"(p.epilogue), "s"(p.wt), "s"(p.dbg));
    asm volatile("" ::"s"(p.ss_in), "s"(p.ss_tiles), "s"(p.ss_inv_h), "s"(p.ss_eps), "s"(p.resid), "s"(p.norm_w), "s"(p.ss_out), "s"(p.tickets));
    if (p.dbg && tid == 0) {
        p.dbg[wgid * 8 + 0] = __builtin_amdgcn_s_memrealtime();
        p.dbg[wgid * 8 + 4] = ((uint64_t)__builtin_amdgcn_s_getreg(20 | (31 << 11)) << 32) | (uint32_t)__builtin_amdgcn_s_getreg(4 | (31 << 11));
    }
    if constexpr (ABL & 32) {  // launch cost only
        return;
    }
    // held in SGPRs until the end (a store here would join the hand-counted vmcnt queues): prologue loads issued / wave 0's first
    // weight unit landed (= the first MFMA can issue)
    uint64_t ts_issued = 0, ts_w0 = 0;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cgl  = wave % CG;
    const int wk   = wave / CG;
    const int l31  = lane & 31;
    const int half = lane >> 5;

    const int cg  = bx * CG + cgl;
    const int cgc = min(cg, p.ncg - 1);
    const int kb0 = by * p.kb_per_split;
    const int nkb = min(p.kb_per_split, p.KB - kb0);
    const int nst = (nkb + S - 1) / S;
    auto      phys = [&](int t) { return min(t, nst - 1); };  // stages past the slice re-read the last one (nobody consumes them)
    // row block (prefill: M > ROWS): rows m0 .. m0 + Mloc of x / y; the x descriptor starts at row m0
    const int m0   = blockIdx.z * ROWS;
    const int Mloc = min(ROWS, p.M - m0);

    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, (int)((size_t)p.KB * p.ncg * kP32Unit), 0x00020000);
    const auto rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (size_t)m0 * p.ldx), 0,
                                                        (int)(((size_t)(Mloc - 1) * p.ldx + p.K) * 2), 0x00020000);
    const int  vw   = lane * 16;
    const int  vs   = 2048 + l31 * 4;

    floatx16 acc[MH];
#pragma unroll
    for (int h = 0; h < MH; ++h) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            acc[h][r] = 0.f;
        }
    }

    // ---- folded RMSNorm, consumer side (p.ss_in): the producing GEMM left, per column tile, partial row sums of squares of the
    // residual rows.  T / ROWS threads per row fetch them (up to 8 loads each) right behind the activation DMA of stage 0 (L2 hits
    // like it, covered by its wait), park their share in LDS before the first barrier; the row factor is finished in the epilogue.
    constexpr int LDSX   = 2 * STG > REDB ? 2 * STG : REDB;  // scratch behind the stage buffers / the reduction image
    constexpr int PARTS  = T / ROWS;
    float*        part_s = (float*)(smem + LDSX);            // [PARTS][ROWS]
    static_assert(PARTS * ROWS * 4 + 16 <= kDec32NormLds, "norm scratch");
    const bool    scaled = p.ss_in != nullptr;               // uniform
    float         ssv[8] = {}, ss_extra = 0.f;
    // one buffer instruction per load: out-of-range tiles read as 0 (descriptor bound = ss_tiles rows), the tile stride rides in an SGPR
    auto ss_issue = [&]() __attribute__((always_inline)) {
        const auto rs_ss = __builtin_amdgcn_make_buffer_rsrc((void*)p.ss_in, 0, p.ss_tiles * p.M * 4, 0x00020000);
        const int  v0    = ((tid / ROWS) * p.M + min(m0 + tid % ROWS, p.M - 1)) * 4;
        const int  step  = PARTS * p.M * 4;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            ssv[u] = bit_cast<float>(__builtin_amdgcn_raw_buffer_load_b32(rs_ss, v0, u * step, 0));
        }
    };
    if (scaled && p.ss_tiles > 8 * PARTS) {  // more tiles than 8 per thread (wide models, narrow workgroups): before anything else
        const int    part = tid / ROWS;
        const float* src  = p.ss_in + min(m0 + tid % ROWS, p.M - 1);
        for (int t = part + 8 * PARTS; t < p.ss_tiles; t += PARTS) {
            ss_extra += src[(size_t)t * p.M];
        }
    }

    u32x4    ring[PF][2];
    uint32_t sring[PF];
    // ---- L2 prefetch of the weight stream (ABL 0x4000 / 0x8000: one / two stages beyond the ring; round 6) ----------------------------
    // The ring holds PF k-blocks per wave: a unit is requested UNR stages before its first use, and VMEM returns in order -- one late HBM
    // line holds up every younger load of the wave at its next counted wait, and the stage barrier passes the stall to the other 15
    // waves.  One more instruction per refill touches the unit the wave will request a stage (two) LATER: lane i reads 4 bytes of the
    // unit's 64-byte sector i (34 sectors = 2176 B), `sc1` (no L1 allocation), into a register nobody reads -- the HBM request is in
    // flight a stage earlier, the ring's own `nt` load of that unit finds the line in (or on its way into) the XCD's L2.  HBM bytes are
    // unchanged; L1 -> L2 requests grow by one 64-lane dword load per 2 KB of weights.
    constexpr bool L2PF = (ABL & 0xc000) != 0;
    constexpr int  PFD  = (ABL & 0x8000) ? 2 : 1;  // stages beyond the ring
    constexpr int  NLD  = L2PF ? 4 : 3;            // compiler-visible VMEM loads per ring slot refill
    static_assert(!L2PF || (ABL & 0x100), "the L2 prefetch is written for the LDS-DMA staging mode");
    uint32_t  pfv[L2PF ? PF : 1];
    const int vpf = min(tid & 63, 33) * 64;
    u32x4    xr[XR];
    int      xoff[XR], xlds[XR];
#pragma unroll
    for (int r = 0; r < XR; ++r) {
        const int q   = tid + T * r;
        const int kbi = q / (ROWS * 16);
        const int row = (q >> 4) % ROWS;
        const int ch  = q & 15;
        xoff[r]       = (min(row, Mloc - 1) * p.ldx + ch * 8) * 2 + kbi * 256;
        xlds[r]       = kbi * KBB + row * 256 + ((ch ^ (row & 15)) << 4);
    }
    // B fragment of 16-k step j: row (l & 31) [+ 32], 16-byte chunk 2j + half, XOR-swizzled by the row
    int coff[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        coff[j] = l31 * 256 + (((2 * j + half) ^ (l31 & 15)) << 4);
    }
    uint32_t m1024 = 0x64006400u, m64 = 0x54005400u;
    asm volatile("" : "+v"(m1024), "+v"(m64));  // magic numbers in VGPRs: one v_and_or_b32 per pair

    // LDS-DMA staging: a stage image = S * ROWS / 4 pieces of 1 KiB (4 rows x 256 B), piece pc = r * WAVES + wave.  The
    // DMA writes lane L to slot L of the piece (wave-linear), so the XOR swizzle sits on the SOURCE address: lane L
    // fetches chunk (L & 15) ^ (row & 15) of row 4 pc + (L >> 4).
    constexpr bool DMA = (ABL & 0x100) != 0;
    constexpr int  NPC = S * ROWS / 4;
    constexpr int  DR  = NPC / WAVES;
    static_assert(!DMA || NPC % WAVES == 0, "DMA pieces per wave");
    int            doff[DMA ? DR : 1];
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)smem);
    if constexpr (DMA) {
#pragma unroll
        for (int r = 0; r < DR; ++r) {
            const int pc  = r * WAVES + wave;
            const int kbi = pc / (ROWS / 4);
            const int row = (pc % (ROWS / 4)) * 4 + (lane >> 4);
            const int ch  = (lane & 15) ^ (row & 15);
            doff[r]       = (min(row, Mloc - 1) * p.ldx + ch * 8) * 2 + kbi * 256;
        }
    }
    // one DMA instruction per piece; M0 = LDS byte address of the piece (saved / restored: the compiler owns M0)
#define D32_DMA_X(t, buf)                                                                                         \
    _Pragma("unroll") for (int r = 0; r < DR; ++r)                                                                \
    {                                                                                                             \
        unsigned       keep_;                                                                                     \
        const unsigned dst_ = lds0 + (buf)*STG + (r * WAVES + wave) * 1024;                                       \
        const int      so_  = (kb0 + (t)*S) * 256;                                                                \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"                                       \
                     "buffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"                               \
                     : "=&s"(keep_)                                                                               \
                     : "v"(doff[r]), "s"(rs_x), "s"(dst_), "s"(so_)                                               \
                     : "memory");                                                                                 \
    }

    // block `b` (relative to kb0) of this wave -> unit offset; blocks past the slice are clamped (their scales are zeroed)
#define D32_LOAD_W(slot, b)                                                                                       \
    {                                                                                                             \
        const int uo_ = ((kb0 + min((b), nkb - 1)) * p.ncg + cgc) * kP32Unit;                                     \
        ring[slot][0] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, vw, uo_, /*nt*/ 2);                           \
        ring[slot][1] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, vw + 1024, uo_, /*nt*/ 2);                    \
        sring[slot]   = __builtin_amdgcn_raw_buffer_load_b32(rs_w, vs, uo_, 0);                                   \
        if constexpr (L2PF) {                                                                                     \
            const int up_ = ((kb0 + min((b) + PFD * S, nkb - 1)) * p.ncg + cgc) * kP32Unit;                       \
            pfv[slot]     = __builtin_amdgcn_raw_buffer_load_b32(rs_w, vpf, up_, /*sc1*/ 16);                     \
        }                                                                                                         \
    }
#define D32_LOAD_X(t)                                                                                             \
    _Pragma("unroll") for (int r = 0; r < XR; ++r)                                                                \
    {                                                                                                             \
        xr[r] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, xoff[r], (kb0 + (t)*S) * 256, 0);                     \
    }
#define D32_STORE_X(buf)                                                                                          \
    _Pragma("unroll") for (int r = 0; r < XR; ++r)                                                                \
    {                                                                                                             \
        *(u32x4*)(smem + (buf)*STG + xlds[r]) = xr[r];                                                            \
    }

    if (nst > 0) {
        // VMEM returns in order, so the ISSUE order decides what every counted wait also waits for.  Steady state at the
        // top of stage t (oldest first): weights of stage t .. t+UNR-2, x of stage t+1, weights of stage t+UNR-1.  The
        // prologue builds exactly that queue (hipcc merges the loop-entry edge and the back edge conservatively: a
        // prologue with fewer loads in flight than the steady state turns every wait of the loop into vmcnt(0)).
        if constexpr (DMA) {
            // queue: x(0) DMA pieces, then the whole weight ring; the DMA is invisible to hipcc's waitcnt pass, so its
            // completion is waited for by hand: everything older than the 3 * PF ring loads
            D32_DMA_X(phys(0), 0);
            if (scaled) {  // uniform: the sums-of-squares loads ride between the DMA and the ring: covered by the DMA's own wait below
                ss_issue();
                __builtin_amdgcn_sched_barrier(0);
            }
            constexpr int PF0 = PF;
#pragma unroll
            for (int q = 0; q < PF0; ++q) {
                D32_LOAD_W(q, phys(q / BPS) * S + wk + (q % BPS) * WK);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (p.dbg) {
                ts_issued = __builtin_amdgcn_s_memrealtime();
            }
            asm volatile("s_waitcnt vmcnt(%0)" ::"i"((ABL & 0x1000) ? 0 : NLD * PF0) : "memory");
        }
        else {
            if (scaled) {
                ss_issue();
            }
            D32_LOAD_X(phys(0));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < UNR - 1; ++u) {
#pragma unroll
                for (int i = 0; i < BPS; ++i) {
                    D32_LOAD_W(u * BPS + i, phys(u) * S + wk + i * WK);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            D32_STORE_X(0);
            __builtin_amdgcn_sched_barrier(0);
            D32_LOAD_X(phys(1));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < BPS; ++i) {
                D32_LOAD_W((UNR - 1) * BPS + i, phys(UNR - 1) * S + wk + i * WK);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // folded RMSNorm, consumer side: this thread's share of its row's tiles, in a fixed order (tiles part, part + PARTS, ... then
        // the tail); the row factor is finished per thread in the epilogue
        if (scaled) {
            float t = 0.f;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                t += ssv[u];  // tiles past the producer's count came back as 0
            }
            part_s[tid] = t + ss_extra;  // [tid / ROWS][tid % ROWS]
        }
        __syncthreads();
        if (p.dbg && tid == 0) {
            p.dbg[wgid * 8 + 1] = __builtin_amdgcn_s_memrealtime();
        }

        // one stage = compute on buffer (u & 1) with ring slots u*BPS.. , then staging + refills, then the barrier
        auto stage = [&](auto U, const int t, auto REM) __attribute__((always_inline)) {
            constexpr int  u   = decltype(U)::value;
            constexpr bool rem = decltype(REM)::value;  // a stage of the remainder (after the last whole unrolled body)
            const int buf = u & 1;  // UNR is even: parity of t
            if constexpr (DMA && !(ABL & 8)) {
                // First make hipcc wait for this stage's ring slots HERE (their first use), then start the DMA of stage
                // t+1 into the other buffer: the waitcnt pass does not see the DMA, so any of its counted waits that came
                // after these instructions would also wait for them.  The DMA lands while the stage computes.
#pragma unroll
                for (int i = 0; i < BPS; ++i) {
                    asm volatile("" ::"v"(ring[u * BPS + i][0]), "v"(ring[u * BPS + i][1]), "v"(sring[u * BPS + i]));
                    if constexpr (L2PF) {  // issued right behind them, UNR stages ago: keeps the load alive and hipcc's counts exact
                        asm volatile("" ::"v"(pfv[u * BPS + i]));
                    }
                }
                if (p.dbg && t == 0) {
                    ts_w0 = __builtin_amdgcn_s_memrealtime();
                }
                // Not behind the last stage: nobody would read it, and the refills that the counted wait below relies on
                // are dead code there (hipcc drops them in the remainder stage), so that DMA could still be landing when the
                // epilogue lays the reduction image over the stage buffers (seen as flaky rows 96..127 of row blocks, r02).
                if (t + 1 < nst) {  // uniform
                    D32_DMA_X(phys(t + 1), buf ^ 1);
                }
            }
#pragma unroll
            for (int i = 0; i < BPS; ++i) {
                const int     slot = u * BPS + i;
                const int     kbi  = wk + i * WK;
                const int     b    = phys(t) * S + kbi;
                const bool    live = b < nkb;
                const half2_t pr   = bit_cast<half2_t>(live ? sring[slot] : 0u);
                const half2_t s2   = {pr[0], pr[0]};
                const half2_t z2   = {pr[1], pr[1]};
                const char*   xb   = smem + buf * STG + kbi * KBB;
                if constexpr ((ABL & 0x800) != 0) {
                    // Explicit fragment pipeline: hipcc schedules a compiler-visible LDS read right in front of the MFMA
                    // that consumes it (seen in the ISA -- every 16-k step then exposes one LDS round trip per wave), so the
                    // reads are inline asm here: step j+1's fragments are requested before step j's MFMAs, and a counted
                    // lgkmcnt that names its registers (no use can be hoisted above it) retires step j's.
                    half8_t        f0[MH], f1[MH];
                    const unsigned xa = lds0 + buf * STG + kbi * KBB;
                    auto           rd = [&](half8_t(&f)[MH], int j) __attribute__((always_inline)) {
                        const unsigned ad = xa + (unsigned)coff[j];
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
                        else if constexpr (MH == 2) {
                            asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(f[0]), "+v"(f[1]) : "i"(n));
                        }
                        else {
                            asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]) : "i"(n));
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    };
                    rd(f0, 0);
                    static_for<8>([&](auto J) {
                        constexpr int  j  = decltype(J)::value;
                        half8_t(&cur)[MH] = (j & 1) ? f1 : f0;
                        half8_t(&nxt)[MH] = (j & 1) ? f0 : f1;
                        if constexpr (j + 1 < 8) {
                            rd(nxt, j + 1);
                        }
                        const half8_t a = dequant8_p32(ring[slot][j >> 2][j & 3], s2, z2, m1024, m64);
                        wt(cur, std::integral_constant<int, (j + 1 < 8) ? MH : 0>{});
#pragma unroll
                        for (int h = 0; h < MH; ++h) {
                            acc[h] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, cur[h], acc[h], 0, 0, 0);
                        }
                    });
                }
                else {
                half8_t       bq[MH], bn[MH];
#pragma unroll
                for (int h = 0; h < MH; ++h) {
                    bq[h] = (ABL & 4) ? bit_cast<half8_t>(ring[slot][1]) : *(const half8_t*)(xb + h * 8192 + coff[0]);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const uint32_t wj = ring[slot][j >> 2][j & 3];
                    half8_t        a;
                    if constexpr (ABL & 1) {
                        const uint32_t rw = wj ^ bit_cast<uint32_t>(pr);
                        a                 = bit_cast<half8_t>(u32x4{rw, rw, rw, rw});
                    }
                    else {
                        a = dequant8_p32(wj, s2, z2, m1024, m64);
                    }
                    if (j + 1 < 8) {  // fragments of the next 16-k step: one step of LDS latency hidden per wave
#pragma unroll
                        for (int h = 0; h < MH; ++h) {
                            bn[h] = (ABL & 4) ? bit_cast<half8_t>(ring[slot][0]) : *(const half8_t*)(xb + h * 8192 + coff[j + 1]);
                        }
                        if constexpr (ABL & 0x200) {
                            __builtin_amdgcn_sched_barrier(0x7f);  // everything but DS instructions may cross
                        }
                    }
                    if constexpr (ABL & 0x400) {
                        __builtin_amdgcn_s_setprio(1);
                    }
#pragma unroll
                    for (int h = 0; h < MH; ++h) {
                        if constexpr (ABL & 2) {
                            asm volatile("" ::"v"(a), "v"(bq[h]));  // operands stay live, no matrix work
                        }
                        else {
                            acc[h] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bq[h], acc[h], 0, 0, 0);
                        }
                    }
                    if constexpr (ABL & 0x400) {
                        __builtin_amdgcn_s_setprio(0);
                    }
                    if (j + 1 < 8) {
#pragma unroll
                        for (int h = 0; h < MH; ++h) {
                            bq[h] = bn[h];
                        }
                    }
                }
                }
            }
            // x of stage t+1 (loaded one stage ago) -> the other buffer; then the loads of stage t+2 into the same
            // registers, then this stage's ring slots are refilled for stage t+UNR.  All unconditional: past the last
            // stage the x descriptor returns zeros / the weight unit is clamped and nobody reads the result, while a
            // branch around a load makes the waitcnt pass assume the not-taken path (every later counted wait then
            // over-waits by the skipped loads).
            if constexpr (!DMA && !(ABL & 8)) {
                D32_STORE_X(buf ^ 1);
                D32_LOAD_X(phys(t + 2));
            }
            if constexpr (!(ABL & 16)) {
#pragma unroll
                for (int i = 0; i < BPS; ++i) {
                    D32_LOAD_W(u * BPS + i, phys(t + UNR) * S + wk + i * WK);
                }
            }
            if constexpr (DMA && !(ABL & 8)) {
                // my DMA pieces of stage t+1 have landed when at most the 3 * BPS refills issued after them are in flight.  In
                // the remainder those refills are dead code (nobody consumes them, hipcc drops them): drain instead.
                asm volatile("s_waitcnt vmcnt(%0)" ::"i"((rem || (ABL & (16 | 0x2000))) ? 0 : NLD * BPS) : "memory");
            }
            __syncthreads();
        };
        // Whole unrolled bodies first, WITHOUT an exit inside: with a `break` between the stages of a body hipcc sees a
        // path "stage u=0 -> latch -> stage u=0" (it cannot know the loop ends there) on which slot 0 was refilled a
        // moment ago, and drains vmcnt(0) at the top of every body (seen in the ISA).  The remainder runs once.
        int t0 = 0;
        for (; t0 + UNR <= nst; t0 += UNR) {
            static_for<UNR>([&](auto U) { stage(U, t0 + decltype(U)::value, std::false_type{}); });
        }
        static_for<UNR>([&](auto U) {
            if (t0 + decltype(U)::value < nst) {  // uniform over the workgroup
                stage(U, t0 + decltype(U)::value, std::true_type{});
            }
        });
    }
#undef D32_DMA_X
#undef D32_LOAD_W
#undef D32_LOAD_X
#undef D32_STORE_X

    if (p.dbg && tid == 0) {
        p.dbg[wgid * 8 + 2] = __builtin_amdgcn_s_memrealtime();
    }
    if constexpr (ABL & 64) {  // no epilogue
        if (acc[0][0] == 12345.f) {
            p.y[tid] = (half_t)acc[0][1];
        }
        return;
    }
    // ---- the WK k-phase partial tiles meet in LDS: red[wk][row][c4 ^ (row & 7)] (floatx4 units, CG*8 per row) --------
    // lane holds, per half h and register r: row m = 32h + (l & 31), column 32 cgl + 8 (r >> 2) + 4 (l >> 5) + (r & 3)
    {
        constexpr int C4 = CG * 8;  // floatx4 units per row
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
        if (p.dbg && tid == 0) {
            p.dbg[wgid * 8 + 7] = __builtin_amdgcn_s_memrealtime();
        }
        constexpr int NE = ROWS * C4;  // floatx4 elements of the output tile
        static_assert(NE % T == 0, "whole epilogue passes");
        const int     ncol0 = bx * CG * 32;
        auto tile_sum = [&](int m, int c4) __attribute__((always_inline)) {
            floatx4 a = red[m * C4 + (c4 ^ (m & 7))];
#pragma unroll
            for (int k = 1; k < WK; ++k) {  // fixed order: deterministic
                a += red[(k * ROWS + m) * C4 + (c4 ^ (m & 7))];
            }
            return a;
        };
        if constexpr (MH <= 2)  // (decode row blocks only: the 128-row prefill tile never produces for a folded norm)
        if (p.epilogue == 3) {
            // ---- folded RMSNorm, producer side: residual add + next norm's weight + per-tile sums of squares -------------------
            const int      splits = gridDim.y;
            const size_t   slab   = (size_t)p.M * p.N;
            unsigned*      flag   = (unsigned*)(smem + LDSX + kDec32NormLds - 16);
            // the residual rows and the norm weight do not depend on anybody's arrival: in flight before the slabs are parked
            constexpr int NI = NE / T;
            half4_t       r4v[NI], g4v[NI];
#pragma unroll
            for (int it = 0; it < NI; ++it) {
                const int    e = it * T + tid, m = e / C4, c4 = e % C4;
                const int    n = ncol0 + c4 * 4;
                const bool   ok = m < Mloc && n < p.N;
                r4v[it] = *(const half4_t*)(p.resid + ((size_t)m0 + (ok ? m : 0)) * p.N + (ok ? n : 0));
                g4v[it] = *(const half4_t*)(p.norm_w + (ok ? n : 0));
            }
            if (splits > 1) {
                // every slice parks its fp32 tile write-through, then takes a ticket: the LAST arriver of the tile sums the slices in
                // slice order (its own from LDS -- the same bits it stored) and runs the epilogue; the others are done
#pragma unroll
                for (int e0 = 0; e0 < NE; e0 += T) {
                    const int e = e0 + tid, m = e / C4, c4 = e % C4;
                    const int n = ncol0 + c4 * 4;
                    if (m < Mloc && n < p.N) {
                        store_wt((floatx4*)(p.partial + (size_t)by * slab + ((size_t)m0 + m) * p.N + n), tile_sum(m, c4), 1);
                    }
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                unsigned* const tk = p.tickets + blockIdx.z * gridDim.x + bx;
                if (tid == 0) {
                    *flag = __hip_atomic_fetch_add(tk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                __syncthreads();
                if (*flag != (unsigned)(splits - 1)) {
                    if (p.dbg && tid == 0) {
                        p.dbg[wgid * 8 + 3] = __builtin_amdgcn_s_memrealtime();
                        p.dbg[wgid * 8 + 5] = ts_issued;
                        p.dbg[wgid * 8 + 6] = ts_w0;
                    }
                    return;
                }
                if (tid == 0) {
                    __hip_atomic_store(tk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // every other slice has arrived
                }
            }
            // the other slices' slabs of EVERY pass of this thread are requested before any is consumed (one memory round trip on the last
            // arriver's critical path, not one per pass); the first four slices' loads are clamped and unconditional
            floatx4 vv[NI][4];
            if (splits > 1) {
#pragma unroll
                for (int it = 0; it < NI; ++it) {
                    const int    e = it * T + tid, m = e / C4, c4 = e % C4;
                    const int    n = ncol0 + c4 * 4;
                    const bool   ok = m < Mloc && n < p.N;
                    const float* src = p.partial + ((size_t)m0 + (ok ? m : 0)) * p.N + (ok ? n : 0);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        vv[it][u] = load_agent(src + (size_t)min(u, splits - 1) * slab);
                    }
                }
            }
#pragma unroll
            for (int e0 = 0; e0 < NE; e0 += T) {
                const int    e = e0 + tid, m = e / C4, c4 = e % C4;
                const int    n = ncol0 + c4 * 4;
                const bool   ok = m < Mloc && n < p.N;
                const size_t mg = (size_t)m0 + (ok ? m : 0);
                const int    nc = ok ? n : 0;
                const floatx4 own = tile_sum(m, c4);
                floatx4       a   = own;
                if (splits > 1) {
                    // slice order, from zero: the bits of the reduce-norm kernel (norm_row.h); the own slice comes from LDS
                    const float* src = p.partial + mg * p.N + nc;
                    a = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (u < splits) {
                            a += u == by ? own : vv[e0 / T][u];
                        }
                    }
                    for (int sl = 4; sl < splits; ++sl) {
                        a += sl == by ? own : load_agent(src + (size_t)sl * slab);
                    }
                }
                const half4_t hc = {(half_t)a[0], (half_t)a[1], (half_t)a[2], (half_t)a[3]};  // the GEMM's fp16 output rounding
                half4_t       r4 = r4v[e0 / T];
                const half4_t g4 = g4v[e0 / T];
                r4               = r4 + hc;  // fp16 add, one rounding per element
                half4_t xg;
                float   ss = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float f = (float)r4[q];
                    ss            = __builtin_fmaf(f, f, ss);
                    xg[q]         = (half_t)fminf(fmaxf(f * (float)g4[q], -65504.f), 65504.f);
                }
                if (ok) {
                    *(half4_t*)(p.resid + mg * p.N + nc) = r4;
                    *(half4_t*)(p.y + mg * p.ldy + nc)   = xg;
                }
                else {
                    ss = 0.f;
                }
                // the C4 threads of row m are C4 consecutive lanes (C4 = 16 / 32 / 64 divides 64): butterfly in a fixed pattern
#pragma unroll
                for (int d = 1; d < C4; d <<= 1) {
                    ss += __shfl_xor(ss, d);
                }
                if (ok && c4 == 0) {
                    p.ss_out[(size_t)bx * p.M + mg] = ss;
                }
            }
            if (p.dbg && tid == 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                p.dbg[wgid * 8 + 3] = __builtin_amdgcn_s_memrealtime();
                p.dbg[wgid * 8 + 5] = ts_issued;
                p.dbg[wgid * 8 + 6] = ts_w0;
            }
            return;
        }
        // ---- split-K merged in the launch (p.merge; fp16 / gated-SiLU epilogues): park, ticket, the last arriver of the tile goes on --
        constexpr int NIg = NE / T;
        floatx4       vm[MH <= 2 ? NIg : 1][4];
        bool          merged = false;
        if constexpr (MH <= 2) {
            if (p.merge && gridDim.y > 1) {  // uniform
                const int    splits = gridDim.y;
                const size_t slab   = (size_t)p.M * p.N;
                unsigned*    flag   = (unsigned*)(smem + LDSX + kDec32NormLds - 16);
#pragma unroll
                for (int e0 = 0; e0 < NE; e0 += T) {
                    const int e = e0 + tid, m = e / C4, c4 = e % C4;
                    const int n = ncol0 + c4 * 4;
                    if (m < Mloc && n < p.N) {
                        store_wt((floatx4*)(p.partial + (size_t)by * slab + ((size_t)m0 + m) * p.N + n), tile_sum(m, c4), 1);
                    }
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                unsigned* const tk = p.tickets + blockIdx.z * gridDim.x + bx;
                if (tid == 0) {
                    *flag = __hip_atomic_fetch_add(tk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                __syncthreads();
                if (*flag != (unsigned)(splits - 1)) {
                    if (p.dbg && tid == 0) {
                        p.dbg[wgid * 8 + 3] = __builtin_amdgcn_s_memrealtime();
                        p.dbg[wgid * 8 + 5] = ts_issued;
                        p.dbg[wgid * 8 + 6] = ts_w0;
                    }
                    return;
                }
                if (tid == 0) {
                    __hip_atomic_store(tk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // every other slice has arrived
                }
                // all passes' slab loads of the first four slices in flight before any is consumed (clamped, unconditional)
#pragma unroll
                for (int it = 0; it < NIg; ++it) {
                    const int    e = it * T + tid, m = e / C4, c4 = e % C4;
                    const int    n = ncol0 + c4 * 4;
                    const bool   ok = m < Mloc && n < p.N;
                    const float* src = p.partial + ((size_t)m0 + (ok ? m : 0)) * p.N + (ok ? n : 0);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        vm[it][u] = load_agent(src + (size_t)min(u, splits - 1) * slab);
                    }
                }
                merged = true;
            }
        }
#pragma unroll
        for (int e0 = 0; e0 < NE; e0 += T) {
            const int e = e0 + tid;
            const int m  = e / C4;
            const int c4 = e % C4;
            floatx4   a  = tile_sum(m, c4);
            if constexpr (MH <= 2) {
                if (merged) {  // slice order, from zero: the bits of splitk_reduce_kernel; the own slice comes from LDS
                    const int     splits = gridDim.y;
                    const floatx4 own    = a;
                    const int     n_     = ncol0 + c4 * 4;
                    const bool    ok     = m < Mloc && n_ < p.N;
                    const float*  src    = p.partial + ((size_t)m0 + (ok ? m : 0)) * p.N + (ok ? n_ : 0);
                    a                    = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (u < splits) {
                            a += u == by ? own : vm[e0 / T][u];
                        }
                    }
                    for (int sl = 4; sl < splits; ++sl) {
                        a += sl == by ? own : load_agent(src + (size_t)sl * (size_t)p.M * p.N);
                    }
                }
            }
            if (scaled) {  // inv[m] = 1 / sqrt(sum over tiles / H + eps): every thread of a row adds the same PARTS numbers in the same order
                float t = part_s[m];
#pragma unroll
                for (int q = 1; q < PARTS; ++q) {
                    t += part_s[q * ROWS + m];
                }
                const float iv = 1.0f / __builtin_sqrtf(t * p.ss_inv_h + p.ss_eps);
                a              = a * floatx4{iv, iv, iv, iv};
            }
            const int n = ncol0 + c4 * 4;
            if (m >= Mloc || n >= p.N) {
                continue;
            }
            const size_t mg = (size_t)m0 + m;  // row of y
            if (p.epilogue == 2) {
                floatx4* dst = (floatx4*)(p.partial + ((size_t)by * p.M + mg) * p.N + n);
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
                half2_t*    dst = (half2_t*)(p.y + mg * p.ldy + (n >> 1));
                if (p.wt & 2) {
                    store_wt(dst, o);
                }
                else {
                    *dst = o;
                }
            }
            else {
                half4_t  o   = {(half_t)a[0], (half_t)a[1], (half_t)a[2], (half_t)a[3]};
                half4_t* dst = (half4_t*)(p.y + mg * p.ldy + n);
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
        p.dbg[wgid * 8 + 5] = ts_issued;
        p.dbg[wgid * 8 + 6] = ts_w0;
    }
}

static int env_int2(const char* name, int dflt)
{
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}

template<int MH, int CG, int WK, int S, int PF, int ABL = 0>
static int launch_dec32_one(const Dec32Params& p, dim3 grid, hipStream_t st)
{
    constexpr int stage = 2 * S * 32 * MH * 256;
    constexpr int red   = WK * 32 * MH * CG * 128;
    constexpr int lds   = (stage > red ? stage : red) + kDec32NormLds;  // + the folded-norm scratch (a few KB)
    if (const int rc = ensure_dynamic_lds((const void*)gemm_dec32_kernel<MH, CG, WK, S, PF, ABL>, lds)) {
        return rc;
    }
    gemm_dec32_kernel<MH, CG, WK, S, PF, ABL><<<grid, CG * WK * 64, lds, st>>>(p);
    TM_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---- prefill tile: 128 rows x 512 columns per workgroup, TWO weight fragments per activation-fragment read -----------------
// Measured on the 128 x 256 tile above (profiles/r02_bench_gemm_prefill_ablation.txt): with loads and dequant removed the
// loop still tops out at 62..71 % of the MFMA peak, because every v_mfma_f32_32x32x16 needs one 1 KB x fragment out of LDS
// (4 SIMDs x 1 MFMA / 32 clk x 1 KB = 128 B/clk = the LDS peak): the LDS pipe is co-critical with the matrix pipe.  Here a
// wave owns 64 columns (two P32 units per k-block) x 128 rows: one ds_read_b128 feeds two MFMAs, 128 accumulator registers
// per lane, 8 waves (two per SIMD, 256 registers each).  One k-block per LDS stage (32 KB, double buffered), x staged
// through registers (the LDS-DMA costs 30 % at this tile size), weights through a two-k-block register ring.
// grid = (ceil(N / 512), splits, ceil(M / 128)); the epilogue stores straight from the accumulators.
template<int ABL = 0>
__global__ __launch_bounds__(512) void gemm_pre64_kernel(Dec32Params p)
{
    constexpr int MH = 4, CG = 8, NB = 2, T = 512, ROWS = 128;
    constexpr int KBB = ROWS * 256;  // LDS bytes of one k-block of x
    constexpr int XR  = ROWS * 16 / T;
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
    int       cgc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        cgc[nb] = min((int)(blockIdx.x * NB + nb) * CG + wave, p.ncg - 1);
    }
    const int kb0  = blockIdx.y * p.kb_per_split;
    const int nkb  = min(p.kb_per_split, p.KB - kb0);
    const int m0   = blockIdx.z * ROWS;
    const int Mloc = min(ROWS, p.M - m0);

    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, (int)((size_t)p.KB * p.ncg * kP32Unit), 0x00020000);
    const auto rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (size_t)m0 * p.ldx), 0,
                                                        (int)(((size_t)(Mloc - 1) * p.ldx + p.K) * 2), 0x00020000);
    const int  vw   = lane * 16;
    const int  vs   = 2048 + l31 * 4;

    floatx16 acc[NB][MH];
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
    u32x4    ring[2][NB][2];
    uint32_t sring[2][NB];
    u32x4    xr[XR];
    // staging: thread -> (row = tid / 16 + 32 r, 16-byte chunk tid % 16); rows 32 apart share the swizzle term, so the LDS
    // address of piece r is xlds0 + r * 8192 (an immediate) and only the global offsets need registers
    int       xoff[XR];
    const int xlds0 = (tid >> 4) * 256 + (((tid & 15) ^ ((tid >> 4) & 15)) << 4);
#pragma unroll
    for (int r = 0; r < XR; ++r) {
        xoff[r] = (min((tid >> 4) + 32 * r, Mloc - 1) * p.ldx + (tid & 15) * 8) * 2;
    }
    // B fragment of 16-k step j: row (l & 31) [+ 32 h], chunk (2j + half) ^ (row & 15) = 2j ^ (half ^ (row & 15)): one
    // v_xor with an immediate per step instead of eight address registers
    const int frow = l31 * 256;
    const int fsw  = (half ^ (l31 & 15)) << 4;
    uint32_t m1024 = 0x64006400u, m64 = 0x54005400u;
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

    if (nkb > 0) {
        // issue order = steady-state order (oldest first at the top of stage t: W(t), x(t+1), W(t+1)); see gemm_dec32_kernel
        P64_LOAD_X(0);
        __builtin_amdgcn_sched_barrier(0);
        P64_LOAD_W(0, 0);
        __builtin_amdgcn_sched_barrier(0);
        P64_STORE_X(0);
        __builtin_amdgcn_sched_barrier(0);
        P64_LOAD_X(1);
        __builtin_amdgcn_sched_barrier(0);
        P64_LOAD_W(1, 1);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        if (p.dbg && tid == 0) {
            p.dbg[wgid * 8 + 1] = __builtin_amdgcn_s_memrealtime();
            p.dbg[wgid * 8 + 5] = __builtin_amdgcn_s_memtime();  // shader-clock counter: (d6 - d5) / loop time = the clock
        }
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
            // Two dequantised fragments live (a0: first column half, a1: second).  The VALU work of a dequant (14 packed
            // ops) is interleaved INTO the MFMA stream of the same wave -- an MFMA occupies the matrix pipe for 32 cycles but
            // the issue port for 4, so ~4 VALU ops fit behind each one: a1(j) is built behind the four MFMAs that use a0(j),
            // a0(j+1) behind the four that use a1(j).  Back-to-back "dequant block, then MFMA block" (the first version of
            // this kernel, and what hipcc emits on its own) measured additive: loop 102 us = 55 us of MFMA + 42 us of
            // everything else (profiles/r02_pre64_phase_traces.txt).
            // experiments (TM_D32_ABL, results are garbage; profiles/r04_gemm_experiments_session2.txt): 8 = the raw codes as fp16
            // subnormals on the matrix pipe instead of the dequantised operand, 16 = the accumulators rescaled once per k-block
            // (what a per-group scale applied on the accumulator side would cost)
            auto dq = [&](uint32_t w, int nb) __attribute__((always_inline)) {
                if constexpr (ABL & 8) {
                    const uint32_t m = 0x000f000fu;
                    return bit_cast<half8_t>(u32x4{w & m, (w >> 4) & m, (w >> 8) & m, (w >> 12) & m});
                }
                else {
                    return dequant8_p32(w, s2[nb], z2[nb], m1024, m64);
                }
            };
            if constexpr (ABL & 16) {
                typedef float floatx2 __attribute__((ext_vector_type(2)));
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    const float   rt = 1.0f + (float)s2[nb][0];
                    const floatx2 rr = {rt, rt};
#pragma unroll
                    for (int h = 0; h < MH; ++h) {
#pragma unroll
                        for (int r = 0; r < 8; ++r) {
                            floatx2 v         = {acc[nb][h][2 * r], acc[nb][h][2 * r + 1]};
                            v                 = v * rr;
                            acc[nb][h][2 * r] = v[0], acc[nb][h][2 * r + 1] = v[1];
                        }
                    }
                }
            }
            half8_t a0 = dq(ring[u][0][0][0], 0), a1;
            static_for<8>([&](auto J) {
                constexpr int  j  = decltype(J)::value;
                half8_t(&cur)[MH] = (j & 1) ? f1 : f0;
                half8_t(&nxt)[MH] = (j & 1) ? f0 : f1;
                if constexpr (j + 1 < 8) {
                    rd(nxt, j + 1);
                }
                wt(cur, std::integral_constant<int, (j + 1 < 8) ? MH : 0>{});
                a1 = dq(ring[u][1][j >> 2][j & 3], 1);
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
                    a0 = dq(ring[u][0][(j + 1) >> 2][(j + 1) & 3], 0);
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
                if constexpr (!(ABL & 4)) {
#pragma unroll
                    for (int g = 0; g < 2 * MH; ++g) {  // 8 x (1 MFMA, up to 4 VALU)
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                    }
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
    }
#undef P64_LOAD_W
#undef P64_LOAD_X
#undef P64_STORE_X
    if (p.dbg && tid == 0) {
        p.dbg[wgid * 8 + 2] = __builtin_amdgcn_s_memrealtime();
        p.dbg[wgid * 8 + 6] = __builtin_amdgcn_s_memtime();
    }
    // ---- epilogue: straight from the accumulators (no k-phases to merge here, so no LDS round trip and no barrier: the LDS
    // image of gemm_dec32_kernel cost 11.7 us of a 114 us workgroup).  Lane holds, per half h and register r: row
    // m = 32h + (l & 31), column 32 cg + 8 (r >> 2) + 4 (l >> 5) + (r & 3): the two lane halves write adjacent 8-byte pieces,
    // the 4 register groups complete a 64-byte row segment, the neighbouring wave the other half of the 128-byte line.
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int ncol0 = ((blockIdx.x * NB + nb) * CG + wave) * 32;
#pragma unroll
        for (int h = 0; h < MH; ++h) {
            const int m = 32 * h + l31;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int n = ncol0 + 8 * g4 + 4 * half;
                if (m >= Mloc || n >= p.N) {
                    continue;
                }
                const floatx4 a  = {acc[nb][h][4 * g4], acc[nb][h][4 * g4 + 1], acc[nb][h][4 * g4 + 2], acc[nb][h][4 * g4 + 3]};
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

template<int ABL>
static int launch_pre64_one(const Dec32Params& p, dim3 grid, hipStream_t st)
{
    constexpr int lds = 96 * 1024;  // two 32 KB stages; > 80 KB so that exactly one workgroup (8 waves x 256 registers) owns the CU
    if (const int rc = ensure_dynamic_lds((const void*)gemm_pre64_kernel<ABL>, lds)) {
        return rc;
    }
    gemm_pre64_kernel<ABL><<<grid, 512, lds, st>>>(p);
    TM_HIP_CHECK(hipGetLastError());
    return 0;
}

// The default structure: LDS-DMA staging of x + the explicit fragment pipeline (measured on MI355X, tools/trace_dec32.py /
// tools/bench_gemm.py: w1w3 main loop at M = 64 16.5 us with register staging + ds_write -> 15.3 us with the DMA -> 14.9 us
// with the inline-asm fragment reads; M = 8192: 1.03 -> 1.09 PF/s; the LDS-read scheduling fence and s_setprio variants
// measured neutral, the rotated k walk neutral at M = 64 and -5..-10 % at M = 8192: it breaks the L2 reuse of the weights
// between row blocks)
constexpr int kD32Mode = 0x900;
#ifdef TM_EXPERIMENTS
constexpr bool l2pf_compiled = true;
#else
constexpr bool l2pf_compiled = false;
#endif

// TM_EXPERIMENTS (compile-time, off in the shipped library): the timing / structure ablations of tools/trace_dec32.py and
// tools/bench_gemm.py (TM_D32_ABL) -- ~40 further instantiations of the kernel whose results are garbage by design.
template<int MH>
static int launch_dec32_shape(const Dec32Params& p, dim3 grid, int shape, hipStream_t st)
{
#ifdef TM_EXPERIMENTS
    static const int abl = env_int2("TM_D32_ABL", -1);
#endif
    // TM_D32_L2PF = 1 / 2 (TM_EXPERIMENTS builds): the weight stream's L2 prefetch one / two stages beyond the ring (ABL 0x4000 / 0x8000; shapes 0, 3 and
    // their 32-row forms).  Measured negative in round 6 (profiles/r06_l2pf_*: parity-green, w1w3's loop 14.9 -> 16.1 us, driver line -1.5 .. -2 %):
    // the loop does not wait for late weight lines -- one more request per 2 KB of weights in the CU's load path costs more than it hides.
#ifdef TM_EXPERIMENTS
    static const int l2pf = env_int2("TM_D32_L2PF", 0);
#else
    constexpr int l2pf = 0;
#endif
    if constexpr (MH == 4) {
        // row blocks of 128 rows x 256 columns, 8 waves of 32 columns over the whole k slice (M > 64): every dequantised
        // weight fragment feeds 4 MFMAs, every x fragment read from LDS feeds 32 columns
        if (shape == 5) {  // 128 x 512 tile, two weight fragments per x-fragment read (gemm_pre64_kernel)
#ifdef TM_EXPERIMENTS
            if (abl == 2) return launch_pre64_one<2>(p, grid, st);  // timing: no MFMA
            if (abl == 8) return launch_pre64_one<8>(p, grid, st);  // timing: raw codes on the matrix pipe
            if (abl == 24) return launch_pre64_one<24>(p, grid, st);  // + accumulators rescaled per k-block
#endif
            return launch_pre64_one<0>(p, grid, st);
        }
        if (shape != 4) {
            set_last_error("gemm_dec32: the 128-row tiles are shapes 4 and 5");
            return 1;
        }
#ifdef TM_EXPERIMENTS
        switch (abl) {
#define D32_CASE(v) case v: return launch_dec32_one<4, 8, 1, 2, 4, v>(p, grid, st)
            D32_CASE(0); D32_CASE(0x100); D32_CASE(0x101); D32_CASE(0x104); D32_CASE(0x108); D32_CASE(0x10d); D32_CASE(0x11d);
            D32_CASE(0x102); D32_CASE(0x1900); D32_CASE(0x2900); D32_CASE(0x3900); D32_CASE(0x800);
#undef D32_CASE
            default: break;
        }
#endif
        return launch_dec32_one<4, 8, 1, 2, 4, kD32Mode>(p, grid, st);
    }
    else
    switch (shape) {
        case 0: {  // 16 waves: 4 column groups x 4 k-phases, one k-block per wave per stage
#ifdef TM_EXPERIMENTS
            if constexpr (MH == 2) {
                switch (abl) {
#define D32_CASE(v) case v: return launch_dec32_one<MH, 4, 4, 4, 2, v>(p, grid, st)
                    D32_CASE(0); D32_CASE(0x200); D32_CASE(0x300); D32_CASE(0x400); D32_CASE(0x500); D32_CASE(0x700);
                    D32_CASE(1); D32_CASE(2); D32_CASE(4); D32_CASE(8); D32_CASE(16); D32_CASE(7); D32_CASE(15); D32_CASE(31);
                    D32_CASE(32); D32_CASE(64); D32_CASE(95); D32_CASE(24); D32_CASE(25); D32_CASE(26); D32_CASE(27);
                    D32_CASE(28); D32_CASE(29); D32_CASE(30); D32_CASE(0x107); D32_CASE(0x118); D32_CASE(0x11f);
                    D32_CASE(0x100); D32_CASE(0x918); D32_CASE(0x800);
#undef D32_CASE
                    default: break;
                }
            }
#endif
            if constexpr (l2pf_compiled) {
                if (l2pf == 1) return launch_dec32_one<MH, 4, 4, 4, 2, kD32Mode | 0x4000>(p, grid, st);
                if (l2pf == 2) return launch_dec32_one<MH, 4, 4, 4, 2, kD32Mode | 0x8000>(p, grid, st);
            }
            return launch_dec32_one<MH, 4, 4, 4, 2, kD32Mode>(p, grid, st);
        }
        case 1:  // 16 waves: 8 column groups x 2 k-phases (256 columns per workgroup), two k-blocks per wave per stage
            return launch_dec32_one<MH, 8, 2, 4, 4, kD32Mode>(p, grid, st);
        case 2:  // 8 waves: 4 column groups x 2 k-phases
            return launch_dec32_one<MH, 4, 2, 4, 4, kD32Mode>(p, grid, st);
        case 3:  // 8 waves: 2 column groups x 4 k-phases (64 columns per workgroup)
            if constexpr (l2pf_compiled) {
                if (l2pf == 1) return launch_dec32_one<MH, 2, 4, 4, 2, kD32Mode | 0x4000>(p, grid, st);
                if (l2pf == 2) return launch_dec32_one<MH, 2, 4, 4, 2, kD32Mode | 0x8000>(p, grid, st);
            }
            return launch_dec32_one<MH, 2, 4, 4, 2, kD32Mode>(p, grid, st);
        case kShapeWide2:  // 16 waves: 8 column groups x 2 k-phases (256 columns per workgroup) on 2-k-block stages, ring depth 2: shape 0's
                           // per-wave rhythm (one k-block per wave and stage) without shape 1's register spills (60 B / lane at 128 VGPRs)
            return launch_dec32_one<MH, 8, 2, 2, 2, kD32Mode>(p, grid, st);
        default: break;
    }
    set_last_error("gemm_dec32: unknown shape");
    return 1;
}

// Shapes 6 .. 9 are the decode shapes 3, 0, 2, 1 on 32-ROW blocks (grid.z = ceil(M / 32), the MH = 1 instantiations): twice the
// workgroups of half the LDS footprint (64 KB: two per CU), every weight unit read by the two row halves of its column tile
// (the second reader hits L2: the halves are gridDim.x * gridDim.y workgroups apart, the same XCD whenever that product is a
// multiple of 8).  They trade dequantisation work (one MFMA per fragment instead of two) for parallelism without split-K
// slabs -- which side wins is measured (tune_decode_gemms), not guessed.
static int dec32_base_shape(int shape)
{
    static const int base[4] = {3, 0, 2, 1};
    if (dec32_is_merge_shape(shape)) {
        shape -= kShapeMerge;
    }
    return shape >= 6 && shape <= 9 ? base[shape - 6] : shape;
}

// shape -> (column groups, k-blocks per stage)
static void dec32_shape_dims(int shape, int* cg, int* s)
{
    static const int cgs[6] = {4, 8, 4, 2, 8, 16};
    if (shape == kShapeLC) {
        *cg = 4;
        *s  = 4;
        return;
    }
    if (shape == kShapePre256 || shape == kShapeF16) {
        *cg = 8;
        *s  = 1;
        return;
    }
    shape = dec32_base_shape(shape);
    if (shape == kShapeWide2) {
        *cg = 8;
        *s  = 2;
        return;
    }
    *cg = cgs[shape < 0 || shape > 5 ? 0 : shape];
    *s  = shape == 5 ? 1 : (shape == 4 ? 2 : 4);
}

static bool dec32_on(bool big)
{
    static const int on     = env_int2("TM_GEMM_D32", 1);
    static const int on_big = env_int2("TM_GEMM_D32_PREFILL", 1);  // M > 64: the 128-row tiles of the same kernel family
    return on && (!big || on_big);
}

bool dec32_supported(const LinearWeight& w, int M)
{
    return dec32_on(M > 64) && w.type == 0 && w.packed32 != nullptr && M >= 1 && w.N % 32 == 0 && w.K % 128 == 0;
}

bool dec32_serves_every_m(int K, int N)
{
    return dec32_on(true) && N % 32 == 0 && K % 128 == 0;
}

// ---- measured dispatch (reference: gemm::Gemm::Run's DispatchCache, kernels/gemm/gemm.cu:92-224, filled by the warm-up
// tuning of turbomind.cc:363-487 under TM_GEMM_TUNE and carried between runs by TM_GEMM_EXPORT / TM_GEMM_IMPORT).  Here the
// key is (K, N, M) of a decode-batch linear, the value its workgroup shape and split-K count; the engine's tuner
// (engine_tune.hip: tune_decode_gemms) fills it by timing every candidate as a hipGraph over the model's own layer weights
// TOGETHER with the kernel that consumes the result (a split-K GEMM pays at the boundary, not inside the kernel).
static std::map<std::tuple<int, int, int, int>, std::pair<int, int>> g_d32_table;  // (role, K, N, M) -> (shape, splits)
static std::mutex                                               g_d32_mutex;  // engines tune / import while others launch

void dec32_table_set(int K, int N, int M, int shape, int splits, int role)
{
    std::lock_guard<std::mutex> lk(g_d32_mutex);
    g_d32_table[std::make_tuple(role, K, N, M)] = std::make_pair(shape, splits);
}

bool dec32_table_get(int K, int N, int M, int* shape, int* splits, int role)
{
    std::lock_guard<std::mutex> lk(g_d32_mutex);
    auto                        it = g_d32_table.find(std::make_tuple(role, K, N, M));
    if (it == g_d32_table.end() && role != 0) {
        it = g_d32_table.find(std::make_tuple(0, K, N, M));  // an entry without a role serves every role
    }
    if (it == g_d32_table.end()) {
        return false;
    }
    *shape  = it->second.first;
    *splits = it->second.second;
    return true;
}

void dec32_table_clear()
{
    {
        std::lock_guard<std::mutex> lk(g_d32_mutex);
        g_d32_table.clear();
    }
    gen_table_clear();  // the general-kernel / grouped entries (`G` lines) are part of the same dispatch state
}

// text, one line per entry: K N M shape splits [role]   (role 0 / absent: any linear of that shape)
int dec32_table_export(const char* path)
{
    FILE* f = fopen(path, "w");
    if (!f) {
        set_last_error(std::string("cannot write ") + path);
        return 1;
    }
    std::lock_guard<std::mutex> lk(g_d32_mutex);
    for (const auto& kv : g_d32_table) {
        fprintf(f, "%d %d %d %d %d %d\n", std::get<1>(kv.first), std::get<2>(kv.first), std::get<3>(kv.first), kv.second.first,
                kv.second.second, std::get<0>(kv.first));
    }
    gen_table_export_lines(f);  // `G ...` lines: the general kernel / grouped GEMMs (older parsers skip them)
    fclose(f);
    return 0;
}

int dec32_table_import(const char* path)
{
    FILE* f = fopen(path, "r");
    if (!f) {
        set_last_error(std::string("cannot read ") + path);
        return 1;
    }
    int  K, N, M, shape, splits, n = 0, skipped = 0;
    char line[160];
    while (fgets(line, sizeof line, f)) {
        if (line[0] == 'G') {
            const bool ok = gen_table_import_line(line);
            n += ok ? 1 : 0;
            skipped += ok ? 0 : 1;
            continue;
        }
        if (line[0] == '#' || line[0] == '\n') {
            continue;
        }
        int       role = 0;
        const int got  = sscanf(line, "%d %d %d %d %d %d", &K, &N, &M, &shape, &splits, &role);
        if (got < 5 || role < 0 || role > 4) {
            ++skipped;
            continue;
        }
        const bool big = M > 64;
        const bool lc  = shape == kShapeLC && M <= 64;
        const bool p256 = (shape == kShapePre256 || (shape == kShapeF16 && splits == 1)) && M > 64;  // what launch_linear_dec32 accepts (the tuner only proposes them for M, N >= 256)
        const bool mrg = dec32_is_merge_shape(shape) && M <= 64 && splits >= 2;  // in-launch merged split-K (decode batches only)
        const bool w2k = shape == kShapeWide2 && M <= 64;
        if (K > 0 && N > 0 && M > 0 && M == dec32_m_bucket(M) && shape >= 0 && (shape <= 9 || lc || p256 || mrg || w2k) && splits >= 1 && splits <= 16
            && (big ? shape >= 4 : (shape != 4 && shape != 5)) && !(shape == 5 && N < 512) && K % 128 == 0 && N % 32 == 0) {
            dec32_table_set(K, N, M, shape, splits, role);
            ++n;
        }
        else {
            ++skipped;
        }
    }
    fclose(f);
    if (skipped) {  // a table written by another build (shapes this build does not have): say so instead of silently running heuristics
        fprintf(stderr, "[tm] %s: %d dispatch line(s) not understood by this build were skipped (%d imported)\n", path, skipped, n);
    }
    return n > 0 ? 0 : 1;
}

// every (shape, splits) the decode kernels can run this linear with at M <= 256 rows: whole stages per slice, <= 512 workgroups
// The dispatch table is keyed by the exact row count up to 256 (decode batches) and by power-of-two buckets above (prefill
// forwards of an admission have arbitrary token counts: 512, 1024, ... 8192 stand for everything up to them)
int dec32_m_bucket(int M)
{
    if (M <= 256) {
        return M;
    }
    int b = 512;
    while (b < M && b < 8192) {
        b <<= 1;
    }
    return b;
}

int dec32_candidates(const LinearWeight& w, int M, int (*out)[2], int cap)
{
    const int ncg = w.N / 32, KB = w.K / 128;
    int       n   = 0;
    for (int shape = 0; shape <= kShapeWide2 && M <= 8192; ++shape) {
        if (shape == kShapeWide2 && M > 64) {
            continue;
        }
        const bool rows32 = shape >= 6 && shape <= 9;  // 32-row blocks on grid.z: any M
        if (rows32 ? (M <= 32 || M > 1024) : ((shape == 4 || shape == 5) != (M > 64))) {
            continue;  // one row block: identical to the base shape / 64-row shapes take M <= 64, 128-row tiles M > 64; beyond
                       // 1024 rows a 32-row block re-reads every weight unit > 32 times: never competitive
        }
        if (shape == 5 && w.N < 512) {
            continue;
        }
        int       cgn, S;
        dec32_shape_dims(shape, &cgn, &S);
        const int tiles = (ncg + cgn - 1) / cgn * (rows32 ? (M + 31) / 32 : (shape == 4 || shape == 5) ? (M + 127) / 128 : 1);
        for (int s = 1; s <= 16; ++s) {
            int per = (KB + s - 1) / s;
            per     = (per + S - 1) / S * S;
            if ((KB + per - 1) / per != s) {
                continue;  // not a distinct slicing
            }
            if ((s > 1 && tiles * s > (rows32 ? 512 : 320)) || tiles * s < 32 || (M > 256 && s > 4)) {
                continue;  // split-K beyond one (32-row shapes: two) workgroup(s) per CU / a handful of workgroups / slabs of MBs per slice
            }
            if (n < cap) {
                out[n][0] = shape;
                out[n][1] = s;
                ++n;
            }
        }
    }
    if (M <= 64) {
        // in-launch merged split-K (kShapeMerge + shape, round 6): 2 .. 4 slices of the <= 64-row tiles.  The last arriver of a tile reads
        // (slices - 1) x 64 x columns x 4 bytes on its critical path -- beyond 4 slices that is longer than the reduce launch it replaces
        const int base = n;
        for (int i = 0; i < base; ++i) {
            const int sh = out[i][0], sp = out[i][1];
            if (dec32_fold_shape(sh) && sh < kShapeMerge && sp >= 2 && sp <= 4 && n < cap) {
                out[n][0] = kShapeMerge + sh;
                out[n][1] = sp;
                ++n;
            }
        }
    }
    if (M <= 64) {  // the loader / consumer kernel (gemm_decode_lc.hip): 128-column tiles, whole stages of 4 k-blocks per slice
        const int tiles = (ncg + 3) / 4;
        for (int s = 1; s <= 16; ++s) {
            int per = (KB + s - 1) / s;
            per     = (per + 3) / 4 * 4;
            if ((KB + per - 1) / per != s || (s > 1 && tiles * s > 320) || tiles * s < 32) {
                continue;
            }
            if (n < cap) {
                out[n][0] = kShapeLC;
                out[n][1] = s;
                ++n;
            }
        }
    }
    if (M >= 1024 && M <= 8192 && w.N >= 256 && w.image16 != nullptr && n < cap) {  // gemm_prefill_f16.hip: the resident fp16 image, no split-K
        out[n][0] = kShapeF16;
        out[n][1] = 1;
        ++n;
    }
    if (M >= 256 && M <= 8192 && w.N >= 256) {  // gemm_prefill.hip: 256 x 256 tiles (<= 4 slices: slabs of MBs each)
        const int tiles = (ncg + 7) / 8 * ((M + 255) / 256);
        for (int s = 1; s <= 4; ++s) {
            const int per = (KB + s - 1) / s;
            if ((KB + per - 1) / per != s || (s > 1 && tiles * s > 512) || per < 8) {
                continue;
            }
            if (n < cap) {
                out[n][0] = kShapePre256;
                out[n][1] = s;
                ++n;
            }
        }
    }
    return n;
}

// Tiling for the decode GEMM: `shape` (see launch_dec32_shape) and the split-K count.  One workgroup per CU (128 KB of
// LDS): take the widest split that keeps <= 256 workgroups and >= one stage (4 k-blocks) per slice.
void dec32_pick(const LinearWeight& w, int M, int* shape_out, int* splits_out)
{
    dec32_pick_ex(w, M, shape_out, splits_out, true);
}

void dec32_pick_ex(const LinearWeight& w, int M, int* shape_out, int* splits_out, bool use_table)
{
    const int ncg = w.N / 32;
    const int KB  = w.K / 128;
    if (use_table && dec32_table_get(w.K, w.N, dec32_m_bucket(M), shape_out, splits_out, w.role)) {
        return;  // measured on this machine for exactly this problem (M <= 256) / for this size class of forwards (above)
    }
    static const int env_shape  = env_int2("TM_D32_SHAPE", -1);  // read once: this runs on every eager launch
    static const int env_splits = env_int2("TM_D32_SPLITS", 0);
    int              shape      = env_shape;
    if (shape < 0 || shape > 3) {
        shape = 0;
    }
    if (M > 64) {
        // M <= 256 (batch-128 decode, small admissions): 256-column tiles keep more workgroups alive; beyond that the
        // 512-column tile with two weight fragments per x-fragment read (TM_PRE64_MIN_M: first M that takes it)
        static const int pre64_from = env_int2("TM_PRE64_MIN_M", 257);
        // The 128 x 512 tile only when it alone puts ~one workgroup on most CUs; otherwise the 128 x 256 tile with at most 4
        // split-K slices (round 3, measured by the tuner at the Llama-3-8B shapes, us per layer incl. the consumer --
        // M = 512: w_qkv 49.3 -> 36.5, wo 44.9 -> 28.5, w2 74.2 -> 54.2; M = 1024: 66.8 -> 48.1, 61.4 -> 41.5, 105.5 -> 92.6;
        // M = 2048: wo 81.6 -> 61.2; from M = 4096 the wide tile wins everywhere: profiles/r03_gemm_tune_prefill_classes.txt)
        const int wgs5 = (ncg + 15) / 16 * ((M + 127) / 128);
        shape          = (M >= pre64_from && w.N >= 512 && wgs5 >= 192) ? 5 : 4;
    }
    // Narrow projections at a full decode batch (33 .. 64 rows, N <= 8192: w_qkv, wo, w2 of an 8B model): 64-column tiles
    // with as little split-K as still gives ~256 workgroups.  Split-K costs more at the kernel boundary than it buys inside
    // the kernel -- in a back-to-back chain the launch after a split-K GEMM starts 4.1 .. 6.6 us after its last workgroup
    // ended (MBs of dirty fp32 slabs) against 1.2 us behind a kernel that leaves fp16 outputs
    // (profiles/r02_gemm_boundary_gap.txt) -- so the k range stays on chip as far as the workgroup count allows:
    //   K <= 8192: shape 6 (32-row x 64-column tiles, two row blocks), splits = 256 / tiles  (w_qkv: 192 tiles x 1, wo: 128 x 2)
    //   K >  8192: shape 3 (64-row x 64-column tiles),                  splits = 256 / tiles  (w2: 64 tiles x 4)
    // These are the winners of the measured dispatch on the Llama-3-8B shapes (tune_decode_gemms, GEMM + consumer per layer:
    // wo 14.3 -> 10.5 us, w2 18.6 -> 17.1 us, w_qkv 12.2 -> 9.8 us; profiles/r02_gemm_tune_measurements.txt); with
    // TM_GEMM_TUNE=1 the engine measures instead of trusting this rule.  The two row halves of a shape-6 column tile are
    // gridDim.x * gridDim.y workgroups apart -- the same XCD when that is a multiple of 8: the second reader hits L2.
    // Round 4: the same rule holds for the TP-shard shapes (one rank of TP = 2 / 8 of Llama-3-8B / -70B: N = 768 .. 8192, K = 512 ..
    // 8192, ragged k-block counts) -- the tuner picks shape 6 for every one of them (profiles/r04_gemm_experiments_session2.txt,
    // call23: w2 1792 x 4096 6.2 us against 10.6 for the old fall-back, w_qkv 4096 x 768 7.7 against 10.8) -- with slices of at
    // least 16 k-blocks: below that a slab boundary costs more than the workgroups it adds.
    static const int rowhalf = env_int2("TM_D32_ROWHALF", 1);
    if (rowhalf && env_shape < 0 && M > 32 && M <= 64 && ncg >= 8 && ncg <= 256) {
        const int nshape = KB <= 64 ? 6 : 3;
        const int tiles  = ((ncg + 1) / 2) * (nshape == 6 ? 2 : 1);
        int       sp     = 1;
        for (int s2 = 2; s2 <= 16 && tiles * s2 <= 256; ++s2) {
            const int per = ((KB + s2 - 1) / s2 + 3) / 4 * 4;
            if ((KB + per - 1) / per == s2 && per >= 16) {
                sp = s2;
            }
        }
        *shape_out  = nshape;
        *splits_out = env_splits > 0 ? env_splits : sp;
        return;
    }
    int cgn, S;
    dec32_shape_dims(shape, &cgn, &S);
    const int col_wgs = (ncg + cgn - 1) / cgn * ((M + 127) / 128);
    int       splits  = 1;
    static const int min_kb = env_int2("TM_D32_MIN_KB", 4);
    for (int s = 2; s <= 16; ++s) {  // the engine's slab workspace holds 16 splits
        int per = (KB + s - 1) / s;
        per     = (per + S - 1) / S * S;
        const int eff = (KB + per - 1) / per;  // splits after rounding the slice to whole stages
        if (eff != s) {
            continue;
        }
        if (col_wgs * s <= 256 && per >= (M > 64 ? 8 : min_kb) && (M <= 256 || s <= 4)) {
            splits = s;  // (prefill-sized forwards: slabs of MBs per slice -- never more than 4)
        }
    }
    splits      = env_splits > 0 ? env_splits : splits;
    *shape_out  = shape;
    *splits_out = splits < 1 ? 1 : (splits > KB ? KB : splits);
}

// y / slabs as launch_linear: *slabs_out = number of fp32 slabs written into `workspace` (1 = direct epilogue)
// (shape, M) pairs a folded launch can run: see launch_linear_dec32
bool dec32_fold_shape_m(int shape, int M, bool producer)
{
    if (M <= 64) {
        return dec32_fold_shape(shape);
    }
    if (dec32_is_merge_shape(shape)) {
        shape -= kShapeMerge;
        if (producer) {
            return false;
        }
    }
    return M <= kFoldMaxRows && ((shape >= 6 && shape <= 9) || (shape == 4 && !producer));
}

bool dec32_fold_shape(int shape)
{
    if (dec32_is_merge_shape(shape)) {
        shape -= kShapeMerge;  // the same kernels; consumers only (a folded producer merges in the launch whatever its shape says)
    }
    return (shape >= 0 && shape <= 3) || (shape >= 6 && shape <= 9) || shape == kShapeWide2;  // gemm_dec32_kernel with <= 64-row blocks
}

int launch_linear_dec32(const LinearWeight& w, const half_t* x, int ldx, half_t* y, int ldy, int M, bool gated_silu, int shape,
                        int splits, float* workspace, int* slabs_out, hipStream_t st, NormFold* nf)
{
    const bool produce = nf && nf->resid != nullptr;
    const bool consume = nf && nf->ss_in != nullptr;
    bool       merge   = false;
    if (dec32_is_merge_shape(shape)) {  // split-K merged in the launch (fp16 / gated epilogues); a producer does that anyway
        shape -= kShapeMerge;
        merge = !produce && splits > 1;
        TM_REQUIRE(!merge || (nf && nf->tickets && (M <= 64 || (M <= kFoldMaxRows && shape >= 6 && shape <= 9))),
                   "merged split-K: arrival counters (NormFold::tickets), M <= 64 (32-row-block tiles: <= 128)");
    }
    // folded RMSNorm: M <= 64 on every <= 64-row decode tile; 64 < M <= kFoldMaxRows (round 6, BASELINE config 3 = batch 128) on the 32-row-block
    // tiles 6..9 (producer and consumer: one ticket / one row of sums per (column tile, row block)) and, consumer only, on the 128-row tile 4
    TM_REQUIRE(!nf || !(produce || consume)
                   || (M <= 64 ? dec32_fold_shape(shape) : (M <= kFoldMaxRows && ((shape >= 6 && shape <= 9) || (shape == 4 && !produce)))),
               "folded RMSNorm: decode tiles 0..3 / 6..10 at M <= 64; 32-row-block tiles 6..9 (and tile 4 as a consumer) up to 128 rows");
    TM_REQUIRE(!produce || (!gated_silu && nf->norm_w && nf->ss_out && ldy % 4 == 0), "folded RMSNorm, producer: residual, norm weight, sums");
    TM_REQUIRE(!consume || (nf->ss_tiles >= 1 && nf->inv_h > 0.f), "folded RMSNorm, consumer: tiles and 1 / H");
    TM_REQUIRE(w.packed32 != nullptr && w.N % 32 == 0, "decode GEMM: P32 layout missing");
    if (shape == kShapeF16 && (w.image16 == nullptr || M <= 64)) {
        shape = kShapePre256;  // a table entry for a linear without the fp16 image (operator-level handle, TM_PREFILL_F16_IMAGE=0): the fused 256 x 256 tile
    }
    TM_REQUIRE(M >= 1 && shape >= 0 && (shape <= 9 || shape == kShapeWide2 || shape == kShapeLC || shape == kShapePre256 || shape == kShapeF16)
                   && (shape >= 6 || (M <= 64) == (shape < 4)) && (shape != kShapeLC || M <= 64) && (shape != kShapePre256 || M > 64)
                   && (shape != kShapeWide2 || M <= 64),
               "decode GEMM: shapes 0..3, 10 and 11 take M <= 64, shapes 4 / 5 / 12 take M > 64, shapes 6..9 any M");
    TM_REQUIRE(ldx % 8 == 0, "x rows must be 16-byte aligned");
    int cgn, S;
    dec32_shape_dims(shape, &cgn, &S);
    Dec32Params p{};
    p.x       = x;
    p.ldx     = ldx;
    p.wp      = shape == kShapeF16 ? (const void*)w.image16 : w.packed32;
    if (shape == kShapeF16) {
        splits = 1;
    }
    p.y       = y;
    p.ldy     = ldy;
    p.partial = workspace;
    p.M       = M;
    p.N       = w.N;
    p.K       = w.K;
    p.KB      = w.K / 128;
    p.ncg     = w.N / 32;
    int per   = (p.KB + splits - 1) / splits;
    per       = (per + S - 1) / S * S;
    per       = per > p.KB ? p.KB : per;
    splits    = (p.KB + per - 1) / per;
    TM_REQUIRE(splits == 1 || workspace != nullptr, "split-K needs a workspace");
    p.kb_per_split = per;
    merge          = merge && splits > 1;
    p.epilogue     = produce ? 3 : (splits > 1 && !merge) ? 2 : (gated_silu ? 1 : 0);
    if (merge) {
        p.merge   = 1;
        p.tickets = nf->tickets;
    }
    if (produce) {
        TM_REQUIRE(splits == 1 || nf->tickets != nullptr, "folded RMSNorm, producer with split-K: arrival counters");
        p.resid   = nf->resid;
        p.norm_w  = nf->norm_w;
        p.ss_out  = nf->ss_out;
        p.tickets = nf->tickets;
    }
    if (consume) {
        p.ss_in    = nf->ss_in;
        p.ss_tiles = nf->ss_tiles;
        p.ss_inv_h = nf->inv_h;
        p.ss_eps   = nf->eps;
    }
    static const int wt = env_int2("TM_D32_WT", 1);  // measured (tools/trace_boundary.py, profiles/r02_gemm_boundary_gap.txt): -0.4..-0.9 us per split-K launch
    p.wt           = wt;
    dim3      grid((p.ncg + cgn - 1) / cgn, splits,
                   (shape == kShapePre256 || shape == kShapeF16) ? (M + 255) / 256 : (shape == kShapeLC || shape == kShapeWide2) ? 1 : shape >= 6 ? (M + 31) / 32 : shape >= 4 ? (M + 127) / 128 : 1);
    static const char* const role_tag[6] = {"gemm", "w_qkv", "wo", "w1w3", "w2", "lm_head"};
    p.dbg        = gemm_trace_for((size_t)grid.x * grid.y * grid.z, role_tag[w.role >= 0 && w.role <= 5 ? w.role : 0], grid.x, grid.y, grid.z);
    const int rc = shape == kShapeF16 ? launch_f16_256(p, grid, st) :
                   shape == kShapePre256 ? launch_pre256(p, grid, st) :
                   shape == kShapeLC ? launch_dec_lc(p, grid, st) :
                   shape == kShapeWide2 ? (M <= 32 ? launch_dec32_shape<1>(p, grid, shape, st) : launch_dec32_shape<2>(p, grid, shape, st)) :
                   shape >= 6 ? launch_dec32_shape<1>(p, grid, dec32_base_shape(shape), st) :
                   shape >= 4 ? launch_dec32_shape<4>(p, grid, shape, st) :
                   M <= 32    ? launch_dec32_shape<1>(p, grid, shape, st) :
                                launch_dec32_shape<2>(p, grid, shape, st);
    if (rc) {
        return rc;
    }
    if (produce) {
        nf->tiles_out = (int)grid.x;
        splits        = 1;  // the slabs were consumed inside the launch
    }
    if (merge) {
        splits = 1;
    }
    if (slabs_out) {
        *slabs_out = splits;
    }
    return 0;
}

}  // namespace tmk
