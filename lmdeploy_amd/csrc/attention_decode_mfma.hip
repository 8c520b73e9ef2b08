// Split-K paged flash-decode for int8 / int4 KV on the matrix cores (gfx950) -- the kernel the headline metric runs.
//
// Same contract as attention_decode.hip (which stays the path for fp16 KV): replaces dispatchDecoding +
// invokeReduceV3 (src/turbomind/kernels/attention/decoding.cu:12-39, attention_universal.h:355-553,
// impl_81616.h:313-349,510-591, reduce.cu:13-226).
//
// Why a second kernel: the VALU version spends ~1400 wave-instructions per 16 KB of KV (dequant + dot products) and is
// instruction-issue bound at ~45 % of the HBM roofline.  Here both contractions run on v_mfma_f32_16x16x32_f16 and
// the per-element dequantisation disappears algebraically: with (s_t, z_t) the per-token scale / zero,
//     S[h,t]  = sum_d q[h,d] (s_t kq[t,d] + z_t)  = s_t * (q . kq[t]) + z_t * sum_d q[h,d]
//     O[h,d]  = sum_t P[h,t] (s_t vq[t,d] + z_t)  = sum_t (P s_t)[h,t] vq[t,d]  +  sum_t P[h,t] z_t
// so the MFMAs contract the RAW integer codes -- turned into fp16 with ONE v_perm per two codes and no arithmetic at
// all: the byte b becomes the fp16 SUBNORMAL 0x00bb = b * 2^-24, which the matrix core multiplies exactly; the factor
// 2^24 is folded into the K scale and into the final O -- and the scales touch 4 values per lane per tile instead of
// 256.  (Round 1 used 0x64bb = 1024 + b and removed 1024 * sum_t (P s_t) at the end: the full-size parity test of
// round 2 found the cancellation -- O ~ 1031 * sum P against a signal of ~7 * sum P for int4 -- costing up to 9e-3
// absolute on 3 of 2048 (sequence, head) pairs at ctx ~1.6k; the subnormal form has no bias and is also cheaper.)  This is NOT the reference's rounding sequence
// (no per-element fp16 rounding of k^ / v^): results agree with it to ~1e-3 relative, inside the stated tolerance
// (reference Compare thresholds rtol 1e-2 / atol 1e-4, kernels/attention/test_utils.h:12-14).
//
// Data path per wave and 64-token cache block: the block's pointer by a scalar load from the block table (one block ahead), 16
// coalesced 16-B global_load (8 whole 128-B rows per instruction; address space 1 spelled out -- FLAT loads can only be waited for
// with vmcnt(0) lgkmcnt(0), see load_tile) -> registers (next block in flight while this one is contracted) -> wave-private LDS image [token][128 B] with the
// 16-B chunks XOR-swizzled by (token>>1)&7 -> K operand by ds_read_b64 (token rows, conflict free), V operand by
// ds_read_b64_tr_b8 (the hardware byte transpose: 8 tokens of one head-dim column per lane).
// The score tile comes out of the MFMA as S^T with 4 consecutive tokens per lane; the second contraction uses the
// k-slot mapping (g, e) -> token 4g+e (e<4) / 16+4g+(e-4), so P feeds it without any cross-lane movement and the
// transpose-read simply addresses those rows.
#include "tm_common.h"
#include "tm_kernels.h"
#include <type_traits>

namespace tmk {

typedef int v2i __attribute__((ext_vector_type(2)));

// byte b -> fp16 bit pattern 0x00bb: the SUBNORMAL b * 2^-24, exact, no bias to remove afterwards (selector byte 0x0c =
// constant 0x00).  The f16 MFMA consumes subnormal inputs exactly (checked by the parity tests: a flush would zero every
// score), so both contractions run on code * 2^-24 and the factor 2^24 is folded into the K scale / the final O.
__device__ __forceinline__ half8_t bytes8_to_f16_subnormal(u32x2 w)
{
    const uint32_t a0 = __builtin_amdgcn_perm(0u, w[0], 0x0c010c00u);
    const uint32_t a1 = __builtin_amdgcn_perm(0u, w[0], 0x0c030c02u);
    const uint32_t a2 = __builtin_amdgcn_perm(0u, w[1], 0x0c010c00u);
    const uint32_t a3 = __builtin_amdgcn_perm(0u, w[1], 0x0c030c02u);
    return bit_cast<half8_t>(u32x4{a0, a1, a2, a3});
}
constexpr float kTwo24 = 16777216.0f;

constexpr int kWaveLds = 16384 + 512;  // K image 8 KB | V image 8 KB | (k_param, v_param) per token

#define GLOBAL_AS __attribute__((address_space(1)))
#define CONST_AS __attribute__((address_space(4)))

// 8 consecutive columns of row b of the qkv GEMM output: fp16 result, or the in-order sum of the fp32 split-K slabs
// rounded to fp16 exactly like the GEMM epilogue / splitk_reduce_kernel would.  Split in an issue half (all loads of
// the first four slabs in flight, nothing consumed) and a finish half, so that the q, new-K/V and first cache-block
// loads of the prologue overlap instead of paying one memory round trip each.
struct QkvRaw {
    floatx4 a0[4], a1[4];
};

// Branch-free (round 5): fp16 input and slab input issue the same 8 loads (fp16: eight times the same 16 bytes).  A branch around the loads
// makes the loaded registers phi values, the compiler copies them (v_mov) where the branches meet -- and a copy waits for the load: the
// cache-block loads behind it were issued one memory round trip late (seen in the ISA).  Address space 1 spelled out for the same reason
// as in load_tile.
__device__ __forceinline__ QkvRaw qkv_issue(const DecodeAttnParams& p, int b, int col)
{
    QkvRaw       r;
    const bool   f16in = p.qkv_splits == 0;
    const char*  base  = f16in ? (const char*)(p.qkv_f16 + (size_t)b * p.qkv_n + col) : (const char*)(p.qkv_slabs + (size_t)b * p.qkv_n + col);
    const size_t slab  = f16in ? 0 : (size_t)p.batch * p.qkv_n * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const char* src = base + (size_t)min(j, max(p.qkv_splits - 1, 0)) * slab;
        r.a0[j]         = *(const GLOBAL_AS floatx4*)src;
        r.a1[j]         = *(const GLOBAL_AS floatx4*)(src + (f16in ? 0 : 16));
    }
    return r;
}

__device__ __forceinline__ half8_t qkv_finish(const DecodeAttnParams& p, int b, int col, const QkvRaw& r)
{
    float acc[8] = {};
#pragma unroll
    for (int j = 0; j < 4; ++j) {  // in slab order: bit-identical to splitk_reduce_kernel
        if (j < p.qkv_splits) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[e] += r.a0[j][e];
                acc[4 + e] += r.a1[j][e];
            }
        }
    }
    if (p.qkv_splits > 4) {  // deeper split-K than the engine uses for this projection: plain loop
        const size_t slab = (size_t)p.batch * p.qkv_n;
        const float* base = p.qkv_slabs + (size_t)b * p.qkv_n + col;
        for (int s = 4; s < p.qkv_splits; ++s) {
            const floatx4 a0 = *(const floatx4*)(base + s * slab);
            const floatx4 a1 = *(const floatx4*)(base + s * slab + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[e] += a0[e];
                acc[4 + e] += a1[e];
            }
        }
    }
    half8_t o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        o[e] = (half_t)acc[e];
    }
    const half8_t direct = bit_cast<half8_t>(r.a0[0]);
    return p.qkv_splits == 0 ? direct : o;
}

// interleaved-pair RoPE in fp16 (rotary_embedding.h:169-181): cs = 4 (cos, sin) pairs for these 8 channels
__device__ __forceinline__ half8_t rope8(half8_t x, half8_t cs)
{
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const half_t c = cs[2 * i], s = cs[2 * i + 1];
        const half_t x0 = x[2 * i], x1 = x[2 * i + 1];
        const half_t a0 = c * x0, a1 = s * x1, b0 = c * x1, b1 = s * x0;
        x[2 * i]     = a0 - a1;
        x[2 * i + 1] = b0 + b1;
    }
    return x;
}

// 32 int4 codes (16 bytes, 4 words of 8 nibbles in the cache's order [0,2,4,6,1,3,5,7], quantization.h:459-471) -> 32
// bytes in element order: lo / hi nibbles of a word are the elements (0,4,1,5) / (2,6,3,7)
__device__ __forceinline__ void expand_u4x32(u32x4 w, u32x4& a, u32x4& b)
{
    uint32_t o[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint32_t lo = w[j] & 0x0f0f0f0fu;
        const uint32_t hi = (w[j] >> 4) & 0x0f0f0f0fu;
        // v_perm_b32(hi, lo, sel): selector bytes 0-3 pick from lo, 4-7 from hi
        o[2 * j]     = __builtin_amdgcn_perm(hi, lo, 0x06040200u);  // (e0, e1, e2, e3) = (lo.b0, lo.b2, hi.b0, hi.b2)
        o[2 * j + 1] = __builtin_amdgcn_perm(hi, lo, 0x07050301u);  // (e4, e5, e6, e7) = (lo.b1, lo.b3, hi.b1, hi.b3)
    }
    a = u32x4{o[0], o[1], o[2], o[3]};
    b = u32x4{o[4], o[5], o[6], o[7]};
}

// BITS = 8: the cache bytes are the codes.  BITS = 4: the block is half as large in HBM; the nibbles are expanded to
// one byte per code when the block is written to the wave's LDS image, everything after that is the int8 path
// (scales / zeros are per token either way, quantization.h:316-366).
template<bool FUSED, int BITS>
__global__ __launch_bounds__(256, 2) void decode_attention_i8_mfma_kernel(DecodeAttnParams p, int head_chunks, int hpw)
{
    static_assert(BITS == 8 || BITS == 4, "int8 / int4 KV");
    constexpr int D = 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // every kernel argument of the prologue in ONE batch of scalar loads: hipcc otherwise fetches late-used fields behind branches
    // -- five dependent s_load round trips in front of the first cache-block load (seen in the ISA, round 5)
    asm volatile("" ::"s"(p.q), "s"(p.q_stride), "s"(p.out), "s"(p.k_len), "s"(p.batch), "s"(p.q_heads), "s"(p.scale_log2), "s"(p.splits),
                 "s"(p.partial_o), "s"(p.partial_ml), "s"(p.cache.block_ptrs), "s"(p.cache.cu_block_nums), "s"(p.cache.block_stride),
                 "s"(p.cache.layer_offset), "s"(p.cache.layout.kv_heads), "s"(p.cache.layout.head_dim), "s"(p.cache.layout.block_len),
                 "s"(p.cache.layout.bits));
    asm volatile("" ::"s"(p.qkv_slabs), "s"(p.qkv_f16), "s"(p.qkv_splits), "s"(p.qkv_n), "s"(p.cos_sin), "s"(p.max_pos), "s"(p.dbg),
                 "s"(head_chunks), "s"(hpw));
    const KvLayout L = p.cache.layout;

    const int kv_head = blockIdx.x / head_chunks;
    const int chunk   = blockIdx.x - kv_head * head_chunks;
    const int b       = blockIdx.y;
    const int split   = blockIdx.z;
    const int group   = p.q_heads / L.kv_heads;
    const int head0   = kv_head * group + chunk * hpw;

    const int wgid = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    if (p.dbg && threadIdx.x == 0) {
        p.dbg[wgid * 8 + 0] = __builtin_amdgcn_s_memrealtime();
        // XCC_ID (hwreg 20) in the high word, HW_ID (hwreg 4: wave / SIMD / CU / SE) in the low word
        p.dbg[wgid * 8 + 4] = ((uint64_t)__builtin_amdgcn_s_getreg(20 | (31 << 11)) << 32) | (uint32_t)__builtin_amdgcn_s_getreg(4 | (31 << 11));
    }
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i16  = lane & 15;
    const int g    = lane >> 4;

    char* Kt = smem + wave * kWaveLds;
    char* Vt = Kt + 8192;
    char* Pm = Kt + 16384;

    // Context length and block pointers are SCALAR loads (constant address space spelled out: the table and k_len are read-only during the
    // launch): a few hundred ns each through the scalar cache instead of a VMEM round trip, and no vector load whose result the block
    // loop would have to wait for with vmcnt(0).  Round 4 kept 64 pointers in a VGPR (v_readlane per block) and re-fetched the window
    // with a vector load; the pointer of the NEXT block to prefetch is now loaded one iteration ahead (ptr_next below).
    const int       bstride = p.cache.block_stride;
    const uint64_t* blocks  = p.cache.block_ptrs
                             + (bstride > 0 ? (size_t)b * bstride : (size_t) * (const CONST_AS int*)(p.cache.cu_block_nums + b));
    auto block_ptr = [&](int tile) -> const char* {  // tile: wave-uniform
        return (const char*)*(const CONST_AS uint64_t*)(blocks + __builtin_amdgcn_readfirstlane(tile));
    };

    const int ctx        = *(const CONST_AS int*)(p.k_len + b);
    const int tiles      = (ctx + 63) >> 6;
    const int per_split  = (tiles + p.splits - 1) / p.splits;
    const int tile_begin = split * per_split;
    const int tile_end   = min(tile_begin + per_split, tiles);

    // ---- q^T fragments (B operand of S^T = K q^T): lane (head = i16, g) holds q[head][32dd + 8g .. +8) -----------
    half8_t    qf[4];
    float      q1 = 0.f;  // sum_d q[head][d]
    const bool hv = i16 < hpw;

    floatx4 O[8];
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) {
        O[dt] = floatx4{0.f, 0.f, 0.f, 0.f};
    }
    float m = -INFINITY, lsum = 0.f, zacc = 0.f;  // per head column i16, partial over this lane's tokens
    const float sc = p.scale_log2;

    const int       koff   = L.k_data(kv_head, 0);
    const int       voff   = L.v_data(kv_head, 0);
    const int       kpoff  = L.k_param(kv_head, 0);
    const int       vpoff  = L.v_param(kv_head, 0);

    constexpr int NR = BITS == 8 ? 8 : 4;  // 16-byte loads per lane per K (V) block
    u32x4    kreg[NR] = {}, vreg[NR] = {};
    uint32_t kpr = 0, vpr = 0;
    // FIRST: the first block a wave touches -- the only one that can be the (partial) newest block of the sequence.  Rows past
    // the context are masked to exactly nothing below whatever the registers hold, so those lanes re-read the last valid row
    // instead (same cache lines: no HBM traffic, no branches) -- on average half a block of HBM reads per (sequence, kv head)
    // and launch, all of it on the critical path of the wave that owns one block more than the others.
    // FIRST is issued by EVERY wave, branch-free (a wave without a block -- tile < tile_begin -- reads row 0 of the clamped block with every
    // lane and never uses it): behind a branch, the static s_waitcnt counts of the prologue would have to assume the worst.
    auto load_tile = [&](const char* blk, int tile, auto FIRST) {
        const char* base = blk + p.cache.layer_offset;
        const int   last = decltype(FIRST)::value ? (tile < tile_begin ? 0 : min(63, ctx - tile * 64 - 1)) : 63;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            // int8: token (lane/8 + 8r), 16-byte chunk lane%8 of its 128 bytes; int4: token (lane/4 + 16r), chunk lane%4 of 64
            const int row = BITS == 8 ? (lane >> 3) + 8 * r : (lane >> 2) + 16 * r;
            const int off = BITS == 8 ? min(row, last) * 128 + (lane & 7) * 16 : min(row, last) * 64 + (lane & 3) * 16;
            // non-temporal: every cache byte is read ONCE per launch by ONE workgroup -- measured (round 3, call 8;
            // profiles/r03_attention_nontemporal_loads.txt) 32.4 -> 30.3 us at ctx 1040, 43.3 -> 39.5 us at ctx 1536
            // address space 1 spelled out: a pointer read from the block table is "generic" to the compiler, the loads were FLAT ones
            // (both counters, waited for with vmcnt(0) lgkmcnt(0)) until round 5
            kreg[r]       = __builtin_nontemporal_load((const GLOBAL_AS u32x4*)(base + koff + off));
            vreg[r]       = __builtin_nontemporal_load((const GLOBAL_AS u32x4*)(base + voff + off));
        }
        kpr = *(const GLOBAL_AS uint32_t*)(base + kpoff + lane * 4);
        vpr = *(const GLOBAL_AS uint32_t*)(base + vpoff + lane * 4);
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            if constexpr (BITS == 8) {
                const int t   = (lane >> 3) + 8 * r;
                const int pos = t * 128 + (((lane & 7) ^ ((t >> 1) & 7)) << 4);
                *(u32x4*)(Kt + pos) = kreg[r];
                *(u32x4*)(Vt + pos) = vreg[r];
            }
            else {
                // 32 nibbles -> the two 16-byte chunks 2c, 2c+1 of the token's 128-byte image row
                const int t  = (lane >> 2) + 16 * r;
                const int c  = (lane & 3) * 2;
                const int sw = (t >> 1) & 7;
                u32x4     a, b;
                expand_u4x32(kreg[r], a, b);
                *(u32x4*)(Kt + t * 128 + ((c ^ sw) << 4))       = a;
                *(u32x4*)(Kt + t * 128 + (((c + 1) ^ sw) << 4)) = b;
                expand_u4x32(vreg[r], a, b);
                *(u32x4*)(Vt + t * 128 + ((c ^ sw) << 4))       = a;
                *(u32x4*)(Vt + t * 128 + (((c + 1) ^ sw) << 4)) = b;
            }
        }
        *(u32x2*)(Pm + lane * 8) = u32x2{kpr, vpr};
    };

    // ---- fused prologue -----------------------------------------------------------------------------------------
    // q: wave w builds the 32-channel slice dd = w (slab sum -> fp16 -> RoPE); the four slices are exchanged through
    // LDS.  New token's K/V: wave 0 of the WG that owns the newest block, lanes 0-15 = K row, 16-31 = V row (32-63
    // mirror them); same arithmetic as kv_rope_store_kernel (bit-exact cache bytes).  The codes go to the cache
    // block AND, after the newest block's image has been written to LDS, are patched into that image: no global
    // read-after-write.  All loads (q slabs, k/v slabs, RoPE rows, first cache block) are issued before any is used.
    const bool owns_newest = FUSED && wave == 0 && tile_end == tiles && tile_end > tile_begin;
    u32x2      nq          = {0u, 0u};
    uint32_t   npar        = 0;
    const int  nti         = (ctx - 1) & 63;
    int        tile        = tile_end - 1 - wave;  // newest -> oldest, waves interleaved
    // pointers of this wave's first and second block (scalar loads; the clamp keeps the address valid when the wave has fewer blocks)
    // block tiles - 1 always exists for ctx >= 1 (the contract: tm_decode_attention* require k_len >= 1; the engine parks free slots on a
    // dummy block with k_len = 1).  The outer max keeps the index at 0 for a k_len = 0 entry all the same: its first block-table entry is
    // read (and must be a mapped block), never blocks[-1]
    auto        in_range = [&](int t) { return max(min(max(t, tile_begin), tiles - 1), 0); };
    const char* ptr_cur  = block_ptr(in_range(tile));
    const char* ptr_next = block_ptr(in_range(tile - 4));
    if constexpr (FUSED) {
        // One fixed order, no branch around a load (round 5): q slice, the new token's K / V slice (every wave loads one, the owner uses
        // it), the RoPE rows, then the first cache block -- all in flight together, waited for one by one with exact counts.  Before, each
        // of these waited for the one in front of it (k_len -> pointer window -> qkv slabs -> cache block: four VMEM round trips).
        const int     pos   = min(ctx - 1, p.max_pos - 1);
        const int     l16   = lane & 15;
        const bool    isv   = (lane & 16) != 0;
        const int     qcol  = (head0 + (hv ? i16 : 0)) * D + wave * 32 + g * 8;
        const int     kvcol = (p.q_heads + (isv ? L.kv_heads : 0) + kv_head) * D + l16 * 8;
        // no RoPE table: the rows are read from the (valid, unused) start of the output instead of being skipped
        const half_t* cs    = p.cos_sin ? (const half_t*)p.cos_sin + (size_t)pos * D : (const half_t*)p.out;
        const QkvRaw  rq    = qkv_issue(p, b, qcol);
        const QkvRaw  rk    = qkv_issue(p, b, kvcol);
        const half8_t csq   = *(const GLOBAL_AS half8_t*)(cs + wave * 32 + g * 8);
        const half8_t csk   = *(const GLOBAL_AS half8_t*)(cs + l16 * 8);
        asm volatile("" ::: "memory");  // the scheduler keeps this order
        load_tile(ptr_cur, tile, std::true_type{});
        asm volatile("" ::: "memory");
        half8_t t = qkv_finish(p, b, qcol, rq);
        if (p.cos_sin) {
            t = rope8(t, csq);
        }
        if (!hv) {
            t = half8_t{};
        }
        *(half8_t*)(smem + 4 * kWaveLds + (wave * 64 + lane) * 16) = t;
        if (owns_newest) {
            half8_t x = qkv_finish(p, b, kvcol, rk);
            if (!isv && p.cos_sin) {
                x = rope8(x, csk);
            }
            float mx = -INFINITY, mnn = -INFINITY;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                mx  = fmaxf(mx, (float)x[e]);
                mnn = fmaxf(mnn, -(float)x[e]);
            }
            mx  = fmaxf(mx, dpp_f32<DPP_XOR1>(mx));
            mx  = fmaxf(mx, dpp_f32<DPP_XOR2>(mx));
            mx  = fmaxf(mx, dpp_f32<DPP_HMIRR>(mx));
            mx  = fmaxf(mx, dpp_f32<DPP_ROR8>(mx));
            mnn = fmaxf(mnn, dpp_f32<DPP_XOR1>(mnn));
            mnn = fmaxf(mnn, dpp_f32<DPP_XOR2>(mnn));
            mnn = fmaxf(mnn, dpp_f32<DPP_HMIRR>(mnn));
            mnn = fmaxf(mnn, dpp_f32<DPP_ROR8>(mnn));
            const float  mn    = -mnn;
            const half_t scale = (half_t)((mx - mn) * (1.0f / (float)((1 << BITS) - 1)));
            const half_t zero  = (half_t)mn;
            const half_t inv   = (half_t)(1.0f / (float)scale);
            uint32_t     qv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const half_t d = x[e] - zero;
                const half_t y = d * inv;
                float        r = __builtin_rintf((float)y);
                r              = (r == r) ? r : 0.0f;
                r              = fminf(fmaxf(r, 0.0f), BITS == 8 ? 255.0f : 15.0f);
                qv[e]          = (uint32_t)r;
            }
            nq[0] = qv[0] | (qv[1] << 8) | (qv[2] << 16) | (qv[3] << 24);
            nq[1] = qv[4] | (qv[5] << 8) | (qv[6] << 16) | (qv[7] << 24);
            npar  = bit_cast<uint32_t>(half2_t{scale, zero});
            char* blk = (char*)ptr_cur + p.cache.layer_offset;  // the owner's first block IS the newest one
            if (lane < 32) {
                if constexpr (BITS == 8) {
                    *(u32x2*)(blk + (isv ? L.v_data(kv_head, nti) : L.k_data(kv_head, nti)) + l16 * 8) = nq;
                }
                else {  // nibble i of the word = element [0,2,4,6,1,3,5,7][i] (quantization.h:459-471)
                    const uint32_t w4 = qv[0] | (qv[2] << 4) | (qv[4] << 8) | (qv[6] << 12) | (qv[1] << 16) | (qv[3] << 20)
                                        | (qv[5] << 24) | (qv[7] << 28);
                    *(uint32_t*)(blk + (isv ? L.v_data(kv_head, nti) : L.k_data(kv_head, nti)) + l16 * 4) = w4;
                }
                if (l16 == 0) {
                    *(uint32_t*)(blk + (isv ? L.v_param(kv_head, nti) : L.k_param(kv_head, nti))) = npar;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int dd = 0; dd < 4; ++dd) {
            qf[dd] = *(const half8_t*)(smem + 4 * kWaveLds + (dd * 64 + lane) * 16);
        }
    }
    else {
        load_tile(ptr_cur, tile, std::true_type{});
        const half_t* qp = p.q + (size_t)b * p.q_stride + (size_t)(head0 + (hv ? i16 : 0)) * D;
#pragma unroll
        for (int dd = 0; dd < 4; ++dd) {
            half8_t t = *(const half8_t*)(qp + dd * 32 + g * 8);
            if (!hv) {
                t = half8_t{};
            }
            qf[dd] = t;
        }
    }
#pragma unroll
    for (int dd = 0; dd < 4; ++dd) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            q1 += (float)qf[dd][e];
        }
    }
    q1 += __shfl_xor(q1, 16);
    q1 += __shfl_xor(q1, 32);
    if (p.dbg && threadIdx.x == 0) {
        p.dbg[wgid * 8 + 1] = __builtin_amdgcn_s_memrealtime();
    }
    bool patch = owns_newest;
    for (; tile >= tile_begin; tile -= 4) {
        store_tile();  // wave-private LDS: in-order DS pipeline, no barrier needed
        if (FUSED && patch) {  // wave-uniform, first iteration of wave 0 only: the new token's codes -> LDS image
            patch = false;
            if (lane < 32) {
                const int l16 = lane & 15;
                char*     img = lane >= 16 ? Vt : Kt;
                *(u32x2*)(img + nti * 128 + ((((l16 >> 1) ^ ((nti >> 1) & 7))) << 4) + (l16 & 1) * 8) = nq;
                if (l16 == 0) {
                    *(uint32_t*)(Pm + nti * 8 + (lane >= 16 ? 4 : 0)) = npar;
                }
            }
        }
        const int ntok = min(64, ctx - tile * 64);
        if (tile - 4 >= tile_begin) {
            load_tile(ptr_next, tile - 4, std::false_type{});  // next block streams in while this one is contracted
            ptr_next = block_ptr(in_range(tile - 8));          // and the pointer of the one after it (scalar load, used next iteration)
        }

        // ---- S^T = K q^T on raw codes -----------------------------------------------------------------------
        floatx4 S[4];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            S[tt]       = floatx4{0.f, 0.f, 0.f, 0.f};
            const int t = 16 * tt + i16;
#pragma unroll
            for (int dd = 0; dd < 4; ++dd) {
                const int   u   = 4 * dd + g;
                const int   off = t * 128 + ((((u >> 1) ^ ((t >> 1) & 7))) << 4) + (u & 1) * 8;
                const u32x2 kb  = *(const u32x2*)(Kt + off);
                const half8_t a = bytes8_to_f16_subnormal(kb);  // code * 2^-24, exact
                S[tt]           = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, qf[dd], S[tt], 0, 0, 0);
            }
        }

        // ---- scales, mask, online softmax (lane: head column i16, tokens 16tt + 4g + r) ----------------------
        // Per score: s = ks*2^24*R + kz*q1 ; p = exp2(s*c - m*c) ; L += p ; Z += p*vz ; P' = h(p*vs).
        // Explicit fmaf() with fp16 operands lowers to v_fma_mix_f32 (no separate converts); the validity mask is
        // only evaluated for the single partial block of a sequence (wave-uniform branch).
        uint32_t kp[4][4], vp[4][4];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            const u32x4 pa_ = *(const u32x4*)(Pm + (16 * tt + 4 * g) * 8);       // tokens +0, +1
            const u32x4 pb_ = *(const u32x4*)(Pm + (16 * tt + 4 * g) * 8 + 16);  // tokens +2, +3
            kp[tt][0] = pa_[0], kp[tt][1] = pa_[2], kp[tt][2] = pb_[0], kp[tt][3] = pb_[2];
            vp[tt][0] = pa_[1], vp[tt][1] = pa_[3], vp[tt][2] = pb_[1], vp[tt][3] = pb_[3];
        }
        const bool partial = ntok < 64;  // wave-uniform: only the newest block of a sequence can be partial
        float sv[4][4];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const half2_t kk = bit_cast<half2_t>(kp[tt][r]);
                sv[tt][r]        = __builtin_fmaf((float)kk[0] * kTwo24, S[tt][r], (float)kk[1] * q1);
            }
        }
        if (partial) {  // tokens past the context: score -inf, V params (0, 0) -> contribute exactly nothing
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (16 * tt + 4 * g + r >= ntok) {
                        sv[tt][r] = -INFINITY;
                        vp[tt][r] = 0u;
                    }
                }
            }
        }
        float tmax = -INFINITY;
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                tmax = fmaxf(tmax, sv[tt][r]);
            }
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
        const float mnew  = fmaxf(m, tmax);
        const float alpha = (m == -INFINITY) ? 0.f : fast_exp2((m - mnew) * sc);
        // wave-uniform: did any head's max move?  (alpha == 1 exactly otherwise)
        if (__builtin_amdgcn_readfirstlane((int)__any(mnew != m))) {
            lsum *= alpha;
            zacc *= alpha;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float ar = __shfl(alpha, 4 * g + r);  // O rows are heads 4g + r
#pragma unroll
                for (int dt = 0; dt < 8; ++dt) {
                    O[dt][r] *= ar;
                }
            }
        }
        m = mnew;
        const float msc = mnew * sc;  // finite: every block holds >= 1 valid token

        half8_t pa[2];
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const half2_t vv = bit_cast<half2_t>(vp[tt][r]);
                const float   pf = fast_exp2(__builtin_fmaf(sv[tt][r], sc, -msc));
                lsum += pf;
                zacc            = __builtin_fmaf(pf, (float)vv[1], zacc);
                pa[tt >> 1][(tt & 1) * 4 + r] = (half_t)(pf * (float)vv[0]);
            }
        }

        // ---- O^T... O = P' V on raw codes: k-slot (g, e) -> token 32a + 4g + e (e<4) / 32a + 16 + 4g + e - 4 ----
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int j = i16 >> 1;
            const int T = 32 * a + (j < 4 ? 4 * g + j : 16 + 4 * g + (j - 4));
            const char* vrow = Vt + T * 128 + (i16 & 1) * 8;
            const int   swz  = (T >> 1) & 7;
#pragma unroll
            for (int dt = 0; dt < 8; ++dt) {
                const v2i vb = __builtin_amdgcn_ds_read_tr8_b64_v2i32(
                    (__attribute__((address_space(3))) v2i*)(vrow + ((dt ^ swz) << 4)));
                const half8_t bq = bytes8_to_f16_subnormal(u32x2{(uint32_t)vb[0], (uint32_t)vb[1]});
                O[dt]            = __builtin_amdgcn_mfma_f32_16x16x32_f16(pa[a], bq, O[dt], 0, 0, 0);
            }
        }
    }

    if (p.dbg && lane == 0) {  // slots 2, 6, 7: waves 0, 1, 3 done with their blocks (wave 0 owns the newest block and the prologue)
        if (wave != 2) {
            p.dbg[wgid * 8 + (wave == 0 ? 2 : wave == 1 ? 6 : 7)] = __builtin_amdgcn_s_memrealtime();
        }
    }
    // ---- per-wave totals, then merge the 4 waves through LDS ------------------------------------------------
    lsum += __shfl_xor(lsum, 16);
    lsum += __shfl_xor(lsum, 32);
    zacc += __shfl_xor(zacc, 16);
    zacc += __shfl_xor(zacc, 32);

    __syncthreads();  // every wave is done with its K/V image: the region is reused for the merge
    if (p.dbg && threadIdx.x == 0) {
        p.dbg[wgid * 8 + 5] = __builtin_amdgcn_s_memrealtime();  // all four waves done
    }
    float* sm_o  = (float*)smem;                     // [4][16][128]
    float* sm_ml = (float*)(smem + 4 * 16 * D * 4);  // [4][16][2]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int   h  = 4 * g + r;
        const float za = __shfl(zacc, h);
        if (h < hpw) {
#pragma unroll
            for (int dt = 0; dt < 8; ++dt) {
                sm_o[(wave * 16 + h) * D + dt * 16 + i16] = __builtin_fmaf(O[dt][r], kTwo24, za);
            }
        }
    }
    if (g == 0 && i16 < hpw) {
        sm_ml[(wave * 16 + i16) * 2]     = m;
        sm_ml[(wave * 16 + i16) * 2 + 1] = lsum;
    }
    __syncthreads();

    for (int idx = threadIdx.x; idx < hpw * D; idx += 256) {
        const int h = idx / D;
        const int d = idx - h * D;
        float     ms = -INFINITY;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            ms = fmaxf(ms, sm_ml[(w * 16 + h) * 2]);
        }
        float o = 0.f, l = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float mw = sm_ml[(w * 16 + h) * 2];
            const float wt = (mw == -INFINITY) ? 0.f : fast_exp2((mw - ms) * sc);
            o += wt * sm_o[(w * 16 + h) * D + d];
            l += wt * sm_ml[(w * 16 + h) * 2 + 1];
        }
        const int hq = head0 + h;
        if (p.splits == 1) {
            p.out[(size_t)b * p.q_heads * D + (size_t)hq * D + d] = (half_t)(o / l);
        }
        else {
            const size_t slot         = ((size_t)b * p.q_heads + hq) * p.splits + split;
            p.partial_o[slot * D + d] = o;
            if (d == 0) {
                p.partial_ml[slot * 2]     = ms;
                p.partial_ml[slot * 2 + 1] = l;
            }
        }
    }
    if (p.dbg && threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        p.dbg[wgid * 8 + 3] = __builtin_amdgcn_s_memrealtime();
    }
}


int launch_decode_attention_i8_mfma(const DecodeAttnParams& p_in, hipStream_t st)
{
    DecodeAttnParams p     = p_in;
    const int        group = p.q_heads / p.cache.layout.kv_heads;
    int       hpw   = group;
    while (hpw > 16) {  // largest divisor of the GQA group that fits the 16 MFMA columns
        int d = 2;
        while (hpw % d) {
            ++d;
        }
        hpw /= d;
    }
    const int chunks = group / hpw;
    // A variant with TWO kv heads per workgroup (one workgroup per CU, 9 / 8 / 8 / 9 blocks per wave instead of 5 / 4 / 4 / 4, one or two
    // blocks in flight per wave) was built and measured in round 5 and is in the history (commit eff45b9): parity-green, 38.2-41.4 us
    // against this kernel's 34.4 at ctx 1040 -- four waves per CU issue the per-block arithmetic more slowly than eight
    // (profiles/r05_attention_pair_kernel.txt).
    dim3      grid(p.cache.layout.kv_heads * chunks, p.batch, p.splits);
    p.dbg = gemm_trace_for((size_t)grid.x * grid.y * grid.z, "attn", grid.x, grid.y, grid.z);
    // 4 wave-private images (+ 4 KB q exchange for the fused prologue); the merge buffers overlay the images
    static_assert(4 * kWaveLds + 4096 > 4 * 16 * 128 * 4 + 4 * 16 * 2 * 4, "merge buffers must fit");
    const int lds = 4 * kWaveLds + 4096;
    const void* const k = (p.qkv_slabs || p.qkv_f16) ?
                              (p.cache.layout.bits == 4 ? (const void*)decode_attention_i8_mfma_kernel<true, 4> :
                                                          (const void*)decode_attention_i8_mfma_kernel<true, 8>) :
                              (p.cache.layout.bits == 4 ? (const void*)decode_attention_i8_mfma_kernel<false, 4> :
                                                          (const void*)decode_attention_i8_mfma_kernel<false, 8>);
    if (const int rc = ensure_dynamic_lds(k, lds)) {
        return rc;
    }
    if (p.qkv_slabs || p.qkv_f16) {
        TM_REQUIRE(p.qkv_n % 8 == 0 && (p.qkv_splits == 0) == (p.qkv_slabs == nullptr), "fused qkv input");
        if (p.cache.layout.bits == 4) {
            decode_attention_i8_mfma_kernel<true, 4><<<grid, 256, lds, st>>>(p, chunks, hpw);
        }
        else {
            decode_attention_i8_mfma_kernel<true, 8><<<grid, 256, lds, st>>>(p, chunks, hpw);
        }
    }
    else {
        if (p.cache.layout.bits == 4) {
            decode_attention_i8_mfma_kernel<false, 4><<<grid, 256, lds, st>>>(p, chunks, hpw);
        }
        else {
            decode_attention_i8_mfma_kernel<false, 8><<<grid, 256, lds, st>>>(p, chunks, hpw);
        }
    }
    TM_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace tmk
