// Small HBM-bound helpers: embedding lookup, greedy argmax, row gather, unfused SiLU*up, and the
// group-wise u4 quantiser used to fabricate synthetic AWQ weights on the device.
//
// Replaces: embeddingLookup (src/turbomind/kernels/decoding_kernels.cu via language_model.cc:232),
//           top-k(=1) sampling (generation/sampling.cc:92-183), CollectHiddenStates
//           (unified_decoder.cc:355-372), Activation (kernels/activation.cu:27-130),
//           QuantizeGroupwise / IntegralQuantizer (kernels/quantization.cu:384-440,515-676).
#include "tm_common.h"
#include "tm_kernels.h"
#include <map>
#include <mutex>

namespace tmk {

__global__ __launch_bounds__(256) void embedding_kernel(half_t* __restrict__ out,
                                                        const half_t* __restrict__ table,
                                                        const int* __restrict__ ids,
                                                        int H,
                                                        int vocab)
{
    const int t  = blockIdx.x;
    int       id = ids[t];
    id           = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    for (int v = threadIdx.x; v < H / 8; v += 256) {
        *(half8_t*)(out + (size_t)t * H + v * 8) = *(const half8_t*)(table + (size_t)id * H + v * 8);
    }
}

int launch_embedding(half_t* out, const half_t* table, const int* ids, int T, int H, int vocab, hipStream_t st)
{
    TM_REQUIRE(H % 8 == 0, "H % 8");
    if (T == 0) {
        return 0;
    }
    embedding_kernel<<<T, 256, 0, st>>>(out, table, ids, H, vocab);
    TM_HIP_CHECK(hipGetLastError());
    return 0;
}

__global__ __launch_bounds__(256) void gather_rows_kernel(half_t* __restrict__ out,
                                                          const half_t* __restrict__ in,
                                                          const int* __restrict__ rows,
                                                          int H)
{
    const int r = rows[blockIdx.x];
    for (int v = threadIdx.x; v < H / 8; v += 256) {
        *(half8_t*)(out + (size_t)blockIdx.x * H + v * 8) = *(const half8_t*)(in + (size_t)r * H + v * 8);
    }
}

int launch_gather_rows(half_t* out, const half_t* in, const int* rows, int n, int H, hipStream_t st)
{
    if (n == 0) {
        return 0;
    }
    gather_rows_kernel<<<n, 256, 0, st>>>(out, in, rows, H);
    TM_HIP_CHECK(hipGetLastError());
    return 0;
}

// greedy: argmax over fp32-cast logits, lowest index wins ties. One 1024-thread workgroup per row.
__global__ __launch_bounds__(1024) void argmax_kernel(int* __restrict__ out_ids,
                                                      half_t* __restrict__ out_val,
                                                      const half_t* __restrict__ logits,
                                                      int V,
                                                      int ld,
                                                      int id_offset)
{
    __shared__ float sv[16];
    __shared__ int   si[16];
    const int     row  = blockIdx.x;
    const half_t* lp   = logits + (size_t)row * ld;
    float         best = -INFINITY;
    int           bi   = 0x7fffffff;
    const int     nvec = V / 8;
    for (int v = threadIdx.x; v < nvec; v += 1024) {
        const half8_t x = *(const half8_t*)(lp + (size_t)v * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float f = (float)x[e];
            if (f > best) {  // ascending scan: strict > keeps the lowest index
                best = f;
                bi   = v * 8 + e;
            }
        }
    }
    for (int i = nvec * 8 + threadIdx.x; i < V; i += 1024) {
        const float f = (float)lp[i];
        if (f > best || (f == best && i < bi)) {
            best = f;
            bi   = i;
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float ov = __shfl_xor(best, off);
        const int   oi = __shfl_xor(bi, off);
        if (ov > best || (ov == best && oi < bi)) {
            best = ov;
            bi   = oi;
        }
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        sv[wave] = best;
        si[wave] = bi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w) {
            if (sv[w] > best || (sv[w] == best && si[w] < bi)) {
                best = sv[w];
                bi   = si[w];
            }
        }
        out_ids[row] = (bi == 0x7fffffff ? 0 : bi) + id_offset;
        if (out_val) {
            out_val[row] = (half_t)best;
        }
    }
}

int launch_argmax(int* out_ids, half_t* out_val, const half_t* logits, int B, int V, int ld, int id_offset, hipStream_t st)
{
    TM_REQUIRE(ld % 8 == 0, "logits rows must be 16-byte aligned");
    if (B == 0) {
        return 0;
    }
    argmax_kernel<<<B, 1024, 0, st>>>(out_ids, out_val, logits, V, ld, id_offset);
    TM_HIP_CHECK(hipGetLastError());
    return 0;
}

// unfused activation on a [M][2*inter] buffer laid out [gate | up]
// Pseudo-random fp16 values in (-amp, amp) -- stand-in activations for the start-up GEMM tuner.  Not zeros: the matrix pipe
// of MI355X is power-limited, and a GEMM over all-zero activations runs at the full 2.4 GHz instead of the ~1.6 GHz it sustains
// on real data (measured: the fused w1w3 prefill tile 1152 us on zeros, 1697 us on N(0,1) rows) -- timings on zeros rank
// compute-bound candidates wrongly.
__global__ __launch_bounds__(256) void fill_uniform_f16_kernel(half_t* __restrict__ out, size_t n, float amp, uint32_t seed)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) {
        return;
    }
    uint32_t h = (uint32_t)i * 0x9E3779B1u + (uint32_t)(i >> 32) + seed;  // one round of a 32-bit integer hash
    h ^= h >> 16;
    h *= 0x7feb352du;
    h ^= h >> 15;
    h *= 0x846ca68bu;
    h ^= h >> 16;
    out[i] = (half_t)(((float)(h >> 8) * (1.0f / 8388608.0f) - 1.0f) * amp);
}

int launch_fill_uniform_f16(half_t* out, size_t n, float amp, uint32_t seed, hipStream_t st)
{
    if (n == 0) {
        return 0;
    }
    fill_uniform_f16_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(out, n, amp, seed);
    TM_HIP_CHECK(hipGetLastError());
    return 0;
}

__global__ __launch_bounds__(256) void silu_mul_kernel(half_t* __restrict__ out,
                                                       const half_t* __restrict__ gate_up,
                                                       int M,
                                                       int inter)
{
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)M * inter / 8) {
        return;
    }
    const int     m = idx / (inter / 8);
    const int     c = (idx - (size_t)m * (inter / 8)) * 8;
    const half8_t g = *(const half8_t*)(gate_up + (size_t)m * 2 * inter + c);
    const half8_t u = *(const half8_t*)(gate_up + (size_t)m * 2 * inter + inter + c);
    half8_t       o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float gf = (float)g[e];
        o[e]           = (half_t)((gf / (1.0f + __builtin_expf(-gf))) * (float)u[e]);
    }
    *(half8_t*)(out + (size_t)m * inter + c) = o;
}

int launch_silu_mul(half_t* out, const half_t* gate_up, int M, int inter, hipStream_t st)
{
    TM_REQUIRE(inter % 8 == 0, "inter % 8");
    const size_t total = (size_t)M * inter / 8;
    if (total == 0) {
        return 0;
    }
    silu_mul_kernel<<<(total + 255) / 256, 256, 0, st>>>(out, gate_up, M, inter);
    TM_HIP_CHECK(hipGetLastError());
    return 0;
}

// IntegralQuantizer<half,4>: groups of `group` along K for every output column n.
//   scale_ = max(max-min, 1e-5)/15 ; zero_ = clamp(-rint(min/scale_), 0, 15)
//   q = clamp(rint(x/scale_) + zero_, 0, 15) ; d = h(q - zero_) * h(scale_)
// Output in the boundary layout: qweight int32 [K][N/8] (nibble j of word c = column 8c+j).
__global__ __launch_bounds__(256) void quantize_groupwise_u4_kernel(int32_t* __restrict__ qweight,
                                                                    half_t* __restrict__ scales,
                                                                    half_t* __restrict__ zeros,
                                                                    half_t* __restrict__ dequant,
                                                                    const half_t* __restrict__ w,
                                                                    int K,
                                                                    int N,
                                                                    int group)
{
    // one thread per (group, 8 consecutive columns)
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int    nc  = N / 8;
    if (idx >= (size_t)(K / group) * nc) {
        return;
    }
    const int gi = idx / nc;
    const int c  = idx - (size_t)gi * nc;
    float     mn[8], mx[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        mn[e] = INFINITY;
        mx[e] = -INFINITY;
    }
    for (int k = gi * group; k < (gi + 1) * group; ++k) {
        const half8_t x = *(const half8_t*)(w + (size_t)k * N + c * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            mn[e] = fminf(mn[e], (float)x[e]);
            mx[e] = fmaxf(mx[e], (float)x[e]);
        }
    }
    float sc[8];
    int   zp[8];
    half8_t s_out, z_out;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        sc[e]    = fmaxf(mx[e] - mn[e], 1e-5f) / 15.0f;
        int z    = (int)(-__builtin_rintf(mn[e] / sc[e]));
        zp[e]    = z < 0 ? 0 : (z > 15 ? 15 : z);
        s_out[e] = (half_t)sc[e];
        z_out[e] = (half_t)(float)zp[e];
    }
    *(half8_t*)(scales + (size_t)gi * N + c * 8) = s_out;
    *(half8_t*)(zeros + (size_t)gi * N + c * 8)  = z_out;
    for (int k = gi * group; k < (gi + 1) * group; ++k) {
        const half8_t x = *(const half8_t*)(w + (size_t)k * N + c * 8);
        uint32_t      word = 0;
        half8_t       d;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            int q = (int)__builtin_rintf((float)x[e] / sc[e]) + zp[e];
            q     = q < 0 ? 0 : (q > 15 ? 15 : q);
            word |= (uint32_t)q << (4 * e);
            d[e] = (half_t)(float)(q - zp[e]) * s_out[e];
        }
        qweight[(size_t)k * nc + c] = (int32_t)word;
        if (dequant) {
            *(half8_t*)(dequant + (size_t)k * N + c * 8) = d;
        }
    }
}

int launch_quantize_groupwise_u4(int32_t* qweight, half_t* scales, half_t* zeros, half_t* dequant, const half_t* w,
                                 int K, int N, int group, hipStream_t st)
{
    TM_REQUIRE(N % 8 == 0 && K % group == 0, "N % 8 == 0, K % group == 0");
    const size_t total = (size_t)(K / group) * (N / 8);
    if (total == 0) {
        return 0;
    }
    quantize_groupwise_u4_kernel<<<(total + 255) / 256, 256, 0, st>>>(qweight, scales, zeros, dequant, w, K, N, group);
    TM_HIP_CHECK(hipGetLastError());
    return 0;
}

int ensure_dynamic_lds(const void* kernel, int bytes)
{
    static std::mutex                               mu;
    static std::map<std::pair<const void*, int>, int> done;  // (kernel, device) -> bytes granted
    int dev = 0;
    TM_HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    auto                        it = done.find({kernel, dev});
    if (it != done.end() && it->second >= bytes) {
        return 0;
    }
    TM_HIP_CHECK(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    done[{kernel, dev}] = bytes;
    return 0;
}

}  // namespace tmk
