// Mixture-of-experts routing for gfx950: router (gate) + top-k + routing tables + weighted combine.
//
// Replaces: MoeFfnLayerImpl::Gate, invokeMoeGate_V2 (softmax / top-k / norm_topk / routed_scale and the f2n / en2f /
// offsets tables) and invokeMoeCombine (src/turbomind/models/llama/moe_ffn_layer.cc:43-53,133-325;
// kernels/gemm/moe_utils_v2.cu:355-690).  The expert FFNs run as grouped GEMMs (gemm_w4a16.hip, `groups` descriptors)
// over the token rows listed per expert.
//   gate:   logits[t][e] = sum_h x[t][h] * Wg[h][e] in fp32; top-k on the logits (ties: lower expert id);
//           norm_topk: w_j = exp(l_j - max) / sum over the SELECTED experts, else softmax over all experts;
//           w_j *= routed_scale.
//   route:  for every expert the tokens that selected it, in ascending token order:
//           offsets[e] .. offsets[e+1] index the flat pair list; f2n[f] = token; en2f[j][t] = f.
//   combine: out[t] = fp16( sum_j w[t][j] * float(y[en2f[j][t]]) ).
#include "tm_common.h"
#include "tm_kernels.h"

#define TM_TRY_RC(expr)       \
    do {                     \
        const int _rc = (expr); \
        if (_rc) {           \
            return _rc;      \
        }                    \
    } while (0)

namespace tmk {

constexpr int kMaxExperts = 64;
constexpr int kMaxTopK    = 8;

// one 256-thread workgroup per token; Wg fp16 [H][E] (input-major like every other weight)
__global__ __launch_bounds__(256) void moe_gate_kernel(int* __restrict__ topk_ids,      // [T][k]
                                                       float* __restrict__ topk_w,      // [T][k]
                                                       float* __restrict__ logits_out,  // [T][E] or nullptr
                                                       const half_t* __restrict__ x,
                                                       int ldx,
                                                       const half_t* __restrict__ wg,
                                                       int H,
                                                       int E,
                                                       int k,
                                                       int norm_topk,
                                                       float routed_scale)
{
    __shared__ float part[4][kMaxExperts];
    __shared__ float logit[kMaxExperts];
    const int        t    = blockIdx.x;
    const int        lane = threadIdx.x & 63;
    const int        wave = threadIdx.x >> 6;
    const half_t*    xr   = x + (size_t)t * ldx;
    for (int e = 0; e < E; ++e) {
        float acc = 0.f;
        for (int h = threadIdx.x; h < H; h += 256) {
            acc = __builtin_fmaf((float)xr[h], (float)wg[(size_t)h * E + e], acc);
        }
        acc = group_sum<64>(acc);
        if (lane == 0) {
            part[wave][e] = acc;
        }
    }
    __syncthreads();
    if (threadIdx.x < E) {
        logit[threadIdx.x] = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
        if (logits_out) {
            logits_out[(size_t)t * E + threadIdx.x] = logit[threadIdx.x];
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int      ids[kMaxTopK];
        float    val[kMaxTopK];
        uint64_t taken = 0;
        float    mx    = -INFINITY;
        for (int e = 0; e < E; ++e) {
            mx = fmaxf(mx, logit[e]);
        }
        for (int j = 0; j < k; ++j) {
            int   best = -1;
            float bv   = -INFINITY;
            for (int e = 0; e < E; ++e) {
                if (!((taken >> e) & 1) && (best < 0 || logit[e] > bv)) {
                    best = e;
                    bv   = logit[e];
                }
            }
            taken |= 1ull << best;
            ids[j] = best;
            val[j] = bv;
        }
        float denom = 0.f;
        if (norm_topk) {
            for (int j = 0; j < k; ++j) {
                denom += __builtin_expf(val[j] - mx);
            }
        }
        else {
            for (int e = 0; e < E; ++e) {
                denom += __builtin_expf(logit[e] - mx);
            }
        }
        const float inv = 1.0f / denom;
        for (int j = 0; j < k; ++j) {
            topk_ids[(size_t)t * k + j] = ids[j];
            topk_w[(size_t)t * k + j]   = __builtin_expf(val[j] - mx) * inv * routed_scale;
        }
    }
}

// single workgroup: routing tables.  offsets [E+1], f2n [T*k], en2f [k][T]
__global__ __launch_bounds__(1024) void moe_route_kernel(int* __restrict__ offsets, int* __restrict__ f2n, int* __restrict__ en2f,
                                                         const int* __restrict__ topk_ids, int T, int E, int k)
{
    __shared__ int s_scan[16];
    __shared__ int s_base;
    const int      tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) {
        s_base     = 0;
        offsets[0] = 0;
    }
    __syncthreads();
    for (int e = 0; e < E; ++e) {
        int running = s_base;
        for (int t0 = 0; t0 < T; t0 += 1024) {
            const int t = t0 + tid;
            int       j = -1;
            if (t < T) {
                for (int q = 0; q < k; ++q) {
                    if (topk_ids[(size_t)t * k + q] == e) {
                        j = q;
                    }
                }
            }
            const int flag = j >= 0;
            int       x    = flag;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int y = __shfl_up(x, d);
                if (lane >= d) {
                    x += y;
                }
            }
            if (lane == 63) {
                s_scan[wave] = x;
            }
            __syncthreads();
            int before = 0, total = 0;
            for (int w = 0; w < 16; ++w) {
                if (w < wave) {
                    before += s_scan[w];
                }
                total += s_scan[w];
            }
            if (flag) {
                const int f             = running + before + x - 1;
                f2n[f]                  = t;
                en2f[(size_t)j * T + t] = f;
            }
            running += total;
            __syncthreads();
        }
        if (tid == 0) {
            s_base         = running;
            offsets[e + 1] = running;
        }
        __syncthreads();
    }
}

// out[t][h] = fp16( sum_j w[t][j] * y[en2f[j][t]][h] ); 8 columns per thread
__global__ __launch_bounds__(256) void moe_combine_kernel(half_t* __restrict__ out, int ldo, const half_t* __restrict__ y, int ldy,
                                                          const float* __restrict__ topk_w, const int* __restrict__ en2f, int T,
                                                          int H, int k)
{
    const int t = blockIdx.y;
    const int h = (blockIdx.x * 256 + threadIdx.x) * 8;
    if (h >= H) {
        return;
    }
    float acc[8] = {};
    for (int j = 0; j < k; ++j) {
        const int     f = en2f[(size_t)j * T + t];
        const float   w = topk_w[(size_t)t * k + j];
        const half8_t v = *(const half8_t*)(y + (size_t)f * ldy + h);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            acc[e] = __builtin_fmaf(w, (float)v[e], acc[e]);
        }
    }
    half8_t o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        o[e] = (half_t)acc[e];
    }
    *(half8_t*)(out + (size_t)t * ldo + h) = o;
}

int launch_moe_gate(int* topk_ids, float* topk_w, float* logits_out, const half_t* x, int ldx, const half_t* wg, int T, int H,
                    int E, int k, bool norm_topk, float routed_scale, hipStream_t st)
{
    TM_REQUIRE(E >= 1 && E <= kMaxExperts && k >= 1 && k <= kMaxTopK && k <= E, "moe: 1 <= top_k <= experts <= 64, top_k <= 8");
    if (T == 0) {
        return 0;
    }
    moe_gate_kernel<<<T, 256, 0, st>>>(topk_ids, topk_w, logits_out, x, ldx, wg, H, E, k, norm_topk ? 1 : 0, routed_scale);
    TM_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_moe_route(int* offsets, int* f2n, int* en2f, const int* topk_ids, int T, int E, int k, hipStream_t st)
{
    moe_route_kernel<<<1, 1024, 0, st>>>(offsets, f2n, en2f, topk_ids, T, E, k);
    TM_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_moe_combine(half_t* out, int ldo, const half_t* y, int ldy, const float* topk_w, const int* en2f, int T, int H, int k,
                       hipStream_t st)
{
    TM_REQUIRE(H % 8 == 0, "moe combine: H % 8 == 0");
    if (T == 0) {
        return 0;
    }
    moe_combine_kernel<<<dim3((H / 8 + 255) / 256, T), 256, 0, st>>>(out, ldo, y, ldy, topk_w, en2f, T, H, k);
    TM_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---- one MoE FFN block (host object shared by the C-ABI operator and the engine) -------------------------------------
size_t moe_workspace_bytes(const MoeBlock& m, int tokens)
{
    const size_t pairs = (size_t)tokens * m.top_k;
    size_t       b     = 0;
    b += pairs * 4;                  // topk_ids
    b += pairs * 4;                  // topk_w
    b += (size_t)(m.experts + 1) * 4;  // offsets
    b += pairs * 4;                  // f2n
    b += pairs * 4;                  // en2f
    b  = (b + 255) / 256 * 256;
    b += pairs * m.inter * 2;        // act  [pairs][I]
    b  = (b + 255) / 256 * 256;
    b += pairs * m.hidden * 2;       // y2   [pairs][H]
    b  = (b + 255) / 256 * 256;
    // fp8 x fp8 experts: codes + scales of x (per token) and of the gated-SiLU output (per (token, expert) row)
    b += fp8_act_workspace_bytes(tokens, m.hidden) + fp8_act_workspace_bytes((int)pairs, m.inter);
    return b + 256;
}

int moe_prepare(MoeBlock& m, hipStream_t st)
{
    TM_REQUIRE((int)m.w13.size() == m.experts && (int)m.w2.size() == m.experts && m.gate, "moe: gate and every expert must be set");
    TM_TRY_RC(moe_build_groups(&m.groups13, m.w13.data(), m.experts, st));
    TM_TRY_RC(moe_build_groups(&m.groups2, m.w2.data(), m.experts, st));
    if (fp8_mfma_supported(m.w13[0]) && fp8_mfma_supported(m.w2[0])) {  // device tables of the experts' P8 unit pointers
        std::vector<const void*> h13(m.experts), h2(m.experts);
        for (int e = 0; e < m.experts; ++e) {
            TM_REQUIRE(m.w13[e].packed8 && m.w2[e].packed8, "moe: fp8 expert without its P8 image");
            h13[e] = m.w13[e].packed8;
            h2[e]  = m.w2[e].packed8;
        }
        if (!m.groups13_p8) {
            TM_HIP_CHECK(hipMalloc(&m.groups13_p8, sizeof(void*) * m.experts));
            TM_HIP_CHECK(hipMalloc(&m.groups2_p8, sizeof(void*) * m.experts));
        }
        TM_HIP_CHECK(hipMemcpyAsync(m.groups13_p8, h13.data(), sizeof(void*) * m.experts, hipMemcpyHostToDevice, st));
        TM_HIP_CHECK(hipMemcpyAsync(m.groups2_p8, h2.data(), sizeof(void*) * m.experts, hipMemcpyHostToDevice, st));
        TM_HIP_CHECK(hipStreamSynchronize(st));
    }
    return 0;
}

// out[t] = sum_j w_j * W2_e( silu(W1_e x_t) * (W3_e x_t) ) over the top_k experts e of token t
int moe_forward(const MoeBlock& m, half_t* out, int ldo, const half_t* x, int ldx, int tokens, void* workspace, int* topk_ids_out,
                float* topk_w_out, hipStream_t st)
{
    TM_REQUIRE(m.groups13 && m.groups2, "moe: not prepared");
    if (tokens == 0) {
        return 0;
    }
    const size_t pairs = (size_t)tokens * m.top_k;
    char*        w     = (char*)workspace;
    int*         ids   = (int*)w;
    float*       tw    = (float*)(w + pairs * 4);
    int*         offs  = (int*)(w + pairs * 8);
    int*         f2n   = offs + (m.experts + 1);
    int*         en2f  = f2n + pairs;
    size_t       o     = ((char*)(en2f + pairs) - w + 255) / 256 * 256;
    half_t*      act   = (half_t*)(w + o);
    o                  = (o + pairs * m.inter * 2 + 255) / 256 * 256;
    half_t*      y2    = (half_t*)(w + o);
    TM_TRY_RC(launch_moe_gate(ids, tw, nullptr, x, ldx, m.gate, tokens, m.hidden, m.experts, m.top_k, m.norm_topk, m.routed_scale, st));
    TM_TRY_RC(launch_moe_route(offs, f2n, en2f, ids, tokens, m.experts, m.top_k, st));
    // expert FFNs: gathered rows of x -> act (gated SiLU fused) -> y2, both grouped over the experts
    const int hint = (int)((pairs + m.experts - 1) / m.experts);  // expected rows per expert
    if (m.groups13_p8) {
        // e4m3 experts on the fp8 matrix cores (the reference's fp8 path: QuantizeSymm + fp8 GEMM, LlamaLinear.cu:67-127):
        // x is quantised once per token, the gated-SiLU output once per (token, expert) row
        o                  = (o + pairs * m.hidden * 2 + 255) / 256 * 256;
        uint8_t*     xq    = (uint8_t*)(w + o);
        const int    ldsx1 = (tokens + 3) / 4 * 4;
        float*       sx1   = (float*)(xq + (size_t)tokens * m.hidden);
        o                  = (o + fp8_act_workspace_bytes(tokens, m.hidden) + 255) / 256 * 256;
        uint8_t*     aq    = (uint8_t*)(w + o);
        const int    ldsx2 = ((int)pairs + 3) / 4 * 4;
        float*       sx2   = (float*)(aq + pairs * m.inter);
        TM_REQUIRE(ldx == m.hidden || tokens == 1, "moe fp8: x must be row-contiguous");
        TM_TRY_RC(launch_quant_fp8_rows(xq, sx1, x, ldx, tokens, m.hidden, ldsx1, st));
        TM_TRY_RC(launch_linear_fp8_grouped(m.w13[0], m.groups13_p8, m.experts, xq, sx1, ldsx1, tokens, act, m.inter, tokens, hint, true,
                                            offs, f2n, st));
        TM_TRY_RC(launch_quant_fp8_rows(aq, sx2, act, m.inter, (int)pairs, m.inter, ldsx2, st));
        TM_TRY_RC(launch_linear_fp8_grouped(m.w2[0], m.groups2_p8, m.experts, aq, sx2, ldsx2, (int)pairs, y2, m.hidden, tokens, hint, false,
                                            offs, nullptr, st));
    }
    else {
        TM_TRY_RC(launch_linear_grouped(m.w13[0], m.groups13, m.experts, x, ldx, tokens, act, m.inter, tokens, hint, true, offs, f2n, st));
        TM_TRY_RC(launch_linear_grouped(m.w2[0], m.groups2, m.experts, act, m.inter, (int)pairs, y2, m.hidden, tokens, hint, false, offs,
                                        nullptr, st));
    }
    TM_TRY_RC(launch_moe_combine(out, ldo, y2, m.hidden, tw, en2f, tokens, m.hidden, m.top_k, st));
    if (topk_ids_out) {
        TM_HIP_CHECK(hipMemcpyAsync(topk_ids_out, ids, pairs * 4, hipMemcpyDeviceToDevice, st));
    }
    if (topk_w_out) {
        TM_HIP_CHECK(hipMemcpyAsync(topk_w_out, tw, pairs * 4, hipMemcpyDeviceToDevice, st));
    }
    return 0;
}

void moe_free(MoeBlock& m)
{
    for (auto& l : m.w13) {
        linear_weight_free(l);
    }
    for (auto& l : m.w2) {
        linear_weight_free(l);
    }
    for (void* q : {(void*)m.gate, m.groups13, m.groups2, m.groups13_p8, m.groups2_p8}) {
        if (q) {
            (void)hipFree(q);
        }
    }
    m.gate = nullptr;
    m.groups13 = m.groups2 = nullptr;
    m.groups13_p8 = m.groups2_p8 = nullptr;
}

}  // namespace tmk
