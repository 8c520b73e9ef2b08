// Prefill-sized forwards of a dense W4A16 linear through the vendor library's fp16 GEMM.
//
// Replaces (for M >= 512 only): the large-batch end of gemm::Gemm::Run (src/turbomind/kernels/gemm/gemm.cu:257-344) --
// the reference keeps its fused dequant tiles there; on MI355X the matrix pipe is power-limited (~1.6 GHz under MFMA
// load) and every VALU dequant op of the fused kernel competes with it: measured on the Llama-3-8B shapes at M = 8192
// (profiles/r03_prefill_gemm_vs_library.txt) the fused 128 x 512 tile of gemm_decode.hip reaches 1.10 .. 1.22 PFLOP/s,
// hipBLASLt on fp16 weights in [N][K] order 1.42 .. 1.49.  A prefill chunk is compute bound, so spending HBM bandwidth
// to save VALU work is the right trade there (the opposite of decode): the weights of ONE linear are dequantised to an
// fp16 [N][K] image -- bit for bit the A fragments the fused kernel builds in registers (dequant8_p32) -- either per
// call into a scratch buffer (K*N*2.5 bytes of traffic, ~3 % of the GEMM at M = 8192) or once at load
// (LinearWeight::f16_nk: 288 GB of HBM hold the u4 image for decode AND the fp16 image for prefill of an 8B..30B model),
// and the product is a plain library GEMM ("TN": D^T[N][M] = W[N][K] x^T).  The gated-SiLU epilogue of w1w3 runs as a
// separate pass over row chunks small enough for the intermediate to stay in the Infinity Cache.
//
// The library is bound at run time (dlopen): the process may already hold a copy (PyTorch ships one), and a host
// without it keeps the fused kernels -- f16_library_available() says which.  Candidate "shape 10" of the measured
// dispatch (dec32_candidates / tune_decode_gemms): the tuner times it against the fused tiles per (K, N, size class).
#include "tm_common.h"
#include "tm_kernels.h"
#include "p32_layout.h"
#include <dlfcn.h>
#include <hipblaslt/hipblaslt.h>
#include <map>
#include <mutex>
#include <stdlib.h>
#include <tuple>

#define TM_TRY_RC(expr)          \
    do {                         \
        const int rc_ = (expr);  \
        if (rc_) {               \
            return rc_;          \
        }                        \
    } while (0)

namespace tmk {

namespace {

struct LtApi {
    void* lib = nullptr;
    decltype(&hipblasLtCreate)                       create           = nullptr;
    decltype(&hipblasLtMatmulDescCreate)             desc_create      = nullptr;
    decltype(&hipblasLtMatmulDescSetAttribute)       desc_set         = nullptr;
    decltype(&hipblasLtMatmulDescDestroy)            desc_destroy     = nullptr;
    decltype(&hipblasLtMatrixLayoutCreate)           layout_create    = nullptr;
    decltype(&hipblasLtMatrixLayoutDestroy)          layout_destroy   = nullptr;
    decltype(&hipblasLtMatmulPreferenceCreate)       pref_create      = nullptr;
    decltype(&hipblasLtMatmulPreferenceSetAttribute) pref_set         = nullptr;
    decltype(&hipblasLtMatmulPreferenceDestroy)      pref_destroy     = nullptr;
    decltype(&hipblasLtMatmulAlgoGetHeuristic)       heuristic        = nullptr;
    decltype(&hipblasLtMatmul)                       matmul           = nullptr;
    bool ok = false;
};

const LtApi& lt_api()
{
    static LtApi api = [] {
        LtApi       a;
        const char* env = getenv("TM_HIPBLASLT_PATH");
        // the SONAME first: an already loaded copy (PyTorch's) is reused instead of mapping a second one
        const char* names[] = {env, "libhipblaslt.so.1", "libhipblaslt.so", "/opt/rocm/lib/libhipblaslt.so.1", "/opt/rocm/lib/libhipblaslt.so"};
        for (const char* n : names) {
            if (n && *n && (a.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL))) {
                break;
            }
        }
        if (!a.lib) {
            return a;
        }
#define TM_LT_SYM(field, name) a.field = (decltype(a.field))dlsym(a.lib, name)
        TM_LT_SYM(create, "hipblasLtCreate");
        TM_LT_SYM(desc_create, "hipblasLtMatmulDescCreate");
        TM_LT_SYM(desc_set, "hipblasLtMatmulDescSetAttribute");
        TM_LT_SYM(desc_destroy, "hipblasLtMatmulDescDestroy");
        TM_LT_SYM(layout_create, "hipblasLtMatrixLayoutCreate");
        TM_LT_SYM(layout_destroy, "hipblasLtMatrixLayoutDestroy");
        TM_LT_SYM(pref_create, "hipblasLtMatmulPreferenceCreate");
        TM_LT_SYM(pref_set, "hipblasLtMatmulPreferenceSetAttribute");
        TM_LT_SYM(pref_destroy, "hipblasLtMatmulPreferenceDestroy");
        TM_LT_SYM(heuristic, "hipblasLtMatmulAlgoGetHeuristic");
        TM_LT_SYM(matmul, "hipblasLtMatmul");
#undef TM_LT_SYM
        a.ok = a.create && a.desc_create && a.desc_set && a.desc_destroy && a.layout_create && a.layout_destroy && a.pref_create
               && a.pref_set && a.pref_destroy && a.heuristic && a.matmul;
        return a;
    }();
    return api;
}

constexpr size_t kLtWorkspace = 64u << 20;  // the library's own scratch (split-K / stream-K variants of its kernels)

struct LtPlan {
    hipblasLtMatmulDesc_t   desc = nullptr;
    hipblasLtMatrixLayout_t a = nullptr, b = nullptr, d = nullptr;
    hipblasLtMatmulAlgo_t   algo{};
    size_t                  ws = 0;
};

std::mutex                                                         g_lt_mutex;
std::map<int, hipblasLtHandle_t>                                   g_lt_handles;  // per device
std::map<std::tuple<int, int, int, int, int, int>, LtPlan>         g_lt_plans;    // (device, N, M, K, ldx, ldy)

#define TM_LT_CHECK(expr)                                                                          \
    do {                                                                                           \
        const hipblasStatus_t s_ = (expr);                                                         \
        if (s_ != HIPBLAS_STATUS_SUCCESS) {                                                        \
            ::tmk::set_last_error(std::string("hipBLASLt: " #expr " -> status ") + std::to_string((int)s_)); \
            return 2; /* TM_FAIL */                                                                \
        }                                                                                          \
    } while (0)

// D^T [N][M] (ld = ldy) = W [N][K] * x^T: column-major A = the [N][K] image read as K x N (transposed), B = x read as K x M
int lt_plan(int N, int M, int K, int ldx, int ldy, hipblasLtHandle_t* handle, LtPlan* out)
{
    const LtApi& api = lt_api();
    int          dev = 0;
    TM_HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_lt_mutex);
    auto                        hit = g_lt_handles.find(dev);
    if (hit == g_lt_handles.end()) {
        hipblasLtHandle_t h = nullptr;
        TM_LT_CHECK(api.create(&h));
        hit = g_lt_handles.emplace(dev, h).first;
    }
    *handle        = hit->second;
    const auto key = std::make_tuple(dev, N, M, K, ldx, ldy);
    auto       pit = g_lt_plans.find(key);
    if (pit != g_lt_plans.end()) {
        *out = pit->second;
        return 0;
    }
    LtPlan p;
    // (a failure below returns with the descriptors created so far destroyed)
    auto build = [&]() -> int {
        TM_LT_CHECK(api.desc_create(&p.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F));
        const int32_t opt = HIPBLAS_OP_T, opn = HIPBLAS_OP_N;
        TM_LT_CHECK(api.desc_set(p.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &opt, sizeof(opt)));
        TM_LT_CHECK(api.desc_set(p.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &opn, sizeof(opn)));
        TM_LT_CHECK(api.layout_create(&p.a, HIP_R_16F, (uint64_t)K, (uint64_t)N, (int64_t)K));
        TM_LT_CHECK(api.layout_create(&p.b, HIP_R_16F, (uint64_t)K, (uint64_t)M, (int64_t)ldx));
        TM_LT_CHECK(api.layout_create(&p.d, HIP_R_16F, (uint64_t)N, (uint64_t)M, (int64_t)ldy));
        return 0;
    };
    auto drop = [&]() {
        if (p.desc) {
            (void)api.desc_destroy(p.desc);
        }
        for (hipblasLtMatrixLayout_t l : {p.a, p.b, p.d}) {
            if (l) {
                (void)api.layout_destroy(l);
            }
        }
    };
    if (const int rc = build()) {
        drop();
        return rc;
    }
    hipblasLtMatmulPreference_t pref = nullptr;
    if (api.pref_create(&pref) != HIPBLAS_STATUS_SUCCESS) {
        drop();
        ::tmk::set_last_error("hipBLASLt: hipblasLtMatmulPreferenceCreate failed");
        return 2;
    }
    const uint64_t max_ws = kLtWorkspace;
    (void)api.pref_set(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &max_ws, sizeof(max_ws));
    hipblasLtMatmulHeuristicResult_t res[8];
    int                              found = 0;
    const hipblasStatus_t            hs    = api.heuristic(*handle, p.desc, p.a, p.b, p.d, p.d, pref, 8, res, &found);
    (void)api.pref_destroy(pref);
    int pick = -1;
    for (int i = 0; hs == HIPBLAS_STATUS_SUCCESS && i < found && pick < 0; ++i) {
        if (res[i].state == HIPBLAS_STATUS_SUCCESS && res[i].workspaceSize <= kLtWorkspace) {
            pick = i;
        }
    }
    if (pick < 0) {
        drop();
        ::tmk::set_last_error("hipBLASLt: no fp16 GEMM algorithm for this problem (status " + std::to_string((int)hs) + ")");
        return 2;
    }
    p.algo = res[pick].algo;
    p.ws   = res[pick].workspaceSize;
    g_lt_plans.emplace(key, p);
    *out = p;
    return 0;
}

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

// y[m][i] = silu(c[m][2i]) * c[m][2i + 1] (w1 / w3 columns interleaved, as the fused epilogue sees them: epilogue.h:159-176);
// 32 bytes in, 16 bytes out per thread
__global__ __launch_bounds__(256) void silu_mul_interleaved_kernel(half_t* __restrict__ y, int ldy, const half_t* __restrict__ c, int ldc,
                                                                   int rows, int inter)
{
    const int    per_row = inter / 8;
    const size_t idx     = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)rows * per_row) {
        return;
    }
    const int      m  = (int)(idx / per_row);
    const int      i  = (int)(idx % per_row) * 8;
    const half8_t* src = (const half8_t*)(c + (size_t)m * ldc + 2 * i);
    const half8_t  v0 = __builtin_nontemporal_load(src), v1 = __builtin_nontemporal_load(src + 1);
    half8_t        o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float g0 = (float)v0[2 * e], u0 = (float)v0[2 * e + 1];
        const float g1 = (float)v1[2 * e], u1 = (float)v1[2 * e + 1];
        o[e]           = (half_t)(g0 / (1.0f + __builtin_expf(-g0)) * u0);
        o[4 + e]       = (half_t)(g1 / (1.0f + __builtin_expf(-g1)) * u1);
    }
    *(half8_t*)(y + (size_t)m * ldy + i) = o;
}

int gated_chunk_rows(int N, int M)
{
    // The intermediate [rows][N] fp16 is written by the GEMM and read once by the SiLU pass.  Measured at w1w3 of Llama-3-8B,
    // M = 8192 (profiles/r03_prefill_gemm_vs_library.txt): 1536-row chunks (96 MB, Infinity-Cache resident) cost more in GEMM
    // efficiency (six small launches) than they save in the pass -- so: as few chunks as a 256 MB intermediate allows
    // (two 4096-row launches there: the library is as fast at 4096 rows as at 8192).
    static const int mb = [] {
        const char* v = getenv("TM_F16_GATED_CHUNK_MB");
        return v && atoi(v) > 0 ? atoi(v) : 256;
    }();
    int rows = (int)(((size_t)mb << 20) / ((size_t)N * 2));
    rows     = rows / 256 * 256;
    rows     = rows < 256 ? 256 : rows;
    if (rows >= M) {
        return M;
    }
    const int chunks = (M + rows - 1) / rows;  // equal chunks rather than a small last one
    return ((M + chunks - 1) / chunks + 255) / 256 * 256;
}

size_t align256(size_t b)
{
    return (b + 255) / 256 * 256;
}

}  // namespace

bool f16_library_available()
{
    static const bool on = [] {
        const char* v = getenv("TM_GEMM_F16_LIBRARY");
        return !(v && atoi(v) == 0);
    }();
    return on && lt_api().ok;
}

size_t f16_image_bytes(int K, int N)
{
    return (size_t)K * N * sizeof(half_t);
}

size_t f16_library_workspace_bytes(int K, int N, int M, bool gated, bool resident)
{
    size_t b = kLtWorkspace;
    if (!resident) {
        b += align256(f16_image_bytes(K, N));
    }
    if (gated) {
        b += align256((size_t)gated_chunk_rows(N, M) * N * sizeof(half_t));
    }
    return b;
}

int launch_dequant_p32_f16(half_t* out_nk, const LinearWeight& w, hipStream_t st)
{
    TM_REQUIRE(w.type == 0 && w.packed32 != nullptr && w.N % 32 == 0 && w.K % 128 == 0, "fp16 image: a u4 linear with its P32 image");
    const int ncg = w.N / 32, KB = w.K / 128;
    dequant_p32_f16_kernel<<<dim3((ncg + 3) / 4, KB), 256, 0, st>>>(out_nk, (const char*)w.packed32, KB, ncg);
    TM_HIP_CHECK(hipGetLastError());
    return 0;
}

int linear_weight_build_f16_image(LinearWeight& w, hipStream_t st)
{
    if (w.f16_nk) {
        return 0;
    }
    TM_HIP_CHECK(hipMalloc((void**)&w.f16_nk, f16_image_bytes(w.K, w.N)));
    return launch_dequant_p32_f16(w.f16_nk, w, st);
}

int launch_linear_f16_library(const LinearWeight& w, const half_t* x, int ldx, half_t* y, int ldy, int M, bool gated_silu, void* ws,
                              size_t ws_bytes, hipStream_t st)
{
    TM_REQUIRE(f16_library_available(), "the vendor GEMM library is not loadable in this process (TM_GEMM_F16_LIBRARY / TM_HIPBLASLT_PATH)");
    TM_REQUIRE(w.type == 0 && w.packed32 != nullptr && w.N % 32 == 0 && w.K % 128 == 0, "library GEMM: a dense u4 linear with its P32 image");
    TM_REQUIRE(ws != nullptr && ws_bytes >= f16_library_workspace_bytes(w.K, w.N, M, gated_silu, w.f16_nk != nullptr)
                   && ((uintptr_t)ws & 255) == 0,
               "library GEMM: workspace of f16_library_workspace_bytes(), 256-byte aligned");
    TM_REQUIRE(ldx % 8 == 0 && ldy % 8 == 0, "library GEMM: 16-byte aligned x and y rows");
    const LtApi& api = lt_api();
    char*        cur = (char*)ws;
    void* const  lws = cur;
    cur += kLtWorkspace;
    const half_t* wnk = w.f16_nk;
    if (!wnk) {
        TM_TRY_RC(launch_dequant_p32_f16((half_t*)cur, w, st));
        wnk = (const half_t*)cur;
        cur += align256(f16_image_bytes(w.K, w.N));
    }
    const float one = 1.0f, zero = 0.0f;
    if (!gated_silu) {
        hipblasLtHandle_t h;
        LtPlan            p;
        TM_TRY_RC(lt_plan(w.N, M, w.K, ldx, ldy, &h, &p));
        TM_LT_CHECK(api.matmul(h, p.desc, &one, wnk, p.a, x, p.b, &zero, y, p.d, y, p.d, &p.algo, lws, kLtWorkspace, st));
        return 0;
    }
    half_t* const tmp   = (half_t*)cur;
    const int     chunk = gated_chunk_rows(w.N, M);
    for (int m0 = 0; m0 < M; m0 += chunk) {
        const int         rows = M - m0 < chunk ? M - m0 : chunk;
        hipblasLtHandle_t h;
        LtPlan            p;
        TM_TRY_RC(lt_plan(w.N, rows, w.K, ldx, w.N, &h, &p));
        TM_LT_CHECK(api.matmul(h, p.desc, &one, wnk, p.a, x + (size_t)m0 * ldx, p.b, &zero, tmp, p.d, tmp, p.d, &p.algo, lws, kLtWorkspace, st));
        const size_t total = (size_t)rows * (w.N / 16);
        silu_mul_interleaved_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(y + (size_t)m0 * ldy, ldy, tmp, w.N, rows, w.N / 2);
        TM_HIP_CHECK(hipGetLastError());
    }
    return 0;
}

}  // namespace tmk
