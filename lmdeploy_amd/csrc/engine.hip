// Static-batch decode engine around the HIP kernels: the MI355X-native counterpart of
//   TurboMind::CreateEngine / ModelExecutor::Run / LanguageModel::Forward / UnifiedDecoder::Forward
//   (src/turbomind/turbomind.cc:281-361, engine/model_executor.cc:68-101, models/language_model.cc:493-542,
//    models/llama/unified_decoder.cc:163-380, unified_attention_layer.cc:365-441, LlamaFfnLayer.cc:28-91).
//
// One process per GPU.  All per-step state (context lengths, current token ids, generated tokens, step
// counter) lives in device memory so a whole decode step is a fixed kernel sequence that is captured ONCE in
// a hipGraph and replayed (the reference keeps ids on the device too: language_model.cc:540).  Tensor
// parallelism: column-parallel w_qkv / w1w3, row-parallel wo / w2 followed by an RCCL all-reduce of the
// fp16 [M,H] partial sums (comm/nccl/nccl.cu:356-398), lm_head sharded over the vocabulary with a
// (value, index) all-gather instead of gathering logits.
//
// This file: engine configuration, weight slots / synthetic weights / weight processing, start-up allocation, fetch / stats /
// destroy.  The forward pass lives in engine_forward.hip, collectives in engine_comm.hip, the start-up tuner in engine_tune.hip,
// continuous batching and the engine thread in engine_serve.hip (shared declarations: engine_internal.h).
#include "engine_internal.h"

namespace tmk {

// ---------------------------------------------------------------------------------------------
// device-side helpers for the engine
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// out[i] = h(mean + std * N(0,1)), counter-based (Box-Muller on splitmix64), 2 values per thread
__global__ void fill_normal_kernel(half_t* out, size_t n, float mean, float stddev, uint64_t seed)
{
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (i >= n) {
        return;
    }
    const uint64_t r  = splitmix64(seed ^ (i * 0x9E3779B97F4A7C15ull + 0x1234567ull));
    const float    u1 = ((float)(uint32_t)(r >> 40) + 1.0f) * (1.0f / 16777217.0f);  // (0,1)
    const float    u2 = (float)(uint32_t)((r >> 8) & 0xffffffu) * (1.0f / 16777216.0f);
    const float    rr = sqrtf(-2.0f * __logf(u1));
    float          s, c;
    __sincosf(6.283185307179586f * u2, &s, &c);
    out[i] = (half_t)(mean + stddev * rr * c);
    if (i + 1 < n) {
        out[i + 1] = (half_t)(mean + stddev * rr * s);
    }
}

__global__ void fill_fp8_kernel(uint8_t* out, size_t n, uint64_t seed)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        uint64_t z = (i + 1) * 0x9E3779B97F4A7C15ull + seed * 0xD1B54A32D192ED03ull;
        z          = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z          = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        uint8_t b  = (uint8_t)(z >> 33);
        if ((b & 0x7f) == 0x7f) {
            b &= 0xf7;  // never NaN
        }
        out[i] = b;
    }
}

__global__ void fill_const_f32_kernel(float* out, size_t n, float v)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        out[i] = v;
    }
}

static int64_t slot_bytes_linear(const tm_engine* e, int K, int N, const char* part)
{
    if (!strcmp(part, "qweight")) return (int64_t)K * N / 2;
    if (!strcmp(part, "scales") || !strcmp(part, "zeros")) return (int64_t)(K / e->cfg.model.group_size) * N * 2;
    if (!strcmp(part, "weight")) return (int64_t)K * N * 2;
    if (!strcmp(part, "weight_fp8")) return (int64_t)K * N;
    if (!strcmp(part, "scales_fp8")) return (int64_t)(K / 128) * ((N + 127) / 128) * 4;
    return 0;
}

static void add_linear(tm_engine* e, LinearSlots& l, const std::string& prefix, int K, int N, int type, int role = 0)
{
    l.prefix  = prefix;
    l.w.role  = role;  // dispatch-table key next to (K, N, M): the tuner times each role with its own consumer
    l.w.K     = K;
    l.w.N     = N;
    l.w.group = e->cfg.model.group_size;
    l.w.type  = type;
    if (type == TM_WEIGHT_U4) {
        for (const char* part : {"qweight", "scales", "zeros"}) {
            e->slots[prefix + "." + part].bytes = slot_bytes_linear(e, K, N, part);
        }
    }
    else if (type == TM_WEIGHT_FP8) {  // e4m3 [K][N] + fp32 128x128 block scales (weight_format.py:349-393)
        e->slots[prefix + ".weight"].bytes = slot_bytes_linear(e, K, N, "weight_fp8");
        e->slots[prefix + ".scales"].bytes = slot_bytes_linear(e, K, N, "scales_fp8");
    }
    else {
        e->slots[prefix + ".weight"].bytes = slot_bytes_linear(e, K, N, "weight");
    }
}

static int ensure_slot(tm_engine* e, const std::string& name, Slot** out)
{
    auto it = e->slots.find(name);
    TM_REQUIRE(it != e->slots.end(), "unknown weight slot: " + name);
    Slot& s = it->second;
    if (!s.dev) {
        TM_HIP_CHECK(hipMalloc(&s.dev, s.bytes));
    }
    *out = &s;
    return 0;
}

static int fill_normal(tm_engine* e, void* dst, size_t n, float mean, float stddev, uint64_t seed)
{
    const size_t thr = (n + 1) / 2;
    fill_normal_kernel<<<(thr + 255) / 256, 256, 0, e->stream>>>((half_t*)dst, n, mean, stddev, seed);
    TM_HIP_CHECK(hipGetLastError());
    return 0;
}

static int prepare_linear(tm_engine* e, LinearSlots& l)
{
    if (l.w.type == TM_WEIGHT_U4) {
        Slot &q = e->slots[l.prefix + ".qweight"], &s = e->slots[l.prefix + ".scales"], &z = e->slots[l.prefix + ".zeros"];
        TM_REQUIRE(q.filled && s.filled && z.filled, "weight not loaded: " + l.prefix);
        // dense linears whose every M goes to gemm_decode.hip keep ONE HBM image (P32); MoE experts run the grouped mode of
        // gemm_kernel and keep the 16-column image as well
        const bool p32_only = l.prefix.find(".experts.") == std::string::npos && dec32_serves_every_m(l.w.K, l.w.N);
        TM_TRY(linear_weight_prepare_u4(l.w, (const int32_t*)q.dev, (const half_t*)s.dev, (const half_t*)z.dev, e->stream, p32_only));
        // the resident fp16 image for prefill-sized forwards (gemm_prefill_f16.hip): dense linears of engines whose prefill forwards are
        // large enough to use it (the 256 x 256 tile is proposed from 1024 rows); TM_PREFILL_F16_IMAGE=1 builds it
        // (opt-in: measured +15 % on w_qkv, +2 % on wo, -1 .. -5 % on w1w3 / w2 against the fused tiles at M = 8192 -- the tile is bound by the
        // CU's load path, not by the dequantisation it removes: profiles/r06_prefill_f16_image_ablations.txt -- for 2 bytes per parameter)
        static const bool f16img = [] {
            const char* v = getenv("TM_PREFILL_F16_IMAGE");
            return v && atoi(v) != 0;
        }();
        if (f16img && p32_only && l.w.packed32 && e->cfg.max_prefill_token_num >= 1024 && l.w.N >= 256) {
            TM_TRY(linear_weight_build_f16_image(l.w, e->stream));
        }
        TM_HIP_CHECK(hipStreamSynchronize(e->stream));
        for (Slot* p : {&q, &s, &z}) {
            TM_HIP_CHECK(hipFree(p->dev));
            p->dev = nullptr;
        }
    }
    else if (l.w.type == TM_WEIGHT_FP8) {
        Slot &w = e->slots[l.prefix + ".weight"], &s = e->slots[l.prefix + ".scales"];
        TM_REQUIRE(w.filled && s.filled, "weight not loaded: " + l.prefix);
        const bool gated = l.prefix.size() >= 5 && l.prefix.compare(l.prefix.size() - 5, 5, ".w1w3") == 0;
        TM_TRY(linear_weight_prepare_fp8(l.w, (const uint8_t*)w.dev, (const float*)s.dev, gated, e->stream));
        TM_HIP_CHECK(hipStreamSynchronize(e->stream));
        for (Slot* p : {&w, &s}) {
            TM_HIP_CHECK(hipFree(p->dev));
            p->dev = nullptr;
        }
    }
    else {
        Slot& w = e->slots[l.prefix + ".weight"];
        TM_REQUIRE(w.filled, "weight not loaded: " + l.prefix);
        TM_TRY(linear_weight_prepare_f16(l.w, (const half_t*)w.dev, e->stream));
        TM_HIP_CHECK(hipStreamSynchronize(e->stream));
        TM_HIP_CHECK(hipFree(w.dev));
        w.dev = nullptr;
    }
    return 0;
}

}  // namespace tmk

extern "C" {

int tm_engine_create(tm_engine** out, const tm_engine_config* c)
{
    TM_REQUIRE(out && c, "null pointer");
    const tm_model_config& m = c->model;
    TM_REQUIRE(m.head_dim == 128, "head_dim must be 128");
    TM_REQUIRE(m.group_size == 128, "AWQ group size must be 128");
    TM_REQUIRE(c->tp >= 1 && c->rank >= 0 && c->rank < c->tp, "0 <= rank < tp");
    TM_REQUIRE(m.q_heads % c->tp == 0 && m.inter % c->tp == 0 && m.vocab % c->tp == 0, "heads/inter/vocab % tp");
    TM_REQUIRE(m.kv_heads % c->tp == 0 || c->tp % m.kv_heads == 0, "kv_heads vs tp");
    TM_REQUIRE(c->quant_policy == 0 || c->quant_policy == 4 || c->quant_policy == 8,
               "quant_policy in {0,4,8} (lmdeploy/messages.py:351-358)");
    TM_REQUIRE(c->cache_block_seq_len == 64, "cache_block_seq_len must be 64");
    TM_REQUIRE(m.weight_type == TM_WEIGHT_U4 || m.weight_type == TM_WEIGHT_F16 || m.weight_type == TM_WEIGHT_FP8, "weight_type");
    TM_REQUIRE(c->max_batch_size >= 1 && c->max_batch_size <= 1024 && c->session_len >= 1,
               "1 <= max_batch_size <= 1024, session_len >= 1");
    TM_HIP_CHECK(hipSetDevice(c->device));

    auto* e        = new tm_engine();
    e->cfg         = *c;
    e->hidden      = m.hidden;
    e->q_heads     = m.q_heads / c->tp;
    e->kv_heads    = std::max(1, m.kv_heads / c->tp);
    e->inter       = m.inter / c->tp;
    e->vocab_local = m.vocab / c->tp;
    {
        const char* fc = getenv("TM_FORCE_COMM");
        e->use_comm    = c->tp > 1 || (fc && atoi(fc));
        const char* ms    = getenv("TM_MIXED_STEP");
        const char* gc    = getenv("TM_GRAPH_COMM");
        e->mixed_steps_on = !(ms && !atoi(ms));
        e->graph_comm     = !(gc && !atoi(gc));
    }
    e->qkv_n       = (e->q_heads + 2 * e->kv_heads) * e->D;
    TM_REQUIRE((e->inter * 1) % 128 == 0 && (e->q_heads * e->D) % 128 == 0 && m.hidden % 128 == 0,
               "K dims must be multiples of 128 after TP sharding");
    TM_REQUIRE(m.moe_experts == 0 || (m.moe_experts <= 64 && m.moe_top_k >= 1 && m.moe_top_k <= 8 && m.moe_top_k <= m.moe_experts
                                      && m.weight_type != TM_WEIGHT_F16),
               "moe: 1 <= top_k <= experts <= 64, top_k <= 8, u4 or fp8 expert weights");
    TM_HIP_CHECK(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));

    e->layers.resize(m.layers);
    for (int i = 0; i < m.layers; ++i) {
        const std::string p = "layers." + std::to_string(i);
        Layer&            L = e->layers[i];
        add_linear(e, L.qkv, p + ".attention.w_qkv", m.hidden, e->qkv_n, m.weight_type, 1);
        add_linear(e, L.wo, p + ".attention.wo", e->q_heads * e->D, m.hidden, m.weight_type, 2);
        if (m.moe_experts > 0) {
            L.is_moe = true;
            L.ex13.resize(m.moe_experts);
            L.ex2.resize(m.moe_experts);
            e->slots[p + ".moe_ffn.gate.weight"].bytes = (int64_t)m.hidden * m.moe_experts * 2;  // fp16 [H][E], replicated
            for (int x = 0; x < m.moe_experts; ++x) {
                const std::string q = p + ".moe_ffn.experts." + std::to_string(x);
                add_linear(e, L.ex13[x], q + ".w1w3", m.hidden, 2 * e->inter, m.weight_type);
                add_linear(e, L.ex2[x], q + ".w2", e->inter, m.hidden, m.weight_type);
            }
        }
        else {
            add_linear(e, L.w13, p + ".feed_forward.w1w3", m.hidden, 2 * e->inter, m.weight_type, 3);
            add_linear(e, L.w2, p + ".feed_forward.w2", e->inter, m.hidden, m.weight_type, 4);
        }
        e->slots[p + ".attention_norm.weight"].bytes = (int64_t)m.hidden * 2;
        e->slots[p + ".ffn_norm.weight"].bytes       = (int64_t)m.hidden * 2;
    }
    e->slots["tok_embeddings.weight"].bytes = (int64_t)m.vocab * m.hidden * 2;  // replicated
    e->slots["norm.weight"].bytes           = (int64_t)m.hidden * 2;
    add_linear(e, e->output, "output", m.hidden, e->vocab_local, TM_WEIGHT_F16, 5);
    *out = e;
    return 0;
}

int64_t tm_engine_weight_bytes(tm_engine* e, const char* name)
{
    if (!e || !name) {
        return 0;
    }
    auto it = e->slots.find(name);
    if (it == e->slots.end()) {
        set_last_error(std::string("unknown weight slot: ") + name);
        return 0;
    }
    return it->second.bytes;
}

int tm_engine_weight_copy(tm_engine* e, const char* name, const void* host_src, int64_t bytes)
{
    TM_REQUIRE(e && name && host_src, "null pointer");
    TM_REQUIRE(!e->weights_ready, "weights already processed");
    Slot* s = nullptr;
    TM_TRY(ensure_slot(e, name, &s));
    TM_REQUIRE(bytes == s->bytes, std::string("byte size mismatch for ") + name + ": got " + std::to_string(bytes)
                                      + " expected " + std::to_string(s->bytes));
    TM_HIP_CHECK(hipMemcpy(s->dev, host_src, bytes, hipMemcpyHostToDevice));
    s->filled = true;
    return 0;
}

int tm_engine_init_synthetic(tm_engine* e, uint64_t seed)
{
    TM_REQUIRE(e, "null pointer");
    TM_REQUIRE(!e->weights_ready, "weights already processed");
    const tm_model_config& m = e->cfg.model;
    uint64_t               sd = seed * 1000003ull + 17;
    half_t*                master = nullptr;  // fp16 master of the largest linear, reused
    size_t                 master_elems = 0;
    auto                   linear = [&](LinearSlots& l) -> int {
        const size_t n = (size_t)l.w.K * l.w.N;
        if (l.w.type == TM_WEIGHT_F16) {
            Slot* w = nullptr;
            TM_TRY(ensure_slot(e, l.prefix + ".weight", &w));
            TM_TRY(fill_normal(e, w->dev, n, 0.f, 0.1f / std::sqrt((float)l.w.K), ++sd));
            w->filled = true;
            return 0;
        }
        if (l.w.type == TM_WEIGHT_FP8) {
            // random e4m3 codes (no NaN), one constant block scale: E|code value| ~ 30 -> weights ~ 0.1 / sqrt(K)
            Slot *w = nullptr, *sc = nullptr;
            TM_TRY(ensure_slot(e, l.prefix + ".weight", &w));
            TM_TRY(ensure_slot(e, l.prefix + ".scales", &sc));
            fill_fp8_kernel<<<(n + 255) / 256, 256, 0, e->stream>>>((uint8_t*)w->dev, n, ++sd);
            TM_HIP_CHECK(hipGetLastError());
            const size_t nsc = (size_t)sc->bytes / 4;
            fill_const_f32_kernel<<<(nsc + 255) / 256, 256, 0, e->stream>>>((float*)sc->dev, nsc, 0.0027f / std::sqrt((float)l.w.K));
            TM_HIP_CHECK(hipGetLastError());
            w->filled = sc->filled = true;
            TM_TRY(prepare_linear(e, l));
            return 0;
        }
        if (n > master_elems) {
            if (master) {
                TM_HIP_CHECK(hipStreamSynchronize(e->stream));
                TM_HIP_CHECK(hipFree(master));
            }
            TM_HIP_CHECK(hipMalloc((void**)&master, n * 2));
            master_elems = n;
        }
        // TM_SYNTH_WEIGHT_SCALE (diagnostics only: `0` = all-zero linear weights, the low-toggle arm of the power experiment in DESIGN 3.8)
        static const float wscale = [] {
            const char* v = getenv("TM_SYNTH_WEIGHT_SCALE");
            return v ? (float)atof(v) : 1.0f;
        }();
        TM_TRY(fill_normal(e, master, n, 0.f, wscale * 0.1f / std::sqrt((float)l.w.K), ++sd));
        Slot *q = nullptr, *s = nullptr, *z = nullptr;
        TM_TRY(ensure_slot(e, l.prefix + ".qweight", &q));
        TM_TRY(ensure_slot(e, l.prefix + ".scales", &s));
        TM_TRY(ensure_slot(e, l.prefix + ".zeros", &z));
        TM_TRY(launch_quantize_groupwise_u4((int32_t*)q->dev, (half_t*)s->dev, (half_t*)z->dev, nullptr, master, l.w.K,
                                            l.w.N, l.w.group, e->stream));
        q->filled = s->filled = z->filled = true;
        // repack right away so that staging never holds more than one linear
        TM_TRY(prepare_linear(e, l));
        return 0;
    };
    auto vec = [&](const std::string& name, size_t n, float mean, float stddev) -> int {
        Slot* s = nullptr;
        TM_TRY(ensure_slot(e, name, &s));
        TM_TRY(fill_normal(e, s->dev, n, mean, stddev, ++sd));
        s->filled = true;
        return 0;
    };
    for (int i = 0; i < m.layers; ++i) {
        const std::string p = "layers." + std::to_string(i);
        Layer&            L = e->layers[i];
        TM_TRY(linear(L.qkv));
        TM_TRY(linear(L.wo));
        if (L.is_moe) {
            TM_TRY(vec(p + ".moe_ffn.gate.weight", (size_t)m.hidden * m.moe_experts, 0.f, 0.05f));
            for (int x = 0; x < m.moe_experts; ++x) {
                TM_TRY(linear(L.ex13[x]));
                TM_TRY(linear(L.ex2[x]));
            }
        }
        else {
            TM_TRY(linear(L.w13));
            TM_TRY(linear(L.w2));
        }
        TM_TRY(vec(p + ".attention_norm.weight", m.hidden, 1.f, 0.02f));
        TM_TRY(vec(p + ".ffn_norm.weight", m.hidden, 1.f, 0.02f));
    }
    TM_TRY(vec("tok_embeddings.weight", (size_t)m.vocab * m.hidden, 0.f, 0.02f));
    TM_TRY(vec("norm.weight", m.hidden, 1.f, 0.02f));
    TM_TRY(linear(e->output));
    TM_HIP_CHECK(hipStreamSynchronize(e->stream));
    if (master) {
        TM_HIP_CHECK(hipFree(master));
    }
    return 0;
}

int tm_engine_process_weights(tm_engine* e)
{
    TM_REQUIRE(e, "null pointer");
    if (e->weights_ready) {
        return 0;
    }
    TM_HIP_CHECK(hipSetDevice(e->cfg.device));
    const tm_model_config& m = e->cfg.model;
    auto norm = [&](const std::string& name, half_t** dst) -> int {
        Slot& s = e->slots[name];
        TM_REQUIRE(s.filled, "weight not loaded: " + name);
        *dst = (half_t*)s.dev;  // used in place
        return 0;
    };
    for (int i = 0; i < m.layers; ++i) {
        const std::string p = "layers." + std::to_string(i);
        Layer&            L = e->layers[i];
        for (LinearSlots* l : {&L.qkv, &L.wo}) {
            if (!l->w.packed && !l->w.packed32) {
                TM_TRY(prepare_linear(e, *l));
            }
        }
        if (L.is_moe) {
            L.moe.hidden       = m.hidden;
            L.moe.inter        = e->inter;
            L.moe.experts      = m.moe_experts;
            L.moe.top_k        = m.moe_top_k;
            L.moe.norm_topk    = m.moe_norm_topk != 0;
            L.moe.routed_scale = m.moe_routed_scale > 0.f ? m.moe_routed_scale : 1.f;
            L.moe.w13.resize(m.moe_experts);
            L.moe.w2.resize(m.moe_experts);
            for (int x = 0; x < m.moe_experts; ++x) {
                for (LinearSlots* l : {&L.ex13[x], &L.ex2[x]}) {
                    if (!l->w.packed && !l->w.packed32) {
                        TM_TRY(prepare_linear(e, *l));
                    }
                }
                L.moe.w13[x] = L.ex13[x].w;  // the MoE block owns the packed weights from here on
                L.moe.w2[x]  = L.ex2[x].w;
                L.ex13[x].w.packed = nullptr, L.ex13[x].w.sz = nullptr, L.ex13[x].w.packed32 = nullptr, L.ex13[x].w.packed8 = nullptr;
                L.ex2[x].w.packed = nullptr, L.ex2[x].w.sz = nullptr, L.ex2[x].w.packed32 = nullptr, L.ex2[x].w.packed8 = nullptr;
            }
            Slot& g = e->slots[p + ".moe_ffn.gate.weight"];
            TM_REQUIRE(g.filled, "weight not loaded: " + p + ".moe_ffn.gate.weight");
            L.moe.gate = (half_t*)g.dev;  // used in place (freed by moe_free)
            g.dev      = nullptr;
            TM_TRY(moe_prepare(L.moe, e->stream));
        }
        else {
            for (LinearSlots* l : {&L.w13, &L.w2}) {
                if (!l->w.packed && !l->w.packed32) {
                    TM_TRY(prepare_linear(e, *l));
                }
            }
        }
        TM_TRY(norm(p + ".attention_norm.weight", &L.attn_norm));
        TM_TRY(norm(p + ".ffn_norm.weight", &L.ffn_norm));
    }
    TM_TRY(norm("tok_embeddings.weight", &e->tok_embeddings));
    TM_TRY(norm("norm.weight", &e->final_norm));
    if (!e->output.w.packed) {
        TM_TRY(prepare_linear(e, e->output));
    }
    e->weights_ready = true;
    return 0;
}

int tm_engine_start(tm_engine* e)
{
    TM_REQUIRE(e, "null pointer");
    TM_REQUIRE(e->weights_ready, "process_weights first");
    if (e->started) {
        return 0;
    }
    TM_HIP_CHECK(hipSetDevice(e->cfg.device));
    const tm_engine_config& c = e->cfg;
    const tm_model_config&  m = c.model;
    const int bits = c.quant_policy == 0 ? 16 : c.quant_policy;
    e->layout      = KvLayout{e->kv_heads, e->D, c.cache_block_seq_len, bits};
    e->block_bytes = (int64_t)m.layers * e->layout.layer_size();
    e->max_blocks_per_seq = (c.session_len + 63) / 64;

    const int B   = c.max_batch_size;
    e->max_tokens = std::max(B, std::max(64, c.max_prefill_token_num));
    const size_t T = e->max_tokens;
    TM_TRY(dmalloc(&e->d_resid, T * e->hidden));
    TM_TRY(dmalloc(&e->d_x, T * e->hidden));
    TM_TRY(dmalloc(&e->d_tmp, T * e->hidden));
    TM_TRY(dmalloc(&e->d_qkv, T * e->qkv_n));
    TM_TRY(dmalloc(&e->d_attn, T * e->q_heads * e->D));
    TM_TRY(dmalloc(&e->d_act, T * e->inter));
    TM_TRY(dmalloc(&e->d_logits, (size_t)B * e->vocab_local));
    TM_TRY(dmalloc(&e->d_last, (size_t)B * e->hidden));
    // split-K workspace: decode-sized problems only (M <= 64 rows x widest N x 16 slabs)
    e->gemm_ws_bytes = (size_t)16 * 64 * std::max(std::max(e->qkv_n, 2 * e->inter), e->hidden) * sizeof(float);
    TM_HIP_CHECK(hipMalloc((void**)&e->d_gemm_ws, e->gemm_ws_bytes));
    {
        const size_t tiles = (size_t)(e->hidden + 63) / 64;
        TM_TRY(dmalloc(&e->d_ss, tiles * kFoldMaxRows));
        // arrival counters: one per (column tile, row block) of the widest decode linear at its narrowest tile (64 columns, 32 rows)
        const size_t ntk = (size_t)(std::max(std::max(e->qkv_n, 2 * e->inter), e->hidden) + 63) / 64 * (kFoldMaxRows / 32);
        const char* fm   = getenv("TM_FOLD_MAX_M");
        // default 64: at batch 128 (BASELINE config 3) the folded layer is parity-green but measured SLOWER than the reduce-norm launches it
        // removes (InternLM2-20B: 10.83 .. 10.91 vs 10.43 ms per step, profiles/r06_fold128_*): its producers are limited to the 32-row-block
        // tiles (every weight unit read by four row blocks), the 128-row tile with 10 slices + the fused reduce-norm wins w2 by 4 us
        e->fold_max_rows = fm ? std::min(kFoldMaxRows, std::max(1, atoi(fm))) : 64;
        TM_TRY(dmalloc(&e->d_tickets, ntk));
        TM_HIP_CHECK(hipMemset(e->d_tickets, 0, ntk * sizeof(unsigned)));
        // Default 0 since round 6: the folded layer is a departure from the reference's rounding sequence (one fp16 rounding
        // of the normalised activations instead of two) and a default-on departure has to pay more than box noise -- four interleaved runs per arm
        // on one box: 19 278.6 (fold 3) vs 19 283.5 tok/s (fold 0) on the driver command, 17 632.6 vs 17 619.5 over the 1k-out run
        // (profiles/r06_fold_ab.txt).  TM_FOLD_NORM=3 runs it (bit 0: wo -> w1w3, bit 1: w2 -> w_qkv); every test of it stays.
        const char* fold = getenv("TM_FOLD_NORM");
        e->fold_norm     = (!e->use_comm && m.weight_type == 0 && m.moe_experts == 0 && e->hidden % 64 == 0) ? (fold ? atoi(fold) & 3 : 0) : 0;
    }
    if (m.moe_experts > 0) {
        TM_HIP_CHECK(hipMalloc(&e->d_moe_ws, moe_workspace_bytes(e->layers[0].moe, e->max_tokens)));
    }
    // prefill scratch: every sequence padded to a multiple of 64 keys
    e->kflat_stride = ((e->max_tokens + c.session_len + 63) / 64) * 64 + 64 * (std::min(B, e->max_tokens) + 1);
    TM_TRY(dmalloc(&e->d_kflat, (size_t)e->kv_heads * e->kflat_stride * e->D));
    TM_TRY(dmalloc(&e->d_vflat, (size_t)e->kv_heads * e->kflat_stride * e->D));

    // RoPE table
    e->rope_max_pos = c.session_len + 1;
    {
        std::vector<half_t> tab((size_t)e->rope_max_pos * e->D);
        TM_TRY(build_rope_table(tab.data(), e->rope_max_pos, e->D, m.rope_base, m.rope_type, m.rope_factor,
                                m.rope_low_freq_factor, m.rope_high_freq_factor, m.rope_original_max_position));
        TM_TRY(dmalloc(&e->d_rope, tab.size() / 2));
        TM_HIP_CHECK(hipMemcpy(e->d_rope, tab.data(), tab.size() * 2, hipMemcpyHostToDevice));
    }

    TM_TRY(dmalloc(&e->d_ids, (size_t)B));
    TM_TRY(dmalloc(&e->d_next_ids, (size_t)B));
    TM_TRY(dmalloc(&e->d_k_len, (size_t)B));
    TM_TRY(dmalloc(&e->d_cu_q, (size_t)B + 1));
    TM_TRY(dmalloc(&e->d_cu_q_b, (size_t)B + 1));
    TM_TRY(dmalloc(&e->d_cu_koff, (size_t)B + 1));
    TM_TRY(dmalloc(&e->d_rows, (size_t)B));
    TM_TRY(dmalloc(&e->d_generated, (size_t)B * c.session_len));
    TM_TRY(dmalloc(&e->d_step, (size_t)1));
    TM_TRY(dmalloc(&e->d_prefill_ids, T));
    TM_TRY(dmalloc(&e->d_argmax_val, (size_t)B));
    TM_TRY(dmalloc(&e->d_cand, (size_t)B * 2));
    TM_TRY(dmalloc(&e->d_cand_all, (size_t)B * 2 * c.tp));
    TM_TRY(dmalloc(&e->d_block_ptrs, (size_t)B * e->max_blocks_per_seq));
    TM_TRY(dmalloc(&e->d_cu_block_nums, (size_t)B + 1));
    {
        std::vector<int> cu(B + 1);
        for (int b = 0; b <= B; ++b) {
            cu[b] = b * e->max_blocks_per_seq;
        }
        TM_HIP_CHECK(hipMemcpy(e->d_cu_block_nums, cu.data(), cu.size() * 4, hipMemcpyHostToDevice));
    }
    const int max_splits = 16;
    TM_HIP_CHECK(hipMalloc((void**)&e->d_attn_ws, decode_attention_workspace_bytes(B, e->q_heads, e->D, max_splits)));

    // KV pool
    int64_t blocks = c.cache_blocks;
    if (blocks <= 0) {
        size_t free_b = 0, total_b = 0;
        TM_HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
        const float frac = c.cache_max_entry_count > 0.f ? c.cache_max_entry_count : 0.8f;
        blocks           = (int64_t)((double)free_b * frac / (double)e->block_bytes);
        const int64_t need = (int64_t)B * e->max_blocks_per_seq;
        blocks           = std::min(blocks, need);
    }
    TM_REQUIRE(blocks >= 1, "no memory for KV blocks");
    if (hipMalloc((void**)&e->pool, (size_t)blocks * e->block_bytes) != hipSuccess) {
        (void)hipGetLastError();
        set_last_error("KV pool allocation failed");
        return TM_OOM;
    }
    TM_HIP_CHECK(hipMemsetAsync(e->pool, 0, (size_t)blocks * e->block_bytes, e->stream));
    e->num_blocks = blocks;
    e->free_blocks.resize(blocks);
    for (int64_t i = 0; i < blocks; ++i) {
        e->free_blocks[i] = (int)(blocks - 1 - i);
    }
    TM_HIP_CHECK(hipStreamSynchronize(e->stream));
    e->started = true;
    // the reference's switches: TM_GEMM_IMPORT=<file> loads a dispatch table, TM_GEMM_TUNE=1 measures the decode batch
    // (max_batch_size rows) now, TM_GEMM_EXPORT=<file> writes the table
    if (const char* imp = getenv("TM_GEMM_IMPORT")) {
        if (dec32_table_import(imp)) {
            fprintf(stderr, "[tm] TM_GEMM_IMPORT: nothing read from %s\n", imp);
        }
    }
    const char* tune = getenv("TM_GEMM_TUNE");
    if (tune && atoi(tune) && B <= 256) {
        const char* v = getenv("TM_GEMM_TUNE_VERBOSE");
        if (tune_decode_gemms(e, B, v && atoi(v))) {  // not fatal: the heuristic tilings stay
            fprintf(stderr, "[tm] TM_GEMM_TUNE failed (%s); keeping the heuristic tilings\n", tm_last_error());
        }
    }
    if (const char* exp = getenv("TM_GEMM_EXPORT")) {
        TM_TRY(dec32_table_export(exp));
    }
    return 0;
}

int tm_engine_sync(tm_engine* e)
{
    TM_REQUIRE(e, "null pointer");
    TM_HIP_CHECK(hipStreamSynchronize(e->stream));
    TM_TRY(device_marks_fetch(e, false));
    return device_marks_check(e);
}

int tm_engine_fetch(tm_engine* e, int* host_out, int* n_generated)
{
    TM_REQUIRE(e && host_out && n_generated, "null pointer");
    TM_HIP_CHECK(hipStreamSynchronize(e->stream));
    TM_TRY(device_marks_fetch(e, false));
    // the tokens are handed over in any case (those of the steps before a communicator give-up are valid); the status says
    // whether every step behind them was
    TM_HIP_CHECK(hipMemcpy(host_out, e->d_generated, (size_t)e->batch * e->max_new * 4, hipMemcpyDeviceToHost));
    const int rc = device_marks_check(e);
    if (rc && e->steps_valid < 0) {
        e->steps_valid = e->steps_fetched;  // the mark was first seen now: what an earlier, clean fetch reported is known to be valid
    }
    // after a communicator give-up *n_generated = the columns known to be valid (the step count of the last clean fetch): the
    // caller can keep those and must discard the rest; the status says so
    *n_generated = rc ? (e->steps_valid < 0 ? 0 : e->steps_valid) : e->steps_done;
    if (!rc) {
        e->steps_fetched = e->steps_done;
    }
    return rc;
}

int tm_engine_fetch_logits(tm_engine* e, void* host_out)
{
    TM_REQUIRE(e && host_out, "null pointer");
    TM_HIP_CHECK(hipStreamSynchronize(e->stream));
    TM_HIP_CHECK(hipMemcpy(host_out, e->d_logits, (size_t)e->batch * e->vocab_local * 2, hipMemcpyDeviceToHost));
    return 0;
}

int tm_engine_debug_read(tm_engine* e, int what, int a, int b, void* host_out, int64_t bytes)
{
    TM_REQUIRE(e && host_out && e->started, "null pointer / engine not started");
    TM_HIP_CHECK(hipStreamSynchronize(e->stream));
    if (what == 0) {  // residual stream of the last forward: rows [0, a)
        TM_REQUIRE(a >= 1 && a <= e->max_tokens && bytes == (int64_t)a * e->hidden * 2, "residual rows / byte count");
        TM_HIP_CHECK(hipMemcpy(host_out, e->d_resid, (size_t)bytes, hipMemcpyDeviceToHost));
        return 0;
    }
    if (what == 1) {  // KV block b of static-batch sequence a (all layers, the reference's block byte layout)
        TM_REQUIRE(a >= 0 && a < (int)e->h_blocks.size() && b >= 0 && b < (int)e->h_blocks[a].size(), "sequence / block index");
        TM_REQUIRE(bytes == (int64_t)e->block_bytes, "byte count must be the block size");
        TM_HIP_CHECK(hipMemcpy(host_out, e->pool + (int64_t)e->h_blocks[a][b] * e->block_bytes, (size_t)bytes, hipMemcpyDeviceToHost));
        return 0;
    }
    if (what == 2) {  // int64: scheduler steps whose decode rows rode on a prefill forward (mixed steps)
        TM_REQUIRE(bytes == 8, "byte count must be 8");
        *(int64_t*)host_out = e->mixed_steps;
        return 0;
    }
    if (what == 3) {  // int64: decode steps issued while the previous one was unretired (two-phase overlap, TM_ASYNC_STEP)
        TM_REQUIRE(bytes == 8, "byte count must be 8");
        *(int64_t*)host_out = e->overlapped_steps;
        return 0;
    }
    set_last_error("tm_engine_debug_read: unknown selector");
    return 1;
}

tm_stream_t tm_engine_stream(tm_engine* e)
{
    return e ? (tm_stream_t)e->stream : nullptr;
}

int tm_engine_stats(tm_engine* e, int64_t* weight_bytes, int64_t* kv_bytes_per_token, int64_t* num_blocks,
                    int* decode_splits)
{
    TM_REQUIRE(e, "null pointer");
    int64_t wb = 0;
    for (auto& L : e->layers) {
        for (const LinearSlots* l : {&L.qkv, &L.wo, &L.w13, &L.w2}) {
            wb += (int64_t)l->w.packed_bytes + (int64_t)l->w.sz_bytes;
        }
        if (L.is_moe) {
            for (int x = 0; x < L.moe.experts; ++x) {
                wb += (int64_t)L.moe.w13[x].packed_bytes + (int64_t)L.moe.w13[x].sz_bytes + (int64_t)L.moe.w2[x].packed_bytes
                      + (int64_t)L.moe.w2[x].sz_bytes;
            }
        }
    }
    wb += (int64_t)e->output.w.packed_bytes;
    if (weight_bytes) *weight_bytes = wb;
    if (kv_bytes_per_token) *kv_bytes_per_token = e->started ? e->block_bytes / 64 : 0;
    if (num_blocks) *num_blocks = e->num_blocks;
    if (decode_splits) *decode_splits = e->decode_splits;
    return 0;
}

int tm_engine_destroy(tm_engine* e)
{
    if (!e) {
        return 0;
    }
    (void)tm_engine_serve_stop(e);
    (void)hipSetDevice(e->cfg.device);
    if (e->stream) {
        (void)hipStreamSynchronize(e->stream);
    }
    if (e->graph) {
        (void)hipGraphExecDestroy(e->graph);
    }
    if (e->graph_cb) {
        (void)hipGraphExecDestroy(e->graph_cb);
    }
    if (e->aux_stream) {
        (void)hipStreamDestroy(e->aux_stream);
        (void)hipEventDestroy(e->ev_aux_fork);
        (void)hipEventDestroy(e->ev_aux_join);
    }
    for (int i = 0; i < 2; ++i) {
        for (void* q : {(void*)e->h_cb_lp_vals[i], (void*)e->h_cb_lp_idx[i], (void*)e->h_cb_lp_num[i], (void*)e->h_cb_lp_sel[i]}) {
            if (q) {
                (void)hipHostFree(q);
            }
        }
        if (e->h_step_pin[i]) {
            (void)hipHostFree(e->h_step_pin[i]);
            (void)hipEventDestroy(e->ev_step[i]);
        }
    }
    for (void* q : {(void*)e->d_active, (void*)e->d_pf_k_len, (void*)e->d_pf_cu_q, (void*)e->d_pf_block_ptrs, (void*)e->d_first_ids, (void*)e->d_temp, (void*)e->d_topp, (void*)e->d_minp,
                    (void*)e->d_u, (void*)e->d_topk, (void*)e->d_seed, e->d_sample_ws, (void*)e->d_logits_gather, (void*)e->d_logits_full,
                    (void*)e->d_seen, (void*)e->d_lp_rep,
                    (void*)e->d_lp_minlen, (void*)e->d_lp_ban, (void*)e->d_lp_end, (void*)e->d_lpr_vals, (void*)e->d_lpr_idx, (void*)e->d_lpr_num,
                    (void*)e->d_lpr_sel, (void*)e->d_kept, (void*)e->d_cb_lp_vals, (void*)e->d_cb_lp_idx, (void*)e->d_cb_lp_num, (void*)e->d_cb_lp_sel}) {
        if (q) {
            (void)hipFree(q);
        }
    }
    for (auto& L : e->layers) {
        for (LinearSlots* l : {&L.qkv, &L.wo, &L.w13, &L.w2}) {
            linear_weight_free(l->w);
        }
        for (auto& l : L.ex13) {
            linear_weight_free(l.w);
        }
        for (auto& l : L.ex2) {
            linear_weight_free(l.w);
        }
        if (L.is_moe) {
            moe_free(L.moe);
        }
    }
    if (e->d_moe_ws) {
        (void)hipFree(e->d_moe_ws);
    }
    linear_weight_free(e->output.w);
    for (auto& kv : e->slots) {
        if (kv.second.dev) {
            (void)hipFree(kv.second.dev);
        }
    }
    void* bufs[] = {e->pool, e->d_block_ptrs, e->d_cu_block_nums, e->d_resid, e->d_x, e->d_qkv, e->d_attn, e->d_act,
                    e->d_tmp, e->d_logits, e->d_last, e->d_gemm_ws, e->d_attn_ws, e->d_kflat, e->d_vflat, e->d_rope,
                    e->d_ids, e->d_k_len, e->d_cu_q, e->d_cu_q_b, e->d_cu_koff, e->d_rows, e->d_generated, e->d_step,
                    e->d_prefill_ids, e->d_argmax_val, e->d_cand, e->d_cand_all, e->d_next_ids, e->d_ss, e->d_tickets};
    for (void* p : bufs) {
        if (p) {
            (void)hipFree(p);
        }
    }
    for (int r = 0; r < 8; ++r) {
        (void)tm_p2p_segment_close(e->p2p_peer[r], 1);
    }
    (void)tm_p2p_segment_close(e->p2p_seg, 0);
    if (e->p2p_state) {
        (void)hipFree(e->p2p_state);
    }
    if (e->comm) {
        (void)ncclCommDestroy(e->comm);
    }
    if (e->comm_stream) {
        (void)hipStreamDestroy(e->comm_stream);
    }
    for (hipEvent_t ev : e->pipe_events) {
        (void)hipEventDestroy(ev);
    }
    if (e->stream) {
        (void)hipStreamDestroy(e->stream);
    }
    delete e;
    return 0;
}

}  // extern "C"
