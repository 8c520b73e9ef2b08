// Measured GEMM dispatch of the engine (reference: the warm-up tuning of turbomind.cc:363-487 -> gemm::Gemm::Run's DispatchCache,
// kernels/gemm/gemm.cu:92-224; TM_GEMM_TUNE / TM_GEMM_EXPORT / TM_GEMM_IMPORT).
#include "engine_internal.h"

namespace tmk {

}  // namespace tmk

extern "C" {

// ------------------------------------------------------------------------------------------------------------------
// Measured GEMM dispatch (reference: the warm-up tuning of turbomind.cc:363-487 -> gemm::Gemm::Run's DispatchCache,
// kernels/gemm/gemm.cu:92-224; TM_GEMM_TUNE / TM_GEMM_EXPORT / TM_GEMM_IMPORT).  For the decode batch M <= 256 every
// dense linear role of the model (w_qkv, wo, w1w3, w2) is timed with every (workgroup shape, split-K) candidate of the
// decode kernel as ONE hipGraph over the model's own layers -- distinct weights per node, more bytes than the Infinity
// Cache holds, as in a decode step -- and each node is followed by the kernel that consumes its result (the fused split-K
// reduce + residual + RMSNorm for wo / w2, the slab reduce standing in for the attention prologue for w_qkv): a split-K
// GEMM looks cheap in isolation and pays at the kernel boundary (profiles/r02_gemm_boundary_gap.txt).  The winner enters
// the (K, N, M) table that dec32_pick consults first; it replaces the heuristic only when it is >= 7 % faster.
// ------------------------------------------------------------------------------------------------------------------
int tune_decode_gemms(tm_engine* e, int M, bool verbose)
{
    TM_REQUIRE(M >= 1 && M <= e->max_tokens && M == dec32_m_bucket(M),
               "tuning: 1 <= M <= 256 (a decode batch) or a prefill size class 512, 1024, ... 8192, within max_prefill_token_num");
    half_t* const norm_out = M <= e->cfg.max_batch_size ? e->d_last : e->d_x;  // (d_x is the INPUT of w_qkv / w1w3 only)
    hipStream_t st = e->stream;
    struct Role {
        const char*   name;
        int           which;  // 0 qkv, 1 wo, 2 w13, 3 w2
        const half_t* x;
        int           ldx;
        half_t*       y;
        int           ldy;
        bool          gated;
    };
    const Role roles[4] = {{"w_qkv", 0, e->d_x, e->hidden, e->d_qkv, e->qkv_n, false},
                           {"wo", 1, e->d_attn, e->q_heads * e->D, e->d_tmp, e->hidden, false},
                           {"w1w3", 2, e->d_x, e->hidden, e->d_act, e->inter, true},
                           {"w2", 3, e->d_act, e->inter, e->d_tmp, e->hidden, false}};
    hipEvent_t e0, e1;
    TM_HIP_CHECK(hipEventCreate(&e0));
    TM_HIP_CHECK(hipEventCreate(&e1));
    // stand-in activations (the buffers are scratch before the first forward): pseudo-random values of the magnitude a
    // normed hidden state / an attention output / a gated activation has -- NOT zeros, which let the power-limited matrix pipe
    // run ~45 % faster than on real data and mis-rank the compute-bound candidates (launch_fill_uniform_f16)
    TM_TRY(launch_fill_uniform_f16(e->d_x, (size_t)M * e->hidden, 1.7f, 1u, st));
    TM_TRY(launch_fill_uniform_f16(e->d_attn, (size_t)M * e->q_heads * e->D, 0.5f, 2u, st));
    TM_TRY(launch_fill_uniform_f16(e->d_act, (size_t)M * e->inter, 0.5f, 3u, st));
    TM_HIP_CHECK(hipMemsetAsync(e->d_resid, 0, (size_t)M * e->hidden * 2, st));
    TM_HIP_CHECK(hipMemsetAsync(e->d_ss, 0x3f, (size_t)(e->hidden / 64) * kFoldMaxRows * sizeof(float), st));  // finite stand-in sums of squares (0.747)
    int rc = 0;
    for (const Role& r : roles) {
        std::vector<const LinearWeight*> ws;
        for (Layer& L : e->layers) {
            if (r.which >= 2 && L.is_moe) {
                continue;
            }
            const LinearWeight* w = r.which == 0 ? &L.qkv.w : r.which == 1 ? &L.wo.w : r.which == 2 ? &L.w13.w : &L.w2.w;
            if (dec32_supported(*w, M)) {
                ws.push_back(w);
            }
        }
        if (ws.size() < 2) {
            continue;
        }
        const LinearWeight& w0 = *ws[0];
        int                 hs, hp;
        if (dec32_table_get(w0.K, w0.N, M, &hs, &hp, r.which + 1)) {
            continue;  // imported / tuned already
        }
        dec32_pick_ex(w0, M, &hs, &hp, false);
        int       cand[96][2];
        int       nc = dec32_candidates(w0, M, cand, 95);
        bool      has = false;
        for (int i = 0; i < nc; ++i) {
            has = has || (cand[i][0] == hs && cand[i][1] == hp);
        }
        if (!has) {
            cand[nc][0] = hs;
            cand[nc][1] = hp;
            ++nc;
        }
        float best = 1e30f, heur = 1e30f;
        int   bs = hs, bp = hp;
        for (int i = 0; i < nc && !rc; ++i) {
            GemmConfig cfg{};
            cfg.nt        = 2;
            cfg.waves     = 16;
            cfg.kphases   = 1;
            cfg.d32_shape = cand[i][0];
            cfg.splits    = cand[i][1];
            if (gemm_workspace_bytes(M, w0.N, cfg.splits) > e->gemm_ws_bytes) {
                continue;
            }
            const bool norm_consumer = (r.which == 1 || r.which == 3) && !e->use_comm;
            // decode batches of an engine that folds the RMSNorm into the GEMMs (linear_fold_*): the candidates are timed as they
            // will run -- wo / w2 with the residual / sums-of-squares epilogue (and the in-launch slab merge) instead of the
            // reduce-norm launch, w_qkv / w1w3 with the row factor from d_ss -- and only tiles whose kernel carries that code
            // (which == 1 / 2: the wo -> w1w3 pair, bit 0; which == 3 / 0: the w2 -> w_qkv pair, bit 1)
            const bool folded = M <= e->fold_max_rows && !e->layers[0].is_moe && (e->fold_norm & ((r.which == 1 || r.which == 2) ? 1 : 2)) != 0;
            const bool slabs_ok = norm_consumer || (r.which == 0 && e->fuse_qkv);  // folded: who can take fp32 slabs
            const bool mrg = dec32_is_merge_shape(cand[i][0]);  // split-K merged in the launch: nothing for a slab consumer to do
            if (mrg && norm_consumer) {
                continue;  // wo / w2: the folded producer merges in the launch anyway, the unfolded one hands its slabs to the reduce-norm
            }
            const bool reduce_after = folded && !norm_consumer && !slabs_ok && !mrg && cfg.splits > 1 && M > 64;  // linear_fold_consume's reduce launch
            if (folded && (!dec32_fold_shape_m(cand[i][0], M, norm_consumer) || (cfg.splits > 1 && !slabs_ok && !mrg && !reduce_after))) {
                continue;
            }
            cfg.tickets = e->d_tickets;
            auto chain = [&]() -> int {
                if (folded) {
                    const int tiles = e->hidden / 64;
                    for (const LinearWeight* w : ws) {
                        NormFold nf{};
                        if (norm_consumer) {
                            nf.resid = e->d_resid, nf.norm_w = e->final_norm, nf.ss_out = e->d_ss, nf.tickets = e->d_tickets;
                        }
                        else {
                            nf.ss_in = e->d_ss, nf.ss_tiles = tiles, nf.inv_h = 1.0f / (float)e->hidden, nf.eps = e->cfg.model.rms_eps;
                            nf.tickets = e->d_tickets;
                        }
                        int nslab = 1;
                        TM_TRY(launch_linear_dec32(*w, r.x, r.ldx, norm_consumer ? norm_out : r.y, norm_consumer ? e->hidden : r.ldy, M, r.gated,
                                                   cfg.d32_shape, cfg.splits, e->d_gemm_ws, &nslab, st, &nf));
                        if (reduce_after && nslab > 1) {
                            TM_TRY(launch_splitk_reduce(r.y, r.ldy, e->d_gemm_ws, nslab, M, w->N, r.gated, st));
                        }
                    }
                    return 0;
                }
                for (const LinearWeight* w : ws) {
                    int slabs = 1;
                    TM_TRY(launch_linear(*w, r.x, r.ldx, r.y, r.ldy, M, r.gated, cfg, e->d_gemm_ws, norm_consumer && cfg.splits > 1, &slabs, st));
                    if (norm_consumer) {
                        TM_TRY(launch_residual_rmsnorm(norm_out, e->d_resid, slabs > 1 ? nullptr : e->d_tmp, slabs > 1 ? e->d_gemm_ws : nullptr,
                                                       slabs, nullptr, e->final_norm, e->cfg.model.rms_eps, M, e->hidden, st));
                    }
                }
                return 0;
            };
            if (norm_consumer) {  // the chain accumulates into the residual stream: every candidate starts from zero
                TM_HIP_CHECK(hipMemsetAsync(e->d_resid, 0, (size_t)M * e->hidden * 2, st));
            }
            if ((rc = chain())) {  // eager once: lazy module loading, function attributes
                break;
            }
            hipGraph_t     g  = nullptr;
            hipGraphExec_t ge = nullptr;
            float          us = 1e30f;
            // every error path below ends the capture and destroys what was created: a failed candidate must not leave the
            // engine stream capturing (every later launch would fail) nor leak the graph
            auto timed = [&]() -> int {
                TM_HIP_CHECK(hipStreamSynchronize(st));
                TM_HIP_CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
                const int        crc = chain();
                const hipError_t erc = hipStreamEndCapture(st, &g);
                if (crc) {
                    return crc;
                }
                TM_HIP_CHECK(erc);
                TM_HIP_CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
                for (int rep = 0; rep < 6; ++rep) {
                    TM_HIP_CHECK(hipEventRecord(e0, st));
                    TM_HIP_CHECK(hipGraphLaunch(ge, st));
                    TM_HIP_CHECK(hipEventRecord(e1, st));
                    TM_HIP_CHECK(hipEventSynchronize(e1));
                    float ms = 0.f;
                    TM_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
                    if (rep > 0) {
                        us = std::min(us, ms * 1000.f / (float)ws.size());
                    }
                }
                return 0;
            };
            rc = timed();
            if (ge) {
                (void)hipGraphExecDestroy(ge);
            }
            if (g) {
                (void)hipGraphDestroy(g);
            }
            if (rc) {
                break;
            }
            if (cand[i][0] == hs && cand[i][1] == hp) {
                heur = us;
            }
            if (us < best) {
                best = us;
                bs   = cand[i][0];
                bp   = cand[i][1];
            }
            if (verbose) {
                fprintf(stderr, "[tm tune] %-5s K=%d N=%d M=%d shape %d splits %2d: %7.2f us / layer%s\n", r.name, w0.K, w0.N, M, cand[i][0],
                        cand[i][1], us, (cand[i][0] == hs && cand[i][1] == hp) ? "  <- heuristic" : "");
            }
        }
        if (rc) {
            break;
        }
        // keep the heuristic unless the measurement clearly beats it.  7 %: run-to-run spread of a candidate is +-3 % (the same tiling
        // measured 18.35 / 18.72 / 19.13 us in three starts on one box), and the chain here is not the model (the same kernel back
        // to back, its activations hot in L2): with a 3 % bar w2 of Llama-3-8B once flipped from (3, 4) to (6, 2) -- 12 % slower in
        // the model (gpurun_out/profile_r03c: 0.61 vs 0.54 ms per step)
        if (!(best < 0.93f * heur)) {
            bs = hs;
            bp = hp;
        }
        dec32_table_set(w0.K, w0.N, M, bs, bp, r.which + 1);
        if (verbose) {
            fprintf(stderr, "[tm tune] %-5s K=%d N=%d M=%d -> shape %d splits %d (%.2f us; heuristic shape %d splits %d %.2f us)\n", r.name, w0.K,
                    w0.N, M, bs, bp, best, hs, hp, heur);
        }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    TM_HIP_CHECK(hipMemsetAsync(e->d_resid, 0, (size_t)M * e->hidden * 2, st));
    TM_HIP_CHECK(hipStreamSynchronize(st));
    return rc;
}

// ------------------------------------------------------------------------------------------------------------------
// The same for everything that is not a P32 kernel: the dense linears gemm_kernel serves (e4m3 weight-only
// w_qkv / wo / w1w3 / w2, the fp16 lm_head) and the grouped expert GEMMs (row-tile height; u4 through gemm_kernel<GRP>, e4m3 on
// the fp8 matrix cores).  Same method: one hipGraph per candidate over the model's own weights, min of 5 replays, a measured
// winner replaces the heuristic only when it is >= 7 % faster.  Results enter gen_table (tm_kernels.h) and travel in the same
// export / import file as the P32 entries.
// ------------------------------------------------------------------------------------------------------------------
static int time_graph_us(tm_engine* e, const std::function<int()>& chain, float* us_out)
{
    hipStream_t    st = e->stream;
    hipGraph_t     g  = nullptr;
    hipGraphExec_t ge = nullptr;
    hipEvent_t     e0 = nullptr, e1 = nullptr;
    float          us = 1e30f;
    auto           run = [&]() -> int {
        TM_TRY(chain());  // eager once: lazy module loading, function attributes
        TM_HIP_CHECK(hipEventCreate(&e0));
        TM_HIP_CHECK(hipEventCreate(&e1));
        TM_HIP_CHECK(hipStreamSynchronize(st));
        TM_HIP_CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        const int        crc = chain();
        const hipError_t erc = hipStreamEndCapture(st, &g);  // always ends the capture: a failed candidate must not leave the stream capturing
        if (crc) {
            return crc;
        }
        TM_HIP_CHECK(erc);
        TM_HIP_CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int rep = 0; rep < 6; ++rep) {
            TM_HIP_CHECK(hipEventRecord(e0, st));
            TM_HIP_CHECK(hipGraphLaunch(ge, st));
            TM_HIP_CHECK(hipEventRecord(e1, st));
            TM_HIP_CHECK(hipEventSynchronize(e1));
            float ms = 0.f;
            TM_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0) {
                us = std::min(us, ms * 1000.f);
            }
        }
        return 0;
    };
    const int rc = run();
    if (ge) {
        (void)hipGraphExecDestroy(ge);
    }
    if (g) {
        (void)hipGraphDestroy(g);
    }
    if (e0) {
        (void)hipEventDestroy(e0);
    }
    if (e1) {
        (void)hipEventDestroy(e1);
    }
    *us_out = us;
    return rc;
}

static int tune_aux_gemms(tm_engine* e, int M, bool verbose)
{
    hipStream_t st = e->stream;
    const int   Mb = dec32_m_bucket(M);
    int         tv[4];
    // ---- dense linears of the general kernel: per role over the layers (with the consumer of wo / w2), then the lm_head ----
    struct Role {
        const char*   name;
        int           which;  // 0 qkv, 1 wo, 2 w13, 3 w2, 4 lm_head
        const half_t* x;
        int           ldx;
        half_t*       y;
        int           ldy;
        bool          gated;
    };
    half_t* const norm_out = M <= e->cfg.max_batch_size ? e->d_last : e->d_x;
    const Role roles[5] = {{"w_qkv", 0, e->d_x, e->hidden, e->d_qkv, e->qkv_n, false},
                           {"wo", 1, e->d_attn, e->q_heads * e->D, e->d_tmp, e->hidden, false},
                           {"w1w3", 2, e->d_x, e->hidden, e->d_act, e->inter, true},
                           {"w2", 3, e->d_act, e->inter, e->d_tmp, e->hidden, false},
                           {"lm_head", 4, e->d_last, e->hidden, e->d_logits, e->vocab_local, false}};
    TM_TRY(launch_fill_uniform_f16(e->d_x, (size_t)M * e->hidden, 1.7f, 1u, st));
    TM_TRY(launch_fill_uniform_f16(e->d_attn, (size_t)M * e->q_heads * e->D, 0.5f, 2u, st));
    TM_TRY(launch_fill_uniform_f16(e->d_act, (size_t)M * e->inter, 0.5f, 3u, st));
    for (const Role& r : roles) {
        std::vector<const LinearWeight*> ws;
        if (r.which == 4) {
            if (M <= e->cfg.max_batch_size) {  // logits exist for batch-slot rows only
                TM_TRY(launch_fill_uniform_f16(e->d_last, (size_t)M * e->hidden, 1.7f, 4u, st));
                ws.push_back(&e->output.w);
            }
        }
        else {
            for (Layer& L : e->layers) {
                if (r.which >= 2 && L.is_moe) {
                    continue;
                }
                const LinearWeight* w = r.which == 0 ? &L.qkv.w : r.which == 1 ? &L.wo.w : r.which == 2 ? &L.w13.w : &L.w2.w;
                if (!dec32_supported(*w, M)) {
                    ws.push_back(w);
                }
            }
        }
        if (ws.empty() || (r.which < 4 && ws.size() < 2)) {
            continue;
        }
        const LinearWeight& w0 = *ws[0];
        if (gen_table_get(kGenDense + w0.type, w0.role, w0.K, w0.N, Mb, tv)) {
            continue;  // imported / tuned already
        }
        GemmConfig cand[16];
        int        nc = gen_dense_candidates(w0, M, e->gemm_ws_bytes, cand, 15);
        if (nc == 0) {
            continue;
        }
        const GemmConfig heur = gemm_pick_config_general(w0, M);
        auto same = [](const GemmConfig& a, const GemmConfig& b) { return a.nt == b.nt && a.splits == b.splits; };
        bool has = false;
        for (int i = 0; i < nc; ++i) {
            has = has || same(cand[i], heur);
        }
        if (!has) {
            cand[nc++] = heur;
        }
        const bool norm_consumer = (r.which == 1 || r.which == 3) && !e->use_comm;
        float      best = 1e30f, t_heur = 1e30f;
        GemmConfig bc = heur;
        for (int i = 0; i < nc; ++i) {
            const GemmConfig cfg = cand[i];
            if (gemm_workspace_bytes(M, w0.N, cfg.splits) > e->gemm_ws_bytes) {
                continue;
            }
            auto chain = [&]() -> int {
                for (const LinearWeight* w : ws) {
                    int slabs = 1;
                    TM_TRY(launch_linear(*w, r.x, r.ldx, r.y, r.ldy, M, r.gated, cfg, e->d_gemm_ws, norm_consumer && cfg.splits > 1, &slabs, st));
                    if (norm_consumer) {
                        TM_TRY(launch_residual_rmsnorm(norm_out, e->d_resid, slabs > 1 ? nullptr : e->d_tmp, slabs > 1 ? e->d_gemm_ws : nullptr,
                                                       slabs, nullptr, e->final_norm, e->cfg.model.rms_eps, M, e->hidden, st));
                    }
                    else if (r.which == 4) {  // the head's consumer
                        TM_TRY(launch_argmax(e->d_next_ids, nullptr, e->d_logits, M, e->vocab_local, e->vocab_local, 0, st));
                    }
                }
                return 0;
            };
            if (norm_consumer) {
                TM_HIP_CHECK(hipMemsetAsync(e->d_resid, 0, (size_t)M * e->hidden * 2, st));
            }
            float us = 1e30f;
            TM_TRY(time_graph_us(e, chain, &us));
            us /= (float)ws.size();
            if (same(cfg, heur)) {
                t_heur = us;
            }
            if (us < best) {
                best = us;
                bc   = cfg;
            }
            if (verbose) {
                fprintf(stderr, "[tm tune] %-7s K=%d N=%d M=%d general nt %d splits %d: %8.2f us%s\n", r.name, w0.K, w0.N, M, cfg.nt, cfg.splits, us,
                        same(cfg, heur) ? "  <- heuristic" : "");
            }
        }
        if (!(best < 0.93f * t_heur)) {
            bc = heur;
        }
        const int v[4] = {bc.nt, bc.splits, bc.waves, bc.kphases < 1 ? 1 : bc.kphases};
        gen_table_set(kGenDense + w0.type, w0.role, w0.K, w0.N, Mb, v);
        if (verbose) {
            fprintf(stderr, "[tm tune] %-7s K=%d N=%d M=%d -> general nt %d splits %d (%.2f us; heuristic %.2f us)\n", r.name, w0.K, w0.N, M, bc.nt,
                    bc.splits, best, t_heur);
        }
    }
    // ---- grouped expert GEMMs: the row-tile height, first of w1w3 (w2 on its heuristic), then of w2 ----
    std::vector<Layer*> moe;
    for (Layer& L : e->layers) {
        if (L.is_moe) {
            moe.push_back(&L);
        }
    }
    if (moe.size() >= 2 && e->d_moe_ws) {
        auto chain = [&]() -> int {
            for (Layer* L : moe) {
                TM_TRY(moe_forward(L->moe, e->d_tmp, e->hidden, e->d_x, e->hidden, M, e->d_moe_ws, nullptr, nullptr, st));
            }
            return 0;
        };
        for (int which = 0; which < 2; ++which) {
            const LinearWeight& proto = which == 0 ? moe[0]->moe.w13[0] : moe[0]->moe.w2[0];
            const int           kind  = kGenGrouped + proto.type;
            if (gen_table_get(kind, 0, proto.K, proto.N, Mb, tv)) {
                continue;
            }
            int       rows[4];
            const int nc = gen_grouped_candidates(proto, M, rows, 4);
            if (nc < 2) {
                continue;
            }
            float t_heur = 1e30f, best = 1e30f;
            int   br = 0;
            TM_TRY(time_graph_us(e, chain, &t_heur));  // no entry: the launchers' own rule
            for (int i = 0; i < nc; ++i) {
                gen_grouped_rows_override(rows[i]);  // this thread's launches only: nothing transient enters the shared table
                float     us = 1e30f;
                const int rc = time_graph_us(e, chain, &us);
                gen_grouped_rows_override(0);
                if (rc) {
                    return rc;
                }
                if (verbose) {
                    fprintf(stderr, "[tm tune] experts %s K=%d N=%d tokens=%d rows/tile %2d: %9.2f us per MoE FFN (heuristic %.2f)\n",
                            which == 0 ? "w1w3" : "w2", proto.K, proto.N, M, rows[i], us / (float)moe.size(), t_heur / (float)moe.size());
                }
                if (us < best) {
                    best = us;
                    br   = rows[i];
                }
            }
            if (br && best < 0.93f * t_heur) {  // else: no entry, the heuristic stays
                const int v[4] = {br, 0, 0, 0};
                gen_table_set(kind, 0, proto.K, proto.N, Mb, v);
            }
            if (verbose) {
                fprintf(stderr, "[tm tune] experts %s K=%d N=%d tokens=%d -> %s (best %.2f us, heuristic %.2f us per MoE FFN)\n", which == 0 ? "w1w3" : "w2",
                        proto.K, proto.N, M, (br && best < 0.93f * t_heur) ? "measured tile" : "heuristic", best / (float)moe.size(),
                        t_heur / (float)moe.size());
            }
        }
    }
    TM_HIP_CHECK(hipMemsetAsync(e->d_resid, 0, (size_t)M * e->hidden * 2, st));
    TM_HIP_CHECK(hipStreamSynchronize(st));
    return 0;
}

int tm_engine_tune_gemm(tm_engine* e, int M, const char* export_path)
{
    TM_REQUIRE(e && e->started, "engine not started");
    TM_REQUIRE(e->batch == 0 && !e->sched, "tune before the first batch is admitted");
    TM_HIP_CHECK(hipSetDevice(e->cfg.device));
    const char* v = getenv("TM_GEMM_TUNE_VERBOSE");
    TM_TRY(tune_decode_gemms(e, M, v && atoi(v)));
    TM_TRY(tune_aux_gemms(e, M, v && atoi(v)));
    if (export_path && *export_path) {
        return dec32_table_export(export_path);
    }
    return 0;
}

int tm_gemm_import(const char* path)
{
    TM_REQUIRE(path && *path, "path");
    return dec32_table_import(path);
}

int tm_gemm_export(const char* path)
{
    TM_REQUIRE(path && *path, "path");
    return dec32_table_export(path);
}

}  // extern "C"
