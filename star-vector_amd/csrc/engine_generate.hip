// Host driver, part 3 of 5: sv_generate -- the greedy / sampling loop (one decode step captured as a hipGraph and replayed, the
// stop state on device), beam search over the same step, and the standalone beam scorer's C ABI (sv_beam_*).
#include "engine_internal.h"

// sample from e->logits into next_tok, then the bookkeeping kernel
// fused: the lm_head launch in front has already left every row's best (value, lowest index) in e->amax (gemm.hip, SkinnyArgs::amax):
// no argmax launch, finish_step_kernel decodes the keys
static FinishArgs make_finish_args(sv_engine* e, int B, const sv_sampling& sp, int max_new, bool fused) {
    const bool pen = sp.repetition_penalty > 0.f && sp.repetition_penalty != 1.0f;
    FinishArgs f;
    f.pval = sp.do_sample ? nullptr : e->am_val; f.pidx = sp.do_sample ? nullptr : e->am_idx;
    f.amax = fused ? e->amax : nullptr;
    f.next = e->next_tok; f.cur_tok = e->cur_tok; f.unfinished = e->unfinished; f.positions = e->positions;
    f.out_tokens = e->out_tok; f.ld_out = e->out_ld; f.step = e->d_step; f.done = e->d_done; f.n_emitted = e->d_nemit;
    f.stop_ids = e->d_stop; f.n_stop = sp.n_stop; f.eos = sp.eos_token_id; f.pad = sp.pad_token_id; f.B = B;
    f.max_new = max_new;
    f.seen = pen ? e->seen : nullptr; f.seen_words = e->seen_words;
    f.V = e->cfg.vocab; f.bad = e->d_bad;
    return f;
}
static void sample_and_finish(sv_engine* e, int B, const sv_sampling& sp, int max_new, hipStream_t st, bool fused = false) {
    if (fused && e->fin_folded) { e->fin_folded = false; return; }      // the lm_head launch in front did the selection AND the bookkeeping (SkinnyArgs::finish)
    const bool pen = sp.repetition_penalty > 0.f && sp.repetition_penalty != 1.0f;
    const uint32_t* seen = pen ? e->seen : nullptr;
    if (sp.min_new_tokens > 0 && sp.eos_token_id >= 0 && sp.eos_token_id < e->cfg.vocab)
        suppress_token(e->logits, e->Vpad, sp.eos_token_id, e->d_step, sp.min_new_tokens, B, st);
    if (sp.do_sample) {
        SampleArgs sa;
        sa.logits = e->logits; sa.ld = e->Vpad; sa.V = e->cfg.vocab; sa.B = B; sa.temperature = sp.temperature;
        sa.top_p = sp.top_p; sa.top_k = sp.top_k; sa.seed = sp.seed; sa.step = e->d_step; sa.out = e->next_tok; sa.scratch = e->sample_scratch;
        sa.seen = seen; sa.seen_words = e->seen_words; sa.penalty = sp.repetition_penalty;
        launch_sample_top_p(sa, st);
    } else if (!fused) {
        launch_argmax_partial(e->logits, e->Vpad, e->cfg.vocab, e->am_val, e->am_idx, B, seen, e->seen_words,
                              sp.repetition_penalty, st);
    }
    const FinishArgs f = make_finish_args(e, B, sp, max_new, fused);
    launch_finish_step(f, st);
}

// ------------------------------------------------------------------------------------------------
// beam search (num_beams > 1): HF _beam_search restated on device (beam.hip).  The prompt is prefilled ONCE per
// request: its full KV pages are shared by all beams through the block table, only the partially filled tail page
// is private to a beam (HF expands the prompt to B * num_beams rows and prefills every copy).
// ------------------------------------------------------------------------------------------------
static int upload_beam_table(sv_engine* e, int B, int nb, int need, int shared_pages, bool prefill_rows, hipStream_t st) {
    std::vector<int32_t> table((size_t)e->cfg.max_batch * e->pages_per_seq, 0);
    if (prefill_rows) {
        for (int b = 0; b < B; ++b)             // request b writes its prompt into the pages of beam row b * nb
            for (int i = 0; i < need; ++i) table[(size_t)b * e->pages_per_seq + i] = (b * nb) * need + i;
    } else {
        for (int r = 0; r < B * nb; ++r)
            for (int i = 0; i < need; ++i)
                table[(size_t)r * e->pages_per_seq + i] = (i < shared_pages ? (r / nb) * nb : r) * need + i;
    }
    HIPCHECK(hipMemcpyAsync(e->block_table, table.data(), table.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    HIPCHECK(hipStreamSynchronize(st));
    return 0;
}

static void beam_step(sv_engine* e, const BeamKvArgs& kv, int logit_div, hipStream_t st) {
    e->beam.enqueue_step(e->logits, e->Vpad, logit_div, st);
    launch_beam_table_reorder(kv, st);
    launch_beam_tail_copy(kv, st);
}

static int generate_beam(sv_engine* e, const void* dev_embeds, int B, int S0, const sv_sampling* sp, int max_new,
                         int64_t* dev_out_tokens, int32_t* n_generated, hipStream_t st) {
    const sv_config& c = e->cfg;
    const int nb = sp->num_beams, R = B * nb;
    if (nb > BM_MAXNB) return fail(SV_EINVAL, "num_beams %d unsupported (2..%d)", nb, BM_MAXNB);
    if (R > c.max_batch) return fail(SV_EINVAL, "batch %d x num_beams %d exceeds engine max_batch %d", B, nb, c.max_batch);
    if (sp->early_stopping < 0 || sp->early_stopping > 2) return fail(SV_EINVAL, "early_stopping must be 0 (False), 1 (True) or 2 (\"never\")");
    if (!dev_embeds || B < 1) return fail(SV_EINVAL, "generate: bad batch %d", B);
    if (S0 < 1) return fail(SV_EINVAL, "generate: bad prompt length %d", S0);
    const int need = (S0 + max_new + SV_PAGE_TOKENS - 1) / SV_PAGE_TOKENS;
    if (need > e->pages_per_seq) return fail(SV_EINVAL, "sequence length %d exceeds max_seq_len %d", S0 + max_new, c.max_seq_len);

    auto t0 = std::chrono::steady_clock::now();
    HIPCHECK(hipMemsetAsync(e->d_bad, 0, sizeof(int32_t), st));         // a flag left by an earlier, failed call is not this call's
    BeamConfig bc;
    bc.B = B; bc.nb = nb; bc.V = c.vocab; bc.max_new = max_new; bc.eos = sp->eos_token_id; bc.pad = sp->pad_token_id;
    bc.early = sp->early_stopping; bc.length_penalty = sp->length_penalty;
    bc.penalty = sp->repetition_penalty > 0.f ? sp->repetition_penalty : 1.f;
    bc.n_stop = sp->n_stop;
    for (int i = 0; i < sp->n_stop; ++i) bc.stop[i] = sp->stop_ids[i];
    bc.do_sample = sp->do_sample ? 1 : 0; bc.temperature = sp->temperature; bc.top_p = sp->top_p; bc.top_k = sp->top_k;
    bc.seed = sp->seed;
    bc.min_new = sp->min_new_tokens > 0 ? sp->min_new_tokens : 0;
    if (!e->beam.matches(bc)) {
        int r = e->beam.init(bc, e->cur_tok, e->positions, e->d_step, e->d_done);
        if (r) return fail(SV_ENOMEM, "beam scorer allocation failed (hip error %d)", r);
    }
    e->beam.c = bc;
    // prompt pass over the B requests, written into the pages of each request's first beam row
    SVCHECK(upload_beam_table(e, B, nb, need, 0, true, st));
    SVCHECK(prefill_forward(e, (const bf16_t*)dev_embeds, B, S0, st));
    SVCHECK(upload_beam_table(e, B, nb, need, S0 / SV_PAGE_TOKENS, false, st));
    e->cached_B = R;
    {
        int r = e->beam.reset(st);
        if (r) return fail(SV_EHIP, "beam scorer reset failed (hip error %d)", r);
    }
    fill_i32(e->positions, S0 - 1, R, st);      // beam_update adds 1
    BeamKvArgs kv;
    kv.block_table = e->block_table; kv.max_pages = e->pages_per_seq; kv.need = need; kv.parent = e->beam.d.parent;
    kv.step = e->d_step; kv.done = e->d_done; kv.S0 = S0; kv.L_fixed = S0; kv.B = B; kv.nb = nb;
    kv.kv_pool = e->kv_pool; kv.layer_stride = e->layer_stride; kv.kv_head_stride = e->kv_head_stride;
    kv.n_layer = c.n_layer; kv.n_kv = e->nkv; kv.page_bytes = e->page_bytes;
    launch_beam_tail_copy(kv, st);              // the prompt's tail page fans out to beams 1.. (parent = first beam row)
    kv.L_fixed = -1;
    beam_step(e, kv, nb, st);                   // first token: every beam of a request reads the request's logits row
    HIPCHECK(hipMemcpyAsync(&e->h_flags[0], e->d_done, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIPCHECK(hipStreamSynchronize(st));
    auto t1 = std::chrono::steady_clock::now();

    int steps = 0;
    const int chunk = sp->sync_every > 0 ? sp->sync_every : 32;
    hipGraph_t graph = nullptr;
    hipGraphExec_t gexec = nullptr;
    if (!e->h_flags[0] && getenv("SV_NO_GRAPH") == nullptr) {
        hipError_t ce = hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed);
        if (ce == hipSuccess) {
            decode_forward(e, R, st);
            beam_step(e, kv, 1, st);
            ce = hipStreamEndCapture(st, &graph);
            if (ce == hipSuccess && graph) ce = hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0);
        }
        if (ce != hipSuccess) {
            (void)hipGetLastError();
            if (gexec) { (void)hipGraphExecDestroy(gexec); gexec = nullptr; }
            if (graph) { (void)hipGraphDestroy(graph); graph = nullptr; }
            if (getenv("SV_REQUIRE_GRAPH")) return fail(SV_EHIP, "hipGraph capture failed: %s", hipGetErrorString(ce));
        }
    }
    while (!e->h_flags[0]) {
        int n = max_new - 1 - steps;
        if (n <= 0) break;
        if (n > chunk) n = chunk;
        for (int i = 0; i < n; ++i) {
            if (gexec) {
                HIPCHECK(hipGraphLaunch(gexec, st));
            } else {
                decode_forward(e, R, st);
                beam_step(e, kv, 1, st);
            }
        }
        steps += n;
        HIPCHECK(hipMemcpyAsync(&e->h_flags[0], e->d_done, sizeof(int32_t), hipMemcpyDeviceToHost, st));
        HIPCHECK(hipStreamSynchronize(st));
    }
    const double gexec_used = gexec ? 1.0 : 0.0;
    if (gexec) (void)hipGraphExecDestroy(gexec);
    if (graph) (void)hipGraphDestroy(graph);
    // the decode steps of a beam search run the fused MLP launch too: its give-up code must not come back as rc 0 with hypotheses
    // built from void logits
    SVCHECK(check_finite_logits(e, st, "sv_generate (beam search)"));
    if (!e->h_flags[0]) return fail(SV_EHIP, "beam search did not terminate within its budget");
    std::vector<int64_t> toks;
    std::vector<float> scores;
    int L = 0;
    {
        int r = e->beam.finalize(st, toks, L, scores);
        if (r) return fail(SV_EHIP, "beam search bookkeeping failed (code %d)", r);
    }
    HIPCHECK(hipMemcpy2DAsync(dev_out_tokens, (size_t)max_new * sizeof(int64_t), toks.data(), (size_t)max_new * sizeof(int64_t),
                              (size_t)L * sizeof(int64_t), B, hipMemcpyHostToDevice, st));
    HIPCHECK(hipStreamSynchronize(st));
    auto t2 = std::chrono::steady_clock::now();
    *n_generated = L;
    e->timing[0] = std::chrono::duration<double, std::milli>(t1 - t0).count();
    e->timing[1] = std::chrono::duration<double, std::milli>(t2 - t1).count();
    e->timing[2] = (double)steps;
    e->timing_graph = gexec_used;
    return 0;
}

// The selection kernels raise a device flag when a row had no finite logit (a numeric failure upstream: the token they emit
// is then 0 instead of the out-of-range sentinel, so nothing indexes the embedding table out of bounds); the entry points
// turn it into an error instead of returning made-up tokens.
int sveng::check_finite_logits(sv_engine* e, hipStream_t st, const char* who) {
    HIPCHECK(hipMemcpyAsync(&e->h_flags[4], e->d_bad, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIPCHECK(hipStreamSynchronize(st));
    return report_bad_logits(e, st, who, e->h_flags[4]);
}
int sveng::report_bad_logits(sv_engine* e, hipStream_t st, const char* who, int what) {
    if (!what) return 0;
    HIPCHECK(hipMemsetAsync(e->d_bad, 0, sizeof(int32_t), st));
    // (The failed steps may have appended non-finite K / V rows to their pages.  Nothing is scrubbed here: the decode attention clears the stale
    //  V columns of a sequence's current key group itself -- attention.hip, process() -- so a page's next owner never multiplies them.)
    HIPCHECK(hipStreamSynchronize(st));
    if (what == 4 && e->rc_dbg) {
        long long d[8] = {0};
        int32_t stepv = -1;
        (void)hipMemcpy(d, e->rc_dbg, sizeof(d), hipMemcpyDeviceToHost);
        (void)hipMemcpy(&stepv, e->d_step, sizeof(stepv), hipMemcpyDeviceToHost);
        (void)hipMemset(e->rc_dbg, 0, sizeof(d));
        fprintf(stderr, "[sv] rowln_cattn give-up: block %lld wave %lld ks0 %lld waited %lld ticks, x0 %016llx, layer %lld, rows %lld; device step %d\n",
                d[0], d[1], d[2], d[3], (unsigned long long)d[4], d[5], d[6], stepv);
    }
    if (what == 3 || what == 4) {
        e->last_giveup = what;
        if (e->cfg.exclusive_device == 2 && !e->fused_off) {
            // optimistic ownership (sv_config.exclusive_device = 2) did not hold: from here on this engine decodes without the launches that need
            // every block resident.  sv_generate runs the failed call again (below); the step-wise entry points report this one failure.
            e->fused_off = true;
            e->xpa_armed = false;
            for (auto& kv : e->cb_graphs) {      // the continuous-batching step graphs were captured with the fused launches (sv_generate's carry the flag in their key)
                if (kv.second.second) (void)hipGraphExecDestroy(kv.second.second);
                if (kv.second.first) (void)hipGraphDestroy(kv.second.first);
            }
            e->cb_graphs.clear();
            fprintf(stderr, "[starvector_amd] a fused decode launch gave up waiting (code %d): this GPU is shared -- the engine continues WITHOUT the all-blocks-resident "
                            "launches (same tokens, ~8 %% more time per step); create it with exclusive_device = 0 to start that way\n", what);
        }
    }
    if (what == 3 || what == 4)
        return fail(SV_EHIP, "%s: a block of a fused decode launch (code %d: 3 = MLP pair, 4 = row update + c_attn) gave up waiting for its producers (its blocks were not all resident at "
                             "once?  another process or engine on this GPU?); the tokens of this call are void -- create the engine with "
                             "exclusive_device = 0 (SV_EXP bit 512) there", who, what);
    {
        int32_t stepv = -1;
        (void)hipMemcpy(&stepv, e->d_step, sizeof(stepv), hipMemcpyDeviceToHost);
        return fail(SV_EHIP, "%s: a row of logits had no finite value (NaN / Inf in the weights or inputs?) [code %d, device step %d]", who, what, stepv);
    }
}

static int generate_attempt(sv_engine* e, const void* dev_embeds, int32_t B, int32_t S0, const sv_sampling* sp,
                            int64_t* dev_out_tokens, int32_t* n_generated, sv_stream stream);

// sv_config.exclusive_device = 2: OPTIMISTIC ownership.  The fused launches run as if the engine owned its GPU; if one gives up (another process
// holds CUs: the bounded wait of rowops.hip / gemm.hip, never a hang, never tokens) the engine switches them off for good (report_bad_logits) and
// THIS call is run again from its prompt pass.  The kernels with and without the fused launches are bit-identical (tests/test_gpu_e2e.py), the
// sampler's random stream is a function of (seed, step, row): the second attempt re-derives exactly the columns the first one had already handed
// to a streaming callback, and skips them.
extern "C" int sv_generate(sv_engine* e, const void* dev_embeds, int32_t B, int32_t S0, const sv_sampling* sp,
                           int64_t* dev_out_tokens, int32_t* n_generated, sv_stream stream) {
    if (e) { e->last_giveup = 0; e->stream_skip = 0; }
    int rc = generate_attempt(e, dev_embeds, B, S0, sp, dev_out_tokens, n_generated, stream);
    if (rc != 0 && e && e->cfg.exclusive_device == 2 && e->last_giveup && e->fused_off) {
        e->last_giveup = 0;
        rc = generate_attempt(e, dev_embeds, B, S0, sp, dev_out_tokens, n_generated, stream);
    }
    if (e) e->stream_skip = 0;
    return rc;
}

static int generate_attempt(sv_engine* e, const void* dev_embeds, int32_t B, int32_t S0, const sv_sampling* sp,
                            int64_t* dev_out_tokens, int32_t* n_generated, sv_stream stream) {
    SVCHECK(check_ready(e));
    if (!sp || !dev_out_tokens || !n_generated) return fail(SV_EINVAL, "sv_generate: null argument");
    const int max_new = sp->max_length - S0;     // HF: with inputs_embeds, max_length includes the prompt
    if (max_new <= 0) return fail(SV_EINVAL, "max_length (%d) must exceed the prompt length (%d)", sp->max_length, S0);
    if (S0 + max_new > e->cfg.max_seq_len) return fail(SV_EINVAL, "max_length %d exceeds engine max_seq_len %d", sp->max_length, e->cfg.max_seq_len);
    if (sp->n_stop < 0 || sp->n_stop > 16) return fail(SV_EINVAL, "stop sequence length %d unsupported (0..16)", sp->n_stop);
    if (sp->do_sample && !(sp->temperature > 0.f && sp->top_p > 0.f)) return fail(SV_EINVAL, "temperature and top_p must be > 0");
    if (sp->num_beams < 0) return fail(SV_EINVAL, "num_beams must be >= 1");
    std::lock_guard<std::mutex> lk(e->mu);
    HIPCHECK(hipSetDevice(e->cfg.device));
    // order the engine stream after everything already queued on the caller's stream
    HIPCHECK(hipEventRecord(e->gen_event, (hipStream_t)stream));
    hipStream_t st = e->gen_stream;
    HIPCHECK(hipStreamWaitEvent(st, e->gen_event, 0));

    SVCHECK(cb_guard(e, "sv_generate"));
    if (sp->num_beams > 1) {
        if (sp->on_tokens) return fail(SV_EINVAL, "streaming is not supported with beam search (hypotheses are only final at the end; HF refuses too)");
        if (sp->n_stop > 0 && !sp->stop_ids) return fail(SV_EINVAL, "n_stop > 0 but stop_ids is null");
        return generate_beam(e, dev_embeds, B, S0, sp, max_new, dev_out_tokens, n_generated, st);
    }

    auto t0 = std::chrono::steady_clock::now();
    HIPCHECK(hipMemsetAsync(e->d_bad, 0, sizeof(int32_t), st));         // a flag left by an earlier, failed call is not this call's
    SVCHECK(prefill_locked(e, dev_embeds, B, S0, S0 + max_new, st, false));
    // Plain greedy decode (no repetition penalty, no min_length hold, one row tile, bf16 lm_head with the K split over the waves of a
    // block): the selection rides in the lm_head epilogue of every decode step -- one launch less per step, same tokens bit for bit.
    // SV_EXP bit 1024 = the separate argmax launch (A/B).
    bool fused_sel = false;
    {
        int waves = 1, two = 0;
        skinny_plan(e->lm_head.Npad, e->lm_head.Kpad, 1, e->lm_head.fp8 ? 1 : 0, (B + 31) / 32, &waves, &two);
        const bool pen = sp->repetition_penalty > 0.f && sp->repetition_penalty != 1.0f;
        fused_sel = !sp->do_sample && !pen && sp->min_new_tokens <= 0 && B <= 32 && !e->lm_head.fp8 && waves > 1 && !two && !(e->exp & 1024);
    }
    // generation state, one launch: positions = S0 - 1 (finish_step adds 1), unfinished = 1, {step, done, n_emitted} = 0, the folded selection's key slots = 0
    gen_state_init(e->positions, S0 - 1, e->unfinished, B, e->d_step, e->amax, fused_sel ? 64 * SV_AMAX_STRIDE : 0, st);
    if (sp->repetition_penalty > 0.f && sp->repetition_penalty != 1.0f)
        HIPCHECK(hipMemsetAsync(e->seen, 0, (size_t)((B + 31) / 32) * 32 * e->seen_words * sizeof(uint32_t), st));
    if (sp->n_stop > 0) {
        if (!sp->stop_ids) return fail(SV_EINVAL, "n_stop > 0 but stop_ids is null");
        HIPCHECK(hipMemcpyAsync(e->d_stop, sp->stop_ids, sp->n_stop * sizeof(int32_t), hipMemcpyHostToDevice, st));
    }
    sample_and_finish(e, B, *sp, max_new, st);       // first token from the prefill logits
    struct FusedSel { sv_engine* e; ~FusedSel() { e->greedy_fused = false; e->fin_fold = false; e->fin_folded = false; } } fused_guard{e};
    if (fused_sel) {
        e->greedy_fused = true;                      // read by decode_forward (capture and eager launches below); cleared on every exit
        const char* ff = getenv("SV_FINISH_FOLD");   // 0 = the bookkeeping as its own launch (A/B)
        e->fin_fold = !(ff && atoi(ff) == 0);
        e->fin_args = make_finish_args(e, B, *sp, max_new, true);
    }
    // ONE copy of the device's {step, done, n_emitted, bad} block: a call that is over after its first token (budget 1, EOS) has everything
    // it needs from this round trip and takes no other before the tokens go out
    HIPCHECK(hipMemcpyAsync(&e->h_flags[8], e->d_step, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIPCHECK(hipStreamSynchronize(st));
    e->h_flags[0] = e->h_flags[9];
    auto t1 = std::chrono::steady_clock::now();

    int steps = 0;
    const int chunk = sp->sync_every > 0 ? sp->sync_every : 32;
    // streaming: columns [0, steps] are final after every poll (a finished batch may have fewer: n_emitted caps it)
    int streamed = e->stream_skip;             // (an optimistic call's second attempt: the columns its first attempt delivered)
    struct SkipSave { sv_engine* e; int* s; ~SkipSave() { e->stream_skip = *s; } } skip_save{e, &streamed};
    std::vector<int32_t> stream_buf;
    auto stream_upto = [&](int n_cols_final) -> int {
        if (!sp->on_tokens || n_cols_final <= streamed) return 0;
        const int n = n_cols_final - streamed;
        stream_buf.resize((size_t)B * n);
        HIPCHECK(hipMemcpy2DAsync(stream_buf.data(), (size_t)n * sizeof(int32_t), e->out_tok + streamed,
                                  (size_t)e->out_ld * sizeof(int32_t), (size_t)n * sizeof(int32_t), B, hipMemcpyDeviceToHost, st));
        HIPCHECK(hipStreamSynchronize(st));
        sp->on_tokens(sp->user_data, stream_buf.data(), B, streamed, n);
        streamed = n_cols_final;
        return 0;
    };
    const bool use_graph = getenv("SV_NO_GRAPH") == nullptr;
    hipGraphExec_t gexec = nullptr;
    auto drop_multi = [&]() {
        if (e->gen_gexec_multi) { (void)hipGraphExecDestroy(e->gen_gexec_multi); e->gen_gexec_multi = nullptr; }
        if (e->gen_graph_multi) { (void)hipGraphDestroy(e->gen_graph_multi); e->gen_graph_multi = nullptr; }
        e->gen_multi_steps = 0;
    };
    if (!e->h_flags[0] && use_graph) {
        // One decode step is captured as a hipGraph (all kernel arguments are stable device pointers; the step index,
        // positions and stop state live in device memory) and replayed every step.  The instantiated graph is KEPT on the
        // engine and reused by the next call whose batch, budget and sampling parameters are the same (a serving request
        // stream, the benchmark), so a short request does not pay a 172-node capture + instantiate; a call with other
        // parameters replaces it.  Owned by the engine: no early return below can leak it.
        char key[256];
        snprintf(key, sizeof(key), "B%d|n%d|ds%d|T%a|p%a|k%d|seed%llu|eos%d|pad%d|ns%d|rp%a|mn%d|x%d|fo%d|ff%d", B, max_new, sp->do_sample,
                 sp->temperature, sp->top_p, sp->top_k, (unsigned long long)sp->seed, sp->eos_token_id, sp->pad_token_id, sp->n_stop,
                 sp->repetition_penalty, sp->min_new_tokens, e->exp, e->fused_off ? 1 : 0, e->fin_fold ? 1 : 0);
        if (e->gen_gexec && e->gen_graph_key == key) {
            gexec = e->gen_gexec;
        } else {
            if (e->gen_gexec) { (void)hipGraphExecDestroy(e->gen_gexec); e->gen_gexec = nullptr; }
            if (e->gen_graph) { (void)hipGraphDestroy(e->gen_graph); e->gen_graph = nullptr; }
            drop_multi();
            e->gen_graph_key.clear();
            hipError_t ce = hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed);
            if (ce == hipSuccess) {
                decode_forward(e, B, st);
                sample_and_finish(e, B, *sp, max_new, st, fused_sel);
                ce = hipStreamEndCapture(st, &e->gen_graph);
                if (ce == hipSuccess && e->gen_graph) ce = hipGraphInstantiate(&e->gen_gexec, e->gen_graph, nullptr, nullptr, 0);
            }
            if (ce != hipSuccess) {          // fall back to plain launches of the SAME kernels
                (void)hipGetLastError();
                if (e->gen_gexec) { (void)hipGraphExecDestroy(e->gen_gexec); e->gen_gexec = nullptr; }
                if (e->gen_graph) { (void)hipGraphDestroy(e->gen_graph); e->gen_graph = nullptr; }
                if (getenv("SV_REQUIRE_GRAPH")) return fail(SV_EHIP, "hipGraph capture failed: %s", hipGetErrorString(ce));
            } else {
                e->gen_graph_key = key;
                gexec = e->gen_gexec;
                size_t nn = 0;                     // sv_debug_step_plan: the step's launches, counted on the captured graph itself
                e->step_nodes = 0;
                if (hipGraphGetNodes(e->gen_graph, nullptr, &nn) == hipSuccess && nn > 0) {
                    std::vector<hipGraphNode_t> nodes(nn);
                    if (hipGraphGetNodes(e->gen_graph, nodes.data(), &nn) == hipSuccess)
                        for (size_t i = 0; i < nn; ++i) {
                            hipGraphNodeType ty;
                            if (hipGraphNodeGetType(nodes[i], &ty) == hipSuccess && ty == hipGraphNodeTypeKernel) ++e->step_nodes;
                        }
                }
            }
        }
    }
    // Several steps per graph launch.  Inside a replayed graph a kernel follows its predecessor with no measurable gap; from one hipGraphLaunch to
    // the next the GPU idles 8.6 us (rocprofv3 kernel trace of the bench: finish_step_kernel of step i -> the first kernel of step i + 1,
    // profiles/step_gaps_r06.log) -- 0.9 % of StarVector-1B's step.  So the SAME step is captured U times into a second graph (every kernel argument
    // is a stable device pointer and the step index lives on the device: U copies of the node list ARE U consecutive steps) and a chunk of the loop
    // below launches it while at least U steps of the chunk are left, the one-step graph for the rest.  U = the polling chunk (sync_every, at most
    // SV_GRAPH_STEPS = 32 by default; 1 = off): nothing runs that did not run before -- the loop already issues a whole chunk before it looks at the
    // done flag.  Built only for calls with at least 4 U steps in front of them (instantiating 32 x 99 nodes costs milliseconds), kept on the
    // engine with the one-step graph (same key).
    hipGraphExec_t gexec_multi = nullptr;
    int U = 0;
    if (gexec) {
        static const int cap = getenv("SV_GRAPH_STEPS") ? atoi(getenv("SV_GRAPH_STEPS")) : 32;
        U = chunk < cap ? chunk : cap;
        if (U >= 2 && max_new - 1 >= 4 * U) {
            if (e->gen_gexec_multi && e->gen_multi_steps == U) {
                gexec_multi = e->gen_gexec_multi;
            } else {
                drop_multi();
                hipError_t ce = hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed);
                if (ce == hipSuccess) {
                    for (int u = 0; u < U; ++u) {
                        decode_forward(e, B, st);
                        sample_and_finish(e, B, *sp, max_new, st, fused_sel);
                    }
                    ce = hipStreamEndCapture(st, &e->gen_graph_multi);
                    if (ce == hipSuccess && e->gen_graph_multi) ce = hipGraphInstantiate(&e->gen_gexec_multi, e->gen_graph_multi, nullptr, nullptr, 0);
                }
                if (ce != hipSuccess || !e->gen_gexec_multi) { (void)hipGetLastError(); drop_multi(); }      // the one-step graph carries the call
                else { e->gen_multi_steps = U; gexec_multi = e->gen_gexec_multi; }
            }
        }
    }
    // The host looks at the device's done flag after every chunk -- a round trip that leaves the GPU idle for ~0.1 ms -- only when something can
    // END the call before its budget (an EOS id inside the vocabulary, a stop sequence) or wants the columns as they become final (streaming).
    // A fixed-length call (the benchmark's EOS-disabled workload, fixed-length rollouts) has nothing to poll for.
    const bool poll = (sp->eos_token_id >= 0 && sp->eos_token_id < e->cfg.vocab) || sp->n_stop > 0 || sp->on_tokens != nullptr;
    while (!e->h_flags[0]) {
        int n = max_new - 1 - steps;
        if (n <= 0) break;                       // budget exhausted: the device flag is already set
        if (n > chunk) n = chunk;
        for (int i = 0; i < n;) {
            if (gexec_multi && n - i >= U) {
                HIPCHECK(hipGraphLaunch(gexec_multi, st));
                i += U;
            } else if (gexec) {
                HIPCHECK(hipGraphLaunch(gexec, st));
                ++i;
            } else {
                decode_forward(e, B, st);
                sample_and_finish(e, B, *sp, max_new, st, fused_sel);
                ++i;
            }
        }
        steps += n;
        if (!poll) continue;                     // nothing can end this call before its budget: the next chunk follows without a host round trip
        HIPCHECK(hipMemcpyAsync(&e->h_flags[8], e->d_step, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, st));      // {step, done, n_emitted, bad}
        HIPCHECK(hipStreamSynchronize(st));
        e->h_flags[0] = e->h_flags[9];
        if (e->h_flags[11]) break;               // a void step (give-up, non-finite row): nothing of this chunk is handed on; the error is raised below
        if (!e->h_flags[0]) SVCHECK(stream_upto(steps + 1));      // still running: every column so far is final
    }
    const double gexec_used = gexec_multi ? (double)U : gexec ? 1.0 : 0.0;       // steps per graph launch
    if (steps > 0) {                             // (a call that ended at its first token has the block already)
        HIPCHECK(hipMemcpyAsync(&e->h_flags[8], e->d_step, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        HIPCHECK(hipStreamSynchronize(st));
    }
    e->h_flags[1] = e->h_flags[10];
    SVCHECK(report_bad_logits(e, st, "sv_generate", e->h_flags[11]));          // (before the last columns go out: a void call streams nothing more)
    if (e->h_flags[1] >= 1 && e->h_flags[1] <= max_new) SVCHECK(stream_upto(e->h_flags[1]));
    const int n_emit = e->h_flags[1];
    if (n_emit < 1 || n_emit > max_new) return fail(SV_EHIP, "generation bookkeeping failed (n_emitted=%d)", n_emit);
    tokens_to_i64(e->out_tok, e->out_ld, dev_out_tokens, B, n_emit, max_new, st);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipStreamSynchronize(st));
    auto t2 = std::chrono::steady_clock::now();
    *n_generated = n_emit;
    e->timing[0] = std::chrono::duration<double, std::milli>(t1 - t0).count();
    e->timing[1] = std::chrono::duration<double, std::milli>(t2 - t1).count();
    e->timing[2] = (double)steps;
    e->timing_graph = gexec_used;
    return 0;
}

// ------------------------------------------------------------------------------------------------
// C ABI: the beam scorer on its own (parity tests drive it with synthetic logits)
// ------------------------------------------------------------------------------------------------
struct sv_beam { BeamScorer s; int device = 0; };

extern "C" int sv_beam_create(const sv_beam_config* cfg, sv_beam** out) {
    if (!cfg || !out) return fail(SV_EINVAL, "sv_beam_create: null argument");
    if (cfg->batch < 1 || cfg->num_beams < 1 || cfg->num_beams > BM_MAXNB || cfg->vocab < 2 * cfg->num_beams || cfg->max_new < 1)
        return fail(SV_EINVAL, "sv_beam_create: bad shape (batch %d, num_beams %d, vocab %d, max_new %d)", cfg->batch,
                    cfg->num_beams, cfg->vocab, cfg->max_new);
    if (cfg->batch > 1024) return fail(SV_EINVAL, "sv_beam_create: batch > 1024");
    if (cfg->n_stop < 0 || cfg->n_stop > BM_MAXSTOP || (cfg->n_stop > 0 && !cfg->stop_ids)) return fail(SV_EINVAL, "sv_beam_create: bad stop sequence");
    if (cfg->early_stopping < 0 || cfg->early_stopping > 2) return fail(SV_EINVAL, "sv_beam_create: early_stopping must be 0, 1 or 2");
    BeamConfig bc;
    bc.B = cfg->batch; bc.nb = cfg->num_beams; bc.V = cfg->vocab; bc.max_new = cfg->max_new; bc.eos = cfg->eos_token_id;
    bc.pad = cfg->pad_token_id; bc.early = cfg->early_stopping; bc.length_penalty = cfg->length_penalty;
    bc.penalty = cfg->repetition_penalty > 0.f ? cfg->repetition_penalty : 1.f; bc.n_stop = cfg->n_stop;
    for (int i = 0; i < cfg->n_stop; ++i) bc.stop[i] = cfg->stop_ids[i];
    if (cfg->do_sample && !(cfg->temperature > 0.f && cfg->top_p > 0.f)) return fail(SV_EINVAL, "sv_beam_create: temperature and top_p must be > 0");
    bc.do_sample = cfg->do_sample ? 1 : 0; bc.temperature = cfg->do_sample ? cfg->temperature : 1.f;
    bc.top_p = cfg->do_sample ? cfg->top_p : 1.f; bc.top_k = cfg->top_k; bc.seed = cfg->seed;
    bc.min_new = cfg->min_new_tokens > 0 ? cfg->min_new_tokens : 0;
    sv_beam* h = new sv_beam();
    int r = h->s.init(bc, nullptr, nullptr, nullptr, nullptr);
    if (!r) r = h->s.reset(nullptr);
    if (r) { h->s.destroy(); delete h; return fail(SV_EHIP, "sv_beam_create: hip error %d", r); }
    *out = h;
    return 0;
}
extern "C" int sv_beam_destroy(sv_beam* h) {
    if (!h) return 0;
    (void)hipDeviceSynchronize();
    h->s.destroy();
    delete h;
    return 0;
}
extern "C" int sv_beam_step(sv_beam* h, const float* dev_logits, int32_t ld, int32_t* done, int32_t* host_parent,
                            int32_t* host_tokens, float* host_scores, sv_stream stream) {
    if (!h || !dev_logits || !done) return fail(SV_EINVAL, "sv_beam_step: null argument");
    if (ld < h->s.c.V) return fail(SV_EINVAL, "sv_beam_step: ld %d < vocab %d", ld, h->s.c.V);
    hipStream_t st = (hipStream_t)stream;
    h->s.enqueue_step(dev_logits, ld, 1, st);
    HIPCHECK(hipGetLastError());
    const size_t R = (size_t)h->s.R;
    HIPCHECK(hipMemcpyAsync(done, h->s.d.done, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    if (host_parent) HIPCHECK(hipMemcpyAsync(host_parent, h->s.d.parent, R * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    if (host_tokens) HIPCHECK(hipMemcpyAsync(host_tokens, h->s.d.cur_tok, R * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    if (host_scores) HIPCHECK(hipMemcpyAsync(host_scores, h->s.d.run_score, R * sizeof(float), hipMemcpyDeviceToHost, st));
    HIPCHECK(hipStreamSynchronize(st));
    return 0;
}
// the (parent beam, token) columns of the last beam-search sv_generate: [n_steps][batch * num_beams] each
extern "C" int sv_beam_history(sv_engine* e, int32_t* host_parent, int32_t* host_tok, int32_t capacity_steps,
                               int32_t* n_steps, int32_t* rows) {
    if (!e || !n_steps || !rows) return fail(SV_EINVAL, "sv_beam_history: null argument");
    std::lock_guard<std::mutex> lk(e->mu);
    if (e->beam.R <= 0) return fail(SV_ESTATE, "sv_beam_history: no beam search has run on this engine");
    HIPCHECK(hipSetDevice(e->cfg.device));
    int32_t n = 0;
    HIPCHECK(hipMemcpy(&n, e->beam.d.step, sizeof(int32_t), hipMemcpyDeviceToHost));
    *n_steps = n;
    *rows = e->beam.R;
    if (!host_parent || !host_tok) return 0;
    if (n > capacity_steps) return fail(SV_EINVAL, "sv_beam_history: %d steps recorded, capacity %d", n, capacity_steps);
    HIPCHECK(hipMemcpy(host_parent, e->beam.d.hist_parent, (size_t)n * e->beam.R * sizeof(int32_t), hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(host_tok, e->beam.d.hist_tok, (size_t)n * e->beam.R * sizeof(int32_t), hipMemcpyDeviceToHost));
    return 0;
}
extern "C" int sv_beam_finalize(sv_beam* h, int64_t* host_tokens, int32_t* n_generated, float* host_scores, sv_stream stream) {
    if (!h || !host_tokens || !n_generated) return fail(SV_EINVAL, "sv_beam_finalize: null argument");
    std::vector<int64_t> toks;
    std::vector<float> sc;
    int L = 0;
    int r = h->s.finalize((hipStream_t)stream, toks, L, sc);
    if (r) return fail(SV_EHIP, "sv_beam_finalize: bookkeeping failed (code %d)", r);
    memcpy(host_tokens, toks.data(), toks.size() * sizeof(int64_t));
    if (host_scores) memcpy(host_scores, sc.data(), sc.size() * sizeof(float));
    *n_generated = L;
    return 0;
}
