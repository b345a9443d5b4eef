// Host driver, part 2 of 5: the op graphs -- vision tower, adapter, prompt pass, one decode step -- and the forward entry
// points of the C ABI (sv_encode_image, sv_adapter, sv_embed_tokens, sv_prefill, sv_forward_logits, sv_decode_step).
#include "engine_internal.h"

namespace sveng {
// record an event in front of the next launch (tagged with its kind); durations = event deltas
void prof_mark(sv_engine* e, int kind, hipStream_t st) {
    if (!e->prof_on) return;
    if (e->prof_used == e->prof_ev.size()) {
        hipEvent_t ev;
        if (hipEventCreate(&ev) != hipSuccess) return;
        e->prof_ev.push_back(ev);
        e->prof_kind.push_back(kind);
    }
    e->prof_kind[e->prof_used] = kind;
    (void)hipEventRecord(e->prof_ev[e->prof_used++], st);
}
}  // namespace sveng

// ------------------------------------------------------------------------------------------------
// launch helpers
// ------------------------------------------------------------------------------------------------
// `kind`: the stage this GEMM belongs to in a time-to-first-token profile (sv_profile_ttft; prof_mark is a no-op otherwise); the
// remainder-row launch of a peeled GEMM is marked PK_GEMM_TAIL through the launcher's hook
struct TailMarkCtx { sv_engine* e; };
static void tail_mark_cb(void* c, hipStream_t st) { prof_mark(static_cast<TailMarkCtx*>(c)->e, PK_GEMM_TAIL, st); }
// seq_rows: the M rows are whole sequences of that many rows (vision tokens of an image, prompt rows of a request): where gemm_seq_form holds
// the rows a sequence leaves over its 256-row tiles go through the split-K remainder kernel (gemm.hip); last_rows: the M rows are the LAST
// rows of such sequences (compact) and take the kernel they take inside the full problem
static void gemm(sv_engine* e, int kind, const bf16_t* A, int lda, const Linear& l, const bf16_t* R, int ldr, void* C, int ldc, int M,
                 int act, int out_f32, hipStream_t st, int seq_rows = 0, int last_rows = 0) {
    TailMarkCtx tc{e};
    prof_mark(e, kind, st);
    GemmArgs g;
    if (e->prof_on) { g.tail_mark = tail_mark_cb; g.tail_ctx = &tc; }
    g.A = A; g.lda = lda; g.Wp = l.Wp; g.bias = l.bias; g.R = R; g.ldr = ldr; g.C = C; g.ldc = ldc;
    g.M = M; g.N = l.N; g.K = l.Kpad; g.act = act; g.out_f32 = out_f32;
    g.cscale = l.fp8 ? l.wscale : nullptr;
    g.seq_rows = (e->exp & 4194304) ? 0 : seq_rows;          // SV_EXP bit 4194304: A/B, the batch-level remainder of rounds 2-5
    g.splitk_rows = (e->exp & 4194304) ? 0 : last_rows;
    launch_gemm(g, st);
}

// Context splits of the decode attention: a constant of the engine (sized for the engine's max_batch), NOT of the batch of the
// call, so that a sequence's partial results are merged in the same grouping whatever shares the batch with it: a row is
// bit-identical alone, inside a batch and inside a continuous batch at any context length.
int sveng::attn_max_splits_of(int max_batch, int nkv, int num_cus) {
    const int rows = (max_batch < 32 ? max_batch : 32) * nkv;
    const int ms = num_cus / (rows < 1 ? 1 : rows);
    return ms < 1 ? 1 : (ms > 8 ? 8 : ms);
}
static int attn_max_splits(const sv_engine* e) { return attn_max_splits_of(e->cfg.max_batch, e->nkv, e->num_cus); }

// 32-key groups a block takes before another context split joins.  Where the engine's rows x KV heads already give every CU a block
// (StarVector-8B at 64 rows: 256 (row, KV head) pairs), a context of <= 8 groups stays in ONE block (one group per wave: no partial
// results, no ticket, no merge); otherwise 4 (measured best where the splits are what fills the chip,
// profiles/attention_r03_groups_per_block_ab.log).  A constant of the engine like the split cap (same reason).
int sveng::attn_groups_per_block_of(int max_batch, int nkv, int num_cus) {
    return (max_batch < 64 ? max_batch : 64) * nkv >= num_cus ? 8 : 4;
}
static int attn_groups_per_block(const sv_engine* e) {
    return (e->exp & 64) ? 4 : attn_groups_per_block_of(e->cfg.max_batch, e->nkv, e->num_cus);       // SV_EXP bit 64: A/B, always 4
}

static int vision_forward(sv_engine* e, const bf16_t* img, int B, bf16_t* out, hipStream_t st) {
    const sv_config& c = e->cfg;
    const int Dv = c.vit_width, T = e->T, NP = e->NP, M = B * T, Fv = e->vit_F;
    const float eps = e->v2 ? c.vit_eps : c.ln_eps;
    const int act = e->v2 ? ACT_GELU_TANH : ACT_QUICKGELU;       // SigLIP gelu_pytorch_tanh | CLIP QuickGELU
    prof_mark(e, PK_VIT_ROWS, st);
    launch_im2col(img, e->patches, B, c.image_size, c.patch_size, e->conv1.Kpad, st);
    gemm(e, PK_VIT_GEMM, e->patches, e->conv1.Kpad, e->conv1, nullptr, 0, e->patch_out, Dv, B * NP, ACT_NONE, 0, st);
    prof_mark(e, PK_VIT_ROWS, st);
    if (e->v2)      // SigLIP: patches (+conv bias) + learned positions, no class token, no ln_pre
        launch_dec_embed(e->patch_out, e->pos, e->vx, B, NP, Dv, st);
    else
        launch_vit_embed_lnpre(e->patch_out, Dv, e->cls, e->pos, e->ln_pre.g, e->ln_pre.b, e->vx, B, NP, Dv,
                               c.ln_eps, st);
    AttnPrefillArgs at;
    at.q = e->vqkv; at.k = e->vqkv + Dv; at.v = e->vqkv + 2 * Dv;
    at.q_row_stride = 3 * Dv; at.kv_row_stride = 3 * Dv; at.q_head_stride = e->vdh; at.kv_head_stride = e->vdh;
    at.o = e->vattn; at.o_row_stride = Dv; at.B = B; at.S = T; at.H = c.vit_heads; at.head_dim = e->vdh;
    at.kv_group = 1; at.causal = 0; at.scale = 1.0f / sqrtf((float)e->vdh);
    for (int i = 0; i < c.vit_layers; ++i) {
        VitLayer& L = e->vit[i];
        prof_mark(e, PK_VIT_ROWS, st);
        launch_layernorm_rows(e->vx, Dv, L.ln1.g, L.ln1.b, e->vln, Dv, M, Dv, eps, st);
        gemm(e, PK_VIT_GEMM, e->vln, Dv, L.in_proj, nullptr, 0, e->vqkv, 3 * Dv, M, ACT_NONE, 0, st, T);
        prof_mark(e, PK_VIT_ATTN, st);
        launch_attn_prefill(at, st);
        gemm(e, PK_VIT_GEMM, e->vattn, Dv, L.out_proj, e->vx, Dv, e->vx, Dv, M, ACT_NONE, 0, st, T);
        prof_mark(e, PK_VIT_ROWS, st);
        launch_layernorm_rows(e->vx, Dv, L.ln2.g, L.ln2.b, e->vln, Dv, M, Dv, eps, st);
        gemm(e, PK_VIT_GEMM, e->vln, Dv, L.c_fc, nullptr, 0, e->vmlp, Fv, M, act, 0, st, T);
        gemm(e, PK_VIT_GEMM, e->vmlp, Fv, L.c_proj, e->vx, Dv, e->vx, Dv, M, ACT_NONE, 0, st, T);
    }
    prof_mark(e, PK_VIT_ROWS, st);
    launch_layernorm_rows(e->vx, Dv, e->ln_vision.g, e->ln_vision.b, out, Dv, M, Dv, eps, st);
    return 0;
}

// out_batch_stride (elements; 0 = T * D): image b's T visual rows go to out + b * stride, e.g. the head of its [S0][D] prompt block
static int adapter_forward(sv_engine* e, const bf16_t* in, int B, bf16_t* out, hipStream_t st, size_t out_batch_stride = 0) {
    const sv_config& c = e->cfg;
    const int Dv = c.vit_width, D = c.hidden, T = e->T, M = B * T;
    gemm(e, PK_AD_GEMM, in, Dv, e->ad_fc, nullptr, 0, e->a1, 2 * Dv, M, ACT_SWISH, 0, st, T);
    gemm(e, PK_AD_GEMM, e->a1, 2 * Dv, e->ad_proj, nullptr, 0, e->a2, D, M, ACT_NONE, 0, st, T);
    prof_mark(e, PK_AD_NORM, st);
    if (c.adapter_norm == SV_NORM_LAYER)
        launch_plane_layernorm(e->a2, e->ad_w, e->ad_b, out, B, T * D, c.ln_eps, st, out_batch_stride);
    else
        launch_token_batchnorm(e->a2, e->ad_w, e->ad_b, e->ad_rm, e->ad_rv, out, B, T, D, c.ln_eps, st, out_batch_stride);
    return 0;
}

static int ensure_prefill_ws(sv_engine* e, size_t rows) {
    if (rows <= e->pf_rows) return 0;
    const sv_config& c = e->cfg;
    const int D = c.hidden;
    // previous buffers stay in e->allocs (freed at destroy); growth is rare (max_batch * S0)
    SVCHECK(dalloc(e, &e->ph, rows * D));
    SVCHECK(dalloc(e, &e->pln, rows * D));
    SVCHECK(dalloc(e, &e->pqkv, rows * (size_t)e->QKV));
    SVCHECK(dalloc(e, &e->pattn, rows * D));
    SVCHECK(dalloc(e, &e->pmlp, rows * c.n_inner));
    e->pf_rows = rows;
    return 0;
}

int sveng::assign_pages(sv_engine* e, int B, int total_len, hipStream_t st) {
    // page allocator: hand every sequence the pages for its whole budget up front (the decode loop
    // runs without host round-trips, so pages cannot be added mid-flight)
    const int need = (total_len + SV_PAGE_TOKENS - 1) / SV_PAGE_TOKENS;
    if (need > e->pages_per_seq) return fail(SV_EINVAL, "sequence length %d exceeds max_seq_len %d", total_len, e->cfg.max_seq_len);
    e->free_pages.clear();
    for (int p = e->num_pages - 1; p >= 0; --p) e->free_pages.push_back(p);
    // The table goes up from a PINNED host image without a stream synchronise (the pageable copy + hipStreamSynchronize of rounds 1-5 drained the
    // stream in front of every prompt pass: ~25 us of idle GPU on the way to the first token); the image is rewritten only after the upload
    // before it has completed (an event, normally long past).
    if (e->table_pending) { HIPCHECK(hipEventSynchronize(e->table_ev)); e->table_pending = false; }
    const size_t n = (size_t)e->cfg.max_batch * e->pages_per_seq;
    int32_t* table = e->h_table;
    memset(table, 0, n * sizeof(int32_t));
    for (int b = 0; b < B; ++b)
        for (int i = 0; i < need; ++i) {
            table[(size_t)b * e->pages_per_seq + i] = e->free_pages.back();
            e->free_pages.pop_back();
        }
    HIPCHECK(hipMemcpyAsync(e->block_table, table, n * sizeof(int32_t), hipMemcpyHostToDevice, st));
    HIPCHECK(hipEventRecord(e->table_ev, st));
    e->table_pending = true;
    return 0;
}

// lm_head -> e->logits; xp holds ln_f(h) in fragment order
// The fused row-update + c_attn launch (rowops.hip) is on for this engine: then xp_a -- the LayerNorm output buffer its GEMM blocks poll --
// is touched ONLY with write-through stores / L1-bypassing loads (row role, the attention launch's and the lm_head launch's pattern
// stores, the GEMM role's polls), and ln_f's output lives in xp_f.  Invariant: behind every lm_head launch xp_a carries the pattern.
// (First form: a memset node at the head of the step + the prompt pass's ln_f through xp_a with plain stores -- NaN logits in the third
//  sv_generate call of bench.py: a stale line of the buffer in some XCD's L2.)
static bool rc_enabled(const sv_engine* e) {
    // the 6-launch layer (fold6: residual stream in fragment order) or the 7-launch layer of the wide model (row-major residual stream)
    const bool layer_ok = e->fold6 ? (e->fold_ready && !(e->exp & 2)) : e->cfg.hidden > 2048;
    return layer_ok && e->rc_fused_ok && !(e->exp & 8192) && (e->cfg.exclusive_device || (e->exp & 16384)) && !e->fused_off;
}
// the lm_head launch stores the pattern 16 bytes per thread from its first blocks: its grid must cover the buffer (a tiny-vocabulary
// configuration would arm only part of it: then layer 0 of the next step takes the two launches)
static bool lm_head_covers(const sv_engine* e, unsigned bytes) {
    return (size_t)(e->lm_head.Npad / 32 / 3) * 128 * 16 >= bytes;      // at least: three column tiles per block, two waves
}
static void lm_head_logits(sv_engine* e, int MT, const bf16_t* xp, hipStream_t st) {
    SkinnyArgs a;
    memset(&a, 0, sizeof(a));
    if (rc_enabled(e) && MT == 1 && xp != e->xp_a) { a.poison = e->xp_a; a.poison_bytes = (unsigned)((size_t)(e->cfg.hidden / 16) * 1024); }
    e->xpa_armed = a.poison != nullptr && lm_head_covers(e, a.poison_bytes);
    a.xp = xp; a.Wp = e->lm_head.Wp; a.Wq = e->lm_head.Wq; a.wscale = e->lm_head.wscale; a.MT = MT; a.Npad = e->lm_head.Npad; a.K = e->lm_head.Kpad;
    a.splitk = 1; a.out_mode = SK_OUT_F32; a.out_f32 = e->logits; a.ldo = e->Vpad; a.round_bf16 = 1;
    a.N = e->lm_head.N;
    a.col_tiles = e->lm_head.col_tiles;      // the engine's own plan: sv_debug_set_col_tiles never reaches an engine launch
    launch_gemm_skinny(a, st);
}

int sveng::prefill_forward(sv_engine* e, const bf16_t* embeds, int B, int S0, hipStream_t st, int n_keep,
                           bf16_t* dev_scores, const int32_t* table) {
    if (!table) table = e->block_table;           // continuous batching prefills NEW requests through a table of their slots' pages
    const sv_config& c = e->cfg;
    const int D = c.hidden, dh = e->dh, F = c.n_inner, M = B * S0, QKV = e->QKV, nkv = e->nkv;
    const int QD = c.n_head * dh;                      // width of the query block (= D for both model families)
    SVCHECK(ensure_prefill_ws(e, (size_t)M));
    prof_mark(e, PK_PF_ROWS, st);
    if (e->v2)      // StarCoder2: no learned positions (rotary), hidden = inputs_embeds
        HIPCHECK(hipMemcpyAsync(e->ph, embeds, (size_t)M * D * sizeof(bf16_t), hipMemcpyDeviceToDevice, st));
    else
        launch_dec_embed(embeds, e->wpe, e->ph, B, S0, D, st);
    AttnPrefillArgs at;
    at.q = e->pqkv; at.k = e->pqkv + QD; at.v = e->pqkv + QD + nkv * dh;
    at.q_row_stride = QKV; at.kv_row_stride = QKV; at.q_head_stride = dh; at.kv_head_stride = nkv > 1 ? dh : 0;
    at.o = e->pattn; at.o_row_stride = QD; at.B = B; at.S = S0; at.H = c.n_head; at.head_dim = dh;
    at.kv_group = c.n_head / nkv; at.causal = 1; at.scale = 1.0f / sqrtf((float)dh);
    at.window = c.sliding_window > 0 ? c.sliding_window : 0;      // StarCoder2: also inside the prompt pass (prompts longer than the window)
    // The LAST layer of a generation prompt pass (n_keep == 0) is only needed for (a) its K / V rows and (b) the residual stream of
    // the last prompt row, which alone feeds ln_f + lm_head.  So after its attention only the last row of every sequence goes on:
    // attention over the last query tile, then c_proj / ln_2 / c_fc / down-proj on B rows instead of B * S0 (HF computes all rows and
    // reads logits[:, -1]; a row's GEMM / attention result does not depend on the rows around it: bit-identical, test_gpu_e2e.py).
    // SV_EXP bit 32768 = off (A/B).
    const bool prune_last = n_keep == 0 && S0 >= 2 && !(e->exp & 32768);
    // the last prompt row may be one of the rows its sequence leaves over its 256-row tiles: the launcher gives it the kernel it takes in the full layers
    for (int i = 0; i < c.n_layer; ++i) {
        DecLayer& L = e->dec[i];
        prof_mark(e, PK_PF_ROWS, st);
        launch_layernorm_rows(e->ph, D, L.ln1.g, L.ln1.b, e->pln, D, M, D, c.ln_eps, st);
        gemm(e, PK_PF_GEMM, e->pln, D, L.c_attn, nullptr, 0, e->pqkv, QKV, M, ACT_NONE, 0, st, S0);
        prof_mark(e, PK_PF_ATTN, st);                      // RoPE + KV scatter + flash attention
        if (e->v2) launch_rope_prefill(e->pqkv, QKV, M, S0, c.n_head + nkv, dh, e->rope_cos, e->rope_sin, st);
        for (int kh = 0; kh < nkv; ++kh)
            launch_kv_write_prefill(e->pqkv, QKV, QD + kh * dh, QD + nkv * dh + kh * dh,
                                    e->kv_pool + (size_t)i * e->layer_stride + (size_t)kh * e->kv_head_stride,
                                    table, e->pages_per_seq, B, S0, dh, st);
        if (prune_last && i + 1 == c.n_layer) {
            at.last_rows = 1;
            launch_attn_prefill(at, st);
            at.last_rows = 0;
            // compact buffers inside the (now free) LayerNorm workspace: [B][QD] attention rows | [B][D] ln_2 rows; the MLP rows in pmlp
            bf16_t* pa_l = e->pln;
            bf16_t* ln_l = e->pln + (size_t)B * QD;
            prof_mark(e, PK_PF_ROWS, st);
            launch_gather_last_rows(e->pattn, pa_l, B, S0, QD, st);
            launch_gather_last_rows(e->ph, e->hl, B, S0, D, st);
            gemm(e, PK_PF_GEMM, pa_l, QD, L.c_proj, e->hl, D, e->hl, D, B, ACT_NONE, 0, st, S0, 1);
            prof_mark(e, PK_PF_ROWS, st);
            launch_layernorm_rows(e->hl, D, L.ln2.g, L.ln2.b, ln_l, D, B, D, c.ln_eps, st);
            gemm(e, PK_PF_GEMM, ln_l, D, L.c_fc, nullptr, 0, e->pmlp, F, B, ACT_GELU_TANH, 0, st, S0, 1);
            gemm(e, PK_PF_GEMM, e->pmlp, F, L.c_proj2, e->hl, D, e->hl, D, B, ACT_NONE, 0, st, S0, 1);
            prof_mark(e, PK_PF_ROWS, st);
            launch_layernorm_rows_packed(e->hl, D, e->ln_f.g, e->ln_f.b, e->xp_f, B, D, c.ln_eps, st);
            prof_mark(e, PK_PF_LMHEAD, st);
            lm_head_logits(e, (B + 31) / 32, e->xp_f, st);
            return 0;
        }
        launch_attn_prefill(at, st);
        gemm(e, PK_PF_GEMM, e->pattn, QD, L.c_proj, e->ph, D, e->ph, D, M, ACT_NONE, 0, st, S0);
        prof_mark(e, PK_PF_ROWS, st);
        launch_layernorm_rows(e->ph, D, L.ln2.g, L.ln2.b, e->pln, D, M, D, c.ln_eps, st);
        gemm(e, PK_PF_GEMM, e->pln, D, L.c_fc, nullptr, 0, e->pmlp, F, M, ACT_GELU_TANH, 0, st, S0);
        gemm(e, PK_PF_GEMM, e->pmlp, F, L.c_proj2, e->ph, D, e->ph, D, M, ACT_NONE, 0, st, S0);
    }
    if (n_keep > 0) {
        // scoring forward (starvector_arch.py:161-184): ln_f + lm_head over the last n_keep rows of every sequence, as one
        // big-M GEMM; bf16 logits like the reference's bf16 lm_head.  The GEMM writes rows of Vpad columns (the packed
        // weight's padding), the caller's tensor has `vocab` columns.
        const size_t rows = (size_t)B * n_keep;
        const size_t need = rows * (size_t)D * 2 + rows * (size_t)e->Vpad;      // [rows][D] hidden, [rows][D] ln_f, [rows][Vpad]
        if (need > e->score_elems) {
            if (e->score_ws) (void)hipFree(e->score_ws);
            e->score_ws = nullptr; e->score_elems = 0;
            HIPCHECK(hipMalloc(reinterpret_cast<void**>(&e->score_ws), need * sizeof(bf16_t)));
            e->score_elems = need;
        }
        bf16_t* hk = e->score_ws;
        bf16_t* hn = hk + rows * D;
        bf16_t* lg = hn + rows * D;
        prof_mark(e, PK_PF_ROWS, st);
        launch_gather_tail_rows(e->ph, hk, B, S0, n_keep, D, st);
        launch_layernorm_rows(hk, D, e->ln_f.g, e->ln_f.b, hn, D, (int)rows, D, c.ln_eps, st);
        gemm(e, PK_PF_GEMM, hn, D, e->lm_head, nullptr, 0, lg, e->Vpad, (int)rows, ACT_NONE, 0, st);
        HIPCHECK(hipMemcpy2DAsync(dev_scores, (size_t)c.vocab * sizeof(bf16_t), lg, (size_t)e->Vpad * sizeof(bf16_t),
                                  (size_t)c.vocab * sizeof(bf16_t), rows, hipMemcpyDeviceToDevice, st));
    }
    // only the last prompt row feeds ln_f + lm_head (HF computes all rows; same result)
    prof_mark(e, PK_PF_ROWS, st);
    launch_gather_last_rows(e->ph, e->hl, B, S0, D, st);
    launch_layernorm_rows_packed(e->hl, D, e->ln_f.g, e->ln_f.b, e->xp_f, B, D, c.ln_eps, st);
    prof_mark(e, PK_PF_LMHEAD, st);
    lm_head_logits(e, (B + 31) / 32, e->xp_f, st);
    return 0;
}

// Arguments of the decode attention of layer `layer` over the engine's KV pool / block table / positions: the c_attn output
// arrives as `splitk` fp32 slabs `ws` (+ bias, summed by the kernel), the result goes to `out_xp` in fragment order.
void sveng::attn_decode_args(sv_engine* e, int layer, int B, const float* ws, int splitk, const bf16_t* bias, bf16_t* out_xp,
                             AttnDecodeArgs& ad) {
    const sv_config& c = e->cfg;
    const int MT = (B + 31) / 32;
    memset(&ad, 0, sizeof(ad));
    ad.ws = ws; ad.splitk = splitk; ad.ldws = e->ldws; ad.rows_ws = MT * 32; ad.bias = bias;
    ad.pool_layer = e->kv_pool + (size_t)layer * e->layer_stride; ad.block_table = e->block_table;
    ad.max_pages = e->pages_per_seq; ad.positions = e->positions; ad.out_xp = out_xp; ad.out_KS = c.n_head * e->dh / 16;
    ad.window = c.sliding_window;
    ad.B = B; ad.H = c.n_head; ad.head_dim = e->dh; ad.scale = 1.0f / sqrtf((float)e->dh);
    ad.part = e->attn_part; ad.counters = e->attn_cnt;
    ad.max_splits = attn_max_splits(e);
    ad.groups_per_block = attn_groups_per_block(e);
    ad.n_kv = e->nkv; ad.kv_head_stride = e->kv_head_stride; ad.rope_cos = e->rope_cos; ad.rope_sin = e->rope_sin;
    ad.merge_all = (e->exp & 2097152) ? 1 : 0;
}

// One autoregressive step: consumes cur_tok / positions, leaves logits in e->logits.  6 launches per layer + 2 (bf16 weights,
// <= 32 rows):
//   row update (embedding | + bias + residual of the previous down-proj, LN1) | c_attn -> fp32 slabs | attention (sums the
//   slabs, + bias) | c_proj over the whole K: h += ..., partial row statistics | c_fc on the raw h (ln_2 folded, bias + GELU
//   epilogue) | down-proj -> slabs ... | row update (ln_f) | lm_head
// fp8 weights / more than one row tile: 7 launches per layer (c_proj -> slabs | row update (+ bias, + residual, LN2) | c_fc).
void sveng::decode_forward(sv_engine* e, int B, hipStream_t st) {
    const sv_config& c = e->cfg;
    const int D = c.hidden, dh = e->dh, F = c.n_inner, MT = (B + 31) / 32;
    // slab double buffer: A = c_attn slabs (read by attention); B = c_proj / down-proj slabs (read by the row update)
    float* wsA = e->ws;
    float* wsB = e->ws2;
    RowUpdateArgs ru;
    memset(&ru, 0, sizeof(ru));
    const bool fold6 = e->fold6 && e->fold_ready && !(e->exp & 2);
    // Row update + c_attn as ONE launch (rowops.hip rowln_cattn_kernel): on when the engine owns its GPU (like the fused MLP launch: its
    // blocks wait for blocks of the same launch); SV_EXP bit 8192 = off, 16384 = on without exclusive_device (A/B).  The GEMM blocks
    // recognise unwritten activations by the 0xFFFF'FFFF pattern: xp_a gets it from the lm_head launch of the step (or prompt pass) before
    // (layer 0) and from the attention launch of layer i for layer i + 1 -- which must have room for it: 16 bytes per thread of its grid
    // (B >= 10 at StarVector-1B's shapes).  Not inside the profiling legs (they time the row updates and the GEMMs apart).  A call that
    // cannot fuse on an engine that does (see rc_enabled) keeps its LayerNorm outputs out of xp_a.
    const size_t attn_threads_bytes = (size_t)B * e->nkv * attn_max_splits(e) * 512 * 16;
    const bool mlp_pattern = fold6 && e->mlp_fused_ok;          // the attention launch may also have to arm the fused MLP launch's buffer
    // (xpa_armed: the launch that arms layer 0's buffer really ran in front of this step -- not after sv_create, sv_debug_set_exp or
    //  sv_debug_kv_load, whose first step takes the two launches and arms the next one)
    // (round 6, third session: a batch below 10 rows has an attention grid too small for the patterns -- B * 8 splits * 512 threads * 16 bytes -- and ran
    //  the two launches, 123 launches per step instead of 99: slower at batch 1 than at batch 16.  On the 6-launch layer the attention output projection,
    //  256 blocks of 1024 threads whatever the batch, arms the buffers instead: ColsArgs::poison / ::poison2)
    const bool attn_room = (size_t)(D / 16) * 1024 + (mlp_pattern ? (size_t)(F / 16) * 1024 : 0u) <= attn_threads_bytes;
    const bool rc = rc_enabled(e) && e->xpa_armed && MT == 1 && !e->only_skinny && !e->skip_skinny && !e->prof_on && (attn_room || fold6);
    const unsigned xpa_bytes = (unsigned)((size_t)(D / 16) * 1024);
    e->step_rc = false; e->step_mlp = false; e->step_sel = e->greedy_fused;
    bf16_t* const xp_ln = (rc_enabled(e) && !rc) ? e->xp_f : e->xp_a;      // LayerNorm(ln_1) output = the c_attn operand

    ru.h = fold6 ? e->h_xp : e->h_dec; ru.ldh = fold6 ? 0 : D;        // 6-launch layer: the residual stream lives in fragment order
    ru.M = B; ru.D = D; ru.eps = c.ln_eps; ru.xp_out = xp_ln;
    ru.ldws = e->ldws; ru.rows_ws = MT * 32;
    ru.ws = nullptr; ru.wte = e->wte; ru.wpe = e->wpe; ru.tokens = e->cur_tok; ru.positions = e->positions;
    ru.g = e->dec[0].ln1.g; ru.b = e->dec[0].ln1.b;
    auto row_update = [&]() {
        if (e->only_skinny) return;
        prof_mark(e, PK_ROWLN, st);
        launch_row_update_ln(ru, st);
    };
    auto skinny = [&](const bf16_t* xp, const Linear& l, int out_mode, float* ws) {
        SkinnyArgs a;
        memset(&a, 0, sizeof(a));
        a.xp = xp; a.Wp = l.Wp; a.Wq = l.Wq; a.wscale = l.wscale; a.MT = MT; a.Npad = l.Npad; a.K = l.Kpad; a.N = l.N;
        a.out_mode = out_mode; a.col_tiles = l.col_tiles;
        if (out_mode == SK_OUT_PARTIAL) {
            a.splitk = l.splitk; a.ws = ws; a.ldws = e->ldws;
            const int NT = l.Npad / 32;
            // XCD-aware (tile, K slice) assignment: every XCD pulls one K slice of the activations into its L2, not all of them
            // (profiles/skinny_r03_xcd_slice_remap_ab.log: 1121 -> 1110 us per step, tokens identical); SV_EXP bit 16 = off (A/B)
            a.xcd_remap = (!(e->exp & 16) && MT == 1 && l.splitk > 1 && 8 % l.splitk == 0 && (NT * l.splitk) % 8 == 0 &&
                           NT % (8 / l.splitk) == 0) ? 1 : 0;
        }
        else if (out_mode == SK_OUT_PACKED_ACT) { a.splitk = 1; a.bias = l.bias; a.act = ACT_GELU_TANH; a.out_xp = e->xp_mlp; a.out_KS = F / 16;
                                                  a.tail_ws = e->tail_ws; a.tail_cnt = e->tail_cnt; }
        else {
            a.splitk = 1; a.out_f32 = e->logits; a.ldo = e->Vpad; a.round_bf16 = 1;
            if (e->greedy_fused) { a.amax = e->amax; a.amax_rows = B; }      // greedy selection inside the lm_head launch (sv_generate)
            if (e->greedy_fused && e->fin_fold) { a.finish = &e->fin_args; a.fin_cnt = e->fin_cnt; }     // ... and the step's bookkeeping behind it
                    if (rc_enabled(e) && MT == 1 && xp != e->xp_a) { a.poison = e->xp_a; a.poison_bytes = xpa_bytes; }      // the next step's layer 0
            if (!e->skip_skinny) e->xpa_armed = a.poison != nullptr && lm_head_covers(e, xpa_bytes);
        }
        if (e->skip_skinny) return;
        if (out_mode == SK_OUT_F32) e->fin_folded = skinny_head_folds_finish(a);
        prof_mark(e, PK_SKINNY, st);
        launch_gemm_skinny(a, st);
    };
    for (int i = 0; i < c.n_layer; ++i) {
        DecLayer& L = e->dec[i];
        bool rc_done = false;
        if (rc) {
            SkinnyArgs a;
            memset(&a, 0, sizeof(a));
            a.xp = e->xp_a; a.Wp = L.c_attn.Wp; a.MT = MT; a.Npad = L.c_attn.Npad; a.K = L.c_attn.Kpad; a.N = L.c_attn.N;
            a.out_mode = SK_OUT_PARTIAL; a.splitk = L.c_attn.splitk; a.ws = wsA; a.ldws = e->ldws;
            prof_mark(e, PK_SKINNY, st);
            rc_done = launch_rowln_cattn(ru, a, e->d_bad, 500000, st, e->rc_delay, e->rc_dbg, i, e->num_cus) == 0;      // 5 ms budget; a refusal takes the two launches
            e->step_rc = e->step_rc || rc_done;
        }
        if (!rc_done) {
            bf16_t* const xo = rc ? e->xp_f : xp_ln;             // a refused fused launch must not reach the polled buffer with plain stores / loads
            ru.xp_out = xo;
            row_update();                                        // embedding or the previous layer's down-proj -> LN1(h)
            skinny(xo, L.c_attn, SK_OUT_PARTIAL, wsA);
            ru.xp_out = xp_ln;
        }
        // c_fc + down projection in ONE launch (gemm.hip mlp_fused_kernel): on when the engine owns its GPU (sv_config.exclusive_device);
        // SV_EXP bit 128 forces it on, bit 512 off (in-process A/B, tools/ab_exp.py).  It recognises unwritten activations by a pattern
        // that an EARLIER launch of the layer leaves in the buffer: the attention launch (16 bytes per thread: free in a latency-bound
        // kernel), or -- grid too small, or the profiling leg without attention -- the projection kernel (+0.5 us there)
        const bool fused = fold6 && e->mlp_fused_ok && !(e->exp & 512) && (c.exclusive_device || (e->exp & 128)) && !e->fused_off;
        const size_t pat_bytes = (size_t)(F / 16) * 1024;
        const size_t poison_cap = (size_t)B * e->nkv * attn_max_splits(e) * 512 * 16;      // 16 bytes per thread of the attention launch
        const bool attn_poisons = fused && !e->only_skinny && pat_bytes + (rc ? xpa_bytes : 0u) <= poison_cap;
        const bool attn_poisons_xpa = rc && i + 1 < c.n_layer && (fused ? attn_poisons : xpa_bytes <= poison_cap);
        const bool cols_poisons_xpa = rc && fold6 && i + 1 < c.n_layer && !attn_poisons_xpa;      // (rc without room in the attention grid implies fold6: see above)
        if (!e->only_skinny) {
            AttnDecodeArgs ad;
            attn_decode_args(e, i, B, wsA, L.c_attn.splitk, L.c_attn.bias, e->xp_attn, ad);
            if (i == c.n_layer / 2) ad.trace = e->attn_trace;             // SV_ATTN_TRACE=1: one layer in the middle of the step
            if (attn_poisons) { ad.poison = e->xp_mlp; ad.poison_bytes = (unsigned)pat_bytes; }
            if (attn_poisons_xpa) { ad.poison2 = e->xp_a; ad.poison2_bytes = xpa_bytes; }     // the next layer's rowln_cattn launch
            prof_mark(e, PK_ATTN, st);
            launch_attn_decode(ad, st);
        }
        if (fold6) {
            // attention output projection over the whole K per block: h += bf(x W^T + b) in place (+ partial row statistics), then
            // c_fc on the raw h with ln_2 folded into its weights / epilogue: no slabs, no row-update launch (decode_cols.hip)
            ColsArgs ca;
            memset(&ca, 0, sizeof(ca));
            ca.xp = e->xp_attn; ca.Wp = L.c_proj.Wp; ca.bias = L.c_proj.bias; ca.MT = MT; ca.N = L.c_proj.N; ca.K = L.c_proj.Kpad;
            ca.cpb = L.c_proj.cpb; ca.h_xp = e->h_xp; ca.out_KS = D / 16;
            if (fused && !attn_poisons) { ca.poison = e->xp_mlp; ca.poison_bytes = (unsigned)pat_bytes; }
            if (cols_poisons_xpa) { ca.poison2 = e->xp_a; ca.poison2_bytes = xpa_bytes; }
            if (!e->skip_skinny) { prof_mark(e, PK_SKINNY, st); launch_gemm_cols(ca, st); }
            if (fused) {
                MlpFusedArgs ma;
                memset(&ma, 0, sizeof(ma));
                ma.W1 = L.c_fc.Wf; ma.x1 = e->h_xp; ma.N1 = L.c_fc.N; ma.N1pad = L.c_fc.Npad; ma.K1 = L.c_fc.Kpad;
                ma.fold_c1 = L.c_fc.c1; ma.fold_c2 = L.c_fc.c2; ma.fold_D = D; ma.fold_eps = c.ln_eps; ma.act = ACT_GELU_TANH;
                ma.out_xp = e->xp_mlp; ma.out_KS = F / 16;
                ma.W2 = L.c_proj2.Wp; ma.N2 = L.c_proj2.N; ma.N2pad = L.c_proj2.Npad; ma.K2 = L.c_proj2.Kpad; ma.splitk = L.c_proj2.splitk;
                ma.ws = wsB; ma.ldws = e->ldws; ma.rows_ws = MT * 32; ma.err = e->d_bad; ma.spin_ticks = 500000;     // 5 ms at 100 MHz
                ma.trace = (i == c.n_layer / 2) ? e->mlp_trace : nullptr;        // one layer in the middle of the step
                // the launcher re-checks the shapes of THIS layer (sv_create looked at layer 0): a layer it refuses takes the two launches
                // below -- never a silently skipped MLP
                bool done = e->skip_skinny;
                if (!done) { prof_mark(e, PK_SKINNY, st); done = launch_mlp_fused(ma, st) == 0; }
                if (done) {
                    e->step_mlp = e->step_mlp || !e->skip_skinny;
                    const LNp& nx = (i + 1 < c.n_layer) ? e->dec[i + 1].ln1 : e->ln_f;
                    ru.ws = wsB; ru.splitk = L.c_proj2.splitk; ru.bias = L.c_proj2.bias; ru.g = nx.g; ru.b = nx.b;
                    continue;
                }
            }
            SkinnyArgs a;
            memset(&a, 0, sizeof(a));
            a.xp = e->h_xp; a.Wp = L.c_fc.Wf; a.MT = MT; a.Npad = L.c_fc.Npad; a.K = L.c_fc.Kpad; a.N = L.c_fc.N; a.splitk = 1;
            a.out_mode = SK_OUT_PACKED_ACT; a.act = ACT_GELU_TANH; a.out_xp = e->xp_mlp; a.out_KS = F / 16;
            a.fold_c1 = L.c_fc.c1; a.fold_c2 = L.c_fc.c2; a.fold_D = D; a.fold_eps = c.ln_eps;
            if (!e->skip_skinny) { prof_mark(e, PK_SKINNY, st); launch_gemm_skinny(a, st); }
        } else {
            skinny(e->xp_attn, L.c_proj, SK_OUT_PARTIAL, wsB);
            ru.ws = wsB; ru.splitk = L.c_proj.splitk; ru.bias = L.c_proj.bias; ru.g = L.ln2.g; ru.b = L.ln2.b;
            bf16_t* const x2 = rc_enabled(e) ? e->xp_f : e->xp_a;      // an engine with the fused launch keeps xp_a for it alone (write-through / L1-bypass only)
            ru.xp_out = x2;
            row_update();                                        // + bias + residual, LN2
            skinny(x2, L.c_fc, SK_OUT_PACKED_ACT, nullptr);
            ru.xp_out = xp_ln;
        }
        skinny(e->xp_mlp, L.c_proj2, SK_OUT_PARTIAL, wsB);
        const LNp& nxt = (i + 1 < c.n_layer) ? e->dec[i + 1].ln1 : e->ln_f;
        ru.ws = wsB; ru.splitk = L.c_proj2.splitk; ru.bias = L.c_proj2.bias; ru.g = nxt.g; ru.b = nxt.b;
    }
    bf16_t* xp_last = rc_enabled(e) ? e->xp_f : e->xp_a;         // (an engine with the fused launch keeps xp_a for it alone: see rc_enabled)
    ru.xp_out = xp_last;
    row_update();                                                // + bias + residual, ln_f
    skinny(xp_last, e->lm_head, SK_OUT_F32, nullptr);
    prof_mark(e, PK_SAMPLE, st);      // closes the lm_head interval; whatever follows is sampling
}

int sveng::check_ready(sv_engine* e) {
    if (!e) return fail(SV_EINVAL, "null engine");
    int r = sv_weights_complete(e);
    if (r) return fail(SV_ESTATE, "weights incomplete: %s", g_err.c_str());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// C ABI: forward entry points
// ------------------------------------------------------------------------------------------------
extern "C" int sv_encode_image(sv_engine* e, const void* dev_image, int32_t B, void* dev_out, sv_stream stream) {
    SVCHECK(check_ready(e));
    if (!dev_image || !dev_out || B < 1 || B > e->cfg.max_batch) return fail(SV_EINVAL, "sv_encode_image: bad B=%d (max_batch %d) or null pointer", B, e->cfg.max_batch);
    std::lock_guard<std::mutex> lk(e->mu);
    HIPCHECK(hipSetDevice(e->cfg.device));
    SVCHECK(vision_forward(e, (const bf16_t*)dev_image, B, (bf16_t*)dev_out, (hipStream_t)stream));
    HIPCHECK(hipGetLastError());
    return 0;
}

extern "C" int sv_adapter(sv_engine* e, const void* dev_in, int32_t B, void* dev_out, sv_stream stream) {
    SVCHECK(check_ready(e));
    if (!dev_in || !dev_out || B < 1 || B > e->cfg.max_batch) return fail(SV_EINVAL, "sv_adapter: bad B=%d or null pointer", B);
    std::lock_guard<std::mutex> lk(e->mu);
    HIPCHECK(hipSetDevice(e->cfg.device));
    SVCHECK(adapter_forward(e, (const bf16_t*)dev_in, B, (bf16_t*)dev_out, (hipStream_t)stream));
    HIPCHECK(hipGetLastError());
    return 0;
}

extern "C" int sv_embed_tokens(sv_engine* e, const int64_t* dev_ids, int32_t n, void* dev_out, sv_stream stream) {
    SVCHECK(check_ready(e));
    if (!dev_ids || !dev_out || n < 0) return fail(SV_EINVAL, "sv_embed_tokens: bad argument");
    if (n == 0) return 0;
    std::lock_guard<std::mutex> lk(e->mu);
    HIPCHECK(hipSetDevice(e->cfg.device));
    launch_gather_rows(e->wte, dev_ids, (bf16_t*)dev_out, n, e->cfg.hidden, (hipStream_t)stream);
    HIPCHECK(hipGetLastError());
    return 0;
}

// a1 (starvector_base.py:203-221) without the torch.cat: the adapter's visual rows and the prompt's token rows are written straight into
// the caller's [B][T + P][D] inputs_embeds buffer (the one sv_prefill / sv_generate read)
extern "C" int sv_adapter_into(sv_engine* e, const void* dev_in, int32_t B, void* dev_embeds, int32_t S0, sv_stream stream) {
    SVCHECK(check_ready(e));
    if (!dev_in || !dev_embeds || B < 1 || B > e->cfg.max_batch) return fail(SV_EINVAL, "sv_adapter_into: bad B=%d or null pointer", B);
    if (S0 < e->T) return fail(SV_EINVAL, "sv_adapter_into: S0=%d is shorter than the %d visual rows", S0, e->T);
    std::lock_guard<std::mutex> lk(e->mu);
    HIPCHECK(hipSetDevice(e->cfg.device));
    SVCHECK(adapter_forward(e, (const bf16_t*)dev_in, B, (bf16_t*)dev_embeds, (hipStream_t)stream, (size_t)S0 * e->cfg.hidden));
    HIPCHECK(hipGetLastError());
    return 0;
}
extern "C" int sv_embed_tokens_into(sv_engine* e, const int64_t* dev_ids, int32_t B, int32_t P, void* dev_embeds, int32_t S0, int32_t row0,
                                    sv_stream stream) {
    SVCHECK(check_ready(e));
    if (!dev_ids || !dev_embeds || B < 1 || P < 0 || row0 < 0 || row0 + P > S0) return fail(SV_EINVAL, "sv_embed_tokens_into: bad argument (rows %d..%d of %d)", row0, row0 + P, S0);
    if (P == 0) return 0;
    std::lock_guard<std::mutex> lk(e->mu);
    HIPCHECK(hipSetDevice(e->cfg.device));
    const int D = e->cfg.hidden;
    launch_gather_rows(e->wte, dev_ids, (bf16_t*)dev_embeds + (size_t)row0 * D, B * P, D, (hipStream_t)stream, P, (size_t)S0 * D);
    HIPCHECK(hipGetLastError());
    return 0;
}

static int copy_logits_out(sv_engine* e, int B, float* dev_logits, hipStream_t st) {
    HIPCHECK(hipMemcpy2DAsync(dev_logits, (size_t)e->cfg.vocab * sizeof(float), e->logits,
                              (size_t)e->Vpad * sizeof(float), (size_t)e->cfg.vocab * sizeof(float), B,
                              hipMemcpyDeviceToDevice, st));
    return 0;
}

int sveng::cb_guard(sv_engine* e, const char* who) {
    if (e->cb_active) return fail(SV_ESTATE, "%s: a continuous batch holds the KV cache of this engine (sv_cb_reset first)", who);
    return 0;
}

int sveng::prefill_locked(sv_engine* e, const void* dev_embeds, int B, int S0, int total_len, hipStream_t st, bool set_positions) {
    SVCHECK(cb_guard(e, "prefill"));
    if (!dev_embeds || B < 1 || B > e->cfg.max_batch) return fail(SV_EINVAL, "prefill: bad B=%d (max_batch %d)", B, e->cfg.max_batch);
    if (S0 < 1 || S0 > e->cfg.max_seq_len) return fail(SV_EINVAL, "prefill: S0=%d out of range (max_seq_len %d)", S0, e->cfg.max_seq_len);
    SVCHECK(assign_pages(e, B, total_len, st));
    SVCHECK(prefill_forward(e, (const bf16_t*)dev_embeds, B, S0, st));
    if (set_positions) fill_i32(e->positions, S0, B, st);          // (sv_generate sets its own generation state in one launch)
    e->cached_B = B;
    HIPCHECK(hipGetLastError());
    return 0;
}

extern "C" int sv_prefill(sv_engine* e, const void* dev_embeds, int32_t B, int32_t S0, float* dev_logits,
                          sv_stream stream) {
    SVCHECK(check_ready(e));
    if (!dev_logits) return fail(SV_EINVAL, "sv_prefill: null logits pointer");
    std::lock_guard<std::mutex> lk(e->mu);
    HIPCHECK(hipSetDevice(e->cfg.device));
    hipStream_t st = (hipStream_t)stream;
    SVCHECK(prefill_locked(e, dev_embeds, B, S0, e->cfg.max_seq_len, st));
    SVCHECK(copy_logits_out(e, B, dev_logits, st));
    return 0;
}

extern "C" int sv_forward_logits(sv_engine* e, const void* dev_embeds, int32_t B, int32_t S, int32_t n_keep,
                                 void* dev_logits_bf16, sv_stream stream) {
    SVCHECK(check_ready(e));
    if (!dev_embeds || !dev_logits_bf16) return fail(SV_EINVAL, "sv_forward_logits: null argument");
    if (B < 1 || B > e->cfg.max_batch) return fail(SV_EINVAL, "sv_forward_logits: bad B=%d (max_batch %d)", B, e->cfg.max_batch);
    if (S < 1 || S > e->cfg.max_seq_len) return fail(SV_EINVAL, "sv_forward_logits: S=%d out of range (max_seq_len %d)", S, e->cfg.max_seq_len);
    if (n_keep < 1 || n_keep > S) return fail(SV_EINVAL, "sv_forward_logits: n_keep=%d must be in 1..S", n_keep);
    std::lock_guard<std::mutex> lk(e->mu);
    HIPCHECK(hipSetDevice(e->cfg.device));
    hipStream_t st = (hipStream_t)stream;
    SVCHECK(cb_guard(e, "sv_forward_logits"));
    SVCHECK(assign_pages(e, B, S, st));
    SVCHECK(prefill_forward(e, (const bf16_t*)dev_embeds, B, S, st, n_keep, (bf16_t*)dev_logits_bf16));
    fill_i32(e->positions, S, B, st);
    e->cached_B = B;
    HIPCHECK(hipGetLastError());
    return 0;
}

extern "C" int sv_decode_step(sv_engine* e, const int32_t* dev_tokens, int32_t B, float* dev_logits, sv_stream stream) {
    SVCHECK(check_ready(e));
    if (!dev_tokens || !dev_logits) return fail(SV_EINVAL, "sv_decode_step: null pointer");
    std::lock_guard<std::mutex> lk(e->mu);
    SVCHECK(cb_guard(e, "sv_decode_step"));
    if (B != e->cached_B) return fail(SV_ESTATE, "sv_decode_step: B=%d but the cache holds %d sequences", B, e->cached_B);
    HIPCHECK(hipSetDevice(e->cfg.device));
    hipStream_t st = (hipStream_t)stream;
    HIPCHECK(hipMemsetAsync(e->d_bad, 0, sizeof(int32_t), st));         // a flag left by an earlier, failed call is not this call's
    HIPCHECK(hipMemcpyAsync(e->cur_tok, dev_tokens, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    decode_forward(e, B, st);
    add_i32(e->positions, 1, B, st);
    SVCHECK(copy_logits_out(e, B, dev_logits, st));
    HIPCHECK(hipGetLastError());
    // the fused MLP launch reports a give-up through d_bad (code 3): logits computed from its unwritten activations are void
    if (e->mlp_fused_ok || e->rc_fused_ok) SVCHECK(check_finite_logits(e, st, "sv_decode_step"));
    return 0;
}

// ------------------------------------------------------------------------------------------------
// C ABI (measurement surface): where the time to first token goes.  One pass image -> encoder -> adapter -> prompt rows -> prompt
// pass -> lm_head -> first greedy token, with a HIP event in front of every launch (group); out[2k] = ms per pass in stage k,
// out[2k + 1] = launches (groups) per pass, k = 0 encoder GEMMs, 1 encoder attention, 2 encoder row kernels (im2col, embedding +
// ln_pre, LayerNorms), 3 adapter GEMMs, 4 adapter norm + prompt-token gather, 5 decoder prompt-pass GEMMs (tile launches),
// 6 remainder-row launches of peeled GEMMs (encoder, adapter and decoder), 7 prompt-pass attention (+ RoPE, KV scatter), 8 prompt-pass
// row kernels (position embedding, LayerNorms, last-row gather, ln_f), 9 lm_head, 10 first-token selection; out[22] = the
// event-pair overhead subtracted from every interval (ms), out[23] = first event -> last event, raw (ms).  The events themselves
// open gaps between launches, so the stages sum to a little more than an unprofiled TTFT: bench.py prints both.
// dev_image NULL (text2svg): no encoder / adapter; the prompt is dev_ids [B][P] alone.
// ------------------------------------------------------------------------------------------------
extern "C" int sv_profile_ttft(sv_engine* e, const void* dev_image, int32_t B, const int64_t* dev_ids, int32_t P, int32_t iters,
                               double* out24, sv_stream stream) {
    SVCHECK(check_ready(e));
    if (!out24 || !dev_ids || iters < 1 || B < 1 || B > e->cfg.max_batch || P < 1) return fail(SV_EINVAL, "sv_profile_ttft: bad argument");
    std::lock_guard<std::mutex> lk(e->mu);
    SVCHECK(cb_guard(e, "sv_profile_ttft"));
    const sv_config& c = e->cfg;
    const int D = c.hidden, T = dev_image ? e->T : 0, S0 = T + P;
    if (S0 + 1 > c.max_seq_len) return fail(SV_EINVAL, "sv_profile_ttft: prompt of %d rows does not fit max_seq_len %d", S0, c.max_seq_len);
    HIPCHECK(hipSetDevice(c.device));
    hipStream_t st = (hipStream_t)stream;
    struct Bufs { bf16_t* enc = nullptr; bf16_t* emb = nullptr; ~Bufs() { if (enc) (void)hipFree(enc); if (emb) (void)hipFree(emb); } } bufs;
    HIPCHECK(hipMalloc(reinterpret_cast<void**>(&bufs.emb), (size_t)B * S0 * D * sizeof(bf16_t)));
    if (dev_image) HIPCHECK(hipMalloc(reinterpret_cast<void**>(&bufs.enc), (size_t)B * e->T * c.vit_width * sizeof(bf16_t)));
    SVCHECK(assign_pages(e, B, S0 + 1, st));
    for (int k = 0; k < 24; ++k) out24[k] = 0.0;
    double overhead_ms = 0.0;
    {
        struct EvPair { hipEvent_t a = nullptr, b = nullptr; ~EvPair() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); } } ev;
        HIPCHECK(hipEventCreate(&ev.a)); HIPCHECK(hipEventCreate(&ev.b));
        float acc = 0.f;
        for (int i = 0; i < 20; ++i) {
            HIPCHECK(hipEventRecord(ev.a, st)); HIPCHECK(hipEventRecord(ev.b, st));
            HIPCHECK(hipEventSynchronize(ev.b));
            float ms = 0.f; HIPCHECK(hipEventElapsedTime(&ms, ev.a, ev.b)); acc += ms;
        }
        overhead_ms = acc / 20.0;
    }
    struct ProfGuard { sv_engine* e; ~ProfGuard() { e->prof_on = false; } } pg{e};
    for (int it = 0; it < iters + 1; ++it) {
        e->prof_on = true; e->prof_used = 0;
        if (dev_image) {
            SVCHECK(vision_forward(e, (const bf16_t*)dev_image, B, bufs.enc, st));
            SVCHECK(adapter_forward(e, bufs.enc, B, bufs.emb, st, (size_t)S0 * D));
        }
        prof_mark(e, PK_AD_NORM, st);                     // the prompt-token gather travels with the adapter stage (a1)
        launch_gather_rows(e->wte, dev_ids, bufs.emb + (size_t)T * D, B * P, D, st, P, (size_t)S0 * D);
        SVCHECK(prefill_forward(e, bufs.emb, B, S0, st));
        prof_mark(e, PK_FIRST_SAMPLE, st);
        launch_argmax_partial(e->logits, e->Vpad, c.vocab, e->am_val, e->am_idx, B, nullptr, e->seen_words, 1.f, st);
        prof_mark(e, PK_END, st);
        e->prof_on = false;
        HIPCHECK(hipStreamSynchronize(st));
        HIPCHECK(hipGetLastError());
        if (it == 0) continue;                            // warm-up pass (creates the events, lets the big-M tuner settle)
        for (size_t i = 0; i + 1 < e->prof_used; ++i) {
            const int k = e->prof_kind[i] - PK_TTFT_FIRST;
            if (k < 0 || k >= PK_TTFT_COUNT) continue;
            float ms = 0.f;
            HIPCHECK(hipEventElapsedTime(&ms, e->prof_ev[i], e->prof_ev[i + 1]));
            double d = (double)ms - overhead_ms;
            out24[2 * k] += d > 0 ? d : 0;
            out24[2 * k + 1] += 1.0;
        }
        float total = 0.f;
        HIPCHECK(hipEventElapsedTime(&total, e->prof_ev[0], e->prof_ev[e->prof_used - 1]));
        out24[23] += total;
    }
    for (int k = 0; k < 2 * PK_TTFT_COUNT; ++k) out24[k] /= (double)iters;
    out24[23] /= (double)iters;
    out24[22] = overhead_ms;
    fill_i32(e->positions, S0, B, st);
    e->cached_B = B;
    HIPCHECK(hipStreamSynchronize(st));
    return 0;
}
