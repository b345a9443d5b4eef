// Host driver, part 5 of 5: host-side planning probes (callable without a GPU), the pre-processing ABI, decode-step profiling
// and the single-operator test surfaces (sv_op_*, sv_bench_*).
#include "engine_internal.h"

struct TmpBufs {
    std::vector<void*> p;
    ~TmpBufs() { for (void* q : p) (void)hipFree(q); }
    template <typename T> int get(T** out, size_t count) {
        hipError_t r = hipMalloc(reinterpret_cast<void**>(out), count * sizeof(T));
        if (r != hipSuccess) return fail(SV_ENOMEM, "hipMalloc: %s", hipGetErrorString(r));
        p.push_back(*out);
        return 0;
    }
};

// ------------------------------------------------------------------------------------------------
// C ABI: host-side decisions, callable without a GPU (CPU tests)
// ------------------------------------------------------------------------------------------------
// A/B tool surface (tools/ab_exp.py): change the experiment mask of a live engine; captured graphs are keyed on it
extern "C" int sv_debug_set_exp(sv_engine* e, int32_t mask) {
    if (!e) return fail(SV_EINVAL, "null engine");
    std::lock_guard<std::mutex> lk(e->mu);
    e->exp = mask;
    set_mt2x((mask & 131072) ? 0 : (mask & 262144) ? 2 : (mask & 524288) ? 3 : 1);      // process-wide (the launcher has no engine): the A/B tools run one engine
    e->xpa_armed = false;                    // another mask may have run plain stores through xp_a: the next step re-arms it
    for (auto& kv : e->cb_graphs) {          // the continuous-batching step graphs were captured with the old mask
        if (kv.second.second) (void)hipGraphExecDestroy(kv.second.second);
        if (kv.second.first) (void)hipGraphDestroy(kv.second.first);
    }
    e->cb_graphs.clear();
    return 0;
}

extern "C" int sv_debug_step_plan(sv_engine* e, int32_t* out4) {
    if (!e || !out4) return fail(SV_EINVAL, "sv_debug_step_plan: null argument");
    std::lock_guard<std::mutex> lk(e->mu);
    out4[0] = e->gen_gexec ? e->step_nodes : 0;
    out4[1] = e->step_rc ? 1 : 0; out4[2] = e->step_sel ? 1 : 0; out4[3] = e->step_mlp ? 1 : 0;
    return 0;
}

extern "C" int sv_debug_rowln_plan(int32_t D, int32_t N, int32_t splitk, int32_t splitk_ru, int32_t num_cus, int32_t* out3) {
    if (!out3 || D < 16 || N < 1 || splitk < 1 || splitk_ru < 1 || num_cus < 1) return fail(SV_EINVAL, "sv_debug_rowln_plan: bad argument");
    const int Npad = round_up(N, 32);
    const bool fits = rowln_cattn_fits(D, Npad, D, splitk, splitk_ru, num_cus);
    out3[0] = fits ? 1 : 0;
    out3[1] = fits ? D / 16 / splitk / 8 : 0;
    out3[2] = 32 + (Npad / 32) * splitk;
    return 0;
}

extern "C" int sv_debug_skinny_plan(int32_t rows, int32_t N, int32_t K, int32_t splitk, int32_t fp8, int32_t* out2) {
    if (!out2 || rows < 1 || N < 1 || K < 16 || K % 16 || splitk < 1 || (K / 16) % splitk)
        return fail(SV_EINVAL, "sv_debug_skinny_plan: bad argument");
    int waves = 0, two = 0;
    skinny_plan(round_up(N, 32), K, splitk, fp8, (rows + 31) / 32, &waves, &two);
    out2[0] = waves; out2[1] = two;
    return 0;
}

extern "C" int sv_debug_decode_plan(int32_t rows, int32_t N, int32_t K, int32_t fp8, int32_t whole_k, int32_t num_cus, int32_t* out2) {
    if (!out2 || rows < 1 || rows > 64 || N < 1 || K < 16 || K % 16 || num_cus < 1)
        return fail(SV_EINVAL, "sv_debug_decode_plan: bad argument");
    Linear l;
    l.N = N; l.K = K; l.Npad = round_up(N, 32); l.Kpad = K;
    int sk = 1, ct = 1;
    pick_decode_plan(l, (rows + 31) / 32, num_cus, fp8 != 0, false, whole_k != 0, &sk, &ct);
    if (fp8) while (sk > 1 && ((K / 16) % sk != 0 || ((K / 16) / sk) % 4 != 0)) --sk;          // as sv_create does
    out2[0] = sk; out2[1] = ct;
    return 0;
}

extern "C" int sv_debug_rowln_occupancy(int32_t wide, int32_t* blocks_per_cu) {
    if (!blocks_per_cu) return fail(SV_EINVAL, "sv_debug_rowln_occupancy: null argument");
    *blocks_per_cu = rowln_cattn_blocks_per_cu(wide != 0);
    return 0;
}

extern "C" int sv_debug_attn_plan(int32_t max_batch, int32_t n_kv_head, int32_t num_cus, int32_t* out2) {
    if (!out2 || max_batch < 1 || n_kv_head < 1 || num_cus < 1) return fail(SV_EINVAL, "sv_debug_attn_plan: bad argument");
    out2[0] = attn_max_splits_of(max_batch, n_kv_head, num_cus);
    out2[1] = attn_groups_per_block_of(max_batch, n_kv_head, num_cus);
    return 0;
}

// ------------------------------------------------------------------------------------------------
// C ABI: the decode attention on its own, over an engine's REAL paged KV pool / block table / split plan.
// The e2e parity tests cannot discriminate at long contexts (with random weights the attention of thousands of keys is a few
// per cent of the residual stream); here the test chooses q / K / V itself (peaked softmax, needles on chosen pages).
// ------------------------------------------------------------------------------------------------
// dev_kv: bf16 [B][S][2 * n_kv * head_dim] rows (k heads | v heads) for tokens 0..S-1 of `layer` (K as it sits in the cache, i.e. already
// rotated for StarCoder2).  Assigns every row the pages of the whole max_seq_len, writes the rows, sets positions[b] = S -- or
// dev_lens[b] (int32 [B], each <= S) when given: a ragged batch, as continuous batching runs it.
extern "C" int sv_debug_kv_load(sv_engine* e, int32_t layer, const void* dev_kv, int32_t B, int32_t S, const int32_t* dev_lens,
                                sv_stream stream) {
    if (!e || (!dev_kv && S > 0)) return fail(SV_EINVAL, "sv_debug_kv_load: null argument");
    const sv_config& c = e->cfg;
    if (layer < 0 || layer >= c.n_layer || B < 1 || B > c.max_batch || S < 0 || S >= c.max_seq_len)
        return fail(SV_EINVAL, "sv_debug_kv_load: bad layer %d / B %d / S %d (max_batch %d, max_seq_len %d)", layer, B, S, c.max_batch, c.max_seq_len);
    std::lock_guard<std::mutex> lk(e->mu);
    SVCHECK(cb_guard(e, "sv_debug_kv_load"));
    HIPCHECK(hipSetDevice(c.device));
    hipStream_t st = (hipStream_t)stream;
    SVCHECK(assign_pages(e, B, c.max_seq_len, st));
    const int dh = e->dh, nkv = e->nkv;
    if (S > 0)
        for (int kh = 0; kh < nkv; ++kh)
            launch_kv_write_prefill((const bf16_t*)dev_kv, 2 * nkv * dh, kh * dh, nkv * dh + kh * dh,
                                    e->kv_pool + (size_t)layer * e->layer_stride + (size_t)kh * e->kv_head_stride,
                                    e->block_table, e->pages_per_seq, B, S, dh, st);
    if (dev_lens) HIPCHECK(hipMemcpyAsync(e->positions, dev_lens, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    else fill_i32(e->positions, S, B, st);
    e->cached_B = B;
    e->xpa_armed = false;                       // no lm_head launch in front of the step that follows
    e->dbg_pos_hi = S;                          // dev_lens[b] <= S by contract
    HIPCHECK(hipGetLastError());
    return 0;
}

// dev_qkv_f32: fp32 [B][n_head*head_dim + 2*n_kv*head_dim] = the c_attn output of the NEW token (q heads | k heads | v heads, bias
// included, before RoPE); dev_out: bf16 [B][n_head*head_dim].  Runs attn_decode_kernel of `layer` exactly as a decode step does
// (same grid, split cap, groups per block, window, RoPE tables; the new K/V row is appended to the cache at positions[b]);
// advance != 0 steps positions[b] afterwards, so consecutive calls walk a sequence across page boundaries.
extern "C" int sv_debug_attn_decode(sv_engine* e, int32_t layer, const float* dev_qkv_f32, int32_t B, void* dev_out, int32_t advance,
                                    sv_stream stream) {
    if (!e || !dev_qkv_f32 || !dev_out) return fail(SV_EINVAL, "sv_debug_attn_decode: null argument");
    const sv_config& c = e->cfg;
    if (layer < 0 || layer >= c.n_layer) return fail(SV_EINVAL, "sv_debug_attn_decode: bad layer %d", layer);
    std::lock_guard<std::mutex> lk(e->mu);
    SVCHECK(cb_guard(e, "sv_debug_attn_decode"));
    if (B != e->cached_B) return fail(SV_ESTATE, "sv_debug_attn_decode: B=%d but the cache holds %d sequences", B, e->cached_B);
    // the kernel appends the new K/V row at positions[b] and indexes the block table with it: repeated advancing calls must not walk
    // past the sequence's allocation
    if (e->dbg_pos_hi >= c.max_seq_len)
        return fail(SV_EINVAL, "sv_debug_attn_decode: position %d would reach max_seq_len %d", e->dbg_pos_hi, c.max_seq_len);
    HIPCHECK(hipSetDevice(c.device));
    hipStream_t st = (hipStream_t)stream;
    TmpBufs tmp;
    bf16_t* zero_bias;
    SVCHECK(tmp.get(&zero_bias, (size_t)e->ldws));
    HIPCHECK(hipMemsetAsync(zero_bias, 0, (size_t)e->ldws * sizeof(bf16_t), st));
    HIPCHECK(hipMemcpy2DAsync(e->ws, (size_t)e->ldws * sizeof(float), dev_qkv_f32, (size_t)e->QKV * sizeof(float),
                              (size_t)e->QKV * sizeof(float), B, hipMemcpyDeviceToDevice, st));
    AttnDecodeArgs ad;
    attn_decode_args(e, layer, B, e->ws, 1, zero_bias, e->xp_attn, ad);
    launch_attn_decode(ad, st);
    unpack_rows(e->xp_attn, (bf16_t*)dev_out, c.n_head * e->dh, B, c.n_head * e->dh, st);
    if (advance) { add_i32(e->positions, 1, B, st); e->dbg_pos_hi += 1; }
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipStreamSynchronize(st));          // `zero_bias` is freed on return
    return 0;
}

// SV_MLP_TRACE=1 at sv_create: wall-clock stamps (100 MHz ticks) of the fused MLP launch of the middle layer of the LAST decode step,
// host_out [blocks][8] = {start, c_fc loop done, tile published, slice complete, end, XCC id, 0, 0}; returns the number of blocks
extern "C" int sv_debug_mlp_trace(sv_engine* e, int64_t* host_out, int32_t capacity_blocks) {
    if (!e || !host_out) return fail(SV_EINVAL, "sv_debug_mlp_trace: null argument");
    std::lock_guard<std::mutex> lk(e->mu);
    if (!e->mlp_trace) return fail(SV_ESTATE, "sv_debug_mlp_trace: the engine was not created with SV_MLP_TRACE=1 (or the fused MLP launch does not fit it)");
    const int T1 = e->dec[0].c_fc.Npad / 32;
    if (capacity_blocks < T1) return fail(SV_EINVAL, "sv_debug_mlp_trace: capacity %d < %d blocks", capacity_blocks, T1);
    HIPCHECK(hipSetDevice(e->cfg.device));
    HIPCHECK(hipDeviceSynchronize());
    HIPCHECK(hipMemcpy(host_out, e->mlp_trace, (size_t)T1 * 8 * sizeof(int64_t), hipMemcpyDeviceToHost));
    return T1;
}

// SV_ATTN_TRACE=1 at sv_create: wall-clock stamps (100 MHz ticks) of the decode attention launch of the middle layer of the LAST decode step,
// host_out [rows * kv heads * splits][16] = {start, first KV group requested, q in LDS, key groups processed, partial stored + drained, ticket
// drawn, end (0: not the merging block), active splits, key groups, 0...}; an all-zero row = an inactive split.  Returns the number of rows.
extern "C" int sv_debug_attn_trace(sv_engine* e, int64_t* host_out, int32_t capacity_rows) {
    if (!e || !host_out) return fail(SV_EINVAL, "sv_debug_attn_trace: null argument");
    std::lock_guard<std::mutex> lk(e->mu);
    if (!e->attn_trace) return fail(SV_ESTATE, "sv_debug_attn_trace: the engine was not created with SV_ATTN_TRACE=1");
    const int rows = e->cached_B * e->nkv * attn_max_splits_of(e->cfg.max_batch, e->nkv, e->num_cus);
    if (rows < 1 || capacity_rows < rows) return fail(SV_EINVAL, "sv_debug_attn_trace: capacity %d < %d rows", capacity_rows, rows);
    HIPCHECK(hipSetDevice(e->cfg.device));
    HIPCHECK(hipDeviceSynchronize());
    HIPCHECK(hipMemcpy(host_out, e->attn_trace, (size_t)rows * 16 * sizeof(int64_t), hipMemcpyDeviceToHost));
    return rows;
}

// host_out[blocks]: the XCC_ID each block of a 1-D launch of 8-wave blocks ran on (heavy = 1: with the decode attention's LDS footprint and
// 10 us of residence, so that a grid above the CU count is dispatched in rounds).
extern "C" int sv_debug_xcc_map(sv_engine* e, int32_t blocks, int32_t heavy, int32_t* host_out) {
    if (!e || !host_out || blocks < 1 || blocks > (1 << 16)) return fail(SV_EINVAL, "sv_debug_xcc_map: bad argument");
    std::lock_guard<std::mutex> lk(e->mu);
    HIPCHECK(hipSetDevice(e->cfg.device));
    int32_t* d = nullptr;
    HIPCHECK(hipMalloc((void**)&d, (size_t)blocks * 4));
    int rc = launch_xcc_probe(d, blocks, heavy, nullptr);
    if (!rc && (hipDeviceSynchronize() != hipSuccess || hipMemcpy(host_out, d, (size_t)blocks * 4, hipMemcpyDeviceToHost) != hipSuccess)) rc = -1;
    (void)hipFree(d);
    if (rc) return fail(SV_EHIP, "sv_debug_xcc_map: probe launch failed");
    return 0;
}

// A foreign tenant for the safety tests: `blocks` blocks that each pin `lds_bytes` of LDS on a CU for `ms` milliseconds, launched on a
// stream of their own; returns at once.  The caller then decodes on the same GPU.
extern "C" int sv_debug_occupy_cus(sv_engine* e, int32_t blocks, int32_t lds_bytes, int32_t ms) {
    if (!e || blocks < 1 || blocks > 4096 || lds_bytes < 1024 || lds_bytes > 160 * 1024 || ms < 1 || ms > 5000)
        return fail(SV_EINVAL, "sv_debug_occupy_cus: blocks 1..4096, lds_bytes 1 KiB..160 KiB, ms 1..5000");
    std::lock_guard<std::mutex> lk(e->mu);
    HIPCHECK(hipSetDevice(e->cfg.device));
    if (!e->tenant_stream) HIPCHECK(hipStreamCreateWithFlags(&e->tenant_stream, hipStreamNonBlocking));      // kept: hipStreamDestroy would wait for the kernel
    const int rc = launch_occupy(blocks, lds_bytes, ms * 100000, e->tenant_stream);
    if (rc) return fail(SV_EHIP, "sv_debug_occupy_cus: launch failed (%d)", rc);
    return 0;
}

extern "C" int sv_debug_set_gemm_form(int32_t form) {
    if (form < -1 || form > 2) return fail(SV_EINVAL, "sv_debug_set_gemm_form: -1 (tuned), 0 .. 2");
    set_gemm_form(form);
    return 0;
}

// sequence structure sv_op_linear / sv_bench_linear hand to the big-M dispatch (GemmArgs::seq_rows; 0 = none): the test surface of
// the per-sequence remainder (gemm_tailk_kernel)
static std::atomic<int> g_op_seq_rows{0};
extern "C" int sv_debug_set_linear_seq_rows(int32_t seq_rows) {
    if (seq_rows < -65536 || seq_rows > 65536) return fail(SV_EINVAL, "sv_debug_set_linear_seq_rows: |S| <= 65536");
    g_op_seq_rows = seq_rows;
    return 0;
}

extern "C" int sv_debug_set_col_tiles(int32_t col_tiles) {
    if (col_tiles < 0 || col_tiles > 3) return fail(SV_EINVAL, "sv_debug_set_col_tiles: 0..3");
    g_op_col_tiles = col_tiles;
    return 0;
}

extern "C" int sv_debug_set_skinny_form(int32_t form) {
    if (form < 0 || form > 3) return fail(SV_EINVAL, "sv_debug_set_skinny_form: 0..3");
    set_mt2x(form);
    return 0;
}

// 1 / 0: does the projection (N, K, act) take the per-sequence remainder form for sequences of S rows (host arithmetic only; < 0: bad argument)
extern "C" int sv_debug_gemm_seq_form(int32_t S, int32_t N, int32_t K, int32_t act) {
    if (S < 1 || N < 1 || K < 1) return fail(SV_EINVAL, "sv_debug_gemm_seq_form: bad argument");
    return gemm_seq_form(S, N, K, act) ? 1 : 0;
}

extern "C" int sv_debug_gemm_plan(int32_t M, int32_t N, int32_t K, int32_t act, int32_t* out5) {
    if (!out5 || M < 1 || N < 1 || K < 1) return fail(SV_EINVAL, "sv_debug_gemm_plan: bad argument");
    const GemmPlan pl = gemm_plan(M, N, K, act, 1);
    out5[0] = pl.peel; out5[1] = pl.tail_rows; out5[2] = pl.tail_by_tiles; out5[3] = pl.main_256;
    out5[4] = (int32_t)(pl.est_us + 0.5);
    return 0;
}

// ------------------------------------------------------------------------------------------------
// C ABI: image pre-processing on device (no engine handle: it depends on nothing but the pixels)
// ------------------------------------------------------------------------------------------------
extern "C" int sv_preprocess_image(const uint8_t* dev_pixels, int32_t width, int32_t height, int32_t channels,
                                   int32_t out_size, int32_t recipe, const float* mean3, const float* std3,
                                   float* dev_out, sv_stream stream) {
    if (recipe != 0 && recipe != 1) return fail(SV_EINVAL, "sv_preprocess_image: recipe must be 0 (ImageTrainProcessor) or 1 (SigLIP processor)");
    if (!dev_pixels || !dev_out || !mean3 || !std3) return fail(SV_EINVAL, "sv_preprocess_image: null argument");
    if (width < 1 || height < 1 || width > 16384 || height > 16384) return fail(SV_EINVAL, "sv_preprocess_image: bad image size %dx%d", width, height);
    if (channels != 3 && channels != 4) return fail(SV_EINVAL, "sv_preprocess_image: channels must be 3 (RGB) or 4 (RGBA), got %d", channels);
    if (out_size < 1 || out_size > 4096) return fail(SV_EINVAL, "sv_preprocess_image: bad output size %d", out_size);
    for (int c = 0; c < 3; ++c)
        if (!(std3[c] > 0.f)) return fail(SV_EINVAL, "sv_preprocess_image: std must be positive");
    const int r = preprocess_image(dev_pixels, width, height, channels, out_size, recipe, mean3, std3, dev_out, (hipStream_t)stream);
    if (r) return fail(SV_EHIP, "sv_preprocess_image: %s", hipGetErrorString((hipError_t)r));
    return 0;
}

extern "C" int64_t sv_preprocess_workspace_bytes(const int32_t* widths, const int32_t* heights, int32_t n, int32_t out_size,
                                                 int32_t recipe) {
    if (!widths || !heights || n < 1 || out_size < 1 || out_size > 4096 || (recipe != 0 && recipe != 1)) return fail(SV_EINVAL, "sv_preprocess_workspace_bytes: bad argument");
    for (int i = 0; i < n; ++i)
        if (widths[i] < 1 || heights[i] < 1 || widths[i] > 16384 || heights[i] > 16384) return fail(SV_EINVAL, "sv_preprocess_workspace_bytes: bad image size %dx%d", widths[i], heights[i]);
    return (int64_t)preprocess_workspace_bytes(widths, heights, n, out_size, recipe);
}

extern "C" int sv_preprocess_images(const uint8_t* const* dev_pixels, const int32_t* widths, const int32_t* heights,
                                    const int32_t* channels, int32_t n, int32_t out_size, int32_t recipe, const float* mean3,
                                    const float* std3, float* dev_out, void* dev_workspace, int64_t workspace_bytes,
                                    sv_stream stream) {
    if (recipe != 0 && recipe != 1) return fail(SV_EINVAL, "sv_preprocess_images: recipe must be 0 (ImageTrainProcessor) or 1 (SigLIP processor)");
    if (!dev_pixels || !widths || !heights || !channels || !dev_out || !mean3 || !std3 || n < 1) return fail(SV_EINVAL, "sv_preprocess_images: null argument or empty batch");
    if (out_size < 1 || out_size > 4096) return fail(SV_EINVAL, "sv_preprocess_images: bad output size %d", out_size);
    for (int i = 0; i < n; ++i) {
        if (!dev_pixels[i]) return fail(SV_EINVAL, "sv_preprocess_images: image %d is null", i);
        if (widths[i] < 1 || heights[i] < 1 || widths[i] > 16384 || heights[i] > 16384) return fail(SV_EINVAL, "sv_preprocess_images: bad image size %dx%d", widths[i], heights[i]);
        if (channels[i] != 3 && channels[i] != 4) return fail(SV_EINVAL, "sv_preprocess_images: channels must be 3 (RGB) or 4 (RGBA), got %d", channels[i]);
    }
    for (int c = 0; c < 3; ++c)
        if (!(std3[c] > 0.f)) return fail(SV_EINVAL, "sv_preprocess_images: std must be positive");
    const size_t need = preprocess_workspace_bytes(widths, heights, n, out_size, recipe);
    if (workspace_bytes < 0 || (size_t)workspace_bytes < need || (!dev_workspace && need > 256))
        return fail(SV_EINVAL, "sv_preprocess_images: workspace of %lld bytes, %zu needed (sv_preprocess_workspace_bytes)", (long long)workspace_bytes, need);
    const int r = preprocess_images(dev_pixels, widths, heights, channels, n, out_size, recipe, mean3, std3, dev_out, dev_workspace,
                                    (size_t)workspace_bytes, (hipStream_t)stream);
    if (r) return fail(SV_EHIP, "sv_preprocess_images: %s", r < 0 ? "workspace too small" : hipGetErrorString((hipError_t)r));
    return 0;
}

// Per-kernel timing of the decode step with HIP events on the engine stream (eager launches of the
// same kernels the graph replays).  Uses the KV cache / positions left by the last generate or prefill.
// out[2*k] = milliseconds per step spent in kernel class k, out[2*k+1] = launches per step,
// k in {0: skinny GEMM, 1: decode attention, 2: row update + LayerNorm, 3: lm_head-to-end marker}.
extern "C" int sv_profile_decode_step(sv_engine* e, int32_t B, int32_t iters, double* out, sv_stream stream) {
    SVCHECK(check_ready(e));
    if (!out || iters < 1) return fail(SV_EINVAL, "sv_profile_decode_step: bad argument");
    std::lock_guard<std::mutex> lk(e->mu);
    if (B != e->cached_B) return fail(SV_ESTATE, "sv_profile_decode_step: B=%d but the cache holds %d sequences", B, e->cached_B);
    HIPCHECK(hipSetDevice(e->cfg.device));
    HIPCHECK(hipEventRecord(e->gen_event, (hipStream_t)stream));
    hipStream_t st = e->gen_stream;
    HIPCHECK(hipStreamWaitEvent(st, e->gen_event, 0));
    // positions may sit one past the budget after a full generate: step back so the probe stays in range
    add_i32(e->positions, -1, B, st);
    struct PosGuard { sv_engine* e; int B; hipStream_t st; ~PosGuard() { add_i32(e->positions, 1, B, st); (void)hipStreamSynchronize(st); } } pg{e, B, st};
    for (int k = 0; k < 2 * PK_COUNT + 2; ++k) out[k] = 0.0;
    // event-pair overhead: two back-to-back events with nothing in between
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
    for (int it = 0; it < iters + 1; ++it) {
        e->prof_on = true; e->prof_used = 0;
        decode_forward(e, B, st);
        e->prof_on = false;                  // (no HIP call that can return sits between the two assignments)
        HIPCHECK(hipStreamSynchronize(st));
        if (it == 0) continue;              // warm-up pass (also creates the events)
        for (size_t i = 0; i + 1 < e->prof_used; ++i) {
            float ms = 0.f;
            HIPCHECK(hipEventElapsedTime(&ms, e->prof_ev[i], e->prof_ev[i + 1]));
            double d = (double)ms - overhead_ms;      // an event pair with nothing in between costs ~4 us
            if (d < 0) d = 0;
            out[2 * e->prof_kind[i]] += d;
            out[2 * e->prof_kind[i] + 1] += 1.0;
        }
    }
    for (int k = 0; k < 2 * PK_COUNT; ++k) out[k] /= (double)iters;
    out[2 * PK_SAMPLE] = overhead_ms;         // slot 6: time between two back-to-back events with no kernel
    // slot 7: the step's weight-streaming GEMMs alone, back to back between ONE event pair: average
    // dispatch-to-dispatch time per launch (what rocprofv3's kernel trace calls the kernel duration)
    // slot 8: the complement -- every other kernel of the step (row updates, attention) back to back, no GEMMs.  The step
    // minus this chain is what the GEMMs cost IN SITU (bench.py's roofline.avg_launch_us)
    // The two filters are engine state read by decode_forward: the guard clears them and destroys the events on EVERY exit,
    // so a failed HIP call in here can never leave a live engine that silently skips its GEMMs.
    struct ChainGuard {
        sv_engine* e; hipEvent_t a = nullptr, b = nullptr;
        ~ChainGuard() { e->only_skinny = false; e->skip_skinny = false; if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
    } cg{e};
    HIPCHECK(hipEventCreate(&cg.a)); HIPCHECK(hipEventCreate(&cg.b));
    for (int leg = 0; leg < 2; ++leg) {
        e->only_skinny = leg == 0;
        e->skip_skinny = leg == 1;
        decode_forward(e, B, st);
        HIPCHECK(hipEventRecord(cg.a, st));
        for (int it = 0; it < iters; ++it) decode_forward(e, B, st);
        HIPCHECK(hipEventRecord(cg.b, st));
        e->only_skinny = false; e->skip_skinny = false;
        HIPCHECK(hipEventSynchronize(cg.b));
        float ms = 0.f; HIPCHECK(hipEventElapsedTime(&ms, cg.a, cg.b));
        out[leg == 0 ? 2 * PK_SAMPLE + 1 : 2 * PK_COUNT] = (double)ms / iters;       // ms per step for the chain
    }
    return 0;           // ~PosGuard steps the positions forward again and drains the stream
}

extern "C" int sv_last_timing(sv_engine* e, double* out3) {   /* 4 doubles */
    if (!e || !out3) return fail(SV_EINVAL, "null argument");
    out3[0] = e->timing[0]; out3[1] = e->timing[1]; out3[2] = e->timing[2]; out3[3] = e->timing_graph;
    return 0;
}

// ------------------------------------------------------------------------------------------------
// C ABI: single operators (test surface)
// ------------------------------------------------------------------------------------------------

extern "C" int sv_op_layernorm(const void* x, const void* gamma, const void* beta, void* y, int32_t M, int32_t D,
                               float eps, sv_stream stream) {
    if (!x || !gamma || !beta || !y || M < 1 || D < 8 || D % 8) return fail(SV_EINVAL, "sv_op_layernorm: bad argument");
    launch_layernorm_rows((const bf16_t*)x, D, (const bf16_t*)gamma, (const bf16_t*)beta, (bf16_t*)y, D, M, D, eps,
                          (hipStream_t)stream);
    HIPCHECK(hipGetLastError());
    return 0;
}

extern "C" int sv_op_linear(const void* x, const void* W, const void* bias, const void* residual, void* y, int32_t M,
                            int32_t N, int32_t K, int32_t act, int32_t out_f32, sv_stream stream) {
    if (!x || !W || !y || M < 1 || N < 4 || N % 4 || K < 1) return fail(SV_EINVAL, "sv_op_linear: bad argument");
    hipStream_t st = (hipStream_t)stream;
    TmpBufs tmp;
    const int Npad = round_up(N, 32), Kpad = round_up(K, 64);
    bf16_t* Wp;
    SVCHECK(tmp.get(&Wp, (size_t)Npad * Kpad));
    launch_pack_weight(W, 0, Wp, N, K, Npad, Kpad, st);
    const bf16_t* A = (const bf16_t*)x;
    int lda = K;
    if (Kpad != K) {
        bf16_t* xpd;
        SVCHECK(tmp.get(&xpd, (size_t)M * Kpad));
        HIPCHECK(hipMemsetAsync(xpd, 0, (size_t)M * Kpad * 2, st));
        HIPCHECK(hipMemcpy2DAsync(xpd, (size_t)Kpad * 2, x, (size_t)K * 2, (size_t)K * 2, M, hipMemcpyDeviceToDevice, st));
        A = xpd; lda = Kpad;
    }
    GemmArgs g;
    g.A = A; g.lda = lda; g.Wp = Wp; g.bias = (const bf16_t*)bias; g.R = (const bf16_t*)residual; g.ldr = N;
    g.C = y; g.ldc = N; g.M = M; g.N = N; g.K = Kpad; g.act = act; g.out_f32 = out_f32;
    g.seq_rows = g_op_seq_rows.load();
    if (g.seq_rows < 0) { g.seq_rows = -g.seq_rows; g.splitk_rows = 1; }      // the rows are the last rows of sequences of |S| rows
    launch_gemm(g, st);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipStreamSynchronize(st));
    return 0;
}

// Micro-benchmark of the big-M MFMA GEMM alone (zero-filled... no: uniform random bf16 operands, HIP events)
extern "C" int sv_bench_linear(int32_t M, int32_t N, int32_t K, int32_t act, int32_t residual, int32_t iters,
                               double* avg_us, sv_stream stream) {
    if (!avg_us || M < 1 || N < 32 || N % 4 || K < 64 || K % 64 || iters < 1) return fail(SV_EINVAL, "sv_bench_linear: bad argument");
    hipStream_t st = (hipStream_t)stream;
    TmpBufs tmp;
    const int Npad = round_up(N, 32);
    bf16_t *Wp, *A, *C, *bias, *Wsrc;
    SVCHECK(tmp.get(&Wp, (size_t)Npad * K));
    SVCHECK(tmp.get(&Wsrc, (size_t)N * K));
    SVCHECK(tmp.get(&A, (size_t)M * K));
    SVCHECK(tmp.get(&C, (size_t)M * N));
    SVCHECK(tmp.get(&bias, (size_t)N));
    // pseudo-random operands in [-1, 1): zero-filled data would clock ~20 % higher (cdna guide, rule 25)
    fill_random_bf16(A, (size_t)M * K, 1u, 4096, st);
    fill_random_bf16(Wsrc, (size_t)N * K, 2u, 4096, st);
    fill_random_bf16(bias, (size_t)N, 3u, 64, st);
    fill_random_bf16(C, (size_t)M * N, 4u, 4096, st);
    launch_pack_weight(Wsrc, 0, Wp, N, K, Npad, K, st);
    if (const char* fill = getenv("SV_BENCH_FILL")) {
        // evidence for the ceiling claim (cdna guide rule 25): the SAME kernel on zero-filled operands clocks ~20 % higher (DVFS), so a
        // quoted TF/s must say which fill it ran on.  "zero": everything zero; default: uniform random in [-1, 1)
        if (!strcmp(fill, "zero")) {
            HIPCHECK(hipMemsetAsync(A, 0, (size_t)M * K * 2, st));
            HIPCHECK(hipMemsetAsync(Wp, 0, (size_t)Npad * K * 2, st));
            HIPCHECK(hipMemsetAsync(C, 0, (size_t)M * N * 2, st));
            HIPCHECK(hipMemsetAsync(bias, 0, (size_t)N * 2, st));
        }
    }
    GemmArgs g;
    g.A = A; g.lda = K; g.Wp = Wp; g.bias = bias; g.R = residual ? C : nullptr; g.ldr = N; g.C = C; g.ldc = N;
    g.M = M; g.N = N; g.K = K; g.act = act; g.out_f32 = 0;
    g.seq_rows = g_op_seq_rows.load() > 0 ? g_op_seq_rows.load() : 0;
    // SV_BENCH_GEMM_FORM = 0 / 1: one fixed form (128^2 tiles, 256^2 tiles; rows not peeled) instead of the tuned choice
    const char* form_s = getenv("SV_BENCH_GEMM_FORM");
    const int form = form_s ? atoi(form_s) : -1;
    auto run1 = [&]() { if (form >= 0) launch_gemm_fixed(g, form, 0, st); else launch_gemm(g, st); };
    hipEvent_t e0, e1;
    HIPCHECK(hipEventCreate(&e0)); HIPCHECK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) run1();
    HIPCHECK(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i) run1();
    HIPCHECK(hipEventRecord(e1, st));
    HIPCHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    *avg_us = (double)ms * 1e3 / iters;
    HIPCHECK(hipGetLastError());
    return 0;
}

// The 256^2 big-M kernel on its own over random operands, one launch with wall-clock stamps (100 MHz): host_out [blocks * 2][8] =
// {start, K-tile 0 staged, K loop done, epilogue stored and drained, tile m, tile n, wave (0 / 7: one wave of each ping-pong group), 0}.
// form: 1 (round 4's persistent stream-K kernel was form 2: profiles/gemm_trace_r04.log).  Returns the number of blocks (tools/gemm_trace.py:
// where the time of a prefill GEMM goes).
extern "C" int sv_debug_gemm_trace(int32_t M, int32_t N, int32_t K, int32_t act, int32_t form, int64_t* host_out, int32_t capacity_blocks) {
    if (!host_out || M < 256 || N < 256 || N % 8 || K < 64 || K % 64 || form != 1) return fail(SV_EINVAL, "sv_debug_gemm_trace: bad argument (form: 1)");
    const int blocks = ((M + 255) / 256) * ((N + 255) / 256);
    if (capacity_blocks < blocks) return fail(SV_EINVAL, "sv_debug_gemm_trace: capacity %d < %d blocks", capacity_blocks, blocks);
    hipStream_t st = nullptr;
    TmpBufs tmp;
    const int Npad = round_up(N, 32);
    bf16_t *Wp, *A, *C, *bias, *Wsrc;
    long long* tr;
    SVCHECK(tmp.get(&Wp, (size_t)Npad * K));
    SVCHECK(tmp.get(&Wsrc, (size_t)N * K));
    SVCHECK(tmp.get(&A, (size_t)M * K));
    SVCHECK(tmp.get(&C, (size_t)M * N));
    SVCHECK(tmp.get(&bias, (size_t)N));
    SVCHECK(tmp.get(&tr, (size_t)blocks * 16));
    fill_random_bf16(A, (size_t)M * K, 1u, 4096, st);
    fill_random_bf16(Wsrc, (size_t)N * K, 2u, 4096, st);
    fill_random_bf16(bias, (size_t)N, 3u, 64, st);
    launch_pack_weight(Wsrc, 0, Wp, N, K, Npad, K, st);
    HIPCHECK(hipMemsetAsync(tr, 0, (size_t)blocks * 16 * sizeof(long long), st));
    GemmArgs g;
    g.A = A; g.lda = K; g.Wp = Wp; g.bias = bias; g.R = nullptr; g.ldr = N; g.C = C; g.ldc = N;
    g.M = M; g.N = N; g.K = K; g.act = act; g.out_f32 = 0;
    for (int i = 0; i < 20; ++i) launch_gemm_fixed(g, form, 0, st);          // clocks and caches as in a running prefill
    g.trace = tr;
    launch_gemm_fixed(g, form, 0, st);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipStreamSynchronize(st));
    HIPCHECK(hipMemcpy(host_out, tr, (size_t)blocks * 16 * sizeof(long long), hipMemcpyDeviceToHost));
    return blocks;
}

extern "C" int sv_op_linear_skinny(const void* x, const void* W, const void* bias, void* y_f32, int32_t M, int32_t N,
                                   int32_t K, int32_t splitk, sv_stream stream) {
    if (!x || !W || !y_f32 || M < 1 || N < 1 || K < 16 || K % 16 || splitk < 1 || (K / 16) % splitk)
        return fail(SV_EINVAL, "sv_op_linear_skinny: bad argument");
    hipStream_t st = (hipStream_t)stream;
    TmpBufs tmp;
    const int Npad = round_up(N, 32), MT = (M + 31) / 32;
    bf16_t *Wp, *xp;
    float* ws;
    SVCHECK(tmp.get(&Wp, (size_t)Npad * K));
    SVCHECK(tmp.get(&xp, (size_t)MT * 32 * K));
    SVCHECK(tmp.get(&ws, (size_t)splitk * MT * 32 * Npad));
    HIPCHECK(hipMemsetAsync(xp, 0, (size_t)MT * 32 * K * 2, st));
    launch_pack_weight(W, 0, Wp, N, K, Npad, K, st);
    pack_rows((const bf16_t*)x, K, xp, M, K, st);
    SkinnyArgs a;
    memset(&a, 0, sizeof(a));
    a.xp = xp; a.Wp = Wp; a.MT = MT; a.Npad = Npad; a.K = K; a.splitk = splitk; a.out_mode = SK_OUT_PARTIAL;
    a.ws = ws; a.ldws = Npad; a.N = N;
    launch_gemm_skinny(a, st);
    reduce_partials(ws, splitk, MT * 32, Npad, (const bf16_t*)bias,
                                                                 (float*)y_f32, M, N, st);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipStreamSynchronize(st));
    return 0;
}

// the fp8-weight decode GEMM on its own: W [N][K] bf16 is quantised like sv_load_weight does it with weight_dtype = fp8
// (one e4m3 scale per row), y = x . dequant(quant(W))^T + bias in fp32; scale_out [N] (optional) returns the row scales
extern "C" int sv_op_linear_skinny_fp8(const void* x, const void* W, const void* bias, void* y_f32, float* scale_out,
                                       int32_t M, int32_t N, int32_t K, int32_t splitk, sv_stream stream) {
    if (!x || !W || !y_f32 || M < 1 || N < 1 || K < 64 || K % 64 || splitk < 1 || (K / 16) % splitk || ((K / 16) / splitk) % 4)
        return fail(SV_EINVAL, "sv_op_linear_skinny_fp8: bad argument (K %% 64 == 0 and (K/16/splitk) %% 4 == 0 required)");
    hipStream_t st = (hipStream_t)stream;
    TmpBufs tmp;
    const int Npad = round_up(N, 32), MT = (M + 31) / 32;
    bf16_t *Wp, *xp;
    uint8_t* Wq;
    float *ws, *sc;
    SVCHECK(tmp.get(&Wp, (size_t)Npad * K));
    SVCHECK(tmp.get(&Wq, (size_t)Npad * K));
    SVCHECK(tmp.get(&sc, (size_t)Npad));
    SVCHECK(tmp.get(&xp, (size_t)MT * 32 * K));
    SVCHECK(tmp.get(&ws, (size_t)splitk * MT * 32 * Npad));
    HIPCHECK(hipMemsetAsync(xp, 0, (size_t)MT * 32 * K * 2, st));
    launch_pack_weight_fp8(W, 0, Wp, Wq, sc, N, K, Npad, K, st);
    pack_rows((const bf16_t*)x, K, xp, M, K, st);
    SkinnyArgs a;
    memset(&a, 0, sizeof(a));
    a.xp = xp; a.Wp = Wp; a.Wq = Wq; a.wscale = sc; a.MT = MT; a.Npad = Npad; a.K = K; a.splitk = splitk;
    a.out_mode = SK_OUT_PARTIAL; a.ws = ws; a.ldws = Npad; a.N = N;
    launch_gemm_skinny(a, st);
    reduce_partials(ws, splitk, MT * 32, Npad, (const bf16_t*)bias,
                                                                 (float*)y_f32, M, N, st);
    if (scale_out) HIPCHECK(hipMemcpyAsync(scale_out, sc, (size_t)N * sizeof(float), hipMemcpyDeviceToDevice, st));
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipStreamSynchronize(st));
    return 0;
}

// the decode GEMM's two fused epilogues on their own (split-K 1): out_f32 == 0: y[M,N] = act(bf16(x W^T + bias)) as bf16 rows
// (c_fc: N %% 8 == 0); out_f32 != 0: fp32 rows of x W^T rounded to bf16 values, no bias (lm_head)
extern "C" int sv_op_linear_skinny_epi(const void* x, const void* W, const void* bias, void* y, int32_t M, int32_t N, int32_t K,
                                       int32_t act, int32_t out_f32, sv_stream stream) {
    if (!x || !W || !y || M < 1 || N < 1 || K < 16 || K % 16) return fail(SV_EINVAL, "sv_op_linear_skinny_epi: bad argument");
    if (!out_f32 && N % 8) return fail(SV_EINVAL, "bf16 output needs N %% 8 == 0");
    hipStream_t st = (hipStream_t)stream;
    TmpBufs tmp;
    const int Npad = round_up(N, 32), MT = (M + 31) / 32, R = MT * 32;
    bf16_t *Wp, *xp, *oxp;
    float* of;
    SVCHECK(tmp.get(&Wp, (size_t)Npad * K));
    SVCHECK(tmp.get(&xp, (size_t)R * K));
    SVCHECK(tmp.get(&oxp, (size_t)R * Npad));
    SVCHECK(tmp.get(&of, (size_t)R * Npad));
    HIPCHECK(hipMemsetAsync(xp, 0, (size_t)R * K * 2, st));
    launch_pack_weight(W, 0, Wp, N, K, Npad, K, st);
    pack_rows((const bf16_t*)x, K, xp, M, K, st);
    SkinnyArgs a;
    memset(&a, 0, sizeof(a));
    a.xp = xp; a.Wp = Wp; a.MT = MT; a.Npad = Npad; a.K = K; a.splitk = 1; a.N = N;
    if (out_f32) { a.out_mode = SK_OUT_F32; a.out_f32 = of; a.ldo = Npad; a.round_bf16 = 1; }
    else {
        a.out_mode = SK_OUT_PACKED_ACT; a.bias = (const bf16_t*)bias; a.act = act; a.out_xp = oxp; a.out_KS = Npad / 16;
        // the engine's scratch for the tiles beyond the first round of blocks (gemm_skinny_tailsplit_kernel; taken only where launch_gemm_skinny's rule says so)
        float* tws; unsigned* tcnt;
        SVCHECK(tmp.get(&tws, (size_t)SV_TAIL_TILES * 4 * 16 * 64));
        SVCHECK(tmp.get(&tcnt, (size_t)SV_TAIL_TILES));
        HIPCHECK(hipMemsetAsync(tcnt, 0, (size_t)SV_TAIL_TILES * sizeof(unsigned), st));
        a.tail_ws = tws; a.tail_cnt = tcnt;
    }
    launch_gemm_skinny(a, st);
    if (out_f32) HIPCHECK(hipMemcpy2DAsync(y, (size_t)N * 4, of, (size_t)Npad * 4, (size_t)N * 4, M, hipMemcpyDeviceToDevice, st));
    else unpack_rows(oxp, (bf16_t*)y, N, M, N, st);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipStreamSynchronize(st));
    return 0;
}

// the lm_head form of the decode GEMM with the greedy selection folded into its epilogue (gemm.hip, SkinnyArgs::amax): y_f32 [M][N]
// (optional) = the bf16-rounded fp32 logits x W^T, host_idx [M] = the arg-max column of every row as finish_step_kernel decodes it
// (lowest index on ties, NaN never wins; 0x7fffffff for a row without a single comparable score).  M <= 32.
extern "C" int sv_op_lm_head_argmax(const void* x, const void* W, void* y_f32, int32_t* host_idx, int32_t M, int32_t N, int32_t K,
                                    sv_stream stream) {
    if (!x || !W || !host_idx || M < 1 || M > 32 || N < 1 || K < 32 || K % 32) return fail(SV_EINVAL, "sv_op_lm_head_argmax: bad argument (M <= 32, K %% 32 == 0)");
    if (int ar = init_gemm_kernels()) return fail(SV_EHIP, "hipFuncSetAttribute: %s", hipGetErrorString((hipError_t)ar));
    hipStream_t st = (hipStream_t)stream;
    TmpBufs tmp;
    const int Npad = round_up(N, 32);
    int waves = 1, two = 0;
    skinny_plan(Npad, K, 1, 0, 1, &waves, &two);
    if (waves < 2) return fail(SV_ENOTSUP, "sv_op_lm_head_argmax: K = %d leaves one wave per block (the fold needs the K split over waves)", K);
    bf16_t *Wp, *xp;
    float* of;
    unsigned long long* keys;
    SVCHECK(tmp.get(&Wp, (size_t)Npad * K));
    SVCHECK(tmp.get(&xp, (size_t)32 * K));
    SVCHECK(tmp.get(&of, (size_t)32 * Npad));
    SVCHECK(tmp.get(&keys, (size_t)32 * SV_AMAX_STRIDE));
    HIPCHECK(hipMemsetAsync(xp, 0, (size_t)32 * K * 2, st));
    HIPCHECK(hipMemsetAsync(keys, 0, (size_t)32 * SV_AMAX_STRIDE * 8, st));
    launch_pack_weight(W, 0, Wp, N, K, Npad, K, st);
    pack_rows((const bf16_t*)x, K, xp, M, K, st);
    SkinnyArgs a;
    memset(&a, 0, sizeof(a));
    a.xp = xp; a.Wp = Wp; a.MT = 1; a.Npad = Npad; a.K = K; a.splitk = 1; a.N = N;
    a.out_mode = SK_OUT_F32; a.out_f32 = of; a.ldo = Npad; a.round_bf16 = 1; a.amax = keys; a.amax_rows = M;
    launch_gemm_skinny(a, st);
    if (y_f32) HIPCHECK(hipMemcpy2DAsync(y_f32, (size_t)N * 4, of, (size_t)Npad * 4, (size_t)N * 4, M, hipMemcpyDeviceToDevice, st));
    std::vector<unsigned long long> h((size_t)32 * SV_AMAX_STRIDE);
    HIPCHECK(hipMemcpyAsync(h.data(), keys, h.size() * 8, hipMemcpyDeviceToHost, st));
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipStreamSynchronize(st));
    for (int m = 0; m < M; ++m) host_idx[m] = sv_amax_index(h[(size_t)m * SV_AMAX_STRIDE]);
    return 0;
}

// The 6-launch layer's two kernels as one op (decode_cols.hip): h2 = bf16(h + bf16(x Wp^T + bp)) by the slab-free output projection
// (whole K per block, partial row statistics), then y = act(bf16(LN(h2; gamma, beta) Wf^T + bf)) by the decode GEMM on the RAW h2
// with the LayerNorm folded into its weights / epilogue.  Row-major in / out; M <= 32.
extern "C" int sv_op_decode_proj_fold(const void* x, const void* Wp_, const void* bp, const void* h, const void* gamma, const void* beta,
                                      float eps, const void* Wf_, const void* bf_, void* h2_out, void* y_out, int32_t M, int32_t D,
                                      int32_t Kp, int32_t F, int32_t act, sv_stream stream) {
    if (!x || !Wp_ || !h || !gamma || !beta || !Wf_ || !h2_out || !y_out || M < 1 || M > 32 || D < 32 || D % 32 || Kp < 32 || Kp % 32 ||
        F < 8 || F % 8)
        return fail(SV_EINVAL, "sv_op_decode_proj_fold: bad argument (M <= 32, D %% 32 == 0, Kp %% 32 == 0, F %% 8 == 0)");
    if (int ar = init_cols_kernels()) return fail(SV_EHIP, "hipFuncSetAttribute: %s", hipGetErrorString((hipError_t)ar));
    if (int ar = init_gemm_kernels()) return fail(SV_EHIP, "hipFuncSetAttribute: %s", hipGetErrorString((hipError_t)ar));
    hipStream_t st = (hipStream_t)stream;
    TmpBufs tmp;
    const int Fpad = round_up(F, 32), cpb = cols_pick_cpb(D, Kp);
    bf16_t *Wpp, *Wfp, *Wff, *xp, *hxp, *yxp;
    float *c1, *c2;
    SVCHECK(tmp.get(&Wpp, (size_t)D * Kp));
    SVCHECK(tmp.get(&Wfp, (size_t)Fpad * D));
    SVCHECK(tmp.get(&Wff, (size_t)Fpad * D));
    SVCHECK(tmp.get(&xp, (size_t)32 * Kp));
    SVCHECK(tmp.get(&hxp, (size_t)32 * D));
    SVCHECK(tmp.get(&yxp, (size_t)32 * Fpad));
    SVCHECK(tmp.get(&c1, (size_t)Fpad));
    SVCHECK(tmp.get(&c2, (size_t)Fpad));
    HIPCHECK(hipMemsetAsync(xp, 0, (size_t)32 * Kp * 2, st));
    HIPCHECK(hipMemsetAsync(hxp, 0, (size_t)32 * D * 2, st));
    launch_pack_weight(Wp_, 0, Wpp, D, Kp, D, Kp, st);
    launch_pack_weight(Wf_, 0, Wfp, F, D, Fpad, D, st);
    pack_rows((const bf16_t*)x, Kp, xp, M, Kp, st);
    pack_rows((const bf16_t*)h, D, hxp, M, D, st);
    launch_fold_prepare(Wfp, (const bf16_t*)gamma, (const bf16_t*)beta, (const bf16_t*)bf_, Wff, c1, c2, F, Fpad, D, st);
    ColsArgs ca;
    memset(&ca, 0, sizeof(ca));
    ca.xp = xp; ca.Wp = Wpp; ca.bias = (const bf16_t*)bp; ca.MT = 1; ca.N = D; ca.K = Kp; ca.cpb = cpb; ca.h_xp = hxp; ca.out_KS = D / 16;
    if (launch_gemm_cols(ca, st)) return fail(SV_ENOTSUP, "sv_op_decode_proj_fold: no kernel for this shape");
    SkinnyArgs a;
    memset(&a, 0, sizeof(a));
    a.xp = hxp; a.Wp = Wff; a.MT = 1; a.Npad = Fpad; a.K = D; a.N = F; a.splitk = 1; a.out_mode = SK_OUT_PACKED_ACT; a.act = act;
    a.out_xp = yxp; a.out_KS = Fpad / 16; a.fold_c1 = c1; a.fold_c2 = c2; a.fold_D = D;
    a.fold_eps = eps;
    launch_gemm_skinny(a, st);
    unpack_rows(hxp, (bf16_t*)h2_out, D, M, D, st);
    unpack_rows(yxp, (bf16_t*)y_out, F, M, F, st);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipStreamSynchronize(st));
    return 0;
}

// Micro-benchmark of the decode GEMM kernel alone (HIP events, `iters` back-to-back launches on one stream):
// mode 0 = fp32 slabs (split-K `splitk`), 1 = bias + GELU -> fragment order, 2 = fp32 rows rounded to bf16 values (lm_head).
// Weights / activations are zero-filled device buffers (bandwidth only).  NOTE: back-to-back launches of ONE GEMM re-read
// the same weights, so anything below ~200 MB is served by the Infinity Cache / L2 -- an upper bound, not the in-situ time.
extern "C" int sv_bench_decode_linear(int32_t M, int32_t N, int32_t K, int32_t splitk, int32_t mode, int32_t iters, double* avg_us,
                                      sv_stream stream) {
    if (!avg_us || M < 1 || N < 32 || K < 32 || K % 32 || splitk < 1 || (K / 16) % splitk || iters < 1 || mode < 0 || mode > 2)
        return fail(SV_EINVAL, "sv_bench_decode_linear: bad argument");
    if (mode != 0 && splitk != 1) return fail(SV_EINVAL, "sv_bench_decode_linear: modes 1 and 2 need splitk == 1");
    hipStream_t st = (hipStream_t)stream;
    TmpBufs tmp;
    const int Npad = round_up(N, 32), MT = (M + 31) / 32, R = MT * 32;
    bf16_t *Wp, *xp, *oxp, *bias;
    float* ws;
    SVCHECK(tmp.get(&Wp, (size_t)Npad * K));
    SVCHECK(tmp.get(&xp, (size_t)R * K));
    SVCHECK(tmp.get(&oxp, (size_t)R * Npad));
    SVCHECK(tmp.get(&bias, (size_t)Npad));
    SVCHECK(tmp.get(&ws, (size_t)splitk * R * Npad));
    HIPCHECK(hipMemsetAsync(Wp, 0, (size_t)Npad * K * 2, st));
    HIPCHECK(hipMemsetAsync(xp, 0, (size_t)R * K * 2, st));
    HIPCHECK(hipMemsetAsync(bias, 0, (size_t)Npad * 2, st));
    SkinnyArgs a;
    memset(&a, 0, sizeof(a));
    a.xp = xp; a.Wp = Wp; a.bias = bias; a.MT = MT; a.Npad = Npad; a.K = K; a.splitk = splitk; a.N = N;
    a.out_mode = mode; a.act = mode == 1 ? ACT_GELU_TANH : ACT_NONE;
    a.ws = ws; a.ldws = Npad; a.out_xp = oxp; a.out_KS = Npad / 16; a.out_f32 = ws; a.ldo = Npad; a.round_bf16 = 1;
    hipEvent_t e0, e1;
    HIPCHECK(hipEventCreate(&e0)); HIPCHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch_gemm_skinny(a, st);
    HIPCHECK(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i) launch_gemm_skinny(a, st);
    HIPCHECK(hipEventRecord(e1, st));
    HIPCHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    *avg_us = (double)ms * 1e3 / iters;
    HIPCHECK(hipGetLastError());
    return 0;
}

extern "C" int sv_op_cvt_bf16_hw(const float* x, void* y, int64_t n, sv_stream stream) {
    if (!x || !y || n < 1) return fail(SV_EINVAL, "sv_op_cvt_bf16_hw: bad argument");
    launch_cvt_bf16_hw(x, (bf16_t*)y, (size_t)n, (hipStream_t)stream);
    HIPCHECK(hipGetLastError());
    return 0;
}

extern "C" int sv_op_attention(const void* q, const void* k, const void* v, void* out, int32_t B, int32_t S, int32_t H,
                               int32_t Hkv, int32_t head_dim, int32_t causal, float scale, sv_stream stream) {
    if (!q || !k || !v || !out || B < 1 || S < 1 || H < 1 || Hkv < 1 || H % Hkv) return fail(SV_EINVAL, "sv_op_attention: bad argument");
    if (head_dim != 64 && head_dim != 128) return fail(SV_EINVAL, "head_dim %d unsupported (64|128)", head_dim);
    AttnPrefillArgs a;
    a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v;
    a.q_row_stride = H * head_dim; a.kv_row_stride = Hkv * head_dim; a.q_head_stride = head_dim;
    a.kv_head_stride = head_dim; a.o = (bf16_t*)out; a.o_row_stride = H * head_dim; a.B = B; a.S = S; a.H = H;
    a.head_dim = head_dim; a.kv_group = H / Hkv; a.causal = causal; a.scale = scale;
    launch_attn_prefill(a, (hipStream_t)stream);
    HIPCHECK(hipGetLastError());
    return 0;
}

extern "C" int sv_op_plane_layernorm(const void* x, const void* gamma, const void* beta, void* y, int32_t B, int32_t QD,
                                     float eps, sv_stream stream) {
    if (!x || !gamma || !beta || !y || B < 1 || QD < 8 || QD % 8) return fail(SV_EINVAL, "sv_op_plane_layernorm: bad argument");
    launch_plane_layernorm((const bf16_t*)x, (const bf16_t*)gamma, (const bf16_t*)beta, (bf16_t*)y, B, QD, eps,
                           (hipStream_t)stream);
    HIPCHECK(hipGetLastError());
    return 0;
}

extern "C" int sv_op_argmax(const float* logits, int32_t B, int32_t V, int32_t ld, int32_t* out, sv_stream stream) {
    if (!logits || !out || B < 1 || V < 1 || ld < V || ld % 4) return fail(SV_EINVAL, "sv_op_argmax: bad argument (ld must be a multiple of 4)");
    TmpBufs tmp;
    float* pv; int32_t* pi;
    SVCHECK(tmp.get(&pv, (size_t)B * 8));
    SVCHECK(tmp.get(&pi, (size_t)B * 8));
    launch_argmax(logits, ld, V, out, pv, pi, B, (hipStream_t)stream);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}

extern "C" int sv_op_sample(const float* logits, int32_t B, int32_t V, int32_t ld, float temperature, int32_t top_k,
                            float top_p, uint64_t seed, int32_t step, int32_t* out, sv_stream stream);
extern "C" int sv_op_sample_top_p(const float* logits, int32_t B, int32_t V, int32_t ld, float temperature, float top_p,
                                  uint64_t seed, int32_t step, int32_t* out, sv_stream stream) {
    return sv_op_sample(logits, B, V, ld, temperature, 0, top_p, seed, step, out, stream);
}
extern "C" int sv_op_sample(const float* logits, int32_t B, int32_t V, int32_t ld, float temperature, int32_t top_k,
                            float top_p, uint64_t seed, int32_t step, int32_t* out, sv_stream stream) {
    if (!logits || !out || B < 1 || V < 1 || ld < V || !(temperature > 0.f) || !(top_p > 0.f))
        return fail(SV_EINVAL, "sv_op_sample: bad argument");
    hipStream_t st = (hipStream_t)stream;
    TmpBufs tmp;
    int32_t* dstep;
    SVCHECK(tmp.get(&dstep, 1));
    HIPCHECK(hipMemcpyAsync(dstep, &step, sizeof(int32_t), hipMemcpyHostToDevice, st));
    SampleArgs sa;
    sa.logits = logits; sa.ld = ld; sa.V = V; sa.B = B; sa.temperature = temperature; sa.top_p = top_p; sa.top_k = top_k; sa.seed = seed;
    sa.step = dstep; sa.out = out; sa.scratch = nullptr; sa.seen = nullptr; sa.seen_words = 0; sa.penalty = 1.f;
    launch_sample_top_p(sa, st);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipStreamSynchronize(st));
    return 0;
}
