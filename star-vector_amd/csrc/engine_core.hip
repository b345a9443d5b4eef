// Host driver, part 1 of 5: errors, utility kernels, allocation, decode planning, weight registration, lifecycle and
// weight loading of libstarvector_hip.so (see include/starvector_hip.h for the contract and the reference lines each entry
// point replaces).  The engine owns: repacked weights, workspaces, the paged KV pool and its page allocator.
//   engine_core.hip      this file
//   engine_forward.hip   the op graphs (vision, adapter, prompt pass, decode step) + the forward entry points
//   engine_generate.hip  sv_generate: greedy / sampling loop (hipGraph-captured step), beam search, the beam scorer's ABI
//   engine_cb.hip        continuous batching (sv_cb_*)
//   engine_ops.hip       host-side planning probes, pre-processing ABI, profiling, single-operator test surfaces
#include "engine_internal.h"

#include <algorithm>

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
namespace sveng {
std::string& last_error() {
    static thread_local std::string err;
    return err;
}
int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    last_error() = buf;
    return code;
}
}  // namespace sveng

// ------------------------------------------------------------------------------------------------
// small utility kernels local to the driver
// ------------------------------------------------------------------------------------------------
__global__ void fill_i32_kernel(int32_t* p, int32_t v, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
// HF MinLengthLogitsProcessor: while fewer than `min_new` tokens have been generated the EOS logit of every row is -inf
__global__ void suppress_token_kernel(float* logits, int ld, int token, const int32_t* step, int min_new, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B && *step < min_new) logits[(size_t)b * ld + token] = -INFINITY;
}
__global__ void add_i32_kernel(int32_t* p, int32_t v, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] += v;
}
__global__ void tokens_to_i64_kernel(const int32_t* src, int ld, int64_t* dst, int B, int ncols, int dst_ld) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B * ncols) {
        int b = i / ncols, t = i % ncols;
        dst[(size_t)b * dst_ld + t] = src[(size_t)b * ld + t];
    }
}
__global__ void fill_random_bf16_kernel(bf16_t* p, size_t n, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u ^ (seed * 0x9E3779B9u);
        x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
        p[i] = f2bf(((float)(x >> 8) * (1.0f / 8388608.0f)) - 1.0f);
    }
}
// row-major [M][K] -> skinny fragment order
__global__ void pack_rows_kernel(const bf16_t* x, int ldx, bf16_t* xp, int M, int K) {
    const int NC = K >> 3;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < M * NC; i += gridDim.x * blockDim.x) {
        const int c = i % NC, m = i / NC;
        *reinterpret_cast<uint4*>(xp + xp_index(m >> 5, K >> 4, m & 31, c * 8)) =
            *reinterpret_cast<const uint4*>(x + (size_t)m * ldx + c * 8);
    }
}
// fragment order -> row-major [M][K]
__global__ void unpack_rows_kernel(const bf16_t* xp, bf16_t* x, int ldx, int M, int K) {
    const int NC = K >> 3;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < M * NC; i += gridDim.x * blockDim.x) {
        const int c = i % NC, m = i / NC;
        *reinterpret_cast<uint4*>(x + (size_t)m * ldx + c * 8) =
            *reinterpret_cast<const uint4*>(xp + xp_index(m >> 5, K >> 4, m & 31, c * 8));
    }
}
// split-K slabs -> fp32 rows (+ bias), slab order (test surface of the skinny GEMM)
__global__ void reduce_partials_kernel(const float* ws, int splitk, int rows_ws, int ldws, const bf16_t* bias,
                                       float* y, int M, int N) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < M * N; i += gridDim.x * blockDim.x) {
        const int n = i % N, m = i / N;
        float v = 0.f;
        for (int s = 0; s < splitk; ++s) v += ws[((size_t)s * rows_ws + m) * ldws + n];
        if (bias) v += bf2f(bias[n]);
        y[(size_t)m * N + n] = v;
    }
}
// the generation state of a call in ONE launch (it was two fills and four memsets, 5 us each on the way to the first token): positions = pos0,
// unfinished = 1, {step, done, n_emitted} = 0, the folded arg-max key slots = 0 (n_amax = 0: not this call's selection)
__global__ void gen_state_init_kernel(int32_t* positions, int32_t pos0, int32_t* unfinished, int B, int32_t* state3, unsigned long long* amax, int n_amax) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) { positions[i] = pos0; unfinished[i] = 1; }
    if (i < 3) state3[i] = 0;
    if (i < n_amax) amax[i] = 0ull;
}
namespace sveng {
void gen_state_init(int32_t* positions, int32_t pos0, int32_t* unfinished, int B, int32_t* state3, unsigned long long* amax, int n_amax, hipStream_t st) {
    const int n = B > n_amax ? (B > 3 ? B : 3) : (n_amax > 3 ? n_amax : 3);
    gen_state_init_kernel<<<(n + 255) / 256, 256, 0, st>>>(positions, pos0, unfinished, B, state3, amax, n_amax);
}
void fill_i32(int32_t* p, int32_t v, int n, hipStream_t st) { fill_i32_kernel<<<(n + 63) / 64, 64, 0, st>>>(p, v, n); }
void add_i32(int32_t* p, int32_t v, int n, hipStream_t st) { add_i32_kernel<<<(n + 63) / 64, 64, 0, st>>>(p, v, n); }
void suppress_token(float* logits, int ld, int token, const int32_t* step, int min_new, int B, hipStream_t st) {
    suppress_token_kernel<<<(B + 63) / 64, 64, 0, st>>>(logits, ld, token, step, min_new, B);
}
void tokens_to_i64(const int32_t* src, int ld, int64_t* dst, int B, int ncols, int dst_ld, hipStream_t st) {
    tokens_to_i64_kernel<<<(B * ncols + 255) / 256, 256, 0, st>>>(src, ld, dst, B, ncols, dst_ld);
}
void fill_random_bf16(bf16_t* p, size_t n, unsigned seed, int blocks, hipStream_t st) {
    fill_random_bf16_kernel<<<blocks, 256, 0, st>>>(p, n, seed);
}
void pack_rows(const bf16_t* x, int ldx, bf16_t* xp, int M, int K, hipStream_t st) {
    pack_rows_kernel<<<(M * (K / 8) + 255) / 256, 256, 0, st>>>(x, ldx, xp, M, K);
}
void unpack_rows(const bf16_t* xp, bf16_t* x, int ldx, int M, int K, hipStream_t st) {
    unpack_rows_kernel<<<(M * (K / 8) + 255) / 256, 256, 0, st>>>(xp, x, ldx, M, K);
}
void reduce_partials(const float* ws, int splitk, int rows_ws, int ldws, const bf16_t* bias, float* y, int M, int N, hipStream_t st) {
    reduce_partials_kernel<<<(M * N + 255) / 256, 256, 0, st>>>(ws, splitk, rows_ws, ldws, bias, y, M, N);
}

int dev_alloc(sv_engine* e, void** p, size_t bytes, bool zero) {
    if (bytes == 0) bytes = 16;
    hipError_t r = hipMalloc(p, bytes);
    if (r != hipSuccess) return fail(SV_ENOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(r));
    if (zero) {
        r = hipMemset(*p, 0, bytes);
        if (r != hipSuccess) return fail(SV_EHIP, "hipMemset failed: %s", hipGetErrorString(r));
    }
    e->allocs.push_back(*p);
    return 0;
}
}  // namespace sveng

static void reg_linear(sv_engine* e, const std::string& base, Linear* l, int N, int K, int Kalign, bool has_bias) {
    l->N = N; l->K = K; l->Npad = round_up(N, 32); l->Kpad = round_up(K, Kalign);
    Slot w; w.kind = SLOT_LINEAR_W; w.lin = l; w.numel = (size_t)N * K;
    e->slots[base + (base.back() == '.' ? "weight" : "")] = w;
    if (has_bias) {
        Slot b; b.kind = SLOT_RAW; b.raw = &l->bias; b.numel = (size_t)N;
        e->slots[base + "bias"] = b;
    }
}
// one part of a fused projection: rows [row_off, row_off + rows) of `l` (weight) and of its bias
static void reg_linear_part(sv_engine* e, const std::string& base, Linear* l, int row_off, int rows, int K) {
    Slot w; w.kind = SLOT_LINEAR_W; w.lin = l; w.numel = (size_t)rows * K; w.row_off = row_off; w.part_rows = rows;
    e->slots[base + "weight"] = w;
    Slot b; b.kind = SLOT_RAW; b.raw = &l->bias; b.numel = (size_t)rows; b.row_off = row_off; b.part_rows = rows;
    e->slots[base + "bias"] = b;
}
static void reg_raw(sv_engine* e, const std::string& name, bf16_t** p, size_t numel, bool required = true) {
    Slot s; s.kind = SLOT_RAW; s.raw = p; s.numel = numel; s.required = required;
    e->slots[name] = s;
}
static void reg_ln(sv_engine* e, const std::string& base, LNp* ln, size_t n) {
    reg_raw(e, base + "weight", &ln->g, n);
    reg_raw(e, base + "bias", &ln->b, n);
}

// Split-K factor of a decode GEMM whose output goes to fp32 slabs (the consumer sums them in slab order).  Blocks are
// one-per-CU-sized and a CU streams HBM at a capped rate, so what matters is how evenly NT * s equal blocks fall on the chip:
// the time is ceil(NT * s / #CU) rounds of one block, i.e. the busiest CU's share -- NOT the average (8B down-projection at
// split 2: 288 blocks = one full round + 32 blocks at 2 x 590 KB per busy CU: 45.8 us measured against 31 us for the bytes).
// Pick the s <= 8 (the slab buffer and the consumers' limit) with the best fill, preferring fewer slabs on near-ties; each
// block keeps >= 16 k-steps so that its 8 waves still have a stream to pipeline.  `legacy` = the round 1-2 rule (smallest
// power of two that reaches one block per CU) kept for the A/B mask.
static int pick_splitk(int n_tiles, int KS, int num_cus, bool fp8, bool legacy) {
    // small GEMMs (StarVector-1B's c_attn / attention c_proj: < 40 KB per CU) are one latency-bound round trip per wave: the fill
    // model does not describe them, and more slabs only cost their consumer -> the old rule
    if (legacy || (long)n_tiles * KS < 24L * 1024) {
        int want = (256 + n_tiles - 1) / n_tiles;
        int s = 1;
        while (s < want && s < 8) s <<= 1;
        while (s > 1 && (KS % s) != 0) s >>= 1;
        return s;
    }
    int best = 1;
    double best_cost = 1e30;
    for (int s = 1; s <= 8; ++s) {
        if (KS % s) continue;
        const int per = KS / s;
        if (s > 1 && per < 16) continue;
        if (per % (fp8 ? 4 : 2)) continue;                       // the kernels cut a block's K over 2..16 waves (fp8: pairs of k-steps)
        const long nb = (long)n_tiles * s;
        const long rounds = (nb + num_cus - 1) / num_cus;
        const double fill = (double)nb / (double)(rounds * num_cus);          // 1 = every CU equally loaded
        const double cost = 1.0 / fill + 0.015 * s;                            // slabs cost the consumer a little
        if (cost < best_cost - 1e-9) { best_cost = cost; best = s; }
    }
    return best;
}

// 33..64 rows (two row tiles per block): at that height every block re-reads 2 (bf16) / 4 (fp8) activation bytes per weight byte out
// of L2 and the L2 -> CU side bounds the launch (tools/diag/mem_mix.hip), so a block may carry two or three column tiles per loaded
// activation fragment.  (column tiles per block, split-K) are picked together: modelled time = the busiest CU's bytes (weights + activations
// over its blocks) relative to an even spread, plus the slab cost.  Only shapes the two-row-tile kernel takes are candidates.
void sveng::pick_decode_plan(const Linear& l, int MT, int num_cus, bool fp8, bool legacy, bool whole_k, int* splitk, int* col_tiles) {
    const int tiles = l.Npad / 32, KS = l.Kpad / 16;
    *col_tiles = 1;
    if (MT != 2 || legacy || (long)tiles * KS < 24L * 1024) {
        *splitk = whole_k ? 1 : pick_splitk(tiles, KS, num_cus, fp8, legacy);
        return;
    }
    const double wb = fp8 ? 512.0 : 1024.0, ab = 1024.0 * MT;
    const double ideal = (double)tiles * KS * (wb + ab) / num_cus;
    double best_cost = 1e30;
    int best_s = 0, best_nt = 1;
    for (int nt = 1; nt <= 3; ++nt)
        for (int s = 1; s <= (whole_k ? 1 : 8); ++s) {
            if (KS % s) continue;
            const int per = KS / s;
            if (s > 1 && per < 16) continue;
            int waves = 0, two = 0;
            skinny_plan(l.Npad, l.Kpad, s, fp8 ? 1 : 0, MT, &waves, &two);
            if (!two || (nt >= 2 && waves != 8)) continue;
            const long nb = (long)((tiles + nt - 1) / nt) * s;
            const long rounds = (nb + num_cus - 1) / num_cus;
            const double cost = (double)rounds * per * (nt * wb + ab) / ideal + 0.015 * s;
            if (cost < best_cost - 1e-9) { best_cost = cost; best_s = s; best_nt = nt; }
        }
    if (!best_s) { *splitk = whole_k ? 1 : pick_splitk(tiles, KS, num_cus, fp8, legacy); return; }
    *splitk = best_s;
    *col_tiles = best_nt;
}

// StarVector-8B key names: HF SiglipVisionTransformer under model.image_encoder.visual_encoder.*
// (image_encoder.py:41-43) and HF Starcoder2ForCausalLM under model.svg_transformer.transformer.*
// (llm/starcoder2.py:22-27).  q|k|v projections are separate tensors there; they are packed side by side into
// ONE fused projection here (parts with a row offset), so the kernels are the same as for v1.
static void register_v2(sv_engine* e) {
    const sv_config& c = e->cfg;
    const int Dv = c.vit_width, D = c.hidden, F = c.n_inner, dh = e->dh, nkv = e->nkv;
    const std::string pv = "model.image_encoder.visual_encoder.";
    e->conv1.N = Dv; e->conv1.K = e->conv_K; e->conv1.Npad = round_up(Dv, 32); e->conv1.Kpad = round_up(e->conv_K, 64);
    { Slot w; w.kind = SLOT_LINEAR_W; w.lin = &e->conv1; w.numel = (size_t)Dv * e->conv_K;
      e->slots[pv + "embeddings.patch_embedding.weight"] = w; }
    reg_raw(e, pv + "embeddings.patch_embedding.bias", &e->conv1.bias, Dv);
    reg_raw(e, pv + "embeddings.position_embedding.weight", &e->pos, (size_t)e->NP * Dv);
    e->vit.resize(c.vit_layers);
    for (int i = 0; i < c.vit_layers; ++i) {
        const std::string p = pv + "encoder.layers." + std::to_string(i) + ".";
        VitLayer& L = e->vit[i];
        reg_ln(e, p + "layer_norm1.", &L.ln1, Dv);
        reg_ln(e, p + "layer_norm2.", &L.ln2, Dv);
        L.in_proj.N = 3 * Dv; L.in_proj.K = Dv; L.in_proj.Npad = 3 * Dv; L.in_proj.Kpad = Dv;
        reg_linear_part(e, p + "self_attn.q_proj.", &L.in_proj, 0, Dv, Dv);
        reg_linear_part(e, p + "self_attn.k_proj.", &L.in_proj, Dv, Dv, Dv);
        reg_linear_part(e, p + "self_attn.v_proj.", &L.in_proj, 2 * Dv, Dv, Dv);
        reg_linear(e, p + "self_attn.out_proj.", &L.out_proj, Dv, Dv, 64, true);
        reg_linear(e, p + "mlp.fc1.", &L.c_fc, e->vit_F, Dv, 64, true);
        reg_linear(e, p + "mlp.fc2.", &L.c_proj, Dv, e->vit_F, 64, true);
    }
    reg_ln(e, pv + "post_layernorm.", &e->ln_vision, Dv);

    const std::string pa = "model.image_projection.";
    reg_linear(e, pa + "c_fc.", &e->ad_fc, 2 * Dv, Dv, 64, true);
    reg_linear(e, pa + "c_proj.", &e->ad_proj, D, 2 * Dv, 64, true);
    if (c.adapter_norm == SV_NORM_LAYER) {
        reg_raw(e, pa + "norm.weight", &e->ad_w, (size_t)e->T * D);
        reg_raw(e, pa + "norm.bias", &e->ad_b, (size_t)e->T * D);
    } else {
        reg_raw(e, pa + "norm.weight", &e->ad_w, e->T);
        reg_raw(e, pa + "norm.bias", &e->ad_b, e->T);
        reg_raw(e, pa + "norm.running_mean", &e->ad_rm, e->T);
        reg_raw(e, pa + "norm.running_var", &e->ad_rv, e->T);
    }

    const std::string pd = "model.svg_transformer.transformer.model.";
    { Slot s; s.kind = SLOT_WTE; s.raw = &e->wte; s.numel = (size_t)c.vocab * D; e->slots[pd + "embed_tokens.weight"] = s; }
    e->lm_head.N = c.vocab; e->lm_head.K = D; e->lm_head.Npad = round_up(c.vocab, 32); e->lm_head.Kpad = D;
    { Slot s; s.kind = SLOT_LINEAR_W; s.lin = &e->lm_head; s.numel = (size_t)c.vocab * D; s.required = false;
      e->slots["model.svg_transformer.transformer.lm_head.weight"] = s; }
    e->dec.resize(c.n_layer);
    const int QD = c.n_head * dh, KD = nkv * dh;
    for (int i = 0; i < c.n_layer; ++i) {
        const std::string p = pd + "layers." + std::to_string(i) + ".";
        DecLayer& L = e->dec[i];
        reg_ln(e, p + "input_layernorm.", &L.ln1, D);
        reg_ln(e, p + "post_attention_layernorm.", &L.ln2, D);
        L.c_attn.N = e->QKV; L.c_attn.K = D; L.c_attn.Npad = round_up(e->QKV, 32); L.c_attn.Kpad = D;
        reg_linear_part(e, p + "self_attn.q_proj.", &L.c_attn, 0, QD, D);
        reg_linear_part(e, p + "self_attn.k_proj.", &L.c_attn, QD, KD, D);
        reg_linear_part(e, p + "self_attn.v_proj.", &L.c_attn, QD + KD, KD, D);
        reg_linear(e, p + "self_attn.o_proj.", &L.c_proj, D, QD, 64, true);
        reg_linear(e, p + "mlp.c_fc.", &L.c_fc, F, D, 64, true);
        reg_linear(e, p + "mlp.c_proj.", &L.c_proj2, D, F, 64, true);
    }
    reg_ln(e, pd + "norm.", &e->ln_f, D);
}

// ------------------------------------------------------------------------------------------------
// C ABI: lifecycle
// ------------------------------------------------------------------------------------------------
extern "C" int sv_abi_version(void) { return SV_ABI_VERSION; }
extern "C" const char* sv_last_error(void) { return g_err.c_str(); }

extern "C" void sv_config_default_1b(sv_config* c) {
    c->image_size = 224; c->patch_size = 14; c->vit_width = 1024; c->vit_layers = 23; c->vit_heads = 16;
    c->adapter_norm = SV_NORM_LAYER; c->hidden = 2048; c->n_layer = 24; c->n_head = 16; c->n_inner = 8192;
    c->vocab = 49156; c->n_positions = 8192; c->max_batch = 32; c->max_seq_len = 2048; c->ln_eps = 1e-5f;
    c->device = 0;
    c->arch = SV_ARCH_V1; c->n_kv_head = 1; c->rope_theta = 0.f; c->vit_mlp = 4096; c->vit_eps = 1e-5f;
    c->sliding_window = 0; c->weight_dtype = SV_WEIGHT_BF16; c->exclusive_device = 0;
}

extern "C" void sv_config_default_8b(sv_config* c) {
    // siglip_384 = google/siglip-large-patch16-384 (image_encoder.py:35-36), bigcode/starcoder2-7b (llm/starcoder2.py:22)
    c->image_size = 384; c->patch_size = 16; c->vit_width = 1024; c->vit_layers = 24; c->vit_heads = 16;
    c->adapter_norm = SV_NORM_LAYER; c->hidden = 4608; c->n_layer = 32; c->n_head = 36; c->n_inner = 18432;
    c->vocab = 49152 + 5; c->n_positions = 16384; c->max_batch = 16; c->max_seq_len = 4096; c->ln_eps = 1e-5f;
    c->device = 0;
    c->arch = SV_ARCH_V2; c->n_kv_head = 4; c->rope_theta = 1e6f; c->vit_mlp = 4096; c->vit_eps = 1e-6f;
    c->sliding_window = 4096; c->weight_dtype = SV_WEIGHT_BF16; c->exclusive_device = 0;
}

extern "C" int sv_destroy(sv_engine* e) {
    if (!e) return 0;
    (void)hipSetDevice(e->cfg.device);
    (void)hipDeviceSynchronize();
    for (void* p : e->allocs) (void)hipFree(p);
    e->beam.destroy();
    for (auto& kv : e->cb_graphs) { if (kv.second.second) (void)hipGraphExecDestroy(kv.second.second); if (kv.second.first) (void)hipGraphDestroy(kv.second.first); }
    if (e->gen_gexec) (void)hipGraphExecDestroy(e->gen_gexec);
    if (e->gen_graph) (void)hipGraphDestroy(e->gen_graph);
    if (e->gen_gexec_multi) (void)hipGraphExecDestroy(e->gen_gexec_multi);
    if (e->gen_graph_multi) (void)hipGraphDestroy(e->gen_graph_multi);
    if (e->score_ws) (void)hipFree(e->score_ws);
    if (e->h_flags) (void)hipHostFree(e->h_flags);
    if (e->h_table) (void)hipHostFree(e->h_table);
    if (e->table_ev) (void)hipEventDestroy(e->table_ev);
    for (hipEvent_t ev : e->prof_ev) (void)hipEventDestroy(ev);
    if (e->gen_event) (void)hipEventDestroy(e->gen_event);
    if (e->gen_stream) (void)hipStreamDestroy(e->gen_stream);
    if (e->tenant_stream) (void)hipStreamDestroy(e->tenant_stream);
    delete e;
    return 0;
}

extern "C" int sv_create(const sv_config* cfg, sv_engine** out) {
    if (!cfg || !out) return fail(SV_EINVAL, "sv_create: null argument");
    const sv_config& c = *cfg;
    if (c.image_size % c.patch_size) return fail(SV_EINVAL, "image_size %% patch_size != 0");
    if (c.vit_width % c.vit_heads || c.hidden % c.n_head) return fail(SV_EINVAL, "width %% heads != 0");
    const int vdh = c.vit_width / c.vit_heads, dh = c.hidden / c.n_head;
    if (vdh != 64 && vdh != 128) return fail(SV_EINVAL, "ViT head_dim %d unsupported (64|128)", vdh);
    if (dh != 64 && dh != 128) return fail(SV_EINVAL, "decoder head_dim %d unsupported (64|128)", dh);
    const bool v2 = c.arch == SV_ARCH_V2;
    const int nkv = v2 ? c.n_kv_head : 1;
    if (c.arch != SV_ARCH_V1 && c.arch != SV_ARCH_V2) return fail(SV_EINVAL, "unknown arch %d", c.arch);
    if (nkv < 1 || c.n_head % nkv) return fail(SV_EINVAL, "n_head %% n_kv_head != 0");
    if (c.n_head / nkv > 16) return fail(SV_EINVAL, "decode attention supports <= 16 query heads per KV head");
    if (c.vit_width % 64 || c.hidden % 64 || c.n_inner % 64) return fail(SV_EINVAL, "dims must be multiples of 64");
    if (v2 && (c.vit_mlp < 64 || c.vit_mlp % 64 || !(c.rope_theta > 1.f))) return fail(SV_EINVAL, "bad vit_mlp / rope_theta");
    if (c.weight_dtype != SV_WEIGHT_BF16 && c.weight_dtype != SV_WEIGHT_FP8_E4M3)
        return fail(SV_EINVAL, "weight_dtype must be SV_WEIGHT_BF16 (0) or SV_WEIGHT_FP8_E4M3 (1)");
    if (c.sliding_window < 0 || (!v2 && c.sliding_window != 0))
        return fail(SV_EINVAL, "sliding_window must be >= 0 (and 0 for the GPTBigCode decoder)");
    if (c.max_batch < 1 || c.max_seq_len < 2 || c.max_seq_len > c.n_positions)
        return fail(SV_EINVAL, "bad max_batch / max_seq_len");
    hipError_t r = hipSetDevice(c.device);
    if (r != hipSuccess) return fail(SV_EHIP, "hipSetDevice(%d): %s", c.device, hipGetErrorString(r));

    if (int ar = init_attention_kernels()) return fail(SV_EHIP, "hipFuncSetAttribute: %s", hipGetErrorString((hipError_t)ar));
    if (int ar = init_gemm_kernels()) return fail(SV_EHIP, "hipFuncSetAttribute: %s", hipGetErrorString((hipError_t)ar));
    if (int ar = init_cols_kernels()) return fail(SV_EHIP, "hipFuncSetAttribute: %s", hipGetErrorString((hipError_t)ar));

    sv_engine* e = new sv_engine();
    e->cfg = c;
    e->vdh = vdh; e->dh = dh; e->v2 = v2; e->nkv = nkv;
    const int G = c.image_size / c.patch_size;
    e->NP = G * G; e->T = e->NP + (v2 ? 0 : 1);
    const int Dv = c.vit_width, D = c.hidden, F = c.n_inner;
    e->conv_K = 3 * c.patch_size * c.patch_size;
    e->QKV = c.n_head * dh + 2 * nkv * dh;
    e->vit_F = v2 ? c.vit_mlp : 4 * Dv;
    if (v2) register_v2(e);
    else {

    const std::string pv = "model.image_encoder.visual_encoder.";
    reg_linear(e, pv + "conv1.", &e->conv1, Dv, e->conv_K, 64, false);
    reg_raw(e, pv + "class_embedding", &e->cls, Dv);
    reg_raw(e, pv + "positional_embedding", &e->pos, (size_t)e->T * Dv);
    reg_ln(e, pv + "ln_pre.", &e->ln_pre, Dv);
    e->vit.resize(c.vit_layers);
    for (int i = 0; i < c.vit_layers; ++i) {
        const std::string p = pv + "transformer.resblocks." + std::to_string(i) + ".";
        VitLayer& L = e->vit[i];
        reg_ln(e, p + "ln_1.", &L.ln1, Dv);
        reg_ln(e, p + "ln_2.", &L.ln2, Dv);
        // nn.MultiheadAttention packs q|k|v into in_proj_weight / in_proj_bias (no '.weight' suffix)
        L.in_proj.N = 3 * Dv; L.in_proj.K = Dv; L.in_proj.Npad = 3 * Dv; L.in_proj.Kpad = Dv;
        { Slot w; w.kind = SLOT_LINEAR_W; w.lin = &L.in_proj; w.numel = (size_t)3 * Dv * Dv; e->slots[p + "attn.in_proj_weight"] = w; }
        reg_raw(e, p + "attn.in_proj_bias", &L.in_proj.bias, (size_t)3 * Dv);
        reg_linear(e, p + "attn.out_proj.", &L.out_proj, Dv, Dv, 64, true);
        reg_linear(e, p + "mlp.c_fc.", &L.c_fc, 4 * Dv, Dv, 64, true);
        reg_linear(e, p + "mlp.c_proj.", &L.c_proj, Dv, 4 * Dv, 64, true);
    }
    reg_ln(e, "model.image_encoder.ln_vision.", &e->ln_vision, Dv);

    const std::string pa = "model.image_projection.";
    reg_linear(e, pa + "c_fc.", &e->ad_fc, 2 * Dv, Dv, 64, true);
    reg_linear(e, pa + "c_proj.", &e->ad_proj, D, 2 * Dv, 64, true);
    if (c.adapter_norm == SV_NORM_LAYER) {
        reg_raw(e, pa + "norm.weight", &e->ad_w, (size_t)e->T * D);
        reg_raw(e, pa + "norm.bias", &e->ad_b, (size_t)e->T * D);
    } else {
        reg_raw(e, pa + "norm.weight", &e->ad_w, e->T);
        reg_raw(e, pa + "norm.bias", &e->ad_b, e->T);
        reg_raw(e, pa + "norm.running_mean", &e->ad_rm, e->T);
        reg_raw(e, pa + "norm.running_var", &e->ad_rv, e->T);
    }

    const std::string pd = "model.svg_transformer.transformer.transformer.";
    { Slot s; s.kind = SLOT_WTE; s.raw = &e->wte; s.numel = (size_t)c.vocab * D; e->slots[pd + "wte.weight"] = s; }
    reg_raw(e, pd + "wpe.weight", &e->wpe, (size_t)c.n_positions * D);
    e->lm_head.N = c.vocab; e->lm_head.K = D; e->lm_head.Npad = round_up(c.vocab, 32); e->lm_head.Kpad = D;
    { Slot s; s.kind = SLOT_LINEAR_W; s.lin = &e->lm_head; s.numel = (size_t)c.vocab * D; s.required = false;
      e->slots["model.svg_transformer.transformer.lm_head.weight"] = s; }
    e->dec.resize(c.n_layer);
    for (int i = 0; i < c.n_layer; ++i) {
        const std::string p = pd + "h." + std::to_string(i) + ".";
        DecLayer& L = e->dec[i];
        reg_ln(e, p + "ln_1.", &L.ln1, D);
        reg_ln(e, p + "ln_2.", &L.ln2, D);
        reg_linear(e, p + "attn.c_attn.", &L.c_attn, D + 2 * dh, D, 64, true);
        reg_linear(e, p + "attn.c_proj.", &L.c_proj, D, D, 64, true);
        reg_linear(e, p + "mlp.c_fc.", &L.c_fc, F, D, 64, true);
        reg_linear(e, p + "mlp.c_proj.", &L.c_proj2, D, F, 64, true);
    }
    reg_ln(e, pd + "ln_f.", &e->ln_f, D);
    }   // v1 registration

    {
        // decode-path split-K: narrow outputs split K across blocks into fp32 slabs that the next kernel (attention / row update)
        // sums in slab order; c_fc keeps the whole K (its bias + GELU epilogue needs the finished sum)
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, c.device) == hipSuccess && prop.multiProcessorCount > 0) e->num_cus = prop.multiProcessorCount;
        const bool fp8 = c.weight_dtype == SV_WEIGHT_FP8_E4M3;
        const bool legacy = getenv("SV_EXP") && (atoi(getenv("SV_EXP")) & 8);          // A/B: the round 1-2 split rule
        if (fp8) e->lm_head.fp8 = true;                  // decoder Linears + lm_head stream as fp8 at decode time
        { int sk1 = 1; pick_decode_plan(e->lm_head, (c.max_batch + 31) / 32, e->num_cus, fp8, legacy, true, &sk1, &e->lm_head.col_tiles); }
        const bool plan_log = getenv("SV_GEMM_AUTOTUNE_LOG") && atoi(getenv("SV_GEMM_AUTOTUNE_LOG"));
        for (DecLayer& L : e->dec) {
            Linear* ls[4] = {&L.c_attn, &L.c_proj, &L.c_fc, &L.c_proj2};
            for (Linear* l : ls) {
                l->fp8 = fp8;
                const int KS = l->Kpad / 16;
                pick_decode_plan(*l, (c.max_batch + 31) / 32, e->num_cus, fp8, legacy, l == &L.c_fc, &l->splitk, &l->col_tiles);
                if (fp8) {
                    // the fp8 kernel wants an even number (>= 2 per wave pair) of k-steps per wave: shrink split-K until it fits
                    while (l->splitk > 1 && (KS % l->splitk != 0 || (KS / l->splitk) % 4 != 0)) --l->splitk;
                    if ((KS / l->splitk) % 4 != 0) {
                        const int code = fail(SV_ENOTSUP, "fp8 weights: K=%d with split-K %d has no fp8 decode kernel", l->Kpad, l->splitk);
                        sv_destroy(e);
                        return code;
                    }
                }
                if (plan_log && &L == &e->dec[0])
                    fprintf(stderr, "[starvector_hip] decode plan N=%d K=%d rows<=%d %s: split-K %d, column tiles per block %d\n", l->Npad, l->Kpad,
                            32 * ((c.max_batch + 31) / 32), fp8 ? "fp8" : "bf16", l->splitk, l->col_tiles);
            }
        }
        if (plan_log)
            fprintf(stderr, "[starvector_hip] decode plan lm_head N=%d K=%d: column tiles per block %d\n", e->lm_head.Npad, e->lm_head.Kpad,
                    e->lm_head.col_tiles);
    }

    // ---- workspaces ----
    int rc = 0;
    const size_t Mv = (size_t)c.max_batch * e->T;
    const size_t Mp = (size_t)c.max_batch * e->NP;
#define A(call) if (!rc) rc = (call)
    A(dalloc(e, &e->patches, Mp * e->conv1.Kpad));
    A(dalloc(e, &e->patch_out, Mp * Dv));
    A(dalloc(e, &e->vx, Mv * Dv));
    A(dalloc(e, &e->vln, Mv * Dv));
    A(dalloc(e, &e->vqkv, Mv * 3 * Dv));
    A(dalloc(e, &e->vattn, Mv * Dv));
    A(dalloc(e, &e->vmlp, Mv * e->vit_F));
    A(dalloc(e, &e->a1, Mv * 2 * Dv));
    A(dalloc(e, &e->a2, Mv * D));

    e->MT = (c.max_batch + 31) / 32;
    const size_t R = (size_t)e->MT * 32;
    e->Vpad = e->lm_head.Npad;
    e->ldws = round_up(e->QKV, 32);
    if (e->ldws < D) e->ldws = D;
    A(dalloc(e, &e->h_dec, R * D));
    A(dalloc(e, &e->h_xp, R * D));
    A(dalloc(e, &e->hl, R * D));
    A(dalloc(e, &e->xp_a, R * D));
    A(dalloc(e, &e->xp_f, R * D));
    A(dalloc(e, &e->rc_dbg, 8));
    A(dalloc(e, &e->xp_attn, R * D));
    A(dalloc(e, &e->xp_mlp, R * F));
    A(dalloc(e, &e->ws, (size_t)8 * R * e->ldws));
    A(dalloc(e, &e->ws2, (size_t)8 * R * e->ldws));
    A(dalloc(e, &e->logits, R * e->Vpad));
    A(dalloc(e, &e->sample_scratch, R * 4));
    A(dalloc(e, &e->attn_part, R * nkv * attn_decode_part_floats(dh)));
    A(dalloc(e, &e->attn_cnt, R * nkv));
    e->seen_words = e->Vpad / 32;
    A(dalloc(e, &e->seen, R * (size_t)e->seen_words));
    A(dalloc(e, &e->am_val, R * 8));
    A(dalloc(e, &e->am_idx, R * 8));
    A(dalloc(e, &e->amax, (size_t)64 * SV_AMAX_STRIDE));
    A(dalloc(e, &e->cur_tok, R));
    A(dalloc(e, &e->next_tok, R));
    A(dalloc(e, &e->unfinished, R));
    A(dalloc(e, &e->positions, R));
    e->out_ld = c.max_seq_len;
    A(dalloc(e, &e->out_tok, (size_t)c.max_batch * e->out_ld));
    // one 16-byte block {step, done, n_emitted, bad}: the host reads the four in ONE copy (engine_generate.hip)
    A(dalloc(e, &e->d_step, 4));
    e->d_done = e->d_step + 1;
    e->d_nemit = e->d_step + 2;
    e->d_bad = e->d_step + 3;         // raised by the selection kernels when a row has no finite logit
    A(dalloc(e, &e->d_stop, 64));
    A(dalloc(e, &e->tail_ws, (size_t)SV_TAIL_TILES * 4 * 16 * 64));
    A(dalloc(e, &e->tail_cnt, (size_t)SV_TAIL_TILES));          // zeroed: the tickets re-arm themselves
    A(dalloc(e, &e->fin_cnt, 64));                               // zeroed: SkinnyArgs::fin_cnt

    e->page_bytes = kv_page_bytes(dh);
    e->pages_per_seq = (c.max_seq_len + SV_PAGE_TOKENS - 1) / SV_PAGE_TOKENS;
    e->num_pages = c.max_batch * e->pages_per_seq;
    e->trash_page = e->num_pages;                 // one page past the allocatable ones: free slots of a continuous batch write there
    e->kv_head_stride = (size_t)(e->num_pages + 1) * e->page_bytes;
    e->layer_stride = e->kv_head_stride * nkv;
    A(dev_alloc(e, reinterpret_cast<void**>(&e->kv_pool), e->layer_stride * c.n_layer, true));
    A(dalloc(e, &e->block_table, (size_t)c.max_batch * e->pages_per_seq));
    A(dalloc(e, &e->cb_table_pf, (size_t)c.max_batch * e->pages_per_seq));
    A(dalloc(e, &e->cb_slots, (size_t)R));
    A(dalloc(e, &e->cb_map, (size_t)R));
    A(dalloc(e, &e->cb_nlive, 4));
    A(dalloc(e, &e->cb_events, 4));
    e->cb_used.assign(c.max_batch, 0);
    e->cb_pages.assign(c.max_batch, {});
#undef A
    if (!rc) {
        hipError_t hr = hipHostMalloc(reinterpret_cast<void**>(&e->h_flags), 64, hipHostMallocDefault);
        if (hr != hipSuccess) rc = fail(SV_ENOMEM, "hipHostMalloc: %s", hipGetErrorString(hr));
    }
    if (!rc) {
        // the block table's host image, pinned: assign_pages uploads it without a stream synchronise (an event guards its reuse)
        hipError_t hr = hipHostMalloc(reinterpret_cast<void**>(&e->h_table), (size_t)c.max_batch * e->pages_per_seq * sizeof(int32_t), hipHostMallocDefault);
        if (hr == hipSuccess) hr = hipEventCreateWithFlags(&e->table_ev, hipEventDisableTiming);
        if (hr != hipSuccess) rc = fail(SV_ENOMEM, "hipHostMalloc / hipEventCreate (block table image): %s", hipGetErrorString(hr));
    }
    if (!rc) {
        hipError_t hr = hipStreamCreateWithFlags(&e->gen_stream, hipStreamNonBlocking);
        if (hr == hipSuccess) hr = hipEventCreateWithFlags(&e->gen_event, hipEventDisableTiming);
        if (hr != hipSuccess) rc = fail(SV_EHIP, "stream/event creation: %s", hipGetErrorString(hr));
    }
    if (!rc && v2) {
        // rotary tables, computed in float like the reference's Starcoder2RotaryEmbedding and rounded to bf16
        // like its `cos.to(dtype=x.dtype)` (values kept in fp32 storage)
        const int half = dh / 2, npos = c.max_seq_len;
        std::vector<float> hc((size_t)npos * half), hs((size_t)npos * half);
        for (int i = 0; i < half; ++i) {
            const float inv = 1.0f / powf(c.rope_theta, (float)(2 * i) / (float)dh);
            for (int p_ = 0; p_ < npos; ++p_) {
                const float fr = (float)p_ * inv;
                auto bfr = [](float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); u &= 0xffff0000u; float r; memcpy(&r, &u, 4); return r; };
                hc[(size_t)p_ * half + i] = bfr(cosf(fr));
                hs[(size_t)p_ * half + i] = bfr(sinf(fr));
            }
        }
        rc = dalloc(e, &e->rope_cos, hc.size(), false);
        if (!rc) rc = dalloc(e, &e->rope_sin, hs.size(), false);
        if (!rc && (hipMemcpy(e->rope_cos, hc.data(), hc.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
                    hipMemcpy(e->rope_sin, hs.data(), hs.size() * 4, hipMemcpyHostToDevice) != hipSuccess))
            rc = fail(SV_EHIP, "rope table upload failed");
    }
    if (getenv("SV_EXP")) e->exp = atoi(getenv("SV_EXP"));
    set_mt2x((e->exp & 131072) ? 0 : (e->exp & 262144) ? 2 : (e->exp & 524288) ? 3 : 1);      // 33..64-row decode GEMM form (process-wide: gemm.hip g_mt2x)
    if (!rc && getenv("SV_ATTN_TRACE")) rc = dalloc(e, &e->attn_trace, R * (size_t)nkv * 16 * 16);
    // 6 launches per layer (decode_cols.hip): bf16 weights, at most one 32-row tile per launch; SV_EXP bit 2 = A/B, the 7-launch layer.
    // Hidden sizes above 2048 keep the 7-launch layer: every block of the whole-K projection re-reads 32 x K activations from L2,
    // and at StarVector-8B's K = 4608 that costs what the removed row update saves (16 columns per block: 4227 vs 4178 us per
    // step; 18 columns = 256 blocks: 3933 vs 3932, profiles/fold6_r03_8b_ab.log); SV_EXP bit 4 = A/B, the 6-launch layer at any size.
    e->fold6 = c.weight_dtype == SV_WEIGHT_BF16 && e->MT == 1 && (c.n_head * dh) % 32 == 0 && D % 32 == 0 &&
               (c.n_head * dh <= 2048 || (e->exp & 4));
    if (!rc && e->fold6) {
        const int cpb = cols_pick_cpb(D, c.n_head * dh);
        for (DecLayer& L : e->dec) L.c_proj.cpb = cpb;
        // the fused MLP launch (SV_EXP bit 128) needs its F / 32 blocks resident at once (one 8-wave block per CU) and the exact
        // (tile, K slice) geometry of the two kernels it replaces: 16 k-steps per wave in both phases
        const Linear& fc = e->dec[0].c_fc; const Linear& dn = e->dec[0].c_proj2;
        const int T1 = fc.Npad / 32;
        e->mlp_fused_ok = T1 <= e->num_cus && T1 % 8 == 0 && fc.N == fc.Npad && dn.N == dn.Npad && fc.Kpad / 16 == 128 &&
                          dn.splitk >= 1 && 8 % dn.splitk == 0 && dn.Kpad / 16 == dn.splitk * 128 && (dn.Npad / 32) * dn.splitk == T1 &&
                          dn.Kpad == fc.Npad;
        if (e->mlp_fused_ok && getenv("SV_MLP_TRACE")) rc = dalloc(e, &e->mlp_trace, (size_t)T1 * 8);
    }
    if (!rc) {
        // the row update + c_attn launch (rowops.hip): the projection's whole weight share per wave in registers (4 k-steps at
        // StarVector-1B, 9 at StarVector-8B: round 5 carried the launch to the 7-launch layer of the wide model), bf16 weights, one row tile
        const Linear& ca = e->dec[0].c_attn; const Linear& dn = e->dec[0].c_proj2;
        e->rc_fused_ok = c.weight_dtype == SV_WEIGHT_BF16 && e->MT == 1 && !ca.fp8 &&
                         rowln_cattn_fits(D, ca.Npad, ca.Kpad, ca.splitk, dn.splitk, e->num_cus) && rowln_cattn_resident(D > 2048);
        // narrow rows (StarVector-1B): the first poll 3.9 us after block start; wide rows (StarVector-8B): the weight stream is the timer and the
        // GEMM blocks' hold-back in front of it measured best at 0 (profiles/rowln_cattn_r05_ab.log, section 8)
        e->rc_delay = D > 2048 ? 0 : 390;
        if (const char* dl = getenv("SV_RC_DELAY")) { const int v = atoi(dl); if (v >= 0 && v <= 2000) e->rc_delay = v; }
    }
    if (rc) { sv_destroy(e); return rc; }
    *out = e;
    return 0;
}

// ------------------------------------------------------------------------------------------------
// weights
// ------------------------------------------------------------------------------------------------
extern "C" int sv_load_weight(sv_engine* e, const char* name, const void* dev_ptr, int32_t dtype, int32_t ndim,
                              const int64_t* shape, sv_stream stream) {
    if (!e || !name || !dev_ptr) return fail(SV_EINVAL, "sv_load_weight: null argument");
    if (dtype != SV_DTYPE_BF16 && dtype != SV_DTYPE_F32) return fail(SV_EINVAL, "unsupported dtype %d", dtype);
    std::lock_guard<std::mutex> lk(e->mu);
    HIPCHECK(hipSetDevice(e->cfg.device));
    if (strstr(name, ".visual_encoder.head.")) return 0;     // SigLIP pooling head: not on the path (image_encoder.py:109)
    auto it = e->slots.find(name);
    if (it == e->slots.end()) return fail(SV_ENOENT, "unknown weight name '%s'", name);
    Slot& s = it->second;
    size_t numel = 1;
    for (int i = 0; i < ndim; ++i) numel *= (size_t)shape[i];
    if (numel != s.numel)
        return fail(SV_EINVAL, "weight '%s': %zu elements given, %zu expected", name, numel, s.numel);
    hipStream_t st = (hipStream_t)stream;
    const int is_f32 = dtype == SV_DTYPE_F32;
    if (s.kind == SLOT_LINEAR_W || s.kind == SLOT_WTE) {
        Linear* l = s.kind == SLOT_WTE ? &e->lm_head : s.lin;
        const bool pack = !(s.kind == SLOT_WTE && e->lm_head_explicit);
        if (pack && l->fp8) {
            // e4m3 weight-only quantisation: fp8 image for the decode kernels, bf16 image of the same values for the big-M ones
            if (!l->Wp) SVCHECK(dalloc(e, &l->Wp, (size_t)l->Npad * l->Kpad, s.part_rows != 0));
            if (!l->Wq) SVCHECK(dalloc(e, &l->Wq, (size_t)l->Npad * l->Kpad, s.part_rows != 0));
            if (!l->wscale) SVCHECK(dalloc(e, &l->wscale, (size_t)l->Npad, true));
            const int roff = s.part_rows ? s.row_off : 0, rows = s.part_rows ? s.part_rows : l->N;
            if (roff % 32) return fail(SV_EINVAL, "weight '%s': part offset %d is not a multiple of 32", name, roff);
            const size_t tile0 = (size_t)(roff / 32) * (l->Kpad / 16);
            launch_pack_weight_fp8(dev_ptr, is_f32, l->Wp + tile0 * 512, l->Wq + tile0 * 512, l->wscale + roff, rows, l->K,
                                   round_up(rows, 32), l->Kpad, st);
        } else if (pack) {
            if (!l->Wp) SVCHECK(dalloc(e, &l->Wp, (size_t)l->Npad * l->Kpad, s.part_rows != 0));
            if (s.part_rows) {
                // one part of a fused projection: rows [row_off, row_off + part_rows), tile aligned
                if (s.row_off % 32) return fail(SV_EINVAL, "weight '%s': part offset %d is not a multiple of 32", name, s.row_off);
                bf16_t* dst = l->Wp + (size_t)(s.row_off / 32) * (l->Kpad / 16) * 512;
                launch_pack_weight(dev_ptr, is_f32, dst, s.part_rows, l->K, round_up(s.part_rows, 32), l->Kpad, st);
            } else {
                launch_pack_weight(dev_ptr, is_f32, l->Wp, l->N, l->K, l->Npad, l->Kpad, st);
            }
        }
        if (s.kind == SLOT_LINEAR_W && l == &e->lm_head) e->lm_head_explicit = true;
    }
    if (s.kind == SLOT_RAW || s.kind == SLOT_WTE) {
        if (s.part_rows) {          // bias of one part of a fused projection
            Linear* l = nullptr;
            for (auto& kv : e->slots) if (kv.second.kind == SLOT_LINEAR_W && kv.second.lin && &kv.second.lin->bias == s.raw) { l = kv.second.lin; break; }
            const size_t total = l ? (size_t)l->Npad : (size_t)s.row_off + numel;
            if (!*s.raw) SVCHECK(dalloc(e, s.raw, total, true));
            launch_convert_to_bf16(dev_ptr, is_f32, *s.raw + s.row_off, numel, st);
        } else {
            if (!*s.raw) SVCHECK(dalloc(e, s.raw, numel, false));
            launch_convert_to_bf16(dev_ptr, is_f32, *s.raw, numel, st);
        }
    }
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipStreamSynchronize(st));   // the caller may free its tensor right after this returns
    s.loaded = true;
    e->fold_ready = false;                // a (re)loaded tensor invalidates the LayerNorm-folded images (rebuilt by sv_weights_complete)
    return 0;
}

// The tensors this engine expects (name, element count, required): what a loader iterates instead of guessing names -- bench.py's
// synthetic weights, from_pretrained's report of unexpected / missing keys.  Sorted by name, so index i is stable for a config.
static std::vector<std::pair<std::string, const Slot*>> sorted_slots(const sv_engine* e) {
    std::vector<std::pair<std::string, const Slot*>> v;
    v.reserve(e->slots.size());
    for (const auto& kv : e->slots) v.emplace_back(kv.first, &kv.second);
    std::sort(v.begin(), v.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
    return v;
}
extern "C" int sv_weight_count(sv_engine* e) {
    if (!e) return fail(SV_EINVAL, "null engine");
    std::lock_guard<std::mutex> lk(e->mu);
    return (int)e->slots.size();
}
extern "C" int sv_weight_info(sv_engine* e, int32_t index, char* name, int32_t name_cap, int64_t* numel, int32_t* required,
                              int32_t* loaded) {
    if (!e || !name || name_cap < 2 || !numel) return fail(SV_EINVAL, "sv_weight_info: bad argument");
    std::lock_guard<std::mutex> lk(e->mu);          // sv_load_weight flips `loaded` under the same mutex
    const auto v = sorted_slots(e);
    if (index < 0 || index >= (int)v.size()) return fail(SV_EINVAL, "sv_weight_info: index %d out of range (0..%zu)", index, v.size());
    const std::string& n = v[index].first;
    if ((int)n.size() + 1 > name_cap) return fail(SV_EINVAL, "sv_weight_info: name of %zu bytes does not fit %d", n.size() + 1, name_cap);
    memcpy(name, n.c_str(), n.size() + 1);
    *numel = (int64_t)v[index].second->numel;
    if (required) *required = v[index].second->required ? 1 : 0;
    if (loaded) *loaded = v[index].second->loaded ? 1 : 0;
    return 0;
}

extern "C" int sv_weights_complete(sv_engine* e) {
    if (!e) return fail(SV_EINVAL, "null engine");
    {
        std::lock_guard<std::mutex> lk(e->mu);      // (every caller checks readiness BEFORE it takes the engine mutex)
        for (auto& kv : e->slots)
            if (kv.second.required && !kv.second.loaded) return fail(SV_ENOENT, "missing weight '%s'", kv.first.c_str());
    }
    if (e->fold6 && !e->fold_ready) {
        // every tensor is in: build the LayerNorm-folded c_fc images (W' = bf16(W * gamma_2), c1, c2) once
        std::lock_guard<std::mutex> lk(e->mu);
        if (!e->fold_ready) {
            HIPCHECK(hipSetDevice(e->cfg.device));
            for (DecLayer& L : e->dec) {
                Linear& l = L.c_fc;
                if (!l.Wf) SVCHECK(dalloc(e, &l.Wf, (size_t)l.Npad * l.Kpad, false));
                if (!l.c1) SVCHECK(dalloc(e, &l.c1, (size_t)l.Npad, true));
                if (!l.c2) SVCHECK(dalloc(e, &l.c2, (size_t)l.Npad, true));
                launch_fold_prepare(l.Wp, L.ln2.g, L.ln2.b, l.bias, l.Wf, l.c1, l.c2, l.N, l.Npad, l.Kpad, nullptr);
            }
            HIPCHECK(hipGetLastError());
            HIPCHECK(hipDeviceSynchronize());
            e->fold_ready = true;
        }
    }
    return 0;
}
