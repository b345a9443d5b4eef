// Host driver + C ABI of libstarvector_hip.so (see include/starvector_hip.h for the contract and the
// reference lines each entry point replaces).  Owns: repacked weights, workspaces, the paged KV pool
// and its page allocator, the generation loop (hipGraph-captured decode step, device-side stop flag).
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/starvector_hip.h"
#include "kernels.h"
#include "beam.h"

using namespace sv;

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static thread_local std::string g_err;
static int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
#define HIPCHECK(x)                                                                              \
    do {                                                                                         \
        hipError_t _e = (x);                                                                     \
        if (_e != hipSuccess)                                                                    \
            return fail(SV_EHIP, "%s failed: %s (%s:%d)", #x, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)
#define SVCHECK(x)          \
    do {                    \
        int _r = (x);       \
        if (_r) return _r;  \
    } while (0)

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

// ------------------------------------------------------------------------------------------------
// engine state
// ------------------------------------------------------------------------------------------------
struct Linear {
    bf16_t* Wp = nullptr;
    bf16_t* bias = nullptr;
    bool fp8 = false;          // decoder weights quantised to e4m3 at load (sv_config.weight_dtype = 1)
    uint8_t* Wq = nullptr;     //   decode image (launch_pack_weight_fp8); Wp then holds the SAME q values as bf16
    float* wscale = nullptr;   //   per-output-row scale [Npad]
    int N = 0, K = 0, Npad = 0, Kpad = 0;
    int splitk = 1;       // decode-path split-K factor (fp32 slabs summed by the consumer)
    int col_tiles = 1;    // column tiles per block of the two-row-tile decode kernel (33..64 rows; pick_decode_plan)
    int cpb = 8;          // output columns per block of the slab-free output projection (decode_cols.hip)
    bf16_t* Wf = nullptr; // LayerNorm-folded image W' = bf16(W * gamma) (decode_cols.hip; c_fc only), with
    float* c1 = nullptr;  //   c1[n] = sum_k W'[n][k]
    float* c2 = nullptr;  //   c2[n] = sum_k beta[k] W[n][k] + bias[n]
};
struct LNp { bf16_t* g = nullptr; bf16_t* b = nullptr; };
struct VitLayer { LNp ln1, ln2; Linear in_proj, out_proj, c_fc, c_proj; };
struct DecLayer { LNp ln1, ln2; Linear c_attn, c_proj, c_fc, c_proj2; };

enum SlotKind { SLOT_LINEAR_W, SLOT_RAW, SLOT_WTE };
struct Slot {
    SlotKind kind;
    Linear* lin = nullptr;
    bf16_t** raw = nullptr;
    size_t numel = 0;
    bool loaded = false;
    bool required = true;
    int row_off = 0;          // fused projections (q|k|v): first output row of this part inside the Linear
    int part_rows = 0;        // rows of this part (0 = the whole tensor)
};

struct sv_engine {
    sv_config cfg;
    std::mutex mu;
    int T = 0, NP = 0, dh = 0, vdh = 0;
    int nkv = 1, QKV = 0, vit_F = 0;      // KV heads, width of the fused q|k|v projection, ViT MLP width
    bool v2 = false;
    float *rope_cos = nullptr, *rope_sin = nullptr;
    size_t kv_head_stride = 0;
    int conv_K = 0;

    // weights
    Linear conv1;
    bf16_t *cls = nullptr, *pos = nullptr;
    LNp ln_pre, ln_vision;
    std::vector<VitLayer> vit;
    Linear ad_fc, ad_proj;
    bf16_t *ad_w = nullptr, *ad_b = nullptr, *ad_rm = nullptr, *ad_rv = nullptr;
    bf16_t *wte = nullptr, *wpe = nullptr;
    Linear lm_head;
    bool lm_head_explicit = false;
    LNp ln_f;
    std::vector<DecLayer> dec;
    std::unordered_map<std::string, Slot> slots;
    std::vector<void*> allocs;

    // vision workspaces (rows = max_batch * T)
    bf16_t *patches = nullptr, *patch_out = nullptr, *vx = nullptr, *vln = nullptr, *vqkv = nullptr,
           *vattn = nullptr, *vmlp = nullptr, *a1 = nullptr, *a2 = nullptr;
    // prefill workspaces (lazily grown)
    size_t pf_rows = 0;
    bf16_t *ph = nullptr, *pln = nullptr, *pqkv = nullptr, *pattn = nullptr, *pmlp = nullptr;
    // decode workspaces
    int MT = 0, ldws = 0, Vpad = 0;
    bf16_t *h_dec = nullptr, *hl = nullptr, *xp_a = nullptr, *xp_attn = nullptr, *xp_mlp = nullptr;
    bf16_t* h_xp = nullptr;         // residual stream of the decode step in fragment order (6-launch layer)
    bool fold6 = false;             // 6 launches per layer: slab-free attention output projection + ln_2 folded into c_fc
    bool fold_ready = false;
    bool only_skinny = false;       // profiling: enqueue only the weight-streaming GEMMs of a step
    bool skip_skinny = false;       // profiling: enqueue everything BUT the weight-streaming GEMMs
    int exp = 0;                    // SV_EXP bit mask, read once at sv_create (A/B switches of the round's experiments):
                                    //   2 the 7-launch layer (no LayerNorm fold);
                                    //   (1: was the row update as one wave per row: 0.218 vs 0.131 ms per step, removed)
                                    //   8 (at sv_create only) the round 1-2 split-K rule of the decode GEMMs
                                    //   (16 / 32 / 64: 2 / 6 / 8 key groups per attention block: 1186 / 1169 / 1175 vs 1171 us, removed)
                                    //   (1, 2: XCD-aligned weight prefetch by attention's idle waves / spare row-update blocks; 4: one key
                                    //    group per attention block -- all measured slower, profiles/prefetch_r03_*.log, removed)
    float *ws = nullptr, *ws2 = nullptr, *logits = nullptr, *sample_scratch = nullptr, *attn_part = nullptr;
    unsigned* attn_cnt = nullptr;
    float* am_val = nullptr; int32_t* am_idx = nullptr;
    uint32_t* seen = nullptr; int seen_words = 0;      // repetition-penalty bitmap [rows][Vpad/32]
    int32_t *cur_tok = nullptr, *next_tok = nullptr, *unfinished = nullptr, *positions = nullptr,
            *out_tok = nullptr, *d_step = nullptr, *d_done = nullptr, *d_nemit = nullptr, *d_stop = nullptr, *d_bad = nullptr;
    int out_ld = 0;
    int32_t* h_flags = nullptr;   // pinned: [0]=done [1]=n_emitted
    // KV pool
    char* kv_pool = nullptr;
    size_t layer_stride = 0;
    int pages_per_seq = 0, num_pages = 0, page_bytes = 0;
    int32_t* block_table = nullptr;
    std::vector<int> free_pages;
    // beam search (num_beams > 1): device scorer + staging for the tail-page copies
    BeamScorer beam;
    char* beam_staging = nullptr;
    size_t beam_staging_bytes = 0;
    bf16_t* score_ws = nullptr;      // scoring forward: kept hidden rows, their ln_f, bf16 logits [rows][Vpad]
    size_t score_elems = 0;
    int cached_B = 0;
    int num_cus = 256;
    double timing[3] = {0, 0, 0};
    double timing_graph = 0;
    // generation runs on an engine-owned non-blocking stream (the caller's stream may be the legacy
    // null stream, which cannot be captured into a hipGraph); ordered after the caller's stream by an event
    hipStream_t gen_stream = nullptr;
    hipEvent_t gen_event = nullptr;
    // continuous batching: one request per row ("slot")
    bool cb_active = false;
    std::vector<char> cb_used;                    // slot in use (admitted, not yet released)
    std::vector<std::vector<int>> cb_pages;       // pages held by each slot
    CbSlot* cb_slots = nullptr;                   // device [max_batch]
    int32_t *cb_map = nullptr, *cb_nlive = nullptr, *cb_events = nullptr, *cb_table_pf = nullptr;
    int trash_page = 0;                           // what the block-table rows of free slots point at
    std::unordered_map<int, std::pair<hipGraph_t, hipGraphExec_t>> cb_graphs;      // one captured step per row bucket
    // sv_generate: the captured decode step is kept while the next call has the same shape and sampling parameters
    std::string gen_graph_key;
    hipGraph_t gen_graph = nullptr;
    hipGraphExec_t gen_gexec = nullptr;
    // optional per-kernel HIP-event profiling of the decode step (bench.py roofline leg)
    bool prof_on = false;
    std::vector<hipEvent_t> prof_ev;
    std::vector<int> prof_kind;
    size_t prof_used = 0;
};

enum { PK_SKINNY = 0, PK_ATTN = 1, PK_ROWLN = 2, PK_SAMPLE = 3, PK_COUNT = 4 };
// record an event in front of the next launch (tagged with its kind); durations = event deltas
static void prof_mark(sv_engine* e, int kind, hipStream_t st) {
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

static int dev_alloc(sv_engine* e, void** p, size_t bytes, bool zero = true) {
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
template <typename T>
static int dalloc(sv_engine* e, T** p, size_t count, bool zero = true) {
    return dev_alloc(e, reinterpret_cast<void**>(p), count * sizeof(T), zero);
}

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
static void pick_decode_plan(const Linear& l, int MT, int num_cus, bool fp8, bool legacy, bool whole_k, int* splitk, int* col_tiles) {
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
    c->sliding_window = 0; c->weight_dtype = SV_WEIGHT_BF16;
}

extern "C" void sv_config_default_8b(sv_config* c) {
    // siglip_384 = google/siglip-large-patch16-384 (image_encoder.py:35-36), bigcode/starcoder2-7b (llm/starcoder2.py:22)
    c->image_size = 384; c->patch_size = 16; c->vit_width = 1024; c->vit_layers = 24; c->vit_heads = 16;
    c->adapter_norm = SV_NORM_LAYER; c->hidden = 4608; c->n_layer = 32; c->n_head = 36; c->n_inner = 18432;
    c->vocab = 49152 + 5; c->n_positions = 16384; c->max_batch = 16; c->max_seq_len = 4096; c->ln_eps = 1e-5f;
    c->device = 0;
    c->arch = SV_ARCH_V2; c->n_kv_head = 4; c->rope_theta = 1e6f; c->vit_mlp = 4096; c->vit_eps = 1e-6f;
    c->sliding_window = 4096; c->weight_dtype = SV_WEIGHT_BF16;
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
    if (e->beam_staging) (void)hipFree(e->beam_staging);
    if (e->score_ws) (void)hipFree(e->score_ws);
    if (e->h_flags) (void)hipHostFree(e->h_flags);
    for (hipEvent_t ev : e->prof_ev) (void)hipEventDestroy(ev);
    if (e->gen_event) (void)hipEventDestroy(e->gen_event);
    if (e->gen_stream) (void)hipStreamDestroy(e->gen_stream);
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
    A(dalloc(e, &e->cur_tok, R));
    A(dalloc(e, &e->next_tok, R));
    A(dalloc(e, &e->unfinished, R));
    A(dalloc(e, &e->positions, R));
    e->out_ld = c.max_seq_len;
    A(dalloc(e, &e->out_tok, (size_t)c.max_batch * e->out_ld));
    A(dalloc(e, &e->d_step, 4));
    A(dalloc(e, &e->d_done, 4));
    A(dalloc(e, &e->d_nemit, 4));
    A(dalloc(e, &e->d_bad, 4));       // raised by the selection kernels when a row has no finite logit
    A(dalloc(e, &e->d_stop, 64));

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
    // 6 launches per layer (decode_cols.hip): bf16 weights, at most one 32-row tile per launch; SV_EXP bit 2 = A/B, the 7-launch layer.
    // Hidden sizes above 2048 keep the 7-launch layer: every block of the whole-K projection re-reads 32 x K activations from L2,
    // and at StarVector-8B's K = 4608 that costs what the removed row update saves (16 columns per block: 4227 vs 4178 us per
    // step; 18 columns = 256 blocks: 3933 vs 3932, profiles/fold6_r03_8b_ab.log); SV_EXP bit 4 = A/B, the 6-launch layer at any size.
    e->fold6 = c.weight_dtype == SV_WEIGHT_BF16 && e->MT == 1 && (c.n_head * dh) % 32 == 0 && D % 32 == 0 &&
               (c.n_head * dh <= 2048 || (e->exp & 4));
    if (!rc && e->fold6) {
        const int cpb = cols_pick_cpb(D, c.n_head * dh);
        for (DecLayer& L : e->dec) L.c_proj.cpb = cpb;
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

extern "C" int sv_weights_complete(sv_engine* e) {
    if (!e) return fail(SV_EINVAL, "null engine");
    for (auto& kv : e->slots)
        if (kv.second.required && !kv.second.loaded) return fail(SV_ENOENT, "missing weight '%s'", kv.first.c_str());
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

// ------------------------------------------------------------------------------------------------
// launch helpers
// ------------------------------------------------------------------------------------------------
static void gemm(const bf16_t* A, int lda, const Linear& l, const bf16_t* R, int ldr, void* C, int ldc, int M,
                 int act, int out_f32, hipStream_t st) {
    GemmArgs g;
    g.A = A; g.lda = lda; g.Wp = l.Wp; g.bias = l.bias; g.R = R; g.ldr = ldr; g.C = C; g.ldc = ldc;
    g.M = M; g.N = l.N; g.K = l.Kpad; g.act = act; g.out_f32 = out_f32;
    g.cscale = l.fp8 ? l.wscale : nullptr;
    launch_gemm(g, st);
}

// Context splits of the decode attention: a constant of the engine (sized for the engine's max_batch), NOT of the batch of the
// call, so that a sequence's partial results are merged in the same grouping whatever shares the batch with it: a row is
// bit-identical alone, inside a batch and inside a continuous batch at any context length.
static int attn_max_splits_of(int max_batch, int nkv, int num_cus) {
    const int rows = (max_batch < 32 ? max_batch : 32) * nkv;
    const int ms = num_cus / (rows < 1 ? 1 : rows);
    return ms < 1 ? 1 : (ms > 8 ? 8 : ms);
}
static int attn_max_splits(const sv_engine* e) { return attn_max_splits_of(e->cfg.max_batch, e->nkv, e->num_cus); }

// 32-key groups a block takes before another context split joins.  Where the engine's rows x KV heads already give every CU a block
// (StarVector-8B at 64 rows: 256 (row, KV head) pairs), a context of <= 8 groups stays in ONE block (one group per wave: no partial
// results, no ticket, no merge); otherwise 4 (measured best where the splits are what fills the chip,
// profiles/attention_r03_groups_per_block_ab.log).  A constant of the engine like the split cap (same reason).
static int attn_groups_per_block_of(int max_batch, int nkv, int num_cus) {
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
    launch_im2col(img, e->patches, B, c.image_size, c.patch_size, e->conv1.Kpad, st);
    gemm(e->patches, e->conv1.Kpad, e->conv1, nullptr, 0, e->patch_out, Dv, B * NP, ACT_NONE, 0, st);
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
        launch_layernorm_rows(e->vx, Dv, L.ln1.g, L.ln1.b, e->vln, Dv, M, Dv, eps, st);
        gemm(e->vln, Dv, L.in_proj, nullptr, 0, e->vqkv, 3 * Dv, M, ACT_NONE, 0, st);
        launch_attn_prefill(at, st);
        gemm(e->vattn, Dv, L.out_proj, e->vx, Dv, e->vx, Dv, M, ACT_NONE, 0, st);
        launch_layernorm_rows(e->vx, Dv, L.ln2.g, L.ln2.b, e->vln, Dv, M, Dv, eps, st);
        gemm(e->vln, Dv, L.c_fc, nullptr, 0, e->vmlp, Fv, M, act, 0, st);
        gemm(e->vmlp, Fv, L.c_proj, e->vx, Dv, e->vx, Dv, M, ACT_NONE, 0, st);
    }
    launch_layernorm_rows(e->vx, Dv, e->ln_vision.g, e->ln_vision.b, out, Dv, M, Dv, eps, st);
    return 0;
}

static int adapter_forward(sv_engine* e, const bf16_t* in, int B, bf16_t* out, hipStream_t st) {
    const sv_config& c = e->cfg;
    const int Dv = c.vit_width, D = c.hidden, T = e->T, M = B * T;
    gemm(in, Dv, e->ad_fc, nullptr, 0, e->a1, 2 * Dv, M, ACT_SWISH, 0, st);
    gemm(e->a1, 2 * Dv, e->ad_proj, nullptr, 0, e->a2, D, M, ACT_NONE, 0, st);
    if (c.adapter_norm == SV_NORM_LAYER)
        launch_plane_layernorm(e->a2, e->ad_w, e->ad_b, out, B, T * D, c.ln_eps, st);
    else
        launch_token_batchnorm(e->a2, e->ad_w, e->ad_b, e->ad_rm, e->ad_rv, out, B, T, D, c.ln_eps, st);
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

static int assign_pages(sv_engine* e, int B, int total_len, hipStream_t st) {
    // page allocator: hand every sequence the pages for its whole budget up front (the decode loop
    // runs without host round-trips, so pages cannot be added mid-flight)
    const int need = (total_len + SV_PAGE_TOKENS - 1) / SV_PAGE_TOKENS;
    if (need > e->pages_per_seq) return fail(SV_EINVAL, "sequence length %d exceeds max_seq_len %d", total_len, e->cfg.max_seq_len);
    e->free_pages.clear();
    for (int p = e->num_pages - 1; p >= 0; --p) e->free_pages.push_back(p);
    std::vector<int32_t> table((size_t)e->cfg.max_batch * e->pages_per_seq, 0);
    for (int b = 0; b < B; ++b)
        for (int i = 0; i < need; ++i) {
            table[(size_t)b * e->pages_per_seq + i] = e->free_pages.back();
            e->free_pages.pop_back();
        }
    HIPCHECK(hipMemcpyAsync(e->block_table, table.data(), table.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    HIPCHECK(hipStreamSynchronize(st));
    return 0;
}

// lm_head -> e->logits; xp holds ln_f(h) in fragment order
static void lm_head_logits(sv_engine* e, int MT, const bf16_t* xp, hipStream_t st) {
    SkinnyArgs a;
    memset(&a, 0, sizeof(a));
    a.xp = xp; a.Wp = e->lm_head.Wp; a.Wq = e->lm_head.Wq; a.wscale = e->lm_head.wscale; a.MT = MT; a.Npad = e->lm_head.Npad; a.K = e->lm_head.Kpad;
    a.splitk = 1; a.out_mode = SK_OUT_F32; a.out_f32 = e->logits; a.ldo = e->Vpad; a.round_bf16 = 1;
    a.N = e->lm_head.N;
    launch_gemm_skinny(a, st);
}

static int prefill_forward(sv_engine* e, const bf16_t* embeds, int B, int S0, hipStream_t st, int n_keep = 0,
                           bf16_t* dev_scores = nullptr, const int32_t* table = nullptr) {
    if (!table) table = e->block_table;           // continuous batching prefills NEW requests through a table of their slots' pages
    const sv_config& c = e->cfg;
    const int D = c.hidden, dh = e->dh, F = c.n_inner, M = B * S0, QKV = e->QKV, nkv = e->nkv;
    const int QD = c.n_head * dh;                      // width of the query block (= D for both model families)
    SVCHECK(ensure_prefill_ws(e, (size_t)M));
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
    for (int i = 0; i < c.n_layer; ++i) {
        DecLayer& L = e->dec[i];
        launch_layernorm_rows(e->ph, D, L.ln1.g, L.ln1.b, e->pln, D, M, D, c.ln_eps, st);
        gemm(e->pln, D, L.c_attn, nullptr, 0, e->pqkv, QKV, M, ACT_NONE, 0, st);
        if (e->v2) launch_rope_prefill(e->pqkv, QKV, M, S0, c.n_head + nkv, dh, e->rope_cos, e->rope_sin, st);
        for (int kh = 0; kh < nkv; ++kh)
            launch_kv_write_prefill(e->pqkv, QKV, QD + kh * dh, QD + nkv * dh + kh * dh,
                                    e->kv_pool + (size_t)i * e->layer_stride + (size_t)kh * e->kv_head_stride,
                                    table, e->pages_per_seq, B, S0, dh, st);
        launch_attn_prefill(at, st);
        gemm(e->pattn, QD, L.c_proj, e->ph, D, e->ph, D, M, ACT_NONE, 0, st);
        launch_layernorm_rows(e->ph, D, L.ln2.g, L.ln2.b, e->pln, D, M, D, c.ln_eps, st);
        gemm(e->pln, D, L.c_fc, nullptr, 0, e->pmlp, F, M, ACT_GELU_TANH, 0, st);
        gemm(e->pmlp, F, L.c_proj2, e->ph, D, e->ph, D, M, ACT_NONE, 0, st);
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
        launch_gather_tail_rows(e->ph, hk, B, S0, n_keep, D, st);
        launch_layernorm_rows(hk, D, e->ln_f.g, e->ln_f.b, hn, D, (int)rows, D, c.ln_eps, st);
        gemm(hn, D, e->lm_head, nullptr, 0, lg, e->Vpad, (int)rows, ACT_NONE, 0, st);
        HIPCHECK(hipMemcpy2DAsync(dev_scores, (size_t)c.vocab * sizeof(bf16_t), lg, (size_t)e->Vpad * sizeof(bf16_t),
                                  (size_t)c.vocab * sizeof(bf16_t), rows, hipMemcpyDeviceToDevice, st));
    }
    // only the last prompt row feeds ln_f + lm_head (HF computes all rows; same result)
    launch_gather_last_rows(e->ph, e->hl, B, S0, D, st);
    launch_layernorm_rows_packed(e->hl, D, e->ln_f.g, e->ln_f.b, e->xp_a, B, D, c.ln_eps, st);
    lm_head_logits(e, (B + 31) / 32, e->xp_a, st);
    return 0;
}

// One autoregressive step: consumes cur_tok / positions, leaves logits in e->logits.  6 launches per layer + 2 (bf16 weights,
// <= 32 rows):
//   row update (embedding | + bias + residual of the previous down-proj, LN1) | c_attn -> fp32 slabs | attention (sums the
//   slabs, + bias) | c_proj over the whole K: h += ..., partial row statistics | c_fc on the raw h (ln_2 folded, bias + GELU
//   epilogue) | down-proj -> slabs ... | row update (ln_f) | lm_head
// fp8 weights / more than one row tile: 7 launches per layer (c_proj -> slabs | row update (+ bias, + residual, LN2) | c_fc).
static void decode_forward(sv_engine* e, int B, hipStream_t st) {
    const sv_config& c = e->cfg;
    const int D = c.hidden, dh = e->dh, F = c.n_inner, MT = (B + 31) / 32;
    // slab double buffer: A = c_attn slabs (read by attention); B = c_proj / down-proj slabs (read by the row update)
    float* wsA = e->ws;
    float* wsB = e->ws2;
    RowUpdateArgs ru;
    memset(&ru, 0, sizeof(ru));
    const bool fold6 = e->fold6 && e->fold_ready && !(e->exp & 2);
    ru.h = fold6 ? e->h_xp : e->h_dec; ru.ldh = fold6 ? 0 : D;        // 6-launch layer: the residual stream lives in fragment order
    ru.M = B; ru.D = D; ru.eps = c.ln_eps; ru.xp_out = e->xp_a;
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
        else if (out_mode == SK_OUT_PACKED_ACT) { a.splitk = 1; a.bias = l.bias; a.act = ACT_GELU_TANH; a.out_xp = e->xp_mlp; a.out_KS = F / 16; }
        else { a.splitk = 1; a.out_f32 = e->logits; a.ldo = e->Vpad; a.round_bf16 = 1; }
        if (e->skip_skinny) return;
        prof_mark(e, PK_SKINNY, st);
        launch_gemm_skinny(a, st);
    };
    for (int i = 0; i < c.n_layer; ++i) {
        DecLayer& L = e->dec[i];
        row_update();                                            // embedding or the previous layer's down-proj -> LN1(h)
        skinny(e->xp_a, L.c_attn, SK_OUT_PARTIAL, wsA);
        if (!e->only_skinny) {
            AttnDecodeArgs ad;
            memset(&ad, 0, sizeof(ad));
            ad.ws = wsA; ad.splitk = L.c_attn.splitk; ad.ldws = e->ldws; ad.rows_ws = MT * 32; ad.bias = L.c_attn.bias;
            ad.pool_layer = e->kv_pool + (size_t)i * e->layer_stride; ad.block_table = e->block_table;
            ad.max_pages = e->pages_per_seq; ad.positions = e->positions; ad.out_xp = e->xp_attn; ad.out_KS = D / 16;
            ad.window = c.sliding_window;
            ad.B = B; ad.H = c.n_head; ad.head_dim = dh; ad.scale = 1.0f / sqrtf((float)dh);
            ad.part = e->attn_part; ad.counters = e->attn_cnt;
            ad.max_splits = attn_max_splits(e);
            ad.groups_per_block = attn_groups_per_block(e);
            ad.n_kv = e->nkv; ad.kv_head_stride = e->kv_head_stride; ad.rope_cos = e->rope_cos; ad.rope_sin = e->rope_sin;
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
            if (!e->skip_skinny) { prof_mark(e, PK_SKINNY, st); launch_gemm_cols(ca, st); }
            SkinnyArgs a;
            memset(&a, 0, sizeof(a));
            a.xp = e->h_xp; a.Wp = L.c_fc.Wf; a.MT = MT; a.Npad = L.c_fc.Npad; a.K = L.c_fc.Kpad; a.N = L.c_fc.N; a.splitk = 1;
            a.out_mode = SK_OUT_PACKED_ACT; a.act = ACT_GELU_TANH; a.out_xp = e->xp_mlp; a.out_KS = F / 16;
            a.fold_c1 = L.c_fc.c1; a.fold_c2 = L.c_fc.c2; a.fold_D = D; a.fold_eps = c.ln_eps;
            if (!e->skip_skinny) { prof_mark(e, PK_SKINNY, st); launch_gemm_skinny(a, st); }
        } else {
            skinny(e->xp_attn, L.c_proj, SK_OUT_PARTIAL, wsB);
            ru.ws = wsB; ru.splitk = L.c_proj.splitk; ru.bias = L.c_proj.bias; ru.g = L.ln2.g; ru.b = L.ln2.b;
            row_update();                                        // + bias + residual, LN2
            skinny(e->xp_a, L.c_fc, SK_OUT_PACKED_ACT, nullptr);
        }
        skinny(e->xp_mlp, L.c_proj2, SK_OUT_PARTIAL, wsB);
        const LNp& nxt = (i + 1 < c.n_layer) ? e->dec[i + 1].ln1 : e->ln_f;
        ru.ws = wsB; ru.splitk = L.c_proj2.splitk; ru.bias = L.c_proj2.bias; ru.g = nxt.g; ru.b = nxt.b;
    }
    row_update();                                                // + bias + residual, ln_f
    skinny(e->xp_a, e->lm_head, SK_OUT_F32, nullptr);
    prof_mark(e, PK_SAMPLE, st);      // closes the lm_head interval; whatever follows is sampling
}

static int check_ready(sv_engine* e) {
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

static int copy_logits_out(sv_engine* e, int B, float* dev_logits, hipStream_t st) {
    HIPCHECK(hipMemcpy2DAsync(dev_logits, (size_t)e->cfg.vocab * sizeof(float), e->logits,
                              (size_t)e->Vpad * sizeof(float), (size_t)e->cfg.vocab * sizeof(float), B,
                              hipMemcpyDeviceToDevice, st));
    return 0;
}

static int cb_guard(sv_engine* e, const char* who) {
    if (e->cb_active) return fail(SV_ESTATE, "%s: a continuous batch holds the KV cache of this engine (sv_cb_reset first)", who);
    return 0;
}

static int prefill_locked(sv_engine* e, const void* dev_embeds, int B, int S0, int total_len, hipStream_t st) {
    SVCHECK(cb_guard(e, "prefill"));
    if (!dev_embeds || B < 1 || B > e->cfg.max_batch) return fail(SV_EINVAL, "prefill: bad B=%d (max_batch %d)", B, e->cfg.max_batch);
    if (S0 < 1 || S0 > e->cfg.max_seq_len) return fail(SV_EINVAL, "prefill: S0=%d out of range (max_seq_len %d)", S0, e->cfg.max_seq_len);
    SVCHECK(assign_pages(e, B, total_len, st));
    SVCHECK(prefill_forward(e, (const bf16_t*)dev_embeds, B, S0, st));
    fill_i32_kernel<<<(B + 63) / 64, 64, 0, st>>>(e->positions, S0, B);
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
    fill_i32_kernel<<<(B + 63) / 64, 64, 0, st>>>(e->positions, S, B);
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
    HIPCHECK(hipMemcpyAsync(e->cur_tok, dev_tokens, (size_t)B * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    decode_forward(e, B, st);
    add_i32_kernel<<<(B + 63) / 64, 64, 0, st>>>(e->positions, 1, B);
    SVCHECK(copy_logits_out(e, B, dev_logits, st));
    HIPCHECK(hipGetLastError());
    return 0;
}

// sample from e->logits into next_tok, then the bookkeeping kernel
static void sample_and_finish(sv_engine* e, int B, const sv_sampling& sp, int max_new, hipStream_t st) {
    const bool pen = sp.repetition_penalty > 0.f && sp.repetition_penalty != 1.0f;
    const uint32_t* seen = pen ? e->seen : nullptr;
    if (sp.min_new_tokens > 0 && sp.eos_token_id >= 0 && sp.eos_token_id < e->cfg.vocab)
        suppress_token_kernel<<<(B + 63) / 64, 64, 0, st>>>(e->logits, e->Vpad, sp.eos_token_id, e->d_step, sp.min_new_tokens, B);
    if (sp.do_sample) {
        SampleArgs sa;
        sa.logits = e->logits; sa.ld = e->Vpad; sa.V = e->cfg.vocab; sa.B = B; sa.temperature = sp.temperature;
        sa.top_p = sp.top_p; sa.top_k = sp.top_k; sa.seed = sp.seed; sa.step = e->d_step; sa.out = e->next_tok; sa.scratch = e->sample_scratch;
        sa.seen = seen; sa.seen_words = e->seen_words; sa.penalty = sp.repetition_penalty;
        launch_sample_top_p(sa, st);
    } else {
        launch_argmax_partial(e->logits, e->Vpad, e->cfg.vocab, e->am_val, e->am_idx, B, seen, e->seen_words,
                              sp.repetition_penalty, st);
    }
    FinishArgs f;
    f.pval = sp.do_sample ? nullptr : e->am_val; f.pidx = sp.do_sample ? nullptr : e->am_idx;
    f.next = e->next_tok; f.cur_tok = e->cur_tok; f.unfinished = e->unfinished; f.positions = e->positions;
    f.out_tokens = e->out_tok; f.ld_out = e->out_ld; f.step = e->d_step; f.done = e->d_done; f.n_emitted = e->d_nemit;
    f.stop_ids = e->d_stop; f.n_stop = sp.n_stop; f.eos = sp.eos_token_id; f.pad = sp.pad_token_id; f.B = B;
    f.max_new = max_new;
    f.seen = pen ? e->seen : nullptr; f.seen_words = e->seen_words;
    f.V = e->cfg.vocab; f.bad = e->d_bad;
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
    const size_t stage_bytes = (size_t)R * c.n_layer * e->nkv * e->page_bytes;
    if (stage_bytes > e->beam_staging_bytes) {
        if (e->beam_staging) (void)hipFree(e->beam_staging);
        e->beam_staging = nullptr; e->beam_staging_bytes = 0;
        HIPCHECK(hipMalloc(reinterpret_cast<void**>(&e->beam_staging), stage_bytes));
        e->beam_staging_bytes = stage_bytes;
    }
    // prompt pass over the B requests, written into the pages of each request's first beam row
    SVCHECK(upload_beam_table(e, B, nb, need, 0, true, st));
    SVCHECK(prefill_forward(e, (const bf16_t*)dev_embeds, B, S0, st));
    SVCHECK(upload_beam_table(e, B, nb, need, S0 / SV_PAGE_TOKENS, false, st));
    e->cached_B = R;
    {
        int r = e->beam.reset(st);
        if (r) return fail(SV_EHIP, "beam scorer reset failed (hip error %d)", r);
    }
    fill_i32_kernel<<<(R + 63) / 64, 64, 0, st>>>(e->positions, S0 - 1, R);      // beam_update adds 1
    BeamKvArgs kv;
    kv.block_table = e->block_table; kv.max_pages = e->pages_per_seq; kv.need = need; kv.parent = e->beam.d.parent;
    kv.step = e->d_step; kv.done = e->d_done; kv.S0 = S0; kv.L_fixed = S0; kv.B = B; kv.nb = nb;
    kv.kv_pool = e->kv_pool; kv.layer_stride = e->layer_stride; kv.kv_head_stride = e->kv_head_stride;
    kv.n_layer = c.n_layer; kv.n_kv = e->nkv; kv.page_bytes = e->page_bytes; kv.staging = e->beam_staging;
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
static int check_finite_logits(sv_engine* e, hipStream_t st, const char* who) {
    HIPCHECK(hipMemcpyAsync(&e->h_flags[4], e->d_bad, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIPCHECK(hipStreamSynchronize(st));
    if (!e->h_flags[4]) return 0;
    HIPCHECK(hipMemsetAsync(e->d_bad, 0, sizeof(int32_t), st));
    HIPCHECK(hipStreamSynchronize(st));
    return fail(SV_EHIP, "%s: a row of logits had no finite value (NaN / Inf in the weights or inputs?)", who);
}

extern "C" int sv_generate(sv_engine* e, const void* dev_embeds, int32_t B, int32_t S0, const sv_sampling* sp,
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
    SVCHECK(prefill_locked(e, dev_embeds, B, S0, S0 + max_new, st));
    // generation state
    fill_i32_kernel<<<(B + 63) / 64, 64, 0, st>>>(e->positions, S0 - 1, B);     // finish_step adds 1
    fill_i32_kernel<<<(B + 63) / 64, 64, 0, st>>>(e->unfinished, 1, B);
    HIPCHECK(hipMemsetAsync(e->d_step, 0, sizeof(int32_t), st));
    HIPCHECK(hipMemsetAsync(e->d_done, 0, sizeof(int32_t), st));
    HIPCHECK(hipMemsetAsync(e->d_nemit, 0, sizeof(int32_t), st));
    if (sp->repetition_penalty > 0.f && sp->repetition_penalty != 1.0f)
        HIPCHECK(hipMemsetAsync(e->seen, 0, (size_t)((B + 31) / 32) * 32 * e->seen_words * sizeof(uint32_t), st));
    if (sp->n_stop > 0) {
        if (!sp->stop_ids) return fail(SV_EINVAL, "n_stop > 0 but stop_ids is null");
        HIPCHECK(hipMemcpyAsync(e->d_stop, sp->stop_ids, sp->n_stop * sizeof(int32_t), hipMemcpyHostToDevice, st));
    }
    sample_and_finish(e, B, *sp, max_new, st);       // first token from the prefill logits
    HIPCHECK(hipMemcpyAsync(&e->h_flags[0], e->d_done, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIPCHECK(hipStreamSynchronize(st));
    auto t1 = std::chrono::steady_clock::now();

    int steps = 0;
    const int chunk = sp->sync_every > 0 ? sp->sync_every : 32;
    // streaming: columns [0, steps] are final after every poll (a finished batch may have fewer: n_emitted caps it)
    int streamed = 0;
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
    if (!e->h_flags[0] && use_graph) {
        // One decode step is captured as a hipGraph (all kernel arguments are stable device pointers; the step index,
        // positions and stop state live in device memory) and replayed every step.  The instantiated graph is KEPT on the
        // engine and reused by the next call whose batch, budget and sampling parameters are the same (a serving request
        // stream, the benchmark), so a short request does not pay a 172-node capture + instantiate; a call with other
        // parameters replaces it.  Owned by the engine: no early return below can leak it.
        char key[256];
        snprintf(key, sizeof(key), "B%d|n%d|ds%d|T%a|p%a|k%d|seed%llu|eos%d|pad%d|ns%d|rp%a|mn%d|x%d", B, max_new, sp->do_sample,
                 sp->temperature, sp->top_p, sp->top_k, (unsigned long long)sp->seed, sp->eos_token_id, sp->pad_token_id, sp->n_stop,
                 sp->repetition_penalty, sp->min_new_tokens, e->exp);
        if (e->gen_gexec && e->gen_graph_key == key) {
            gexec = e->gen_gexec;
        } else {
            if (e->gen_gexec) { (void)hipGraphExecDestroy(e->gen_gexec); e->gen_gexec = nullptr; }
            if (e->gen_graph) { (void)hipGraphDestroy(e->gen_graph); e->gen_graph = nullptr; }
            e->gen_graph_key.clear();
            hipError_t ce = hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed);
            if (ce == hipSuccess) {
                decode_forward(e, B, st);
                sample_and_finish(e, B, *sp, max_new, st);
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
            }
        }
    }
    while (!e->h_flags[0]) {
        int n = max_new - 1 - steps;
        if (n <= 0) break;                       // budget exhausted: the device flag is already set
        if (n > chunk) n = chunk;
        for (int i = 0; i < n; ++i) {
            if (gexec) {
                HIPCHECK(hipGraphLaunch(gexec, st));
            } else {
                decode_forward(e, B, st);
                sample_and_finish(e, B, *sp, max_new, st);
            }
        }
        steps += n;
        HIPCHECK(hipMemcpyAsync(&e->h_flags[0], e->d_done, sizeof(int32_t), hipMemcpyDeviceToHost, st));
        HIPCHECK(hipStreamSynchronize(st));
        if (!e->h_flags[0]) SVCHECK(stream_upto(steps + 1));      // still running: every column so far is final
    }
    const double gexec_used = gexec ? 1.0 : 0.0;
    HIPCHECK(hipMemcpyAsync(&e->h_flags[1], e->d_nemit, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIPCHECK(hipStreamSynchronize(st));
    if (e->h_flags[1] >= 1 && e->h_flags[1] <= max_new) SVCHECK(stream_upto(e->h_flags[1]));
    SVCHECK(check_finite_logits(e, st, "sv_generate"));
    const int n_emit = e->h_flags[1];
    if (n_emit < 1 || n_emit > max_new) return fail(SV_EHIP, "generation bookkeeping failed (n_emitted=%d)", n_emit);
    tokens_to_i64_kernel<<<(B * n_emit + 255) / 256, 256, 0, st>>>(e->out_tok, e->out_ld, dev_out_tokens, B, n_emit, max_new);
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
// C ABI: continuous batching.  The reference's worker admits up to 5 concurrent requests
// (serve/model_worker.py:161-172,216-229) and runs each as its own HF generate; here they share ONE decode loop: every
// row ("slot") of the batch is a request with its own sampling parameters, budget, EOS and stop sequence, requests join
// (prefill into free slots while the others keep their KV pages) and leave at any step, and the captured decode step is kept
// per row bucket.  A request produces the same tokens as when it runs alone through sv_generate.
// ------------------------------------------------------------------------------------------------
static int cb_bucket(const sv_engine* e) {
    int hi = 0;
    for (int s2 = 0; s2 < e->cfg.max_batch; ++s2) if (e->cb_used[s2]) hi = s2 + 1;
    int b = 8;
    while (b < hi) b <<= 1;
    return b > e->cfg.max_batch ? e->cfg.max_batch : b;
}

static void cb_step_args(sv_engine* e, CbStepArgs& a, const int32_t* map) {
    a.logits = e->logits; a.ld = e->Vpad; a.V = e->cfg.vocab; a.slots = e->cb_slots; a.slot_map = map;
    a.cur_tok = e->cur_tok; a.positions = e->positions; a.out_tokens = e->out_tok; a.ld_out = e->out_ld;
    a.seen = e->seen; a.seen_words = e->seen_words; a.n_live = e->cb_nlive; a.events = e->cb_events; a.bad = e->d_bad;
}

static int cb_begin(sv_engine* e, hipStream_t st) {
    if (e->cb_active) return 0;
    e->free_pages.clear();
    for (int p = e->num_pages - 1; p >= 0; --p) e->free_pages.push_back(p);
    std::fill(e->cb_used.begin(), e->cb_used.end(), 0);
    for (auto& v : e->cb_pages) v.clear();
    const size_t R = (size_t)e->MT * 32;
    std::vector<int32_t> table((size_t)e->cfg.max_batch * e->pages_per_seq, e->trash_page);
    HIPCHECK(hipMemcpyAsync(e->block_table, table.data(), table.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemsetAsync(e->cb_slots, 0, R * sizeof(CbSlot), st));
    HIPCHECK(hipMemsetAsync(e->positions, 0, R * sizeof(int32_t), st));
    HIPCHECK(hipMemsetAsync(e->cur_tok, 0, R * sizeof(int32_t), st));
    HIPCHECK(hipMemsetAsync(e->cb_nlive, 0, sizeof(int32_t), st));
    HIPCHECK(hipMemsetAsync(e->cb_events, 0, sizeof(int32_t), st));
    HIPCHECK(hipStreamSynchronize(st));                  // `table` is a host temporary
    e->cb_active = true;
    e->cached_B = 0;
    return 0;
}

extern "C" int sv_cb_admit(sv_engine* e, const void* dev_embeds, int32_t n, int32_t S0, const sv_cb_request* reqs,
                           int32_t* slots_out, sv_stream stream) {
    SVCHECK(check_ready(e));
    if (!dev_embeds || !reqs || !slots_out || n < 1) return fail(SV_EINVAL, "sv_cb_admit: null argument or empty batch");
    const sv_config& c = e->cfg;
    if (n > c.max_batch) return fail(SV_EINVAL, "sv_cb_admit: %d requests exceed max_batch %d", n, c.max_batch);
    if (S0 < 1) return fail(SV_EINVAL, "sv_cb_admit: bad prompt length %d", S0);
    for (int i = 0; i < n; ++i) {
        const sv_cb_request& r = reqs[i];
        if (r.max_new_tokens < 1 || S0 + r.max_new_tokens > c.max_seq_len)
            return fail(SV_EINVAL, "sv_cb_admit: request %d: prompt %d + max_new_tokens %d out of range (max_seq_len %d)", i, S0, r.max_new_tokens, c.max_seq_len);
        if (r.n_stop < 0 || r.n_stop > SV_CB_MAXSTOP) return fail(SV_EINVAL, "sv_cb_admit: request %d: stop sequence length %d unsupported (0..%d)", i, r.n_stop, SV_CB_MAXSTOP);
        if (r.do_sample && !(r.temperature > 0.f && r.top_p > 0.f)) return fail(SV_EINVAL, "sv_cb_admit: request %d: temperature and top_p must be > 0", i);
    }
    std::lock_guard<std::mutex> lk(e->mu);
    HIPCHECK(hipSetDevice(c.device));
    HIPCHECK(hipEventRecord(e->gen_event, (hipStream_t)stream));
    hipStream_t st = e->gen_stream;
    HIPCHECK(hipStreamWaitEvent(st, e->gen_event, 0));
    SVCHECK(cb_begin(e, st));
    // free slots (lowest first: keeps the row bucket of the decode graph small) and pages for the whole budget
    std::vector<int> slots;
    size_t need_pages = 0;
    for (int s2 = 0; s2 < c.max_batch && (int)slots.size() < n; ++s2) if (!e->cb_used[s2]) slots.push_back(s2);
    for (int i = 0; i < n; ++i) need_pages += (size_t)(S0 + reqs[i].max_new_tokens + SV_PAGE_TOKENS - 1) / SV_PAGE_TOKENS;
    if ((int)slots.size() < n || need_pages > e->free_pages.size())
        return fail(SV_EBUSY, "sv_cb_admit: %d requests need %d slots / %zu KV pages, %zu / %zu are free (release finished slots first)",
                    n, n, need_pages, slots.size(), e->free_pages.size());
    std::vector<int32_t> rows((size_t)n * e->pages_per_seq, e->trash_page);
    std::vector<CbSlot> hs(n);
    std::vector<int32_t> map(n), pos(n, S0 - 1);
    const bool any_pen = [&] { for (int i = 0; i < n; ++i) if (reqs[i].repetition_penalty > 0.f && reqs[i].repetition_penalty != 1.0f) return true; return false; }();
    for (int i = 0; i < n; ++i) {
        const int s2 = slots[i];
        const sv_cb_request& r = reqs[i];
        const int need = (S0 + r.max_new_tokens + SV_PAGE_TOKENS - 1) / SV_PAGE_TOKENS;
        e->cb_pages[s2].clear();
        for (int k = 0; k < need; ++k) {
            rows[(size_t)i * e->pages_per_seq + k] = e->free_pages.back();
            e->cb_pages[s2].push_back(e->free_pages.back());
            e->free_pages.pop_back();
        }
        e->cb_used[s2] = 1;
        CbSlot& h = hs[i];
        memset(&h, 0, sizeof(h));
        h.live = 1; h.step = 0; h.budget = r.max_new_tokens; h.do_sample = r.do_sample ? 1 : 0; h.temperature = r.temperature;
        h.top_p = r.top_p; h.top_k = r.top_k; h.eos = r.eos_token_id; h.pad = r.pad_token_id; h.min_new = r.min_new_tokens;
        h.penalty = r.repetition_penalty > 0.f ? r.repetition_penalty : 1.0f; h.n_stop = r.n_stop; h.seed = r.seed;
        for (int k = 0; k < r.n_stop; ++k) h.stop[k] = r.stop_ids[k];
        map[i] = s2;
        slots_out[i] = s2;
    }
    // Device side.  Any failure below rolls the host bookkeeping back (slots, pages) and parks the slots' device state, so a
    // failed admit leaks nothing and the caller may simply retry: the slots it was told about are NOT in use on error.
    bool nlive_added = false;
    const int rc = [&]() -> int {
    for (int i = 0; i < n; ++i) {
        const int s2 = slots[i];
        HIPCHECK(hipMemcpyAsync(e->block_table + (size_t)s2 * e->pages_per_seq, rows.data() + (size_t)i * e->pages_per_seq,
                                e->pages_per_seq * sizeof(int32_t), hipMemcpyHostToDevice, st));
        HIPCHECK(hipMemcpyAsync(e->cb_slots + s2, &hs[i], sizeof(CbSlot), hipMemcpyHostToDevice, st));
        HIPCHECK(hipMemcpyAsync(e->positions + s2, &pos[i], sizeof(int32_t), hipMemcpyHostToDevice, st));
        if (any_pen) HIPCHECK(hipMemsetAsync(e->seen + (size_t)s2 * e->seen_words, 0, e->seen_words * sizeof(uint32_t), st));
    }
    HIPCHECK(hipMemcpyAsync(e->cb_table_pf, rows.data(), rows.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    HIPCHECK(hipMemcpyAsync(e->cb_map, map.data(), n * sizeof(int32_t), hipMemcpyHostToDevice, st));
    add_i32_kernel<<<1, 64, 0, st>>>(e->cb_nlive, n, 1);
    nlive_added = true;
    // prompt pass of the NEW requests only (their pages through cb_table_pf); the live slots keep decoding afterwards
    SVCHECK(prefill_forward(e, (const bf16_t*)dev_embeds, n, S0, st, 0, nullptr, e->cb_table_pf));
    CbStepArgs a;
    cb_step_args(e, a, e->cb_map);
    launch_cb_step(a, n, st);                              // first token of every new request, from the prefill logits
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipStreamSynchronize(st));                    // the staging vectors above are host temporaries
    return 0;
    }();
    if (rc) {
        const std::string why = g_err;                      // keep the first error's text
        (void)hipStreamSynchronize(st);
        (void)hipGetLastError();
        std::vector<int32_t> trash(e->pages_per_seq, e->trash_page);
        for (int i = n - 1; i >= 0; --i) {                  // pages go back in reverse order: the free list is as it was
            const int s2 = slots[i];
            for (size_t k = e->cb_pages[s2].size(); k-- > 0;) e->free_pages.push_back(e->cb_pages[s2][k]);
            e->cb_pages[s2].clear();
            e->cb_used[s2] = 0;
            slots_out[i] = -1;
            // best effort on the device: the slot is dead and its block-table row points at the trash page again
            (void)hipMemsetAsync(e->cb_slots + s2, 0, sizeof(CbSlot), st);
            (void)hipMemcpyAsync(e->block_table + (size_t)s2 * e->pages_per_seq, trash.data(), trash.size() * sizeof(int32_t),
                                 hipMemcpyHostToDevice, st);
        }
        if (nlive_added) add_i32_kernel<<<1, 64, 0, st>>>(e->cb_nlive, -n, 1);
        (void)hipStreamSynchronize(st);
        (void)hipGetLastError();
        g_err = why;
        return rc;
    }
    return 0;
}

extern "C" int sv_cb_step(sv_engine* e, int32_t n_steps, int32_t* n_live_out, sv_stream stream) {
    SVCHECK(check_ready(e));
    if (n_steps < 1 || !n_live_out) return fail(SV_EINVAL, "sv_cb_step: bad argument");
    std::lock_guard<std::mutex> lk(e->mu);
    if (!e->cb_active) return fail(SV_ESTATE, "sv_cb_step: no continuous batch (sv_cb_admit first)");
    HIPCHECK(hipSetDevice(e->cfg.device));
    hipStream_t st = e->gen_stream;
    const int Bb = cb_bucket(e);
    hipGraphExec_t gexec = nullptr;
    if (getenv("SV_NO_GRAPH") == nullptr) {
        auto it = e->cb_graphs.find(Bb);
        if (it != e->cb_graphs.end()) {
            gexec = it->second.second;
        } else {
            hipGraph_t g = nullptr;
            hipGraphExec_t ge = nullptr;
            hipError_t ce = hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed);
            if (ce == hipSuccess) {
                decode_forward(e, Bb, st);
                CbStepArgs a;
                cb_step_args(e, a, nullptr);
                launch_cb_step(a, Bb, st);
                ce = hipStreamEndCapture(st, &g);
                if (ce == hipSuccess && g) ce = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
            }
            if (ce != hipSuccess) {
                (void)hipGetLastError();
                if (ge) (void)hipGraphExecDestroy(ge);
                if (g) (void)hipGraphDestroy(g);
                if (getenv("SV_REQUIRE_GRAPH")) return fail(SV_EHIP, "hipGraph capture failed: %s", hipGetErrorString(ce));
            } else {
                e->cb_graphs[Bb] = {g, ge};                // kept for the life of the engine: every argument is engine-owned
                gexec = ge;
            }
        }
    }
    for (int i = 0; i < n_steps; ++i) {
        if (gexec) {
            HIPCHECK(hipGraphLaunch(gexec, st));
        } else {
            decode_forward(e, Bb, st);
            CbStepArgs a;
            cb_step_args(e, a, nullptr);
            launch_cb_step(a, Bb, st);
        }
    }
    HIPCHECK(hipMemcpyAsync(&e->h_flags[3], e->cb_nlive, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIPCHECK(hipStreamSynchronize(st));
    *n_live_out = e->h_flags[3];
    SVCHECK(check_finite_logits(e, st, "sv_cb_step"));
    e->timing_graph = gexec ? 1.0 : 0.0;
    return 0;
}

extern "C" int sv_cb_poll(sv_engine* e, int32_t* host_live, int32_t* host_steps, int32_t capacity) {
    if (!e || !host_live || !host_steps) return fail(SV_EINVAL, "sv_cb_poll: null argument");
    std::lock_guard<std::mutex> lk(e->mu);
    if (capacity < e->cfg.max_batch) return fail(SV_EINVAL, "sv_cb_poll: capacity %d < max_batch %d", capacity, e->cfg.max_batch);
    for (int s2 = 0; s2 < e->cfg.max_batch; ++s2) { host_live[s2] = 0; host_steps[s2] = 0; }
    if (!e->cb_active) return 0;
    HIPCHECK(hipSetDevice(e->cfg.device));
    std::vector<CbSlot> hs(e->cfg.max_batch);
    HIPCHECK(hipMemcpyAsync(hs.data(), e->cb_slots, hs.size() * sizeof(CbSlot), hipMemcpyDeviceToHost, e->gen_stream));
    HIPCHECK(hipStreamSynchronize(e->gen_stream));
    for (int s2 = 0; s2 < e->cfg.max_batch; ++s2)
        if (e->cb_used[s2]) { host_live[s2] = hs[s2].live; host_steps[s2] = hs[s2].step; }
    return 0;
}

extern "C" int sv_cb_read(sv_engine* e, int32_t slot, int32_t first, int32_t count, int64_t* host_tokens) {
    if (!e || !host_tokens) return fail(SV_EINVAL, "sv_cb_read: null argument");
    std::lock_guard<std::mutex> lk(e->mu);
    if (!e->cb_active || slot < 0 || slot >= e->cfg.max_batch || !e->cb_used[slot]) return fail(SV_EINVAL, "sv_cb_read: slot %d is not in use", slot);
    if (first < 0 || count < 0 || first + count > e->out_ld) return fail(SV_EINVAL, "sv_cb_read: columns [%d, %d) out of range", first, first + count);
    if (count == 0) return 0;
    HIPCHECK(hipSetDevice(e->cfg.device));
    std::vector<int32_t> tmp(count);
    HIPCHECK(hipMemcpyAsync(tmp.data(), e->out_tok + (size_t)slot * e->out_ld + first, (size_t)count * sizeof(int32_t),
                            hipMemcpyDeviceToHost, e->gen_stream));
    HIPCHECK(hipStreamSynchronize(e->gen_stream));
    for (int i = 0; i < count; ++i) host_tokens[i] = tmp[i];
    return 0;
}

static int cb_release_locked(sv_engine* e, int slot, hipStream_t st) {
    // a slot released while still generating is stopped first (live -> 0, live counter adjusted on the host's view)
    CbSlot h;
    HIPCHECK(hipMemcpyAsync(&h, e->cb_slots + slot, sizeof(CbSlot), hipMemcpyDeviceToHost, st));
    HIPCHECK(hipStreamSynchronize(st));
    if (h.live) add_i32_kernel<<<1, 64, 0, st>>>(e->cb_nlive, -1, 1);
    HIPCHECK(hipMemsetAsync(e->cb_slots + slot, 0, sizeof(CbSlot), st));
    HIPCHECK(hipMemsetAsync(e->positions + slot, 0, sizeof(int32_t), st));
    HIPCHECK(hipMemsetAsync(e->cur_tok + slot, 0, sizeof(int32_t), st));
    fill_i32_kernel<<<(e->pages_per_seq + 63) / 64, 64, 0, st>>>(e->block_table + (size_t)slot * e->pages_per_seq, e->trash_page, e->pages_per_seq);
    for (int pg : e->cb_pages[slot]) e->free_pages.push_back(pg);
    e->cb_pages[slot].clear();
    e->cb_used[slot] = 0;
    HIPCHECK(hipGetLastError());
    return 0;
}

extern "C" int sv_cb_release(sv_engine* e, int32_t slot) {
    if (!e) return fail(SV_EINVAL, "null engine");
    std::lock_guard<std::mutex> lk(e->mu);
    if (!e->cb_active || slot < 0 || slot >= e->cfg.max_batch || !e->cb_used[slot]) return fail(SV_EINVAL, "sv_cb_release: slot %d is not in use", slot);
    HIPCHECK(hipSetDevice(e->cfg.device));
    return cb_release_locked(e, slot, e->gen_stream);
}

extern "C" int sv_cb_reset(sv_engine* e) {
    if (!e) return fail(SV_EINVAL, "null engine");
    std::lock_guard<std::mutex> lk(e->mu);
    if (!e->cb_active) return 0;
    HIPCHECK(hipSetDevice(e->cfg.device));
    for (int s2 = 0; s2 < e->cfg.max_batch; ++s2)
        if (e->cb_used[s2]) SVCHECK(cb_release_locked(e, s2, e->gen_stream));
    HIPCHECK(hipStreamSynchronize(e->gen_stream));
    e->cb_active = false;
    return 0;
}

// ------------------------------------------------------------------------------------------------
// C ABI: host-side decisions, callable without a GPU (CPU tests)
// ------------------------------------------------------------------------------------------------
// A/B tool surface (tools/ab_exp.py): change the experiment mask of a live engine; captured graphs are keyed on it
extern "C" int sv_debug_set_exp(sv_engine* e, int32_t mask) {
    if (!e) return fail(SV_EINVAL, "null engine");
    std::lock_guard<std::mutex> lk(e->mu);
    e->exp = mask;
    for (auto& kv : e->cb_graphs) {          // the continuous-batching step graphs were captured with the old mask
        if (kv.second.second) (void)hipGraphExecDestroy(kv.second.second);
        if (kv.second.first) (void)hipGraphDestroy(kv.second.first);
    }
    e->cb_graphs.clear();
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

extern "C" int sv_debug_attn_plan(int32_t max_batch, int32_t n_kv_head, int32_t num_cus, int32_t* out2) {
    if (!out2 || max_batch < 1 || n_kv_head < 1 || num_cus < 1) return fail(SV_EINVAL, "sv_debug_attn_plan: bad argument");
    out2[0] = attn_max_splits_of(max_batch, n_kv_head, num_cus);
    out2[1] = attn_groups_per_block_of(max_batch, n_kv_head, num_cus);
    return 0;
}

extern "C" int sv_debug_set_col_tiles(int32_t col_tiles) {
    if (col_tiles < 0 || col_tiles > 3) return fail(SV_EINVAL, "sv_debug_set_col_tiles: 0..3");
    g_op_col_tiles = col_tiles;
    return 0;
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
    add_i32_kernel<<<(B + 63) / 64, 64, 0, st>>>(e->positions, -1, B);
    for (int k = 0; k < 2 * PK_COUNT + 2; ++k) out[k] = 0.0;
    // event-pair overhead: two back-to-back events with nothing in between
    double overhead_ms = 0.0;
    {
        hipEvent_t a, b;
        HIPCHECK(hipEventCreate(&a)); HIPCHECK(hipEventCreate(&b));
        float acc = 0.f;
        for (int i = 0; i < 20; ++i) {
            HIPCHECK(hipEventRecord(a, st)); HIPCHECK(hipEventRecord(b, st));
            HIPCHECK(hipEventSynchronize(b));
            float ms = 0.f; HIPCHECK(hipEventElapsedTime(&ms, a, b)); acc += ms;
        }
        overhead_ms = acc / 20.0;
        (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    }
    for (int it = 0; it < iters + 1; ++it) {
        e->prof_on = true; e->prof_used = 0;
        decode_forward(e, B, st);
        e->prof_on = false;
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
    {
        hipEvent_t a, b;
        HIPCHECK(hipEventCreate(&a)); HIPCHECK(hipEventCreate(&b));
        e->only_skinny = true;
        decode_forward(e, B, st);
        HIPCHECK(hipEventRecord(a, st));
        for (int it = 0; it < iters; ++it) decode_forward(e, B, st);
        HIPCHECK(hipEventRecord(b, st));
        e->only_skinny = false;
        HIPCHECK(hipEventSynchronize(b));
        float ms = 0.f; HIPCHECK(hipEventElapsedTime(&ms, a, b));
        (void)hipEventDestroy(a); (void)hipEventDestroy(b);
        out[2 * PK_SAMPLE + 1] = (double)ms / iters;       // ms per step for the GEMM chain
        // slot 8: the complement -- every other kernel of the step (row updates, attention) back to back, no GEMMs.  The step
        // minus this chain is what the GEMMs cost IN SITU (bench.py's roofline.avg_launch_us)
        HIPCHECK(hipEventCreate(&a)); HIPCHECK(hipEventCreate(&b));
        e->skip_skinny = true;
        decode_forward(e, B, st);
        HIPCHECK(hipEventRecord(a, st));
        for (int it = 0; it < iters; ++it) decode_forward(e, B, st);
        HIPCHECK(hipEventRecord(b, st));
        e->skip_skinny = false;
        HIPCHECK(hipEventSynchronize(b));
        HIPCHECK(hipEventElapsedTime(&ms, a, b));
        (void)hipEventDestroy(a); (void)hipEventDestroy(b);
        out[2 * PK_COUNT] = (double)ms / iters;
    }
    add_i32_kernel<<<(B + 63) / 64, 64, 0, st>>>(e->positions, 1, B);
    HIPCHECK(hipStreamSynchronize(st));
    return 0;
}

extern "C" int sv_last_timing(sv_engine* e, double* out3) {   /* 4 doubles */
    if (!e || !out3) return fail(SV_EINVAL, "null argument");
    out3[0] = e->timing[0]; out3[1] = e->timing[1]; out3[2] = e->timing[2]; out3[3] = e->timing_graph;
    return 0;
}

// ------------------------------------------------------------------------------------------------
// C ABI: single operators (test surface)
// ------------------------------------------------------------------------------------------------
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
    fill_random_bf16_kernel<<<4096, 256, 0, st>>>(A, (size_t)M * K, 1u);
    fill_random_bf16_kernel<<<4096, 256, 0, st>>>(Wsrc, (size_t)N * K, 2u);
    fill_random_bf16_kernel<<<64, 256, 0, st>>>(bias, (size_t)N, 3u);
    fill_random_bf16_kernel<<<4096, 256, 0, st>>>(C, (size_t)M * N, 4u);
    launch_pack_weight(Wsrc, 0, Wp, N, K, Npad, K, st);
    GemmArgs g;
    g.A = A; g.lda = K; g.Wp = Wp; g.bias = bias; g.R = residual ? C : nullptr; g.ldr = N; g.C = C; g.ldc = N;
    g.M = M; g.N = N; g.K = K; g.act = act; g.out_f32 = 0;
    hipEvent_t e0, e1;
    HIPCHECK(hipEventCreate(&e0)); HIPCHECK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) launch_gemm(g, st);
    HIPCHECK(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i) launch_gemm(g, st);
    HIPCHECK(hipEventRecord(e1, st));
    HIPCHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    *avg_us = (double)ms * 1e3 / iters;
    HIPCHECK(hipGetLastError());
    return 0;
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
    pack_rows_kernel<<<(M * (K / 8) + 255) / 256, 256, 0, st>>>((const bf16_t*)x, K, xp, M, K);
    SkinnyArgs a;
    memset(&a, 0, sizeof(a));
    a.xp = xp; a.Wp = Wp; a.MT = MT; a.Npad = Npad; a.K = K; a.splitk = splitk; a.out_mode = SK_OUT_PARTIAL;
    a.ws = ws; a.ldws = Npad; a.N = N;
    launch_gemm_skinny(a, st);
    reduce_partials_kernel<<<(M * N + 255) / 256, 256, 0, st>>>(ws, splitk, MT * 32, Npad, (const bf16_t*)bias,
                                                                 (float*)y_f32, M, N);
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
    pack_rows_kernel<<<(M * (K / 8) + 255) / 256, 256, 0, st>>>((const bf16_t*)x, K, xp, M, K);
    SkinnyArgs a;
    memset(&a, 0, sizeof(a));
    a.xp = xp; a.Wp = Wp; a.Wq = Wq; a.wscale = sc; a.MT = MT; a.Npad = Npad; a.K = K; a.splitk = splitk;
    a.out_mode = SK_OUT_PARTIAL; a.ws = ws; a.ldws = Npad; a.N = N;
    launch_gemm_skinny(a, st);
    reduce_partials_kernel<<<(M * N + 255) / 256, 256, 0, st>>>(ws, splitk, MT * 32, Npad, (const bf16_t*)bias,
                                                                 (float*)y_f32, M, N);
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
    pack_rows_kernel<<<(M * (K / 8) + 255) / 256, 256, 0, st>>>((const bf16_t*)x, K, xp, M, K);
    SkinnyArgs a;
    memset(&a, 0, sizeof(a));
    a.xp = xp; a.Wp = Wp; a.MT = MT; a.Npad = Npad; a.K = K; a.splitk = 1; a.N = N;
    if (out_f32) { a.out_mode = SK_OUT_F32; a.out_f32 = of; a.ldo = Npad; a.round_bf16 = 1; }
    else { a.out_mode = SK_OUT_PACKED_ACT; a.bias = (const bf16_t*)bias; a.act = act; a.out_xp = oxp; a.out_KS = Npad / 16; }
    launch_gemm_skinny(a, st);
    if (out_f32) HIPCHECK(hipMemcpy2DAsync(y, (size_t)N * 4, of, (size_t)Npad * 4, (size_t)N * 4, M, hipMemcpyDeviceToDevice, st));
    else unpack_rows_kernel<<<(M * (N / 8) + 255) / 256, 256, 0, st>>>(oxp, (bf16_t*)y, N, M, N);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipStreamSynchronize(st));
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
    pack_rows_kernel<<<(M * (Kp / 8) + 255) / 256, 256, 0, st>>>((const bf16_t*)x, Kp, xp, M, Kp);
    pack_rows_kernel<<<(M * (D / 8) + 255) / 256, 256, 0, st>>>((const bf16_t*)h, D, hxp, M, D);
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
    unpack_rows_kernel<<<(M * (D / 8) + 255) / 256, 256, 0, st>>>(hxp, (bf16_t*)h2_out, D, M, D);
    unpack_rows_kernel<<<(M * (F / 8) + 255) / 256, 256, 0, st>>>(yxp, (bf16_t*)y_out, F, M, F);
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
