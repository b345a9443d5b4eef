// Internal header of the host driver (engine_*.hip): engine state, error plumbing and the functions the translation units
// share.  Not part of the C ABI (include/starvector_hip.h is); nothing here is exported by name to callers.
#pragma once
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
#include "../../include/starvector_hip_debug.h"
#include "kernels.h"
#include "beam.h"

using namespace sv;

// ------------------------------------------------------------------------------------------------
// errors (engine_core.hip owns the thread-local message)
// ------------------------------------------------------------------------------------------------
namespace sveng {
std::string& last_error();
int fail(int code, const char* fmt, ...);
}  // namespace sveng
using sveng::fail;
#define g_err (sveng::last_error())
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
    // sv_config.exclusive_device == 2 ("optimistic"): the fused launches are on until one of them gives up (another tenant holds CUs); then they are off for the
    // rest of the engine's life and the failed sv_generate call is run again without them (engine_generate.hip)
    bool fused_off = false;
    int last_giveup = 0;            // code (3 / 4) of the give-up the last report_bad_logits saw, 0: none
    int stream_skip = 0;            // columns the failed attempt of an optimistic call already handed to the streaming callback
    bool fold6 = false;             // 6 launches per layer: slab-free attention output projection + ln_2 folded into c_fc
    bool fold_ready = false;
    bf16_t* xp_f = nullptr;         // ln_f output of the last row update (the lm_head's operand) when xp_a belongs to the fused row-update + c_attn launch
    long long* rc_dbg = nullptr;    //   [8] what the first GEMM wave to give up saw
    hipStream_t tenant_stream = nullptr;       // sv_debug_occupy_cus (safety tests): the stream the foreign tenant's kernel runs on
    int rc_delay = 390;             //   narrow rows: its GEMM blocks' first poll, 10-ns ticks after block start (3.9 us: measured optimum); wide rows: their
                                    //   hold-back in front of the weight requests (0); SV_RC_DELAY at sv_create
    int step_nodes = 0; bool step_rc = false, step_sel = false, step_mlp = false;      // sv_debug_step_plan: the last captured / launched decode step
    bool xpa_armed = false;         // xp_a carries the "not written yet" pattern, put there by a write-through fill of the lm_head launch in front (layer 0 of the next fused step polls it)
    bool rc_fused_ok = false;       // row update + c_attn as one launch (rowops.hip rowln_cattn_kernel) fits this engine: shapes + all blocks resident
    bool mlp_fused_ok = false;      // the MLP half as one launch (gemm.hip mlp_fused_kernel) fits this engine: shapes + one block per CU
    long long* attn_trace = nullptr;// SV_ATTN_TRACE=1: wall-clock stamps of the decode attention of the middle layer, [rows * kv heads * splits][16]
    long long* mlp_trace = nullptr; // SV_MLP_TRACE=1: wall-clock stamps of the LAST fused MLP launch, [F / 32][8] (sv_debug_mlp_trace)
    bool only_skinny = false;       // profiling: enqueue only the weight-streaming GEMMs of a step
    bool skip_skinny = false;       // profiling: enqueue everything BUT the weight-streaming GEMMs
    int exp = 0;                    // SV_EXP bit mask, read once at sv_create (A/B switches of the round's experiments):
                                    //   2 the 7-launch layer (no LayerNorm fold);
                                    //   (1: was the row update as one wave per row: 0.218 vs 0.131 ms per step, removed)
                                    //   8 (at sv_create only) the round 1-2 split-K rule of the decode GEMMs
                                    //   8192 / 16384 (round 5) the row update and c_attn as ONE launch (rowln_cattn_kernel) forced off / on; default: on iff exclusive_device
                                    //   1024 (round 5) greedy selection as its own launch again (argmax_kernel), not folded into the lm_head epilogue
                                    //   128 / 512 (round 4) the MLP half of a layer as ONE launch (mlp_fused_kernel) forced on / off; default:
                                    //       on iff sv_config.exclusive_device
                                    //   (16 / 32 / 64: 2 / 6 / 8 key groups per attention block: 1186 / 1169 / 1175 vs 1171 us, removed)
                                    //   (1, 2: XCD-aligned weight prefetch by attention's idle waves / spare row-update blocks; 4: one key
                                    //    group per attention block -- all measured slower, profiles/prefetch_r03_*.log, removed)
    float *ws = nullptr, *ws2 = nullptr, *logits = nullptr, *sample_scratch = nullptr, *attn_part = nullptr;
    unsigned* attn_cnt = nullptr;
    float* am_val = nullptr; int32_t* am_idx = nullptr;
    // the step's bookkeeping folded into the lm_head launch (SkinnyArgs::finish): set by sv_generate for the decode steps of a call whose selection is folded
    // (greedy_fused), read by decode_forward; fin_folded = the lm_head launch just issued took the bookkeeping (sample_and_finish launches no finish_step_kernel)
    FinishArgs fin_args; bool fin_fold = false, fin_folded = false; unsigned* fin_cnt = nullptr;
    float* tail_ws = nullptr; unsigned* tail_cnt = nullptr;      // SkinnyArgs::tail_ws / ::tail_cnt (gemm_skinny_tailsplit_kernel): partials + zeroed arrival tickets
    unsigned long long* amax = nullptr;   // greedy selection folded into the lm_head launch: one 64-bit key per row, [64 * SV_AMAX_STRIDE]
    bool greedy_fused = false;            //   on for the decode steps of the current sv_generate call (set and cleared by it)
    uint32_t* seen = nullptr; int seen_words = 0;      // repetition-penalty bitmap [rows][Vpad/32]
    int32_t *cur_tok = nullptr, *next_tok = nullptr, *unfinished = nullptr, *positions = nullptr,
            *out_tok = nullptr, *d_step = nullptr, *d_done = nullptr, *d_nemit = nullptr, *d_stop = nullptr, *d_bad = nullptr;
    int out_ld = 0;
    int32_t* h_flags = nullptr;   // pinned: [0]=done [1]=n_emitted [4]=bad; [8..11] = a snapshot of the device block {step, done, n_emitted, bad}
    int32_t* h_table = nullptr;   // pinned host image of block_table (assign_pages); table_ev = its last upload
    hipEvent_t table_ev = nullptr;
    bool table_pending = false;
    // KV pool
    char* kv_pool = nullptr;
    size_t layer_stride = 0;
    int pages_per_seq = 0, num_pages = 0, page_bytes = 0;
    int32_t* block_table = nullptr;
    std::vector<int> free_pages;
    // beam search (num_beams > 1): device scorer
    BeamScorer beam;
    bf16_t* score_ws = nullptr;      // scoring forward: kept hidden rows, their ln_f, bf16 logits [rows][Vpad]
    size_t score_elems = 0;
    int cached_B = 0;
    int dbg_pos_hi = 0;             // sv_debug_kv_load / sv_debug_attn_decode: upper bound of positions[] (host side), kept below max_seq_len
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
    // the same step SV_GRAPH_STEPS times in ONE graph (engine_generate.hip: a hipGraphLaunch-to-hipGraphLaunch boundary costs 8.6 us of idle GPU per step,
    // a kernel-to-kernel boundary inside a graph nothing measurable -- profiles/step_gaps_r06.log); built for calls long enough to pay for its instantiation
    hipGraph_t gen_graph_multi = nullptr;
    hipGraphExec_t gen_gexec_multi = nullptr;
    int gen_multi_steps = 0;
    // optional per-kernel HIP-event profiling of the decode step (bench.py roofline leg)
    bool prof_on = false;
    std::vector<hipEvent_t> prof_ev;
    std::vector<int> prof_kind;
    size_t prof_used = 0;
};

enum { PK_SKINNY = 0, PK_ATTN = 1, PK_ROWLN = 2, PK_SAMPLE = 3, PK_COUNT = 4,
       // stages of the time to first token (sv_profile_ttft): never marked by decode_forward
       PK_VIT_GEMM = 4, PK_VIT_ATTN, PK_VIT_ROWS, PK_AD_GEMM, PK_AD_NORM, PK_PF_GEMM, PK_GEMM_TAIL, PK_PF_ATTN, PK_PF_ROWS, PK_PF_LMHEAD,
       PK_FIRST_SAMPLE, PK_END, PK_TTFT_FIRST = PK_VIT_GEMM, PK_TTFT_COUNT = PK_END - PK_VIT_GEMM };

namespace sveng {
// engine_core.hip: small utility kernels behind host wrappers, allocation, decode planning
void fill_i32(int32_t* p, int32_t v, int n, hipStream_t st);
void gen_state_init(int32_t* positions, int32_t pos0, int32_t* unfinished, int B, int32_t* state3, unsigned long long* amax, int n_amax, hipStream_t st);
void add_i32(int32_t* p, int32_t v, int n, hipStream_t st);
void suppress_token(float* logits, int ld, int token, const int32_t* step, int min_new, int B, hipStream_t st);
void tokens_to_i64(const int32_t* src, int ld, int64_t* dst, int B, int ncols, int dst_ld, hipStream_t st);
void fill_random_bf16(bf16_t* p, size_t n, unsigned seed, int blocks, hipStream_t st);
void pack_rows(const bf16_t* x, int ldx, bf16_t* xp, int M, int K, hipStream_t st);
void unpack_rows(const bf16_t* xp, bf16_t* x, int ldx, int M, int K, hipStream_t st);
void reduce_partials(const float* ws, int splitk, int rows_ws, int ldws, const bf16_t* bias, float* y, int M, int N, hipStream_t st);
int dev_alloc(sv_engine* e, void** p, size_t bytes, bool zero = true);
template <typename T>
inline int dalloc(sv_engine* e, T** p, size_t count, bool zero = true) {
    return dev_alloc(e, reinterpret_cast<void**>(p), count * sizeof(T), zero);
}
void pick_decode_plan(const Linear& l, int MT, int num_cus, bool fp8, bool legacy, bool whole_k, int* splitk, int* col_tiles);
// engine_forward.hip: the op graphs
void prof_mark(sv_engine* e, int kind, hipStream_t st);
int attn_max_splits_of(int max_batch, int nkv, int num_cus);
int attn_groups_per_block_of(int max_batch, int nkv, int num_cus);
int assign_pages(sv_engine* e, int B, int total_len, hipStream_t st);
int prefill_forward(sv_engine* e, const bf16_t* embeds, int B, int S0, hipStream_t st, int n_keep = 0,
                    bf16_t* dev_scores = nullptr, const int32_t* table = nullptr);
void decode_forward(sv_engine* e, int B, hipStream_t st);
void attn_decode_args(sv_engine* e, int layer, int B, const float* ws, int splitk, const bf16_t* bias, bf16_t* out_xp, AttnDecodeArgs& ad);
int check_ready(sv_engine* e);
int cb_guard(sv_engine* e, const char* who);
int prefill_locked(sv_engine* e, const void* dev_embeds, int B, int S0, int total_len, hipStream_t st, bool set_positions = true);
// engine_generate.hip
int check_finite_logits(sv_engine* e, hipStream_t st, const char* who);
int report_bad_logits(sv_engine* e, hipStream_t st, const char* who, int what);      // what = the d_bad code already read (0: fine)
}  // namespace sveng
using namespace sveng;
