// Internal host-side launchers for the gfx950 kernels (not part of the public C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include "common.h"

namespace sv {

inline int round_up(int a, int b) { return (a + b - 1) / b * b; }

// ---- weights -----------------------------------------------------------------------------------
// src: [N][K] row-major (bf16 if src_is_f32==0 else float).  dst: packed [Npad/32][Kpad/16][64][8].
void launch_pack_weight(const void* src, int src_is_f32, bf16_t* dst, int N, int K, int Npad, int Kpad,
                        hipStream_t st);
void launch_convert_to_bf16(const void* src, int src_is_f32, bf16_t* dst, size_t n, hipStream_t st);
// fp8 (OCP e4m3) weight-only quantisation, one scale per output row: scale[n] = max_k |W[n][k]| / 448,
// q = fp8_rne(W / scale).  Writes the decode image Wq [Npad/32][Kpad/32][64 lanes][16 B] (a lane's two consecutive
// k-steps of 8 fp8 each), the bf16 image of the SAME q values in the usual packed layout (prefill / big-M kernels, which
// apply the scale in their epilogue like the decode kernel does), and scale[Npad] (1 for padding rows).
void launch_pack_weight_fp8(const void* src, int src_is_f32, bf16_t* dst_bf16, uint8_t* dst_q, float* scale, int N, int K,
                            int Npad, int Kpad, hipStream_t st);

// ---- big-M MFMA GEMM:  C[M][N] = epi( A[M][K] . W^T + bias ) (+ residual) ----------------------
struct GemmArgs {
    const bf16_t* A; int lda;        // activations, row-major bf16, K-contiguous
    const bf16_t* Wp;                // packed weight
    const bf16_t* bias;              // [N] or nullptr
    const bf16_t* R; int ldr;        // residual [M][N] or nullptr (added after the activation)
    void* C; int ldc;                // bf16 (out_f32==0) or float
    int M, N, K;                     // K = padded K (multiple of 64) shared by A and Wp
    int act; int out_f32;
    const float* cscale = nullptr;   // fp8 weights: per-output-column scale applied to the accumulator (or nullptr)
    long long* trace = nullptr;      // optional [blocks][8] wall-clock stamps of the 256^2 kernel (tools/gemm_trace.py); nullptr in production
    void (*tail_mark)(void* ctx, hipStream_t st) = nullptr;   // profiling hook: called right before the remainder-row launch of a peeled GEMM
    void* tail_ctx = nullptr;        //   (sv_profile_ttft prices the tails apart from the tile launches); nullptr in production
    // Sequence structure of the rows (round 6, gemm.hip "rows a sequence leaves over"): the M rows are M / seq_rows sequences of seq_rows
    // rows each.  Where gemm_seq_form(seq_rows, N, K, act) holds, the first seq_rows - r rows of EVERY sequence go through the tile
    // kernels (a tile never straddles two sequences) and the last r = seq_rows % 256 rows of every sequence through the split-K
    // remainder kernel -- which kernel computes a row depends on the row's position in its sequence and on the projection, never on
    // the batch around it.  0 = plain row-major problem.
    int seq_rows = 0;
    int seq_tail = 0;                // set by the launcher for the remainder launch: row i -> (i / seq_tail) * seq_rows + seq_rows - seq_tail + i % seq_tail
    int splitk_rows = 0;             // 1: the M rows (compact, contiguous) are LAST rows of sequences of seq_rows rows (the pruned last prompt layer): they take
                                     //    the kernel the same rows take inside the full problem (split-K remainder kernel where gemm_seq_form holds)
};
// r = rows a sequence of S rows leaves over its 256-row tiles when that is a handful (32 sequences leave at most 96 rows) and a full tile exists; else 0
inline int seq_peel_rows(int S) { const int r = S % 256; return (S > 256 && r >= 1 && r <= 3) ? r : 0; }
// Does the projection (N, K, act) take the per-sequence form for sequences of S rows?  A function of the sequence length and the projection ONLY
// (never of the batch): where the dispatch's cost model peels the row remainder of a 32-sequence batch (gemm.hip gemm_plan).
bool gemm_seq_form(int S, int N, int K, int act);
void launch_gemm(const GemmArgs& a, hipStream_t st);
// one fixed configuration (kernel256: 0 = 128^2 tiles, 1 = 256^2; peel: the row remainder over a multiple of 256 in its own launch), no tuning
void launch_gemm_fixed(const GemmArgs& a, int kernel256, int peel, hipStream_t st);
void set_mt2x(int on);                // 33..64-row decode GEMMs: 1 = activations through a wave-private LDS ring (gemm_skinny_mt2x_kernel, default), 0 = both operands in registers
void set_gemm_form(int form);         // -1: tuned (default); 0 / 1: every big-M launch takes that form, rows not peeled
// what launch_gemm decides for a shape (host arithmetic only; tail_on: 0 never peel, 1 cost model, 2 always)
struct GemmPlan { int peel, tail_rows, tail_by_tiles, main_256; double est_us; };
GemmPlan gemm_plan(int M, int N, int K, int act, int tail_on);

// ---- skinny (M<=32 per tile) weight-streaming GEMM --------------------------------------------
enum { SK_OUT_PARTIAL = 0, SK_OUT_PACKED_ACT = 1, SK_OUT_F32 = 2 };
struct SkinnyArgs {
    const bf16_t* xp;                // packed activations [MT][K/16][64][8]
    const bf16_t* Wp;                // packed weight [Npad/32][K/16][64][8]
    const uint8_t* Wq;               // fp8 image of the same weight (launch_pack_weight_fp8) or nullptr: then the fp8 kernel runs
    const float* wscale;             //   with its per-column scales [Npad]
    const bf16_t* bias;              // [N] or nullptr (PACKED_ACT only)
    int MT;                          // number of 32-row tiles
    int Npad, K;                     // Npad multiple of 32, K multiple of 16
    int splitk;                      // >=1; (K/16) must be divisible by splitk*waves
    int out_mode; int act;
    int N;                           // valid columns (<= Npad)
    float* ws; int ldws;             // PARTIAL: fp32 slabs ws[split][MT*32][ldws], summed in slab order by the consumer
    bf16_t* out_xp; int out_KS;      // PACKED_ACT: fragment-order buffer with out_KS = Npad/16 k-steps
    float* out_f32; int ldo;         // F32: [MT*32][ldo]; rounded to bf16 values if round_bf16
    int round_bf16;
    int col_tiles;                   // two-row-tile kernel (33..64 rows): column tiles per block, 1..3; 0 = the launcher's default
    int xcd_remap;                   // 1: XCD-aware (tile, K slice) assignment of a split-K launch (needs splitk | 8, (Npad/32 * splitk) % 8 == 0)
    // LayerNorm fold (decode_cols.hip; split-K 1): xp is the RAW residual stream, Wp the folded image W' = bf16(W * gamma);
    // epilogue x = rstd[m] * (acc - mean[m] * c1[n]) + c2[n], the row statistics accumulated from the activation stream in the kernel
    const float* fold_c1; const float* fold_c2;            // [Npad] (nullptr: off)
    int fold_D; float fold_eps;                            // LayerNorm width (= K) and epsilon
    // Greedy selection folded into the lm_head epilogue (F32 mode, one row tile, bf16 weights): every block leaves the best
    // (value, lowest index) of its 32 columns per row in amax[row * SV_AMAX_STRIDE] through an atomic max on a 64-bit key
    // (sv_amax_key); finish_step_kernel decodes and re-arms it.  nullptr: off.
    unsigned long long* amax; int amax_rows;               // rows < amax_rows take part
    // F32 mode (lm_head), one-row-tile bf16 kernel only: a buffer the launch fills with the 0xFFFF'FFFF pattern through write-through
    // stores, 16 bytes per thread (the LayerNorm output buffer of the fused row-update + c_attn launch of the NEXT decode step)
    void* poison; unsigned poison_bytes;
    // PACKED_ACT, one row tile, whole K (StarVector-8B's c_fc: 576 column tiles on 512 block slots): the tiles beyond the first round of blocks are split four
    // ways along K by a second launch whose blocks leave fp32 partials here and elect a last arriver through `tail_cnt` (gemm_skinny_tailsplit_kernel).
    // Per-engine scratch: [SV_TAIL_TILES][4][16][64] floats + [SV_TAIL_TILES] zeroed counters; nullptr: off.
    float* tail_ws; unsigned* tail_cnt;
    // The step's bookkeeping (finish_step_kernel) folded into the lm_head launch of a greedy step whose selection is folded already (`amax`; persistent-block
    // kernel only): every block drains its key atomics and draws a ticket from `fin_cnt` (zeroed, re-armed by the last arriver); the last block's first wave
    // decodes the keys and runs the bookkeeping.  `finish` is a HOST pointer read by the launcher (the struct rides behind SkinnyArgs in that kernel's
    // arguments); nullptr: off.  skinny_head_folds_finish(a) tells the caller whether the launch takes the bookkeeping with it.
    const struct FinishArgs* finish; unsigned* fin_cnt;
};
bool skinny_head_folds_finish(const SkinnyArgs& a);
#define SV_TAIL_TILES 128
#define SV_AMAX_STRIDE 16          // one 128-byte line per row: the rows' atomics do not share an L2 line
// key = (order-preserving image of the float) << 32 | (0xFFFFFFFF - column): larger value wins, equal values -> LOWER column wins
// (torch.argmax / HF _sample's tie-break, sampling.hip argmax_pair); 0 = "no finite-or-infinite score seen" (NaN never wins).
__host__ __device__ inline unsigned long long sv_amax_key(float v, unsigned col) {
    if (v != v) return 0ull;
    if (v == 0.f) v = 0.f;                                 // -0.0 == +0.0 for the comparison the argmax kernel makes
    union { float f; unsigned u; } c; c.f = v;
    const unsigned o = (c.u & 0x80000000u) ? ~c.u : (c.u | 0x80000000u);
    return ((unsigned long long)o << 32) | (unsigned long long)(0xFFFFFFFFu - col);
}
__host__ __device__ inline int sv_amax_index(unsigned long long key) { return key ? (int)(0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull)) : 0x7fffffff; }
void launch_gemm_skinny(const SkinnyArgs& a, hipStream_t st);
// host arithmetic of the launch: waves per block (how K is cut inside a block = the summation order of a row) and whether a
// block carries two row tiles; a function of the GEMM only for the former (sv_debug_skinny_plan, CPU tests)
void skinny_plan(int Npad, int K, int splitk, int fp8, int MT, int* waves, int* two_row_tiles);
extern std::atomic<int> g_op_col_tiles;       // sv_debug_set_col_tiles: what SkinnyArgs.col_tiles == 0 means (0 = the launcher's default)

int init_gemm_kernels();        // hipFuncSetAttribute for the large-LDS variants (0 = ok)

void launch_cvt_bf16_hw(const float* x, bf16_t* y, size_t n, hipStream_t st);

// ---- the MLP half of a decode layer as one launch (gemm.hip: mlp_fused_kernel; round-4 experiment, SV_EXP bit 128) ----
struct MlpFusedArgs {
    const bf16_t* W1; const bf16_t* x1;      // folded c_fc image W' [N1pad/32][K1/16][64][8], raw residual stream in fragment order
    int N1, N1pad, K1;
    const float* fold_c1; const float* fold_c2; int fold_D; float fold_eps; int act;
    bf16_t* out_xp; int out_KS;              // the GELU output in fragment order (N1pad = 16 * out_KS columns): phase 2's operand;
                                             // filled with 0xFFFF'FFFF by the launch in front (ColsArgs::poison)
    const bf16_t* W2; int N2, N2pad, K2;     // down projection, K2 == N1pad
    int splitk;                              // K slices of the down projection = fp32 slabs
    float* ws; int ldws; int rows_ws;        // slabs [splitk][rows_ws][ldws]
    int* err;                                // set to 3 when a block gives up waiting (never a hang)
    int spin_ticks;                          // wall-clock budget of a wave's wait in 100 MHz ticks (s_memrealtime): on expiry -- or when
                                             // *err is already set by another wave / launch -- the wave gives up (code 3)
    long long* trace;                        // optional [N1pad / 32][8] wall-clock stamps per block (tools/mlp_trace.py); nullptr in production
};
int launch_mlp_fused(const MlpFusedArgs& a, hipStream_t st);

// ---- attention output projection without slabs + LayerNorm fold (decode_cols.hip) ------------------
struct ColsArgs {
    const bf16_t* xp;                // packed activations [MT][K/16][64][8]
    const bf16_t* Wp;                // packed weight [Npad/32][K/16][64][8] (the image the 32-column kernels read)
    const bf16_t* bias;              // [N] or nullptr
    int MT, N, K;                    // K multiple of 32
    int cpb;                         // output columns per block (<= 32): cols_pick_cpb(N, K)
    bf16_t* h_xp; int out_KS;        // residual stream in fragment order, N = 16 * out_KS columns: h = bf(h + bf(x W^T + b)), in place
    void* poison; unsigned poison_bytes; // optional: a buffer the blocks fill with 0xFF bytes (the fused MLP launch behind this one
                                     // recognises unwritten activations by that pattern); a multiple of 16 bytes
    void* poison2; unsigned poison2_bytes;   // optional: the NEXT layer's LayerNorm output buffer (the polled buffer of rowln_cattn_kernel), armed here with
                                     // write-through stores when the attention launch's grid is too small to carry the pattern (batches below 10 rows)
};
int cols_pick_cpb(int N, int K);
int launch_gemm_cols(const ColsArgs& a, hipStream_t st);        // 0 = ok, -1 = unsupported shape
int init_cols_kernels();
void launch_fold_prepare(const bf16_t* Wp, const bf16_t* gamma, const bf16_t* beta, const bf16_t* bias, bf16_t* Wf, float* c1, float* c2,
                         int N, int Npad, int K, hipStream_t st);

// ---- row kernels ---------------------------------------------------------------------------------
void launch_layernorm_rows(const bf16_t* x, int ldx, const bf16_t* g, const bf16_t* b, bf16_t* y, int ldy,
                           int M, int D, float eps, hipStream_t st);
// y_packed (xp layout) variant for the decode path
void launch_layernorm_rows_packed(const bf16_t* x, int ldx, const bf16_t* g, const bf16_t* b, bf16_t* yp,
                                  int M, int D, float eps, hipStream_t st);

// decode row update: v = sum_s ws[s][m][:] + bias ; h = bf(h + bf(v)) (in place) ; xp = LN(h)
struct RowUpdateArgs {
    const float* ws; int splitk; int ldws; int rows_ws;   // partials [splitk][rows_ws][ldws] (or nullptr)
    const bf16_t* bias;
    bf16_t* h; int ldh;                                    // residual stream [M][D] (in/out); ldh == 0: fragment order (xp layout)
    const bf16_t* wte; const bf16_t* wpe;                  // embedding mode (ws == nullptr)
    const int32_t* tokens; const int32_t* positions;       // [M]
    const bf16_t* g; const bf16_t* b; float eps;           // LayerNorm applied to the updated row
    bf16_t* xp_out;                                        // packed LN output
    int M, D;
};
void launch_row_update_ln(const RowUpdateArgs& a, hipStream_t st);
// the row update and the split-K projection that consumes its LayerNorm output (c_attn) as ONE launch (rowops.hip, rowln_cattn_kernel):
// late arguments of the launch; the leading scalars (slabs, bias, residual stream, weights, shapes) travel as kernel parameters
struct RowCattnArgs {
    const bf16_t* g; const bf16_t* b; float eps; int D;
    const bf16_t* wte; const bf16_t* wpe; const int32_t* tokens; const int32_t* positions;      // embedding mode (slabs == nullptr)
    bf16_t* xp_out;                                          // LayerNorm output in fragment order = the projection's operand; pre-filled with
                                                             // the 0xFFFF'FFFF pattern by an earlier launch of the step
    float* ws_out; int ldws_out;                             // the projection's fp32 slabs [splitk][32][ldws]
    int* err; int spin_ticks;                                // give-up code 4 after spin_ticks x 10 ns of waiting (never a hang)
    int delay;                                               // GEMM blocks: 10-ns ticks between block start and the first poll
    int first_round;                                         //   ... for the blocks below this index (the ones resident when the launch starts)
    int ldh;                                                 // WIDE row role: residual row stride (0 = fragment order)
    long long* dbg; int layer;                               // optional [8]: what the first wave to give up saw (reported with the failure)
};
// 0 = launched; -1 = outside the kernel's scope (the caller runs the two launches).  sk.xp must be ru.xp_out.
int launch_rowln_cattn(const RowUpdateArgs& ru, const SkinnyArgs& sk, int* err, int spin_ticks, hipStream_t st, int delay = 390, long long* dbg = nullptr, int layer = 0,
                       int num_cus = 256);
bool rowln_cattn_fits(int D, int Npad, int K, int splitk, int splitk_ru, int num_cus);
int rowln_cattn_blocks_per_cu(bool wide);       // hipOccupancyMaxActiveBlocksPerMultiprocessor of the form on the current device
bool rowln_cattn_resident(bool wide);           // >= 2 (narrow) / 3 (wide) of them: what the launch's waits assume (logs when not)

// ---- embeddings ---------------------------------------------------------------------------------
void launch_im2col(const bf16_t* img, bf16_t* out, int B, int img_size, int patch, int Kpad, hipStream_t st);
void launch_vit_embed_lnpre(const bf16_t* patch_out, int ldp, const bf16_t* cls, const bf16_t* pos,
                            const bf16_t* g, const bf16_t* b, bf16_t* x, int B, int NP, int Dv, float eps,
                            hipStream_t st);
void launch_dec_embed(const bf16_t* emb, const bf16_t* wpe, bf16_t* h, int B, int S0, int D, hipStream_t st);
// per_seq > 0: row r goes to out + (r / per_seq) * seq_stride + (r % per_seq) * D (token rows written behind the visual rows of a prefill buffer)
void launch_gather_rows(const bf16_t* table, const int64_t* ids, bf16_t* out, int n, int D, hipStream_t st, int per_seq = 0, size_t seq_stride = 0);
void launch_gather_last_rows(const bf16_t* h, bf16_t* out, int B, int S0, int D, hipStream_t st);
void launch_gather_tail_rows(const bf16_t* h, bf16_t* out, int B, int S0, int n_keep, int D, hipStream_t st);

// ---- adapter norm -------------------------------------------------------------------------------
// y_batch_stride (elements, 0 = planes back to back): the planes may be written straight into a [B][S0][D] prefill buffer
void launch_plane_layernorm(const bf16_t* x, const bf16_t* g, const bf16_t* b, bf16_t* y, int B, int QD,
                            float eps, hipStream_t st, size_t y_batch_stride = 0);
void launch_token_batchnorm(const bf16_t* x, const bf16_t* w, const bf16_t* b, const bf16_t* rm,
                            const bf16_t* rv, bf16_t* y, int B, int Q, int D, float eps, hipStream_t st, size_t y_batch_stride = 0);

// ---- attention ----------------------------------------------------------------------------------
struct AttnPrefillArgs {
    const bf16_t* q; const bf16_t* k; const bf16_t* v;   // token-major buffers
    int q_row_stride, kv_row_stride;                      // elements between consecutive tokens
    int q_head_stride, kv_head_stride;                    // elements between heads (kv: 0 for MQA)
    bf16_t* o; int o_row_stride;                          // [B*S][H*D]
    int B, S, H, head_dim, kv_group;                      // kv head = head / kv_group
    int causal; float scale;
    int window = 0;                                       // causal only: query q sees keys q - window < k <= q (StarCoder2); 0 = all
    int last_rows = 0;                                    // > 0: only the query tiles holding the last `last_rows` rows of every sequence are launched
    int q_tile0 = 0;                                      //   (set by the launcher: first query tile of the grid)
};
void launch_attn_prefill(const AttnPrefillArgs& a, hipStream_t st);

// paged KV cache geometry (one layer):  page = 64 tokens, K|V in MFMA-fragment order, 32 KiB
#define SV_PAGE_TOKENS 64
struct KvLayout { int head_dim; int page_bytes; };
__host__ __device__ inline int kv_page_bytes(int head_dim) { return SV_PAGE_TOKENS * head_dim * 2 * 2; }

// scatter prefill K/V rows (from the c_attn output) into the paged cache
void launch_kv_write_prefill(const bf16_t* qkv, int row_stride, int k_off, int v_off, char* pool_layer,
                             const int32_t* block_table, int max_pages, int B, int S0, int head_dim,
                             hipStream_t st);

struct AttnDecodeArgs {
    const float* ws; int splitk; int ldws; int rows_ws;   // c_attn output as fp32 split-K slabs [splitk][rows][ldws], summed here in slab order
    const bf16_t* bias;                                    //   + c_attn bias [H*D + 2*n_kv*D]
    char* pool_layer;                                      // this layer's page pool
    const int32_t* block_table; int max_pages;
    const int32_t* positions;                              // [B] index of the new token (= tokens already cached)
    bf16_t* out_xp; int out_KS;                            // packed [MT][H*D/16][64][8]
    int B, H, head_dim; float scale;
    float* part;                                           // [B][splits][32 + 16*D] partial (m, l, O)
    unsigned* counters;                                    // [B * n_kv] arrival tickets, zero between launches
    int max_splits;                                        // cap on active context splits (#CUs / (B*n_kv), <= 16)
    int n_kv;                                              // key/value heads (1 = MQA); grid.x = B * n_kv
    size_t kv_head_stride;                                 // bytes between the page pools of consecutive KV heads
    const float* rope_cos; const float* rope_sin;          // [positions][D/2] rotary tables (nullptr: no RoPE)
    int window;                                            // sliding window: keys pos - window < j <= pos (0 = all)
    int groups_per_block;                                  // 32-key groups a block takes before another context split joins (0 = 4)
    long long* trace;                                      // optional [B * n_kv * max_splits][16] wall-clock stamps (tools/attn_trace.py); nullptr in production
    int merge_all;                                         // 1: every wave takes part in the block's LDS merge (rounds 1-5); 0: only the waves that had a key group
    void* poison2; unsigned poison2_bytes;                 // a second buffer filled the same way, by the threads behind those of the first (the next layer's
                                                           // LayerNorm output buffer: rowln_cattn_kernel)
    void* poison; unsigned poison_bytes;                   // optional: a buffer the launch fills with 0xFF bytes, 16 per thread, before anything else (the
                                                           // "not written yet" pattern of the fused MLP launch later in the layer): this kernel is bound by
                                                           // latencies, a store per thread costs it nothing -- in gemm_cols_resid_kernel it cost 0.5 us
};
// in-place rotary embedding of the q and k heads of a prefill c_attn output (rotate_half convention)
void launch_rope_prefill(bf16_t* qkv, int row_stride, int rows, int S0, int n_heads, int head_dim,
                         const float* cos_t, const float* sin_t, hipStream_t st);
void launch_attn_decode(const AttnDecodeArgs& a, hipStream_t st);
// XCC_ID of every block of a 1-D launch of `blocks` 8-wave blocks (heavy = 1: with the decode attention's LDS footprint and 10 us of residence)
int launch_xcc_probe(int32_t* out, int blocks, int heavy, hipStream_t st);
int launch_occupy(int blocks, int lds_bytes, int ticks, hipStream_t st);     // test tenant: pins LDS of `blocks` CUs for ticks x 10 ns
size_t attn_decode_part_floats(int head_dim);            // floats of `part` per sequence
int init_attention_kernels();   // returns a hipError_t value (0 = ok)

// ---- sampling -----------------------------------------------------------------------------------
// greedy selection: per-slice winners (pval/pidx: [B][8]) then a merge (standalone, or inside finish_step)
// seen / penalty: HF RepetitionPenaltyLogitsProcessor (bitmap of generated ids per row, [B][seen_words]); nullptr = off
void launch_argmax_partial(const float* logits, int ld, int V, float* pval, int32_t* pidx, int B, const uint32_t* seen,
                           int seen_words, float penalty, hipStream_t st);
void launch_argmax(const float* logits, int ld, int V, int32_t* out, float* pval, int32_t* pidx, int B, hipStream_t st);
struct SampleArgs {
    const float* logits; int ld; int V; int B;
    float temperature, top_p; int top_k;                             // top_k <= 0: off
    uint64_t seed; const int32_t* step;                              // device step counter
    int32_t* out; float* scratch;                                    // scratch >= B*4 floats
    const uint32_t* seen; int seen_words; float penalty;             // repetition penalty (nullptr = off)
};
void launch_sample_top_p(const SampleArgs& a, hipStream_t st);

struct FinishArgs {
    const int32_t* next;          // [B] raw sampled ids (sampling path)
    const float* pval; const int32_t* pidx;   // greedy path: [B][8] slice winners merged here (or nullptr)
    unsigned long long* amax;     // greedy path, selection folded into the lm_head launch: keys [B * SV_AMAX_STRIDE] (SkinnyArgs::amax),
                                  // decoded and re-armed (zeroed) here; nullptr: off
    int32_t* cur_tok;             // [B] token fed to the next step
    int32_t* unfinished;          // [B]
    int32_t* positions;           // [B] (+1 each step)
    int32_t* out_tokens; int ld_out;   // [B][max_new]
    int32_t* step;                // device scalar: number of tokens emitted so far
    int32_t* done;                // device scalar: 1 once generation has ended
    int32_t* n_emitted;           // device scalar: final column count
    const int32_t* stop_ids; int n_stop;
    int eos, pad, B, max_new;
    uint32_t* seen; int seen_words;   // repetition-penalty bitmap to update with the emitted ids (nullptr = off)
    int V; int32_t* bad;          // vocabulary size; device flag raised when a row has no valid token (all logits NaN): the id
                                  // fed to the next step is then 0, never an out-of-range row of the embedding table
};
void launch_finish_step(const FinishArgs& a, hipStream_t st);
#ifdef __HIPCC__
// the per-row part of the bookkeeping (row b takes token `nxt` at step t): returns whether the row is still generating
__device__ __forceinline__ int finish_step_row(const FinishArgs& p, int b, int t, int nxt) {
    const int unf = p.unfinished[b];
    if (unf && (unsigned)nxt >= (unsigned)p.V) {     // no finite logit in this row: never index the embedding table with it
        nxt = 0;
        if (p.bad) atomicCAS(p.bad, 0, 1);          // 0 -> 1 only: a code already there (3 = the fused MLP launch gave up) survives
    }
    const int tok = unf ? nxt : p.pad;
    p.out_tokens[(size_t)b * p.ld_out + t] = tok;
    p.cur_tok[b] = tok;
    if (p.seen && tok >= 0) atomicOr(p.seen + (size_t)b * p.seen_words + (tok >> 5), 1u << (tok & 31));
    const int still = unf && tok != p.eos;
    p.unfinished[b] = still;
    p.positions[b] += 1;
    return still;
}
// the call's part (one thread, after every row's): stop sequence on row 0, step counter, end of generation
__device__ __forceinline__ void finish_step_call(const FinishArgs& p, int t, int any_unf) {
    bool fired = false;
    if (p.n_stop > 0 && t + 1 >= p.n_stop) {
        fired = true;
        for (int i = 0; i < p.n_stop; ++i)
            if (p.out_tokens[t + 1 - p.n_stop + i] != p.stop_ids[i]) { fired = false; break; }
    }
    *p.step = t + 1;
    if (fired || !any_unf || t + 1 >= p.max_new) {
        *p.done = 1;
        *p.n_emitted = t + 1;
    }
}
#endif

// ---- continuous batching: one request per row ("slot"), everything per row (sampling.hip) ---------------------------
#define SV_CB_MAXSTOP 16
struct CbSlot {                  // device-resident, one per slot
    int32_t live;                // 1 = generating
    int32_t step;                // tokens emitted so far
    int32_t budget;              // new-token budget of the request
    int32_t do_sample; float temperature, top_p; int32_t top_k;
    int32_t eos, pad, min_new; float penalty;
    int32_t n_stop; int32_t stop[SV_CB_MAXSTOP];
    int32_t pad_[1];
    uint64_t seed;
};
struct CbStepArgs {
    const float* logits; int ld; int V;
    CbSlot* slots; const int32_t* slot_map;      // block b works for slot slot_map[b] (nullptr: b) on logits row b
    int32_t* cur_tok; int32_t* positions; int32_t* out_tokens; int ld_out;
    uint32_t* seen; int seen_words;
    int32_t* n_live; int32_t* events;           // device counters: live slots, finished-slot events
    int32_t* bad;                               // raised when a slot has no valid token (all logits NaN); the id becomes 0
};
void launch_cb_step(const CbStepArgs& a, int nblocks, hipStream_t st);

// ---- image pre-processing (preprocess.hip): uint8 HWC (3 or 4 channels) -> float32 [3][S][S], returns a hipError_t value
int preprocess_image(const uint8_t* dev_pixels, int width, int height, int channels, int out_size, int recipe,
                     const float* mean3, const float* std3, float* dev_out, hipStream_t st);
// batched + stateless: n images (any sizes) -> [n][3][S][S]; the caller owns the workspace (preprocess_workspace_bytes);
// nothing is copied from the host or synchronised inside.  Returns a hipError_t value, or -1 if the workspace is too small.
size_t preprocess_workspace_bytes(const int32_t* widths, const int32_t* heights, int n, int out_size, int recipe);
int preprocess_images(const uint8_t* const* dev_pixels, const int32_t* widths, const int32_t* heights, const int32_t* channels,
                      int n, int out_size, int recipe, const float* mean3, const float* std3, float* dev_out, void* workspace,
                      size_t workspace_bytes, hipStream_t st);

}  // namespace sv
