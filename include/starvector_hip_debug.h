/*
 * libstarvector_hip.so - TEST AND MEASUREMENT surface of the MI355X StarVector engine (not needed to run the path).
 *
 * Compiled into the same shared object as the product ABI (include/starvector_hip.h, included below), kept apart so that a
 * reference-side binding sees only what it has to bind:
 *   sv_op_*                 single operators, one per row of SURVEY.md section 8a: the parity tests put each HIP kernel next to the
 *                           oracle through them (tests/test_gpu_ops.py, tests/test_gpu_fp8.py)
 *   sv_debug_*_plan,
 *   sv_debug_resample_coeffs  host-side decisions of the library (dispatch, split-K, context splits, Pillow's tap tables), callable
 *                           WITHOUT a GPU: the CPU suite pins them
 *   sv_debug_kv_load / sv_debug_attn_decode   the decode attention alone over an engine's real paged pool (tests/test_gpu_long_context.py)
 *   sv_debug_*_trace, sv_debug_xcc_map        in-kernel wall-clock traces and block placement (tools/*_trace.py)
 *   sv_bench_*, sv_profile_decode_step        micro-benchmarks and the HIP-event profile bench.py's roofline leg reads
 *   sv_debug_set_exp / sv_debug_set_col_tiles / sv_debug_set_gemm_form / sv_debug_set_linear_seq_rows
 *                           A/B switches: they CHANGE which kernels a live engine (set_exp) or the process (the other two) launches
 *                           from then on -- every form computes the same bits (tests), but a product deployment has no reason to
 *                           call them.  The environment variable SV_EXP sets the same mask at sv_create.
 * Conventions as in starvector_hip.h.
 */
#ifndef STARVECTOR_HIP_DEBUG_H
#define STARVECTOR_HIP_DEBUG_H

#include "starvector_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Host-side decisions of the library, callable WITHOUT a GPU (the CPU test suite pins them):
 *   sv_debug_resample_coeffs  the fixed-point table sv_preprocess_image feeds its two passes = Pillow's
 *                             precompute_coeffs + normalize_coeffs_8bpc (libImaging/Resample.c) for BICUBIC, box (0, in_size):
 *                             bounds [out_size][2] = (first input index, tap count), taps [out_size][cap]; returns ksize
 *                             (> 0) or SV_EINVAL when cap < ksize
 *   sv_debug_gemm_plan        what the big-M GEMM dispatch does with an M x N x K problem: out5 = {peel the row remainder,
 *                             remainder rows, remainder as a 128^2 tile row (else one wave per 32x32 tile), main part on the
 *                             256^2 kernel, modelled time in us}; every choice computes the same bits */
/*   sv_debug_skinny_plan      how a decode GEMM (rows <= 64, W [N][K], split-K `splitk`, bf16 or fp8 weights) is launched:
 *                             out2 = {waves per block = how K is cut inside a block, i.e. the order in which a row's partial
 *                             sums are added -- a function of the GEMM only, never of `rows`; 1 when a block carries two
 *                             32-row tiles (33..64 rows: the weights are streamed once)} */
int  sv_debug_resample_coeffs(int32_t in_size, int32_t out_size, int32_t* bounds, int32_t* taps, int32_t cap);
int  sv_debug_skinny_plan(int32_t rows, int32_t N, int32_t K, int32_t splitk, int32_t fp8, int32_t* out2);
/*   sv_debug_set_exp          the experiment bit mask of a live engine (what the environment variable SV_EXP sets at sv_create):
 *                             in-process A/B runs of the round's experiments (tools/ab_exp.py) */
int  sv_debug_set_exp(sv_engine* e, int32_t mask);
int  sv_debug_gemm_plan(int32_t M, int32_t N, int32_t K, int32_t act, int32_t* out5);
/*   sv_debug_decode_plan      what sv_create decides for a decoder Linear W [N][K] when the engine decodes `rows` (<= 64) rows at a
 *                             time on a GPU with `num_cus` CUs: out2 = {split-K factor (1 when whole_k: c_fc / lm_head keep the whole
 *                             K for their epilogue), column tiles per block of the two-row-tile kernel (1 for rows <= 32)} -- plain
 *                             host arithmetic (engine_core.hip: pick_splitk / pick_decode_plan), pinned by the CPU tests */
int  sv_debug_decode_plan(int32_t rows, int32_t N, int32_t K, int32_t fp8, int32_t whole_k, int32_t num_cus, int32_t* out2);
/*   sv_debug_attn_plan        the decode attention's context-split constants of an engine: out2 = {the most blocks a sequence's context
 *                             is split over (so that rows x KV heads x splits covers the CUs, <= 8), 32-key groups a block takes
 *                             before another split joins (8 where rows x KV heads alone cover the CUs, else 4)} -- functions of the
 *                             engine's max_batch / KV heads only, never of the call's batch (a row's bits do not depend on its batch) */
int  sv_debug_attn_plan(int32_t max_batch, int32_t n_kv_head, int32_t num_cus, int32_t* out2);
/*   sv_debug_rowln_plan       whether (and how) an engine that owns its GPU runs a decode layer's row update and c_attn projection as ONE
 *                             launch (rowops.hip rowln_cattn_kernel): hidden size D, the projection W [N][D] with
 *                             split-K `splitk`, `splitk_ru` slabs summed by the row update in front, a GPU with `num_cus` CUs;
 *                             out3 = {1 if the shapes fit, k-steps a GEMM wave holds in registers (4: StarVector-1B, 9: StarVector-8B), blocks of
 *                             the launch = 32 row blocks + (N / 32) * splitk} -- narrow rows (D <= 2048) need every block resident at once
 *                             (two per CU), wide rows rely on the row blocks being dispatched first.  Host arithmetic, pinned by the CPU tests */
int  sv_debug_rowln_plan(int32_t D, int32_t N, int32_t splitk, int32_t splitk_ru, int32_t num_cus, int32_t* out3);
/*   sv_debug_rowln_occupancy  blocks of that launch a CU of the current device holds at once (hipOccupancyMaxActiveBlocksPerMultiprocessor of
 *                             rowln_cattn_kernel<4, false> / <9, true> for wide != 0): sv_create turns the fused launch off below 2 / 3;
 *                             needs a GPU */
int  sv_debug_rowln_occupancy(int32_t wide, int32_t* blocks_per_cu);
/*   sv_debug_step_plan        what the LAST sv_generate call's decode step actually was, from the engine itself (bench.py's `launches_per_step`
 *                             and roofline captions; ADVICE r05: not re-derived from the configuration): out4 = {kernel nodes of the captured
 *                             decode-step graph (0: no graph, plain launches), 1 if the layers' row update + c_attn ran as one launch,
 *                             1 if the greedy selection rode in the lm_head launch, 1 if c_fc + down-projection ran as one launch} */
int  sv_debug_step_plan(sv_engine* e, int32_t* out4);
/*   sv_debug_set_col_tiles    column tiles per block (1..3; 0 = the launcher's own choice) the OP-LEVEL decode GEMM entry points
 *                             (sv_op_linear_skinny*, 33..64 rows) launch with from now on, process-wide: lets the parity tests put
 *                             every variant of the two-row-tile kernel next to the one-tile kernels (all bit-identical).  An engine's
 *                             decode loop is not affected (it carries its own plan per Linear). */
int  sv_debug_set_col_tiles(int32_t col_tiles);
/*   sv_debug_set_skinny_form  which kernel the 33..64-row decode GEMMs take from now on, process-wide (all forms are bit-identical; the choice is speed):
 *                             0 = gemm_skinny_mt2_kernel (both operands of the stream in registers, rounds 2-5), 1 = gemm_skinny_mt2x_kernel
 *                             (activations through a wave-private LDS ring, weights by hand-counted register loads; chunk depth by block
 *                             count: the default), 2 / 3 = that kernel's two-blocks-per-CU / one-block-per-CU form wherever the shape allows.
 *                             SV_EXP bits 131072 / 262144 / 524288 select 0 / 2 / 3 at sv_create and in sv_debug_set_exp */
int  sv_debug_set_skinny_form(int32_t form);
/*   sv_debug_set_gemm_form    process-wide: every big-M GEMM launch takes ONE form -- 0 = 128x128 tiles, 1 = 256x256 tiles (rows not peeled),
     2 = 256x256 tiles + the row remainder over a multiple of 256 through the one-wave-per-tile tail kernel; -1 = the tuned choice (default).
     The forms give the same bits; the tests compare them through this switch. */
int  sv_debug_set_gemm_form(int32_t form);
/*   sv_debug_set_linear_seq_rows  process-wide: the sequence structure sv_op_linear / sv_bench_linear hand to the big-M dispatch from now on --
     S > 0: the M rows are M / S sequences of S rows.  Where S leaves 1..3 rows over its 256-row tiles (a 259-row prompt, a 257-token image:
     starvector_base.py:131-170, clip_model.py:140-155 produce exactly those row counts) AND the dispatch's cost model peels the remainder of a
     32-sequence batch for this projection, the tile kernels cover the full tiles of EVERY sequence and gemm_tailk_kernel (8 waves per 32 x 32
     tile, K split over the waves) the rows left over -- what the engine's prompt pass and vision tower launch; S < 0: the M rows are the LAST
     rows of sequences of |S| rows (the pruned last prompt layer) and take the kernel they take inside the full problem; 0 = none (default). */
int  sv_debug_set_linear_seq_rows(int32_t seq_rows);
/*   sv_debug_gemm_seq_form    1 / 0: does the projection (N, K, act) take that per-sequence form for sequences of S rows?  Host arithmetic only
     (callable without a GPU); a function of S and the projection alone, never of the batch: a sequence's tokens do not depend on its neighbours. */
int  sv_debug_gemm_seq_form(int32_t S, int32_t N, int32_t K, int32_t act);
/* The decode attention (gpt_bigcode/modeling_gpt_bigcode.py:151-285, llm/starcoder2.py:22-27 sliding window) on
 * its own, over the engine's real paged KV pool, block table and context-split plan.  Test surface: the caller chooses q / K / V.
 *   sv_debug_kv_load     dev_kv bf16 [B][S][2*n_kv*head_dim] (k heads | v heads, K as cached = after RoPE) -> pages of `layer`;
 *                        positions[b] = S for every row, or dev_lens[b] (int32 [B], <= S; NULL = S): a ragged batch
 *   sv_debug_attn_decode dev_qkv_f32 fp32 [B][n_head*head_dim + 2*n_kv*head_dim] (the new token's c_attn output, before RoPE) ->
 *                        dev_out bf16 [B][n_head*head_dim]; appends the new K/V row at positions[b]; advance != 0: positions += 1 */
/*   sv_debug_mlp_trace   (engine created with SV_MLP_TRACE=1) 100 MHz wall-clock stamps of the fused MLP launch (SV_EXP bit 128) of the
 *                        middle layer of the last decode step: host_out [blocks][8] = {start, c_fc loop done, tile published, slice
 *                        complete, end, XCC id, 0, 0}; returns the block count or a negative error code */
/*   sv_debug_attn_trace  (engine created with SV_ATTN_TRACE=1) the same for the decode attention launch of the middle layer: host_out
 *                        [rows * kv heads * context splits][16] = {start, first KV group requested, q in LDS, key groups processed,
 *                        partial stored + drained, ticket drawn, end (0 unless the merging block), active splits, key groups, 0...} */
int  sv_debug_attn_trace(sv_engine* e, int64_t* host_out, int32_t capacity_rows);
int  sv_debug_mlp_trace(sv_engine* e, int64_t* host_out, int32_t capacity_blocks);
/*   sv_debug_xcc_map     the XCD (XCC_ID) each block of a 1-D launch of `blocks` 8-wave blocks ran on, into host_out[blocks]; heavy = 1 gives
     the blocks the decode attention's LDS footprint and 10 us of residence (a grid above the CU count then runs in rounds, like a 64-row
     attention launch).  Evidence for the XCD-aware block -> tile mappings (block L runs on XCD (L + c) % 8): DESIGN.md sections 3c / 3f. */
int  sv_debug_xcc_map(sv_engine* e, int32_t blocks, int32_t heavy, int32_t* host_out);
/*   sv_debug_occupy_cus  a foreign tenant for the safety test of exclusive_device (tests/test_gpu_safety.py): `blocks` 4-wave blocks on a stream
     of their own, each pinning `lds_bytes` of a CU's LDS for `ms` milliseconds; returns at once.  With 144 KiB per block no block of the
     fused MLP / fused row-update launch fits beside one: those CUs are taken the way another process's kernels would take them, and a decode
     call made meanwhile has to END WITH AN ERROR (give-up code 3 / 4), never hang and never return tokens. */
int  sv_debug_occupy_cus(sv_engine* e, int32_t blocks, int32_t lds_bytes, int32_t ms);
/*   sv_debug_gemm_trace  the 256x256 big-M GEMM kernel (prefill / ViT) on random operands of the given shape, one launch with wall-clock
     stamps (form = 1): host_out [blocks * 2][8] = {start, K-tile 0 staged, K loop done, epilogue stored, tile m, tile n, wave, 0}
     (tools/gemm_trace.py).  Needs no engine.  Returns the number of blocks. */
int  sv_debug_gemm_trace(int32_t M, int32_t N, int32_t K, int32_t act, int32_t form, int64_t* host_out, int32_t capacity_blocks);
int  sv_debug_kv_load(sv_engine* e, int32_t layer, const void* dev_kv, int32_t B, int32_t S, const int32_t* dev_lens,
                      sv_stream stream);
int  sv_debug_attn_decode(sv_engine* e, int32_t layer, const float* dev_qkv_f32, int32_t B, void* dev_out, int32_t advance,
                          sv_stream stream);

/* HIP-event timing of one decode step by kernel class, on the cache state left by the last
 * sv_generate / sv_prefill (bench.py roofline leg).  out: 10 doubles, [2k] = ms per step in class k
 * (event deltas minus the empty event-pair time), [2k+1] = launches per step; k = 0 skinny
 * weight-streaming GEMM, 1 paged decode attention, 2 residual+LayerNorm row update;
 * [6] = ms between two back-to-back events with no kernel; [7] = ms per step of the step's GEMM launches
 * enqueued back to back between ONE event pair (dispatch-to-dispatch); [8] = the same for every OTHER kernel
 * of the step (no GEMMs): step - [8] = what the GEMMs cost in situ; [9] reserved. */
int  sv_profile_decode_step(sv_engine* e, int32_t B, int32_t iters, double* out10, sv_stream stream);

/* Where the time to first token goes: one pass image -> encoder -> adapter -> prompt rows -> prompt pass -> lm_head -> first greedy
 * token with a HIP event in front of every launch (group), averaged over `iters` passes after one warm-up pass.
 * dev_image bf16 [B,3,S,S] (NULL for text2svg: no encoder / adapter), dev_ids int64 [B][P].  out24: [2k] = ms per pass in stage k,
 * [2k+1] = launches (groups) per pass; k = 0 encoder GEMMs, 1 encoder attention, 2 encoder row kernels, 3 adapter GEMMs, 4 adapter
 * norm + prompt-token gather, 5 decoder prompt-pass GEMMs (tile launches), 6 remainder-row launches of peeled GEMMs (all towers),
 * 7 prompt-pass attention (+ RoPE, KV scatter), 8 prompt-pass row kernels (positions, LayerNorms, last-row gather, ln_f), 9 lm_head,
 * 10 first-token selection; [22] = the event-pair overhead subtracted from every interval (ms); [23] = first event -> last event, raw.
 * Leaves the KV cache filled like sv_prefill(B, P or T + P). */
int  sv_profile_ttft(sv_engine* e, const void* dev_image, int32_t B, const int64_t* dev_ids, int32_t P, int32_t iters,
                     double* out24, sv_stream stream);

/* ---- single operators (parity tests; all device pointers, bf16 unless noted) ------------------ */
int  sv_op_layernorm(const void* x, const void* gamma, const void* beta, void* y, int32_t M, int32_t D,
                     float eps, sv_stream stream);
/* y[M,N] = act(x[M,K] . W[N,K]^T + bias) (+ residual); W/bias/residual reference layouts */
int  sv_op_linear(const void* x, const void* W, const void* bias, const void* residual, void* y,
                  int32_t M, int32_t N, int32_t K, int32_t act, int32_t out_f32, sv_stream stream);
/* the decode-path (M<=32 per tile, weight-streaming) implementation of the same contraction */
int  sv_op_linear_skinny(const void* x, const void* W, const void* bias, void* y_f32, int32_t M, int32_t N,
                         int32_t K, int32_t splitk, sv_stream stream);
/* the same contraction with the weight quantised to fp8 e4m3 (one scale per output row, as weight_dtype = SV_WEIGHT_FP8_E4M3
 * does at load): y = x . dequant(quant(W))^T + bias, fp32; scale_out [N] (device, optional) receives the row scales */
int  sv_op_linear_skinny_fp8(const void* x, const void* W, const void* bias, void* y_f32, float* scale_out, int32_t M,
                             int32_t N, int32_t K, int32_t splitk, sv_stream stream);
/* the decode GEMM's fused epilogues on their own (split-K 1): out_f32 == 0: y[M,N] = act(bf16(x W^T + bias)) as bf16 rows
 * (the c_fc form, N %% 8 == 0); out_f32 != 0: float32 rows of x W^T holding bf16-rounded values, bias ignored (the lm_head form) */
int  sv_op_linear_skinny_epi(const void* x, const void* W, const void* bias, void* y, int32_t M, int32_t N, int32_t K,
                             int32_t act, int32_t out_f32, sv_stream stream);
/* the lm_head form with the greedy selection folded into its epilogue (what a plain greedy sv_generate runs per decode step): y_f32 [M][N]
 * (optional, device) = bf16-rounded fp32 logits, host_idx [M] (HOST) = arg-max column per row -- lowest index on ties, NaN never wins,
 * 0x7fffffff when a row has no comparable score; M <= 32 */
int  sv_op_lm_head_argmax(const void* x, const void* W, void* y_f32, int32_t* host_idx, int32_t M, int32_t N, int32_t K, sv_stream stream);
/* the two kernels of the 6-launch decode layer (csrc/decode_cols.hip; gpt_bigcode/modeling_gpt_bigcode.py:694-755) as one op, M <= 32:
 *   h2 = bf16(h + bf16(x[M,Kp] . Wp[D,Kp]^T + bp))                      attention output projection, whole K per block, in place
 *   y  = act(bf16(LayerNorm(h2; gamma, beta, eps) . Wf[F,D]^T + bf))    c_fc on the raw h2, ln_2 folded into weights / epilogue */
int  sv_op_decode_proj_fold(const void* x, const void* Wp, const void* bp, const void* h, const void* gamma, const void* beta, float eps,
                            const void* Wf, const void* bf, void* h2_out, void* y_out, int32_t M, int32_t D, int32_t Kp, int32_t F,
                            int32_t act, sv_stream stream);
/* micro-benchmark of the big-M MFMA GEMM alone (pseudo-random operands): average microseconds per launch */
int  sv_bench_linear(int32_t M, int32_t N, int32_t K, int32_t act, int32_t residual, int32_t iters, double* avg_us,
                     sv_stream stream);
/* micro-benchmark of the decode GEMM kernel alone: average microseconds per launch over `iters` back-to-back launches (HIP
 * events); mode = 0 fp32 slabs (split-K `splitk`), 1 bias + GELU -> fragment order, 2 fp32 logits (modes 1, 2: splitk == 1) */
int  sv_bench_decode_linear(int32_t M, int32_t N, int32_t K, int32_t splitk, int32_t mode, int32_t iters, double* avg_us,
                            sv_stream stream);
/* f32 -> bf16 through the hardware convert used inside the kernels (rounding-mode check) */
int  sv_op_cvt_bf16_hw(const float* x, void* y, int64_t n, sv_stream stream);
/* q,k,v token-major [B,S,H*D] / [B,S,Hkv*D]; out [B,S,H*D] */
int  sv_op_attention(const void* q, const void* k, const void* v, void* out, int32_t B, int32_t S,
                     int32_t H, int32_t Hkv, int32_t head_dim, int32_t causal, float scale,
                     sv_stream stream);
int  sv_op_plane_layernorm(const void* x, const void* gamma, const void* beta, void* y, int32_t B,
                           int32_t QD, float eps, sv_stream stream);
int  sv_op_argmax(const float* logits, int32_t B, int32_t V, int32_t ld, int32_t* out, sv_stream stream);
int  sv_op_sample_top_p(const float* logits, int32_t B, int32_t V, int32_t ld, float temperature,
                        float top_p, uint64_t seed, int32_t step, int32_t* out, sv_stream stream);
/* temperature -> top-k (0 = off) -> top-p -> one multinomial draw per row */
int  sv_op_sample(const float* logits, int32_t B, int32_t V, int32_t ld, float temperature, int32_t top_k,
                  float top_p, uint64_t seed, int32_t step, int32_t* out, sv_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* STARVECTOR_HIP_DEBUG_H */
