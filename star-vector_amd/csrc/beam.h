// Beam search on device (HF GenerationMixin._beam_search, do_sample=False, as the reference drives it with its
// default num_beams=2: starvector_base.py:234,238,293).  The scorer is a standalone component (like HF's
// BeamSearchScorer): it consumes a [B*num_beams][V] fp32 logits matrix per step and keeps every piece of search
// state in device memory, so the decode loop runs as a replayed hipGraph with no per-step host round trip.
// Sequences are not stored per beam: each step appends (parent beam, token) columns and the host backtracks once
// at the end.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>

namespace sv {

#define BM_SPLIT 8       // a logits row is scanned by this many blocks (one CU streams only ~25 GB/s)
#define BM_MAXNB 8       // num_beams <= 8
#define BM_MAXK 16       // candidates kept per request = 2 * num_beams (one EOS id)
#define BM_MAXSTOP 16

struct BeamConfig {
    int B = 0, nb = 0, V = 0, max_new = 0;
    int eos = -1, pad = -1;
    int early = 1;                  // early_stopping: 0 False, 1 True, 2 "never"
    float length_penalty = 1.f;
    float penalty = 1.f;            // repetition penalty (on the log-probs, as HF's beam search applies processors)
    int n_stop = 0;
    int32_t stop[BM_MAXSTOP] = {0};
    int min_new = 0;                // MinLengthLogitsProcessor on the log-probs: EOS at -inf while fewer tokens were generated
    // beam-sample (do_sample with num_beams > 1): warpers on the log-probs, K draws without replacement
    int do_sample = 0;
    float temperature = 1.f, top_p = 1.f;
    int top_k = 0;
    uint64_t seed = 0;
};

// everything the kernels need, passed by value
struct BeamDev {
    int B, nb, K, V, max_new, eos, early, n_stop, min_new;
    float length_penalty, penalty;
    const float* logits; int ld; int logit_div;       // logits row of beam row r = r / logit_div
    float* run_score;                                  // [R] accumulated log-prob of each running beam
    int32_t* cur_tok;                                  // [R] token each running beam feeds to the next step
    int32_t* parent;                                   // [R] flat row each running beam descends from (this step)
    int32_t* hist_parent; int32_t* hist_tok;           // [max_new][R] back-pointers (beam index in request) / tokens
    float* fin_score; int32_t* fin_done; int32_t* fin_step; int32_t* fin_parent; int32_t* fin_tok;   // [B][nb]
    int32_t* can_improve;                              // [B]
    float* stats;                                      // [R][BM_SPLIT][2] slice (max, sum exp)
    float* cand_val; float* cand_key; int32_t* cand_idx;   // [R][BM_SPLIT][K] slice winners (key = ranking value)
    float* top_val; int32_t* top_beam; int32_t* top_tok;   // [B][K] best first
    const float* lenpow;                               // [max_new + 1]  t ** length_penalty
    const int32_t* stop_ids;
    uint32_t* seen; int seen_words;                    // [R][seen_words] ids each running beam has generated (or null)
    int32_t* step; int32_t* done;                      // device scalars
    int32_t* positions;                                // [R] engine position counters (+1 per step) or null
    int do_sample; float inv_temp, top_p; int top_k; uint64_t seed;
    float* warp;                                       // [R][8] per-row warper thresholds (WarpStats)
};

struct BeamScorer {
    BeamConfig c;
    BeamDev d;
    int R = 0, K = 0;
    bool own_scalars = false;
    std::vector<void*> allocs;
    float* h_lenpow = nullptr;

    // ext_*: engine-owned buffers to drive (nullptr: the scorer allocates its own)
    int init(const BeamConfig& cfg, int32_t* ext_cur_tok, int32_t* ext_positions, int32_t* ext_step, int32_t* ext_done);
    int reset(hipStream_t st);                                      // initial search state
    void enqueue_step(const float* logits, int ld, int logit_div, hipStream_t st);
    // best hypothesis per request: tokens [B][max_new] (filled with pad-or-eos), common length L, scores [B]
    int finalize(hipStream_t st, std::vector<int64_t>& tokens, int& L, std::vector<float>& scores);
    void destroy();
    bool matches(const BeamConfig& o) const {
        return R > 0 && c.B == o.B && c.nb == o.nb && c.V == o.V && c.max_new == o.max_new;
    }
};

// KV-cache side of a beam step (engine only).  Full pages are immutable and shared through the block table; only
// the partially filled tail page of a beam whose parent changed is copied (the beams of a request permute among
// themselves: one thread moves the same pieces of all of them, loads before stores -- beam.hip::beam_tail_copy_kernel).
struct BeamKvArgs {
    int32_t* block_table; int max_pages;      // [rows][max_pages]
    int need;                                  // pages owned by each row: own page (r, i) = r * need + i
    const int32_t* parent;                     // [R] flat parent rows
    const int32_t* step; const int32_t* done;  // device scalars
    int S0, L_fixed;                           // cached length L = L_fixed >= 0 ? L_fixed : S0 + *step - 1
    int B, nb;
    char* kv_pool; size_t layer_stride, kv_head_stride; int n_layer, n_kv, page_bytes;
};
void launch_beam_table_reorder(const BeamKvArgs& a, hipStream_t st);
void launch_beam_tail_copy(const BeamKvArgs& a, hipStream_t st);

}  // namespace sv
