// Beam search scorer + KV-cache reordering kernels (see beam.h).
//
// Per step (R = B * num_beams rows, K = 2 * num_beams):
//   beam_row_stats   grid (R, 8)   slice max / sum-exp of the raw logits           (log_softmax, fp32)
//   beam_row_warp    grid R        (beam-sample only) temperature / top-k / top-p thresholds of each row's log-probs
//   beam_row_topk    grid (R, 8)   slice -> LDS as penalty(log_softmax) + running score; K rounds of block argmax
//                                  (beam-sample: + Gumbel noise, so the K winners are K draws without replacement
//                                  from softmax(accumulated scores), in draw order -- Gumbel-top-k == sequential
//                                  sampling without replacement, which is what torch.multinomial does)
//   beam_merge       grid B        num_beams * 8 * K slice winners -> the request's K best (beam, token), best first
//   beam_update      1 block       HF's bookkeeping for every request: hits (EOS / budget / the reference's row-0 stop),
//                                  next running beams, finished slots with the length penalty, early-stop heuristic,
//                                  loop termination -> device `done` flag
//   beam_seen_gather grid B        repetition-penalty bitmaps follow their beams (only when the penalty is on)
#include "beam.h"
#include "kernels.h"
#include "warp.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

namespace sv {

static constexpr float BM_NEG = -1.0e9f;

__device__ __forceinline__ void bm_pair(float& v, int& i, float ov, int oi) {
    if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
}
__device__ __forceinline__ int bm_slice(int V) { return (((V + BM_SPLIT - 1) / BM_SPLIT) + 3) & ~3; }

__global__ __launch_bounds__(256) void beam_row_stats_kernel(BeamDev p) {
    __shared__ float red[4];
    if (*p.done) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row = blockIdx.x;
    const float* x = p.logits + (size_t)(row / p.logit_div) * p.ld;
    const int per = bm_slice(p.V);
    const int beg = blockIdx.y * per, end = min(beg + per, p.V);
    float m = -INFINITY;
    for (int i = beg + tid; i < end; i += 256) m = fmaxf(m, x[i]);
    m = wave_max(m);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.f;
    for (int i = beg + tid; i < end; i += 256) s += expf(x[i] - m);
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    if (tid == 0) {
        float* o = p.stats + ((size_t)row * BM_SPLIT + blockIdx.y) * 2;
        o[0] = m;                                   // -inf for an empty slice (then s = 0)
        o[1] = red[0] + red[1] + red[2] + red[3];
    }
}

// log_softmax row constants from the slice statistics: x -> (x - M) - logS
__device__ __forceinline__ void bm_row_lse(const BeamDev& p, int row, float& M, float& logS) {
    const float* st = p.stats + (size_t)row * BM_SPLIT * 2;
    M = -INFINITY;
#pragma unroll
    for (int s = 0; s < BM_SPLIT; ++s) M = fmaxf(M, st[2 * s]);
    float S = 0.f;
#pragma unroll
    for (int s = 0; s < BM_SPLIT; ++s) S += st[2 * s + 1] > 0.f ? st[2 * s + 1] * expf(st[2 * s] - M) : 0.f;
    logS = logf(S);
}
__device__ __forceinline__ float bm_logprob(const BeamDev& p, const float* x, const uint32_t* seen, float M, float logS, int id) {
    float lp = (x[id] - M) - logS;
    if (seen && ((seen[id >> 5] >> (id & 31)) & 1u)) lp = lp < 0.f ? lp * p.penalty : lp / p.penalty;
    // HF builds [RepetitionPenalty, MinLength] and, under beam search, runs them on the LOG-PROBS (GenerationMixin._beam_search)
    if (id == p.eos && *p.step < p.min_new) lp = -INFINITY;
    return lp;
}

// beam-sample: HF applies the warpers to the (penalised) log-probs of every running beam
template <bool SELECT>
__global__ __launch_bounds__(WP_THREADS) void beam_row_warp_kernel(BeamDev p) {
    __shared__ float red[WP_THREADS / 64];
    if (*p.done) return;
    const int row = blockIdx.x;
    const float* x = p.logits + (size_t)(row / p.logit_div) * p.ld;
    const uint32_t* seen = p.seen ? p.seen + (size_t)row * p.seen_words : nullptr;
    float M, logS;
    bm_row_lse(p, row, M, logS);
    // bm_logprob with everything that does not depend on the column hoisted (the step counter, the penalty, the temperature): the functor is
    // inlined into a 52-values-per-thread register sweep under a 128-VGPR cap (1024 threads)
    const bool hold_eos = *p.step < p.min_new;
    const int eos = p.eos;
    const float pen = p.penalty, inv_pen = 1.0f / p.penalty, inv_temp = p.inv_temp;
    auto sc = [&](int i) {
        float lp = (x[i] - M) - logS;
        if (seen && ((seen[i >> 5] >> (i & 31)) & 1u)) lp = lp < 0.f ? lp * pen : lp / pen;
        if (i == eos && hold_eos) lp = -INFINITY;
        return lp * inv_temp;
    };
    (void)inv_pen;
    float* o = p.warp + (size_t)row * 8;
    if constexpr (SELECT) {
        // TopK by selection (warp.h row_warp_stats_select) leaves the ~k candidates of the row in LDS -- and everything beam_row_topk_kernel would
        // rank lies among them (a filtered score is -inf there).  So the K = 2 num_beams draws are taken HERE, from ~50 values instead of 8 slices of
        // 6145: the same accumulated scores, the same Gumbel noise per (seed, step, row, column), the same order (key, then lowest column); the
        // row's slots of slice 0 get the winners, the other slices' slots are emptied, and o[5] = 1 tells beam_row_topk_kernel's blocks to leave
        // (24.9 -> 4 us per decode step at 64 rows).
        const int kk = p.top_k > 2 ? p.top_k : 2;
        WarpStats w;
        if (p.V <= 49 * WP_THREADS && p.top_k > 0 && kk <= 256 && kk < p.V && row_warp_stats_select<49>(sc, p.V, p.top_k, p.top_p, 2, red, w)) {
            WarpSelSmem& sm = warp_sel_smem();
            const int tid = threadIdx.x, K = p.K;
            __syncthreads();
            const int n = sm.n_cand;
            const float base = p.run_score[row];
            const uint64_t noise_seed = wp_splitmix64(p.seed ^ wp_splitmix64(((uint64_t)(uint32_t)*p.step << 32) | (uint32_t)row));
            float gk = -INFINITY, acc = -INFINITY;
            int id = 0x7fffffff;
            if (tid < n) {
                float s2 = wp_unkey(sm.key[tid]);
                id = sm.idx[tid];
                if (!wp_keep(w, s2)) s2 = -INFINITY;
                acc = s2 + base;
                const uint64_t h = wp_splitmix64(noise_seed + (uint64_t)id);
                const float u = ((float)(h >> 40) + 0.5f) * (1.0f / 16777216.0f);          // (0, 1)
                gk = acc + -logf(-logf(u));                                                // Gumbel(0, 1)
            }
            __syncthreads();                                   // every read of sm.p (TopP) is behind us
            if (tid < n) sm.p[tid] = gk;
            if (tid < 256) sm.hist[tid] = -1;                  // winner of rank r
            __syncthreads();
            if (tid < n && gk > -INFINITY) {
                int rank = 0;
                for (int j = 0; j < n; ++j) {
                    const float gj = sm.p[j];
                    rank += (gj > gk || (gj == gk && sm.idx[j] < id)) ? 1 : 0;
                }
                if (rank < K) sm.hist[rank] = tid;
            }
            __syncthreads();
            if (tid < BM_SPLIT * K) {
                const size_t slot = (size_t)row * BM_SPLIT * K + tid;      // [row][slice][k]: slice 0 holds the winners
                const int c = tid < K ? sm.hist[tid] : -1;
                // (the winner's values are re-formed from LDS exactly as above)
                float ck = -INFINITY, cv = -INFINITY;
                int ci = -1;
                if (c >= 0) {
                    float s2 = wp_unkey(sm.key[c]);
                    if (!wp_keep(w, s2)) s2 = -INFINITY;
                    cv = s2 + base; ck = sm.p[c]; ci = sm.idx[c];
                }
                p.cand_key[slot] = ck; p.cand_val[slot] = cv; p.cand_idx[slot] = ci;
            }
            if (tid == 0) { o[0] = w.kth; o[1] = w.mx; o[2] = w.invZ; o[3] = w.v0; o[4] = w.smin; o[5] = 1.f; }
            return;
        }
    }
    const WarpStats w = row_warp_stats<false, false>(sc, p.V, p.top_k, p.top_p, 2, red);   // min_tokens_to_keep = 2 under beams
    if (threadIdx.x == 0) { o[0] = w.kth; o[1] = w.mx; o[2] = w.invZ; o[3] = w.v0; o[4] = w.smin; o[5] = 0.f; }
}

__global__ __launch_bounds__(256) void beam_row_topk_kernel(BeamDev p) {
    extern __shared__ float sl[];                  // the slice's ranking keys
    __shared__ float rv[4];
    __shared__ int ri[4];
    if (*p.done) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row = blockIdx.x;
    if (p.do_sample && p.warp[(size_t)row * 8 + 5] != 0.f) return;      // beam_row_warp_kernel has taken this row's draws from its candidates
    const float* x = p.logits + (size_t)(row / p.logit_div) * p.ld;
    const int per = bm_slice(p.V);
    const int beg = blockIdx.y * per, end = min(beg + per, p.V);
    const int n = max(end - beg, 0);
    float M, logS;
    bm_row_lse(p, row, M, logS);
    const float base = p.run_score[row];
    const uint32_t* seen = p.seen ? p.seen + (size_t)row * p.seen_words : nullptr;
    WarpStats w;
    if (p.do_sample) {
        const float* o = p.warp + (size_t)row * 8;
        w.kth = o[0]; w.mx = o[1]; w.invZ = o[2]; w.v0 = o[3]; w.smin = o[4];
    }
    const uint64_t noise_seed = p.do_sample ? wp_splitmix64(p.seed ^ wp_splitmix64(((uint64_t)(uint32_t)*p.step << 32) | (uint32_t)row)) : 0ull;
    // accumulated score of continuation `id` of this beam (what HF ranks / samples from)
    auto acc_of = [&](int id) {
        float s = bm_logprob(p, x, seen, M, logS, id);
        if (p.do_sample) {
            s *= p.inv_temp;
            if (!wp_keep(w, s)) s = -INFINITY;
        }
        return s + base;
    };
    for (int i = tid; i < n; i += 256) {
        float key = acc_of(beg + i);
        if (p.do_sample) {
            const uint64_t h = wp_splitmix64(noise_seed + (uint64_t)(beg + i));
            const float u = ((float)(h >> 40) + 0.5f) * (1.0f / 16777216.0f);          // (0, 1)
            key += -logf(-logf(u));                                                    // Gumbel(0, 1)
        }
        sl[i] = key;
    }
    __syncthreads();
    const size_t slot = ((size_t)row * BM_SPLIT + blockIdx.y) * p.K;
    for (int k = 0; k < p.K; ++k) {
        float best = -INFINITY;
        int bi = 0x7fffffff;
        for (int i = tid; i < n; i += 256) bm_pair(best, bi, sl[i], i);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            bm_pair(best, bi, ov, oi);
        }
        if (lane == 0) { rv[wave] = best; ri[wave] = bi; }
        __syncthreads();
        if (tid == 0) {
            for (int w2 = 1; w2 < 4; ++w2) bm_pair(best, bi, rv[w2], ri[w2]);
            const bool ok = bi != 0x7fffffff;
            p.cand_key[slot + k] = ok ? best : -INFINITY;
            p.cand_val[slot + k] = ok ? (p.do_sample ? acc_of(beg + bi) : best) : -INFINITY;
            p.cand_idx[slot + k] = ok ? beg + bi : -1;
            if (ok) sl[bi] = -INFINITY;            // taken
        }
        __syncthreads();
    }
}

// one wave per request: the K best of its num_beams * BM_SPLIT * K slice winners, best first
// (ties: lowest flat index beam * V + token)
__global__ __launch_bounds__(64) void beam_merge_kernel(BeamDev p) {
    if (*p.done) return;
    const int b = blockIdx.x, lane = threadIdx.x;
    const int per_row = BM_SPLIT * p.K;
    const int n = p.nb * per_row;                  // <= 8 * 8 * 16 = 1024
    float v[16], val[16];
    int f[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int i = lane + 64 * j;
        v[j] = -INFINITY;
        val[j] = -INFINITY;
        f[j] = 0x7fffffff;
        if (i < n) {
            const int beam = i / per_row;
            const size_t src = (size_t)(b * p.nb + beam) * per_row + (i % per_row);
            const int tok = p.cand_idx[src];
            if (tok >= 0) { v[j] = p.cand_key[src]; val[j] = p.cand_val[src]; f[j] = beam * p.V + tok; }
        }
    }
    for (int k = 0; k < p.K; ++k) {
        float best = -INFINITY;
        int bf = 0x7fffffff;
#pragma unroll
        for (int j = 0; j < 16; ++j) bm_pair(best, bf, v[j], f[j]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(best, o, 64);
            const int of = __shfl_xor(bf, o, 64);
            bm_pair(best, bf, ov, of);
        }
        float mine = -INFINITY;                     // the winner's unperturbed score lives in one lane
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (f[j] == bf && bf != 0x7fffffff) { mine = val[j]; v[j] = -INFINITY; f[j] = 0x7fffffff; }
        mine = wave_max(mine);
        if (lane == 0) {
            const bool ok = bf != 0x7fffffff;
            p.top_val[b * p.K + k] = ok ? mine : -INFINITY;
            p.top_beam[b * p.K + k] = ok ? bf / p.V : 0;
            p.top_tok[b * p.K + k] = ok ? bf % p.V : 0;
        }
    }
}

// thread b = request b
__global__ void beam_update_kernel(BeamDev p) {
    __shared__ int fired_s;
    if (*p.done) return;
    const int b = threadIdx.x;
    const int t = *p.step;                         // tokens generated before this step
    const int nb = p.nb, K = p.K, R = p.B * nb;
    if (b == 0) {
        // the reference's StoppingCriteriaSub (starvector_base.py:9-20) looks at row 0 of what HF hands it -- under
        // beam search that is the best continuation of request 0 -- and returns a plain bool for the whole batch
        int fired = 0;
        if (p.n_stop > 0 && t + 1 >= p.n_stop) {
            fired = p.top_tok[0] == p.stop_ids[p.n_stop - 1];
            int x = p.top_beam[0];
            for (int i = p.n_stop - 2, s = t - 1; fired && i >= 0; --i, --s) {
                fired = p.hist_tok[(size_t)s * R + x] == p.stop_ids[i];
                x = p.hist_parent[(size_t)s * R + x];
            }
        }
        fired_s = fired;
    }
    __syncthreads();
    int ci = 0, alld = 1, allhit = 1;
    if (b < p.B) {
        float c[BM_MAXK], rs[BM_MAXK];
        int cb[BM_MAXK], ct[BM_MAXK];
        bool hit[BM_MAXK];
        for (int k = 0; k < K; ++k) {
            c[k] = p.top_val[b * K + k];
            cb[k] = p.top_beam[b * K + k];
            ct[k] = p.top_tok[b * K + k];
            hit[k] = fired_s || ct[k] == p.eos || t + 1 >= p.max_new;
            allhit &= hit[k] ? 1 : 0;
            rs[k] = c[k] + (hit[k] ? 1.f : 0.f) * BM_NEG;
        }
        // the num_beams best non-hitting continuations run on
        unsigned used = 0;
        float run0 = 0.f;
        for (int j = 0; j < nb; ++j) {
            int sel = -1;
            for (int k = 0; k < K; ++k)
                if (!((used >> k) & 1u) && (sel < 0 || rs[k] > rs[sel])) sel = k;
            used |= 1u << sel;
            const int r = b * nb + j;
            p.run_score[r] = rs[sel];
            if (j == 0) run0 = rs[sel];
            p.parent[r] = b * nb + cb[sel];
            p.cur_tok[r] = ct[sel];
            p.hist_parent[(size_t)t * R + r] = cb[sel];
            p.hist_tok[(size_t)t * R + r] = ct[sel];
            if (p.positions) p.positions[r] += 1;
        }
        // finished slots: hitting continuations ranked inside the first num_beams, length-penalised
        float ms[BM_MAXNB + BM_MAXK];
        int md[BM_MAXNB + BM_MAXK], mt[BM_MAXNB + BM_MAXK], mp[BM_MAXNB + BM_MAXK], mk[BM_MAXNB + BM_MAXK];
        int full = 1;
        for (int j = 0; j < nb; ++j) {
            ms[j] = p.fin_score[b * nb + j];
            md[j] = p.fin_done[b * nb + j];
            mt[j] = p.fin_step[b * nb + j];
            mp[j] = p.fin_parent[b * nb + j];
            mk[j] = p.fin_tok[b * nb + j];
            full &= md[j];
        }
        const int can = p.can_improve[b];
        const float lpow = p.lenpow[t + 1];
        for (int k = 0; k < K; ++k) {
            const int just = hit[k] && k < nb;
            float f = c[k] / lpow;
            f = f + ((full && p.early == 1) ? 1.f : 0.f) * BM_NEG;
            f = f + (can ? 0.f : 1.f) * BM_NEG;
            f = f + (just ? 0.f : 1.f) * BM_NEG;
            ms[nb + k] = f;
            md[nb + k] = just;
            mt[nb + k] = t;
            mp[nb + k] = cb[k];
            mk[nb + k] = ct[k];
        }
        unsigned used2 = 0;
        float fmin = INFINITY;
        float ns[BM_MAXNB];
        int nd[BM_MAXNB];
        for (int j = 0; j < nb; ++j) {
            int sel = -1;
            for (int k = 0; k < nb + K; ++k)
                if (!((used2 >> k) & 1u) && (sel < 0 || ms[k] > ms[sel])) sel = k;
            used2 |= 1u << sel;
            ns[j] = ms[sel];
            nd[j] = md[sel];
            p.fin_score[b * nb + j] = ms[sel];
            p.fin_done[b * nb + j] = md[sel];
            p.fin_step[b * nb + j] = mt[sel];
            p.fin_parent[b * nb + j] = mp[sel];
            p.fin_tok[b * nb + j] = mk[sel];
            fmin = fminf(fmin, ms[sel]);
            alld &= md[sel];
        }
        // can the best running beam still beat the worst finished hypothesis?
        const int hyp = (p.early == 2 && p.length_penalty > 0.f) ? p.max_new : t + 1;
        const float best_run = run0 / p.lenpow[hyp];
        int any = 0;
        for (int j = 0; j < nb; ++j) any |= best_run > (nd[j] ? fmin : BM_NEG) ? 1 : 0;
        ci = can && any;
        p.can_improve[b] = ci;
    }
    const int any_ci = __syncthreads_or(ci);
    const int all_done = __syncthreads_and(alld);
    const int all_hit = __syncthreads_and(allhit);
    if (b == 0) {
        *p.step = t + 1;
        const bool go_on = any_ci && !(all_done && p.early == 1) && !all_hit;
        if (!go_on) *p.done = 1;
    }
}

__global__ __launch_bounds__(256) void beam_seen_gather_kernel(BeamDev p) {
    extern __shared__ uint32_t bits[];             // [nb][seen_words]
    if (!p.seen) return;
    const int b = blockIdx.x, tid = threadIdx.x, W = p.seen_words, nb = p.nb;
    uint32_t* base = p.seen + (size_t)b * nb * W;
    for (int i = tid; i < nb * W; i += 256) bits[i] = base[i];
    __syncthreads();
    for (int j = 0; j < nb; ++j) {
        const int src = p.parent[b * nb + j] - b * nb;
        const int tok = p.cur_tok[b * nb + j];
        for (int i = tid; i < W; i += 256) {
            uint32_t wv = bits[src * W + i];
            if ((tok >> 5) == i) wv |= 1u << (tok & 31);
            base[(size_t)j * W + i] = wv;
        }
    }
}

__global__ void beam_reset_kernel(BeamDev p) {
    const int R = p.B * p.nb;
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < R; r += gridDim.x * blockDim.x) {
        const int j = r % p.nb;
        p.run_score[r] = j == 0 ? 0.f : BM_NEG;
        p.parent[r] = r - j;                        // every beam starts as a copy of the request's prompt
        p.cur_tok[r] = 0;
        p.fin_score[r] = BM_NEG;
        p.fin_done[r] = 0;
        p.fin_step[r] = -1;
        p.fin_parent[r] = 0;
        p.fin_tok[r] = 0;
        if (r < p.B) p.can_improve[r] = 1;
        if (r == 0) { *p.step = 0; *p.done = 0; }
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
#define BMCHK(x) do { hipError_t _e = (x); if (_e != hipSuccess) return (int)_e; } while (0)

template <typename T>
static int bm_alloc(BeamScorer* s, T** p, size_t count) {
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, (count ? count : 1) * sizeof(T));
    if (e != hipSuccess) return (int)e;
    e = hipMemset(q, 0, (count ? count : 1) * sizeof(T));
    if (e != hipSuccess) return (int)e;
    s->allocs.push_back(q);
    *p = reinterpret_cast<T*>(q);
    return 0;
}

int BeamScorer::init(const BeamConfig& cfg, int32_t* ext_cur_tok, int32_t* ext_positions, int32_t* ext_step,
                     int32_t* ext_done) {
    destroy();
    c = cfg;
    R = c.B * c.nb;
    K = 2 * c.nb;
    memset(&d, 0, sizeof(d));
    d.B = c.B; d.nb = c.nb; d.K = K; d.V = c.V; d.max_new = c.max_new;
    d.logit_div = 1;
#define A(x) do { int _r = (x); if (_r) { destroy(); return _r; } } while (0)
    A(bm_alloc(this, &d.run_score, R));
    A(bm_alloc(this, &d.parent, R));
    A(bm_alloc(this, &d.hist_parent, (size_t)c.max_new * R));
    A(bm_alloc(this, &d.hist_tok, (size_t)c.max_new * R));
    A(bm_alloc(this, &d.fin_score, R));
    A(bm_alloc(this, &d.fin_done, R));
    A(bm_alloc(this, &d.fin_step, R));
    A(bm_alloc(this, &d.fin_parent, R));
    A(bm_alloc(this, &d.fin_tok, R));
    A(bm_alloc(this, &d.can_improve, c.B));
    A(bm_alloc(this, &d.stats, (size_t)R * BM_SPLIT * 2));
    A(bm_alloc(this, &d.cand_val, (size_t)R * BM_SPLIT * K));
    A(bm_alloc(this, &d.cand_key, (size_t)R * BM_SPLIT * K));
    A(bm_alloc(this, &d.cand_idx, (size_t)R * BM_SPLIT * K));
    A(bm_alloc(this, &d.warp, (size_t)R * 8));
    A(bm_alloc(this, &d.top_val, (size_t)c.B * K));
    A(bm_alloc(this, &d.top_beam, (size_t)c.B * K));
    A(bm_alloc(this, &d.top_tok, (size_t)c.B * K));
    float* lp = nullptr;
    A(bm_alloc(this, &lp, (size_t)c.max_new + 1));
    d.lenpow = lp;
    int32_t* stop = nullptr;
    A(bm_alloc(this, &stop, BM_MAXSTOP));
    d.stop_ids = stop;
    d.seen_words = (c.V + 31) / 32;
    A(bm_alloc(this, &d.seen, (size_t)R * d.seen_words));
    if (ext_cur_tok) d.cur_tok = ext_cur_tok; else A(bm_alloc(this, &d.cur_tok, R));
    d.positions = ext_positions;
    if (ext_step && ext_done) { d.step = ext_step; d.done = ext_done; }
    else { A(bm_alloc(this, &d.step, 4)); A(bm_alloc(this, &d.done, 4)); }
#undef A
    return 0;
}

int BeamScorer::reset(hipStream_t st) {
    // per-call parameters (shape-independent: the device buffers are reused across calls of the same shape)
    d.eos = c.eos; d.early = c.early; d.n_stop = c.n_stop; d.length_penalty = c.length_penalty; d.penalty = c.penalty;
    d.min_new = c.min_new;
    d.do_sample = c.do_sample; d.inv_temp = 1.0f / c.temperature; d.top_p = c.top_p; d.top_k = c.top_k; d.seed = c.seed;
    std::vector<float> lp((size_t)c.max_new + 1);
    // HF divides by the Python float (cur_len + 1) ** length_penalty: computed in double, used as an fp32 scalar
    for (int t = 0; t <= c.max_new; ++t) lp[t] = (float)pow((double)t, (double)c.length_penalty);
    BMCHK(hipMemcpyAsync(const_cast<float*>(d.lenpow), lp.data(), lp.size() * sizeof(float), hipMemcpyHostToDevice, st));
    BMCHK(hipMemcpyAsync(const_cast<int32_t*>(d.stop_ids), c.stop, sizeof(c.stop), hipMemcpyHostToDevice, st));
    BMCHK(hipMemsetAsync(d.seen, 0, (size_t)R * d.seen_words * sizeof(uint32_t), st));
    beam_reset_kernel<<<(R + 255) / 256, 256, 0, st>>>(d);
    BMCHK(hipStreamSynchronize(st));               // lp / c.stop are host temporaries
    return 0;
}

void BeamScorer::enqueue_step(const float* logits, int ld, int logit_div, hipStream_t st) {
    BeamDev a = d;
    a.logits = logits; a.ld = ld; a.logit_div = logit_div;
    const bool pen = c.penalty > 0.f && c.penalty != 1.f;
    if (!pen) a.seen = nullptr;
    const int per = (((c.V + BM_SPLIT - 1) / BM_SPLIT) + 3) & ~3;
    beam_row_stats_kernel<<<dim3(R, BM_SPLIT), 256, 0, st>>>(a);
    if (c.do_sample) {
        // SV_BEAM_WARP_SELECT=0: the bisection form at every top_k (A/B switch of warp.h's selection path; read per call, a captured graph keeps its choice)
        const bool sel = !(getenv("SV_BEAM_WARP_SELECT") && atoi(getenv("SV_BEAM_WARP_SELECT")) == 0);
        if (sel) beam_row_warp_kernel<true><<<R, WP_THREADS, 0, st>>>(a);
        else beam_row_warp_kernel<false><<<R, WP_THREADS, 0, st>>>(a);
    }
    beam_row_topk_kernel<<<dim3(R, BM_SPLIT), 256, (size_t)per * sizeof(float), st>>>(a);
    beam_merge_kernel<<<c.B, 64, 0, st>>>(a);
    beam_update_kernel<<<1, ((c.B + 63) / 64) * 64, 0, st>>>(a);
    if (pen) beam_seen_gather_kernel<<<c.B, 256, (size_t)c.nb * a.seen_words * sizeof(uint32_t), st>>>(a);
}

int BeamScorer::finalize(hipStream_t st, std::vector<int64_t>& tokens, int& L, std::vector<float>& scores) {
    int32_t n_steps = 0;
    BMCHK(hipMemcpyAsync(&n_steps, d.step, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    BMCHK(hipStreamSynchronize(st));
    if (n_steps < 1 || n_steps > c.max_new) return -1;
    std::vector<int32_t> hp((size_t)n_steps * R), ht((size_t)n_steps * R), fs(R), fp(R), ft(R);
    std::vector<float> fsc(R);
    BMCHK(hipMemcpyAsync(hp.data(), d.hist_parent, hp.size() * 4, hipMemcpyDeviceToHost, st));
    BMCHK(hipMemcpyAsync(ht.data(), d.hist_tok, ht.size() * 4, hipMemcpyDeviceToHost, st));
    BMCHK(hipMemcpyAsync(fs.data(), d.fin_step, fs.size() * 4, hipMemcpyDeviceToHost, st));
    BMCHK(hipMemcpyAsync(fp.data(), d.fin_parent, fp.size() * 4, hipMemcpyDeviceToHost, st));
    BMCHK(hipMemcpyAsync(ft.data(), d.fin_tok, ft.size() * 4, hipMemcpyDeviceToHost, st));
    BMCHK(hipMemcpyAsync(fsc.data(), d.fin_score, fsc.size() * 4, hipMemcpyDeviceToHost, st));
    BMCHK(hipStreamSynchronize(st));
    // HF: output_fill_value = pad_token_id or eos_token_id[0]  (a pad id of 0 is falsy there)
    const int64_t fill = c.pad > 0 ? c.pad : c.eos;
    tokens.assign((size_t)c.B * c.max_new, fill);
    scores.assign(c.B, 0.f);
    L = 0;
    for (int b = 0; b < c.B; ++b) {
        const int slot = b * c.nb;                 // slot 0 = best finished hypothesis
        const int t_end = fs[slot];
        if (t_end < 0 || t_end >= n_steps) return -2;
        int64_t* row = tokens.data() + (size_t)b * c.max_new;
        row[t_end] = ft[slot];
        int x = fp[slot];
        for (int s = t_end - 1; s >= 0; --s) {
            if (x < 0 || x >= c.nb) return -3;
            row[s] = ht[(size_t)s * R + slot + x];
            x = hp[(size_t)s * R + slot + x];
        }
        if (t_end + 1 > L) L = t_end + 1;
        scores[b] = fsc[slot];
    }
    return 0;
}

void BeamScorer::destroy() {
    for (void* p : allocs) (void)hipFree(p);
    allocs.clear();
    R = 0;
}

// ------------------------------------------------------------------------------------------------
// KV cache: block-table reorder + tail-page copy
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int bm_cached_len(const BeamKvArgs& a) { return a.L_fixed >= 0 ? a.L_fixed : a.S0 + *a.step - 1; }

// one block per request: beam j's full pages become its parent's (the permutation is staged through LDS)
__global__ __launch_bounds__(256) void beam_table_reorder_kernel(BeamKvArgs a) {
    extern __shared__ int32_t tab[];               // [nb][pi]
    if (*a.done) return;
    const int pi = bm_cached_len(a) / SV_PAGE_TOKENS;     // index of the tail page = number of full pages
    const int b = blockIdx.x, nb = a.nb, tid = threadIdx.x;
    int32_t* rows = a.block_table + (size_t)b * nb * a.max_pages;
    for (int i = tid; i < nb * pi; i += 256) tab[i] = rows[(size_t)(i / pi) * a.max_pages + (i % pi)];
    __syncthreads();
    for (int i = tid; i < nb * pi; i += 256) {
        const int j = i / pi;
        const int src = a.parent[b * nb + j] - b * nb;
        rows[(size_t)j * a.max_pages + (i % pi)] = tab[src * pi + (i % pi)];
    }
}

// grid (B, n_layer * n_kv, BM_TAIL_Z): the tail pages of ONE request's beams for one (layer, KV head), a slice of the page per z.  The beams of a
// request permute among themselves, so a page may be both a source and a destination; a THREAD moves the same 16-byte pieces of every beam -- all its
// loads, then all its stores -- and no other thread touches those pieces: one launch, no staging copy (rounds 1-5 ran two launches over a staging
// buffer: stage, then commit; 2 x 11.8 us per decode step at 64 rows).
#define BM_TAIL_Z 4
__global__ __launch_bounds__(256) void beam_tail_copy_kernel(BeamKvArgs a) {
    if (*a.done) return;
    const int L = bm_cached_len(a);
    if (L % SV_PAGE_TOKENS == 0) return;           // the tail page is empty
    const int b = blockIdx.x, nb = a.nb;
    int src[BM_MAXNB];
    bool any = false;
#pragma unroll
    for (int j = 0; j < BM_MAXNB; ++j) {
        src[j] = -1;
        if (j < nb) {
            const int r = b * nb + j, sr = a.parent[r];
            if (sr != r) { src[j] = sr; any = true; }
        }
    }
    if (!any) return;
    const int pi = L / SV_PAGE_TOKENS;
    const int layer = blockIdx.y / a.n_kv, kvh = blockIdx.y % a.n_kv;
    char* pool = a.kv_pool + (size_t)layer * a.layer_stride + (size_t)kvh * a.kv_head_stride;
    const int n16 = (int)(a.page_bytes / 16), per = (n16 + BM_TAIL_Z - 1) / BM_TAIL_Z;
    const int beg = blockIdx.z * per, end = min(beg + per, n16);
    for (int i = beg + threadIdx.x; i < end; i += 256) {
        uint4 v[BM_MAXNB];
#pragma unroll
        for (int j = 0; j < BM_MAXNB; ++j)
            if (src[j] >= 0) v[j] = reinterpret_cast<const uint4*>(pool + ((size_t)src[j] * a.need + pi) * a.page_bytes)[i];
#pragma unroll
        for (int j = 0; j < BM_MAXNB; ++j)
            if (src[j] >= 0) reinterpret_cast<uint4*>(pool + ((size_t)(b * nb + j) * a.need + pi) * a.page_bytes)[i] = v[j];
    }
}

void launch_beam_table_reorder(const BeamKvArgs& a, hipStream_t st) {
    beam_table_reorder_kernel<<<a.B, 256, (size_t)a.nb * a.max_pages * sizeof(int32_t), st>>>(a);
}
void launch_beam_tail_copy(const BeamKvArgs& a, hipStream_t st) {
    beam_tail_copy_kernel<<<dim3(a.B, a.n_layer * a.n_kv, BM_TAIL_Z), 256, 0, st>>>(a);
}

}  // namespace sv
