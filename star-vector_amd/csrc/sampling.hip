// Token selection + generation bookkeeping on device (no per-step host sync).
//   argmax        : greedy (HF _sample with do_sample=False): argmax of the fp32 logits, lowest index on ties
//   sample_top_p  : TemperatureLogitsWarper -> TopKLogitsWarper -> TopPLogitsWarper (warp.h) -> softmax -> one
//                   multinomial draw (counter-based RNG; distributional parity with torch.multinomial)
//   finish_step   : pad-after-EOS, append, EOS bookkeeping, the reference's row-0 stop sequence
//                   (starvector_base.py:9-20), max-length budget -> device "done" flag
#include "kernels.h"
#include "warp.h"

namespace sv {

#define SP_THREADS WP_THREADS

__device__ __forceinline__ void argmax_pair(float& v, int& i, float ov, int oi) {
    if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
}
// RepetitionPenaltyLogitsProcessor: score < 0 ? score * penalty : score / penalty for ids already generated
__device__ __forceinline__ float rep_penalty(float l, int i, const uint32_t* seen_row, float penalty) {
    if (seen_row && ((seen_row[i >> 5] >> (i & 31)) & 1u)) return l < 0.f ? l * penalty : l / penalty;
    return l;
}

// grid (B, AM_SPLIT): one CU streams ~25 GB/s, so a 196 KB logits row is scanned by AM_SPLIT blocks; each
// leaves (max, index) of its slice and finish_step_kernel merges them (lowest index wins ties)
#define AM_SPLIT 8
__global__ __launch_bounds__(256) void argmax_kernel(const float* __restrict__ logits, int ld, int V,
                                                     float* __restrict__ pval, int32_t* __restrict__ pidx,
                                                     const uint32_t* __restrict__ seen, int seen_words, float penalty) {
    __shared__ float sv_[4];
    __shared__ int si_[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* row = logits + (size_t)blockIdx.x * ld;
    const uint32_t* srow = seen ? seen + (size_t)blockIdx.x * seen_words : nullptr;
    const int per = (((V + AM_SPLIT - 1) / AM_SPLIT) + 3) & ~3;
    const int beg = blockIdx.y * per, end = min(beg + per, V);
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = beg + tid * 4; i < end; i += 256 * 4) {
        const float4 v = *reinterpret_cast<const float4*>(row + i);
        const float a[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (i + e < end) argmax_pair(best, bi, rep_penalty(a[e], i + e, srow, penalty), i + e);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        argmax_pair(best, bi, ov, oi);
    }
    if (lane == 0) { sv_[wave] = best; si_[wave] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; ++w) argmax_pair(best, bi, sv_[w], si_[w]);
        pval[blockIdx.x * AM_SPLIT + blockIdx.y] = best;
        pidx[blockIdx.x * AM_SPLIT + blockIdx.y] = bi;
    }
}
__global__ void argmax_merge_kernel(const float* __restrict__ pval, const int32_t* __restrict__ pidx,
                                    int32_t* __restrict__ out, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int s = 0; s < AM_SPLIT; ++s) argmax_pair(best, bi, pval[b * AM_SPLIT + s], pidx[b * AM_SPLIT + s]);
    out[b] = bi;
}
void launch_argmax_partial(const float* logits, int ld, int V, float* pval, int32_t* pidx, int B, const uint32_t* seen,
                           int seen_words, float penalty, hipStream_t st) {
    argmax_kernel<<<dim3(B, AM_SPLIT), 256, 0, st>>>(logits, ld, V, pval, pidx, seen, seen_words, penalty);
}
void launch_argmax(const float* logits, int ld, int V, int32_t* out, float* pval, int32_t* pidx, int B, hipStream_t st) {
    launch_argmax_partial(logits, ld, V, pval, pidx, B, nullptr, 0, 1.f, st);
    argmax_merge_kernel<<<(B + 63) / 64, 64, 0, st>>>(pval, pidx, out, B);
}

// one multinomial draw from softmax(warped scores) of a row, by the whole 1024-thread block; `lg(i)` = the row's score after
// processors and temperature.  The random number is a pure function of (seed, step, row).  Every thread returns the token.
template <class F>
__device__ int sample_row(F lg, int V, int top_k, float top_p, uint64_t seed, uint32_t step, uint32_t rowid, float* red,
                          float* scan, int* result) {
    const int tid = threadIdx.x;
    const WarpStats w = row_warp_stats(lg, V, top_k, top_p, 1, red);                     // TopK -> TopP thresholds
    // multinomial over the survivors, in index order: per-thread contiguous ranges + block scan
    const int per = (V + SP_THREADS - 1) / SP_THREADS;
    const int beg = tid * per, end = min(beg + per, V);
    float mine = 0.f;
    for (int i = beg; i < end; ++i) {
        const float s = lg(i);
        mine += wp_keep(w, s) ? wp_prob(w, s) : 0.f;
    }
    __syncthreads();
    scan[tid] = mine;
    __syncthreads();
    if (tid == 0) {
        float run = 0.f;
        for (int t = 0; t < SP_THREADS; ++t) { const float x = scan[t]; scan[t] = run; run += x; }
        red[0] = run;
        *result = -1;
    }
    __syncthreads();
    const float total = red[0];
    const uint64_t h = wp_splitmix64(seed ^ wp_splitmix64(((uint64_t)step << 32) | rowid));
    const float u = (float)((h >> 40) * (1.0 / 16777216.0)) * total;
    const float base = scan[tid];
    if (mine > 0.f && u >= base && u < base + mine) {
        float run = base;
        int pick = -1;
        for (int i = beg; i < end; ++i) {
            const float s = lg(i);
            if (wp_keep(w, s)) {
                run += wp_prob(w, s);
                pick = i;                 // last survivor seen (guards fp round-off at the range end)
                if (u < run) break;
            }
        }
        *result = pick;
    }
    __syncthreads();
    if (tid == 0 && *result < 0) {
        // round-off fell between ranges: take the most probable token (always a survivor)
        float best = -INFINITY; int bi = 0x7fffffff;        // stays the out-of-range sentinel when no score is finite: the
        for (int i = 0; i < V; ++i) if (lg(i) > best) { best = lg(i); bi = i; }     // caller reports it (never used as an id)
        *result = bi;
    }
    __syncthreads();
    return *result;
}

__global__ __launch_bounds__(SP_THREADS) void sample_top_p_kernel(SampleArgs p) {
    __shared__ float red[SP_THREADS / 64];
    __shared__ float scan[SP_THREADS];
    __shared__ int result;
    const float* row = p.logits + (size_t)blockIdx.x * p.ld;
    const uint32_t* srow = p.seen ? p.seen + (size_t)blockIdx.x * p.seen_words : nullptr;
    const float invT = 1.0f / p.temperature;
    auto lg = [&](int i) { return rep_penalty(row[i], i, srow, p.penalty) * invT; };     // processors, then temperature
    const int tok = sample_row(lg, p.V, p.top_k, p.top_p, p.seed, (uint32_t)p.step[0], (uint32_t)blockIdx.x, red, scan, &result);
    if (threadIdx.x == 0) p.out[blockIdx.x] = tok;
}
void launch_sample_top_p(const SampleArgs& a, hipStream_t st) {
    sample_top_p_kernel<<<a.B, SP_THREADS, 0, st>>>(a);
}

__global__ void finish_step_kernel(FinishArgs p) {
    __shared__ int any_unf;
    if (*p.done) return;
    const int b = threadIdx.x;
    const int t = *p.step;
    if (b == 0) any_unf = 0;
    __syncthreads();
    if (b < p.B) {
        const int unf = p.unfinished[b];
        int nxt;
        if (p.pval) {                       // greedy: merge the AM_SPLIT slice winners of this row
            float best = -INFINITY;
            nxt = 0x7fffffff;
            for (int s = 0; s < AM_SPLIT; ++s) argmax_pair(best, nxt, p.pval[b * AM_SPLIT + s], p.pidx[b * AM_SPLIT + s]);
        } else {
            nxt = p.next[b];
        }
        if (unf && (unsigned)nxt >= (unsigned)p.V) {     // no finite logit in this row: never index the embedding table with it
            nxt = 0;
            if (p.bad) *p.bad = 1;
        }
        const int tok = unf ? nxt : p.pad;
        p.out_tokens[(size_t)b * p.ld_out + t] = tok;
        p.cur_tok[b] = tok;
        if (p.seen && tok >= 0) atomicOr(p.seen + (size_t)b * p.seen_words + (tok >> 5), 1u << (tok & 31));
        const int still = unf && tok != p.eos;
        p.unfinished[b] = still;
        p.positions[b] += 1;
        if (still) atomicOr(&any_unf, 1);
    }
    __syncthreads();
    if (b == 0) {
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
}
void launch_finish_step(const FinishArgs& a, hipStream_t st) {
    int threads = ((a.B + 63) / 64) * 64;
    finish_step_kernel<<<1, threads, 0, st>>>(a);
}

// ------------------------------------------------------------------------------------------------
// Continuous batching (the reference's worker admits 5 concurrent requests, serve/model_worker.py:
// 216-229, and runs them one HF generate each): every row ("slot") of the decode batch is an independent REQUEST with its
// own sampling parameters, budget, EOS and stop sequence.  One block per slot selects the token (greedy argmax, lowest
// index on ties; or temperature -> top-k -> top-p -> multinomial) and does the slot's bookkeeping -- everything is per row,
// so the reference's row-0 stop (starvector_base.py:9-20: right for ONE request per generate call) becomes each request's
// own stop.  The random stream of a slot depends on (its seed, its own step, row 0): a request produces the same tokens
// as when it runs alone through sv_generate.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(SP_THREADS) void cb_step_kernel(CbStepArgs p) {
    __shared__ float red[SP_THREADS / 64];
    __shared__ float scan[SP_THREADS];
    __shared__ int result;
    __shared__ float am_v[SP_THREADS / 64];
    __shared__ int am_i[SP_THREADS / 64];
    const int b = blockIdx.x;
    const int s = p.slot_map ? p.slot_map[b] : b;
    const CbSlot sl = p.slots[s];
    if (!sl.live) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* row = p.logits + (size_t)b * p.ld;
    const uint32_t* srow = (p.seen && sl.penalty > 0.f && sl.penalty != 1.0f) ? p.seen + (size_t)s * p.seen_words : nullptr;
    const bool hold_eos = sl.step < sl.min_new;                     // MinLengthLogitsProcessor
    int tok;
    if (sl.do_sample) {
        const float invT = 1.0f / sl.temperature;
        auto lg = [&](int i) { return (hold_eos && i == sl.eos) ? -INFINITY : rep_penalty(row[i], i, srow, sl.penalty) * invT; };
        tok = sample_row(lg, p.V, sl.top_k, sl.top_p, sl.seed, (uint32_t)sl.step, 0u, red, scan, &result);
    } else {
        float best = -INFINITY;
        int bi = 0x7fffffff;
        for (int i = tid * 4; i < p.V; i += SP_THREADS * 4) {
            const float4 v = *reinterpret_cast<const float4*>(row + i);
            const float a[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (i + e < p.V && !(hold_eos && i + e == sl.eos)) argmax_pair(best, bi, rep_penalty(a[e], i + e, srow, sl.penalty), i + e);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            argmax_pair(best, bi, ov, oi);
        }
        if (lane == 0) { am_v[wave] = best; am_i[wave] = bi; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < SP_THREADS / 64; ++w) argmax_pair(best, bi, am_v[w], am_i[w]);
            result = bi;
        }
        __syncthreads();
        tok = result;
    }
    if (tid != 0) return;
    if ((unsigned)tok >= (unsigned)p.V) {               // no finite logit: never index the embedding table with the sentinel
        tok = 0;
        if (p.bad) *p.bad = 1;
    }
    const int t = sl.step;
    int32_t* out = p.out_tokens + (size_t)s * p.ld_out;
    out[t] = tok;
    p.cur_tok[s] = tok;
    p.positions[s] += 1;
    if (srow) atomicOr(p.seen + (size_t)s * p.seen_words + (tok >> 5), 1u << (tok & 31));
    bool fin = tok == sl.eos || t + 1 >= sl.budget;
    if (!fin && sl.n_stop > 0 && t + 1 >= sl.n_stop) {
        fin = true;
        for (int i = 0; i < sl.n_stop; ++i)
            if (out[t + 1 - sl.n_stop + i] != p.slots[s].stop[i]) { fin = false; break; }      // from memory: indexing the register copy puts it on the stack
    }
    p.slots[s].step = t + 1;
    if (fin) {
        p.slots[s].live = 0;
        atomicSub(p.n_live, 1);
        atomicAdd(p.events, 1);
    }
}
void launch_cb_step(const CbStepArgs& a, int nblocks, hipStream_t st) {
    cb_step_kernel<<<nblocks, SP_THREADS, 0, st>>>(a);
}

}  // namespace sv
