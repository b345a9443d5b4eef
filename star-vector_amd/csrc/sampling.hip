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

// ------------------------------------------------------------------------------------------------
// TopK (k <= 256) -> TopP -> multinomial without the two 32-step bisections of warp.h (each step = a sweep of the row's 48 register
// values per thread + a block reduction: 126 us per step at V = 49157, 3 % of StarVector-8B's decode step).  Selection instead:
//   1. one sweep of the row: mx = the maximum, m_t = every thread's own maximum (thread t owns the scores t, t + 1024, ...)
//   2. T0 = the k-th largest of the 1024 thread maxima: an 8-bit MSB-first radix select (4 passes: LDS histogram of one key per
//      thread, one wave scans the 256 bins).  The k largest thread maxima are k distinct elements >= T0, so the k-th largest
//      element of the ROW is >= T0: everything >= T0 is a candidate, and there are ~k of them (not 49157)
//   3. a second sweep compacts the candidates (key, index) into LDS (block scan of the per-thread counts, thread order = a fixed
//      order); more than TK_CAP of them, or more than 4 in one thread -> general path
//   4. exact k-th largest among the n candidates by counting ranks (n^2 / 1024 LDS reads per thread, n ~ 60): kth = the key with
//      #(greater) < k <= #(greater or equal) -- ties at the k-th value are all kept, like HF's `scores < kth` removal
//   5. softmax over the survivors; TopP: F(c) = the mass of survivors with p <= p_c, v0 = the smallest p_c with F(c) > 1 - top_p,
//      keep p >= v0 (the maximum always: min_tokens_to_keep = 1) -- warp.h's rule, evaluated on the ~k survivors
//   6. one multinomial draw over the kept tokens in INDEX order: the token with the largest index whose exclusive prefix mass is
//      <= u (the same counter-based uniform as the general path)
// Deterministic (fixed summation orders).  Returns -1 when the row is outside its scope (no finite score, too many candidates).
// ------------------------------------------------------------------------------------------------
#define TK_CAP 2048
struct TopkSmem {
    uint32_t key[TK_CAP];
    int idx[TK_CAP];
    float p[TK_CAP];
    int hist[256];
    int wtot[SP_THREADS / 64];
    uint32_t sel_key;
    int sel_need;
    int pick;
};
__device__ __forceinline__ TopkSmem& topk_smem() {
    __shared__ TopkSmem s;
    return s;
}

template <class F>
__device__ int sample_row_topk(F lg, int V, int k, float top_p, uint64_t seed, uint32_t step, uint32_t rowid, float* red) {
    TopkSmem& sm = topk_smem();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float mt = -INFINITY;
    // sweep 1 (strided: thread t owns i = t mod 1024), 8 independent loads per round trip
    constexpr int UNR = 8;
    for (int i0 = tid; i0 < V; i0 += UNR * SP_THREADS) {
        float x[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int i = i0 + u * SP_THREADS;
            x[u] = i < V ? lg(i) : -INFINITY;
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) mt = fmaxf(mt, x[u]);
    }
    const float mx = wp_block_max(mt, red);
    if (!(mx > -INFINITY) || !(mx < INFINITY)) return -1;             // no finite score / NaN / +inf: the general path reports it

    // ---- 2. T0 = k-th largest thread maximum ----
    const uint32_t mkey = wp_key(mt);
    uint32_t prefix = 0u;
    int need = k;
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        if (tid < 256) sm.hist[tid] = 0;
        __syncthreads();
        if (pass == 0 || (mkey >> (shift + 8)) == prefix) atomicAdd(&sm.hist[(mkey >> shift) & 255u], 1);
        __syncthreads();
        if (tid < 64) {                                              // lane l owns bins 255 - 4l .. 252 - 4l (descending keys)
            const int b0 = 255 - 4 * tid;
            const int h0 = sm.hist[b0], h1 = sm.hist[b0 - 1], h2 = sm.hist[b0 - 2], h3 = sm.hist[b0 - 3];
            const int sum4 = h0 + h1 + h2 + h3;
            int inc = sum4;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(inc, o, 64);
                if (tid >= o) inc += t;
            }
            int above = inc - sum4;                                   // keys in the bins above this lane's four
            if (above < need && need <= inc) {
                int d = b0;
                if (need > above + h0) {
                    above += h0; d = b0 - 1;
                    if (need > above + h1) {
                        above += h1; d = b0 - 2;
                        if (need > above + h2) { above += h2; d = b0 - 3; }
                    }
                }
                sm.sel_key = (prefix << 8) | (uint32_t)d;
                sm.sel_need = need - above;
            }
        }
        __syncthreads();
        prefix = sm.sel_key;
        need = sm.sel_need;
    }
    const uint32_t t0 = prefix;

    // ---- 3. candidates >= T0 into LDS (sweep 2: a thread meets ~0.06 of them; more than TK_PER -> general path) ----
    constexpr int TK_PER = 4;
    uint32_t ck[TK_PER];
    int ci[TK_PER];
    int cnt = 0;
    for (int i0 = tid; i0 < V; i0 += UNR * SP_THREADS) {
        uint32_t x[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int i = i0 + u * SP_THREADS;
            x[u] = i < V ? wp_key(lg(i)) : 0u;                      // key 0 < every real key (t0 >= the key of a finite score)
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            if (x[u] >= t0) {
#pragma unroll
                for (int q = 0; q < TK_PER; ++q)
                    if (cnt == q) { ck[q] = x[u]; ci[q] = i0 + u * SP_THREADS; }
                ++cnt;
            }
        }
    }
    int inc = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    if (lane == 63) sm.wtot[wave] = inc;
    if (tid == 0) sm.pick = -1;
    const float over = wp_block_max(cnt > TK_PER ? 1.f : 0.f, red);  // (syncs inside: wtot / pick are visible after it)
    int off = inc - cnt, n = 0;
#pragma unroll
    for (int w = 0; w < SP_THREADS / 64; ++w) {
        const int t = sm.wtot[w];
        off += w < wave ? t : 0;
        n += t;
    }
    if (over > 0.f || n > TK_CAP || n < k) return -1;                 // block-uniform
#pragma unroll
    for (int q = 0; q < TK_PER; ++q)
        if (q < cnt) { sm.key[off + q] = ck[q]; sm.idx[off + q] = ci[q]; }
    __syncthreads();

    // ---- 4. the exact k-th largest key ----
    for (int c = tid; c < n; c += SP_THREADS) {
        const uint32_t kc = sm.key[c];
        int gt = 0, ge = 0;
        for (int j = 0; j < n; ++j) {
            const uint32_t kj = sm.key[j];
            gt += kj > kc ? 1 : 0;
            ge += kj >= kc ? 1 : 0;
        }
        if (gt < k && k <= ge) sm.sel_key = kc;                       // every writer writes the same key
    }
    __syncthreads();
    const uint32_t kth = sm.sel_key;

    // ---- 5. softmax over the survivors, TopP ----
    float e[(TK_CAP + SP_THREADS - 1) / SP_THREADS];
    float zs = 0.f;
#pragma unroll
    for (int q = 0; q < (TK_CAP + SP_THREADS - 1) / SP_THREADS; ++q) {
        const int c = tid + q * SP_THREADS;
        e[q] = (c < n && sm.key[c] >= kth) ? __expf(wp_unkey(sm.key[c]) - mx) : 0.f;
        zs += e[q];
    }
    const float invZ = 1.0f / wp_block_sum(zs, red);
#pragma unroll
    for (int q = 0; q < (TK_CAP + SP_THREADS - 1) / SP_THREADS; ++q) {
        const int c = tid + q * SP_THREADS;
        e[q] *= invZ;
        if (c < n) sm.p[c] = e[q];
    }
    __syncthreads();
    if (top_p < 1.0f) {
        const float cut = 1.0f - top_p;
        float v0c = INFINITY;
#pragma unroll
        for (int q = 0; q < (TK_CAP + SP_THREADS - 1) / SP_THREADS; ++q) {
            const int c = tid + q * SP_THREADS;
            if (c < n && e[q] > 0.f) {
                float f = 0.f;
                for (int j = 0; j < n; ++j) { const float pj = sm.p[j]; f += pj <= e[q] ? pj : 0.f; }
                if (f > cut) v0c = fminf(v0c, e[q]);
            }
        }
        const float v0 = -wp_block_max(-v0c, red);                    // (syncs inside: every F loop has finished)
#pragma unroll
        for (int q = 0; q < (TK_CAP + SP_THREADS - 1) / SP_THREADS; ++q) {
            const int c = tid + q * SP_THREADS;
            if (c < n) {
                const bool keep = e[q] > 0.f && (e[q] >= v0 || sm.key[c] == wp_key(mx));
                e[q] = keep ? e[q] : 0.f;
                sm.p[c] = e[q];
            }
        }
        __syncthreads();
    }

    // ---- 6. multinomial over the kept tokens, index order ----
    float ts = 0.f;
#pragma unroll
    for (int q = 0; q < (TK_CAP + SP_THREADS - 1) / SP_THREADS; ++q) ts += e[q];
    const float total = wp_block_sum(ts, red);
    const uint64_t h = wp_splitmix64(seed ^ wp_splitmix64(((uint64_t)step << 32) | rowid));
    const float u = (float)((h >> 40) * (1.0 / 16777216.0)) * total;
#pragma unroll
    for (int q = 0; q < (TK_CAP + SP_THREADS - 1) / SP_THREADS; ++q) {
        const int c = tid + q * SP_THREADS;
        if (c < n && e[q] > 0.f) {
            const int ic = sm.idx[c];
            float pre = 0.f;
            for (int j = 0; j < n; ++j) pre += sm.idx[j] < ic ? sm.p[j] : 0.f;
            if (pre <= u) atomicMax(&sm.pick, ic);
        }
    }
    __syncthreads();
    const int tok = sm.pick;
    __syncthreads();                                                  // sm is reused by the caller's next row (beam-free: one row per block)
    return tok;
}

// one multinomial draw from softmax(warped scores) of a row, by the whole 1024-thread block; `lg(i)` = the row's score after
// processors and temperature.  The random number is a pure function of (seed, step, row).  Every thread returns the token.
template <class F>
__device__ int sample_row(F lg, int V, int top_k, float top_p, uint64_t seed, uint32_t step, uint32_t rowid, float* red,
                          float* scan, int* result) {
    const int tid = threadIdx.x;
    if (top_k > 0 && top_k <= 256 && top_k < V) {                                        // the reference's default: top_k = 50
        const int t = sample_row_topk(lg, V, top_k, top_p, seed, step, rowid, red);
        if (t >= 0) return t;                                                            // -1: outside its scope (block-uniform)
    }
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
        int nxt;
        if (p.amax) {                       // greedy, selection folded into the lm_head launch: decode the row's key, re-arm the slot
            unsigned long long* slot = p.amax + (size_t)b * SV_AMAX_STRIDE;
            nxt = sv_amax_index(__hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            __hip_atomic_store(slot, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (p.pval) {                // greedy: merge the AM_SPLIT slice winners of this row
            float best = -INFINITY;
            nxt = 0x7fffffff;
            for (int s = 0; s < AM_SPLIT; ++s) argmax_pair(best, nxt, p.pval[b * AM_SPLIT + s], p.pidx[b * AM_SPLIT + s]);
        } else {
            nxt = p.next[b];
        }
        if (finish_step_row(p, b, t, nxt)) atomicOr(&any_unf, 1);
    }
    __syncthreads();
    if (b == 0) finish_step_call(p, t, any_unf);
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
    // block-uniform by construction: say so, and the slot's fields travel as scalar loads into SGPRs instead of occupying ~30 VGPRs
    // across sample_row (round 3: a by-value copy of the slot, 224 B of scratch per lane; now 80 B, all of it inside the general
    // bisection path of sample_row, which holds 48 scores per thread under the 128-VGPR cap of a 1024-thread block)
    const int s = __builtin_amdgcn_readfirstlane(p.slot_map ? p.slot_map[b] : b);
    const CbSlot& sl = p.slots[s];                                  // a reference: the fields are read where they are used (uniform address)
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
        if (p.bad) atomicCAS(p.bad, 0, 1);              // 0 -> 1 only (see finish_step_kernel)
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
