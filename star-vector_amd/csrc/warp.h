// HF logits warpers on device, shared by the sampler (sampling.hip) and beam-sample (beam.hip):
//   TemperatureLogitsWarper -> TopKLogitsWarper -> TopPLogitsWarper   (generation/logits_process.py, the order
//   GenerationMixin._get_logits_processor builds them in), as reached from starvector_base.py:230-232.
// transformers==4.49.0 (the reference's pin, pyproject.toml:18) defaults GenerationConfig.top_k to 50, so the
// reference's sampling is top-k 50 THEN top-p even though it never passes top_k.
//
// One 1024-thread block per row.  `sc(i)` is the row's score after processors and temperature.  The filters are
// thresholds, found by bisection over monotone float bit patterns (no sort):
//   kth  : the top_k-th largest score        -> TopK keeps sc >= kth (ties kept, like HF's `scores < kth` removal)
//   v0   : smallest probability whose at-or-below mass exceeds 1 - top_p, probabilities renormalised over the top-k
//          survivors                          -> TopP keeps p >= v0
//   smin : the min_tokens_to_keep-th largest score (1 for sampling, 2 under beam search) -> always kept
#pragma once
#include "common.h"

namespace sv {

#define WP_THREADS 1024

struct WarpStats { float kth, mx, invZ, v0, smin; };

__device__ __forceinline__ uint32_t wp_key(float f) {          // order-preserving float -> uint
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float wp_unkey(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// block-wide reductions over WP_THREADS threads (every thread gets the result); red: >= 16 words of LDS
__device__ __forceinline__ float wp_block_sum(float v, float* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < WP_THREADS / 64; ++w) t += red[w];
    return t;
}
__device__ __forceinline__ float wp_block_max(float v, float* red) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v = wave_max(v);
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float t = -INFINITY;
#pragma unroll
    for (int w = 0; w < WP_THREADS / 64; ++w) t = fmaxf(t, red[w]);
    return t;
}

// v0 is itself the probability of the smallest kept token, computed in another pass: the same expression compiled in a
// second place may round one ulp lower (different fma contraction around __expf), which would drop exactly that token.
// A relative slack of 2^-18 restores it; a token that close to the threshold is inside HF's own float-cumsum noise.
__device__ __forceinline__ bool wp_keep(const WarpStats& w, float s) {
    if (!(s >= w.kth)) return false;
    return __expf(s - w.mx) * w.invZ >= w.v0 * 0.99999619f || s >= w.smin;
}
__device__ __forceinline__ float wp_prob(const WarpStats& w, float s) { return __expf(s - w.mx) * w.invZ; }

// The thresholds need ~70 passes over the row (two bisections).  The row is read ONCE into registers (NPT values per
// thread, strided: value j of thread t is score t + 1024 j) and every pass runs on registers; `sc` is only evaluated in
// that first sweep.  NPT * 1024 must cover V (StarVector: 49156 / 49157 -> NPT 52).
// ONCE (round 6, beam-sample): `sc` is evaluated exactly once per element -- the statistics min_tokens_to_keep = 2 needs (does the maximum occur
// twice?  else the largest score below it) are taken in the first sweep too, from each thread's own top two, so a heavy score functor (log-prob +
// repetition penalty + min-length hold) is inlined once and the 52 values stay in registers.  Same sums in the same order as the other forms:
// identical thresholds (tests/test_gpu_beam.py).
template <int NPT, bool ONCE = false, class F>
__device__ WarpStats row_warp_stats_regs(F sc, int V, int top_k, float top_p, int min_keep, float* red) {
    const int tid = threadIdx.x;
    WarpStats w;
    float v[NPT];
    float mx = -INFINITY;
    float t1 = -INFINITY, t2 = -INFINITY;               // ONCE: this thread's largest and second largest score (t2 == t1 when it occurs twice)
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
        const int i = tid + j * WP_THREADS;
        v[j] = i < V ? sc(i) : -INFINITY;
        mx = fmaxf(mx, v[j]);
        if (ONCE && i < V) {
            if (v[j] > t1) { t2 = t1; t1 = v[j]; } else if (v[j] > t2) t2 = v[j];
        }
        // ONCE: at most 8 elements' loads in flight at a time (all NPT hoisted to the top cost NPT more registers: scratch under the 128-VGPR cap)
        if (ONCE && (j & 7) == 7) __builtin_amdgcn_sched_barrier(0);
    }
    w.mx = mx = wp_block_max(mx, red);
    float once_cnt = 0.f, once_below = -INFINITY;
    if (ONCE && min_keep >= 2) {
        // how often the block maximum occurs (0, 1 or "2 or more" per thread is all that matters) and the largest score below it
        once_cnt = t1 == mx ? (t2 == mx ? 2.f : 1.f) : 0.f;
        once_below = t1 == mx ? (t2 == mx ? -INFINITY : t2) : t1;
        once_cnt = wp_block_sum(once_cnt, red);
        once_below = wp_block_max(once_below, red);
    }

    // TopK: kth = the k-th largest score, k = max(top_k, min_keep); off when top_k <= 0 or k >= V.  Padding entries are
    // -inf = the smallest key, so they only ever count towards thresholds at the very bottom (k < V keeps them out).
    w.kth = -INFINITY;
    const int k = top_k > 0 ? (top_k > min_keep ? top_k : min_keep) : 0;
    if (k > 0 && k < V) {
        uint32_t lo = 0u, hi = wp_key(mx) + 1u;          // count(key >= lo) >= k > count(key >= hi)
        // (round 6: the keys REPLACE the scores in their registers for the duration of the search -- wp_unkey(wp_key(x)) is x bit for bit.  Written as
        //  wp_key(v[j]) inside the loop, the compiler hoists the NPT loop-invariant keys and keeps 2 NPT values alive under the 128-VGPR cap of a
        //  1024-thread block: 31 spilled registers, reloaded in every sweep)
#pragma unroll
        for (int j = 0; j < NPT; ++j) v[j] = __uint_as_float(wp_key(v[j]));
        while (hi - lo > 1u) {
            const uint32_t mid = lo + ((hi - lo) >> 1);
            float c = 0.f;
#pragma unroll
            for (int j = 0; j < NPT; ++j) c += __float_as_uint(v[j]) >= mid ? 1.f : 0.f;
            c = wp_block_sum(c, red);                     // exact: counts < 2^24
            if (c >= (float)k) lo = mid; else hi = mid;
        }
        w.kth = wp_unkey(lo);
#pragma unroll
        for (int j = 0; j < NPT; ++j) v[j] = wp_unkey(__float_as_uint(v[j]));
    }
    // probabilities over the top-k survivors (unnormalised e = exp(s - mx), 0 for the filtered and the padding)
    float z = 0.f;
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
        v[j] = (v[j] >= w.kth && v[j] > -INFINITY) ? __expf(v[j] - mx) : (v[j] == mx ? 1.f : 0.f);
        z += v[j];
    }
    z = wp_block_sum(z, red);
    w.invZ = 1.0f / z;

    // TopP: ascending sort, drop while cumulative mass <= 1 - top_p  ==  keep p >= v0
    w.v0 = 0.f;
    if (top_p < 1.0f) {
        uint32_t lo = 0u, hi = 0x3f800000u;               // (lo, hi]
        const float cut = 1.0f - top_p;
        // (the probabilities replace the unnormalised values in their registers, for the same reason: v[j] * invZ is loop-invariant; the same products,
        //  formed once.  Beam-sample's warper: 1561 -> 1549 us per decode step at 64 rows, thresholds bit-identical.  An 8-way form of this search -- seven
        //  thresholds per sweep, one pair of barriers for their seven sums, same boundary because the sums are monotone in the threshold -- was built and
        //  is SLOWER: the accumulators push the sweep back into spills, profiles/beam_warp_r06.log)
#pragma unroll
        for (int j = 0; j < NPT; ++j) v[j] = v[j] * w.invZ;
        while (hi - lo > 1u) {
            const uint32_t mid = lo + ((hi - lo) >> 1);
            const float thr = __uint_as_float(mid);
            float f = 0.f;
#pragma unroll
            for (int j = 0; j < NPT; ++j) f += v[j] <= thr ? v[j] : 0.f;
            f = wp_block_sum(f, red);
            if (f > cut) hi = mid; else lo = mid;
        }
        w.v0 = __uint_as_float(hi);
    }
    // min_tokens_to_keep: 1 -> the maximum; 2 -> the second largest (== the maximum if it occurs twice).  v[] now holds
    // exp(s - mx): monotone in s, the maximum maps to exactly 1, so the second largest s is mx + log(second largest e)
    // only approximately -- recover it from the scores instead (one more sweep of `sc`, beam-sample only).
    w.smin = mx;
    if (min_keep >= 2) {
        if constexpr (ONCE) {
            if (once_cnt < 2.f) w.smin = once_below;
        } else {
            float cnt = 0.f, below = -INFINITY;
#pragma unroll
            for (int j = 0; j < NPT; ++j) {
                const int i = tid + j * WP_THREADS;
                if (i < V) {
                    const float s = sc(i);
                    if (s == mx) cnt += 1.f; else below = fmaxf(below, s);
                }
            }
            cnt = wp_block_sum(cnt, red);
            below = wp_block_max(below, red);
            if (cnt < 2.f) w.smin = below;
        }
    }
    return w;
}

// ------------------------------------------------------------------------------------------------
// TopK by SELECTION for the register form (0 < top_k <= 256: HF's effective default is top_k = 50, so the reference's default
// generate_im2svg call -- beam-sample, num_beams 2, top-p 0.9 -- runs TopK 50 THEN TopP on every beam row of every step).  The two bisections above
// are 32 + 30 sweeps of NPT register values per thread, each with a block reduction: ~140 us per decode step at 64 rows x 49157 columns.  After
// TopK only ~k scores are alive, so (the sampler's scheme, sampling.hip::sample_row_topk, on the values already in registers):
//   1. T0 = the k-th largest of the 1024 THREAD maxima (8-bit MSB-first radix select, one key per thread): k distinct elements are >= T0, so
//      the row's k-th largest is too -- everything >= T0 is a candidate, ~k of them
//   2. the candidates' keys go to LDS in thread order; kth = the key with #(greater) < k <= #(greater or equal)  (= what the bisection finds:
//      the largest threshold that still keeps k scores; ties at the k-th value all survive)
//   3. Z: the SAME register sweep and block sum as above (invZ bit-identical); the candidates' e = exp(s - mx) ride to LDS with it
//   4. TopP on the candidates: F(c) = mass of candidates with p <= p_c (summed in LDS order), v0 = the smallest p_c with F(c) > 1 - top_p,
//      1.0 when there is none -- the bisection's fixed point, with another (fixed) summation order inside F
// Returns false, block-uniformly, when the row is outside its scope (no finite maximum, more than WS_CAP candidates: ties across T0): the
// caller takes the bisections.
// ------------------------------------------------------------------------------------------------
#define WS_CAP 1024
struct WarpSelSmem {
    uint32_t key[WS_CAP];
    float p[WS_CAP];
    int idx[WS_CAP];                 // column of candidate c (beam-sample folds its top-K selection onto the candidates: beam.hip)
    int n_cand;
    int hist[256];
    int wtot[WP_THREADS / 64];
    uint32_t sel_key;
    int sel_need;
};
__device__ __forceinline__ WarpSelSmem& warp_sel_smem() {
    __shared__ WarpSelSmem s;
    return s;
}

template <int NPT, class F>
__device__ bool row_warp_stats_select(F sc, int V, int top_k, float top_p, int min_keep, float* red, WarpStats& w) {
    WarpSelSmem& sm = warp_sel_smem();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k = top_k > min_keep ? top_k : min_keep;                 // the caller checked 0 < top_k, k <= 256, k < V
    float v[NPT];
    float t1 = -INFINITY, t2 = -INFINITY;                              // this thread's largest and second largest score
#pragma unroll
    for (int j = 0; j < NPT; ++j) {
        const int i = tid + j * WP_THREADS;
        v[j] = i < V ? sc(i) : -INFINITY;
        if (i < V) {
            if (v[j] > t1) { t2 = t1; t1 = v[j]; } else if (v[j] > t2) t2 = v[j];
        }
        if ((j & 7) == 7) __builtin_amdgcn_sched_barrier(0);
    }
    const float mx = wp_block_max(t1, red);
    if (!(mx > -INFINITY) || !(mx < INFINITY)) return false;
    w.mx = mx;
    w.smin = mx;
    if (min_keep >= 2) {
        float cnt2 = t1 == mx ? (t2 == mx ? 2.f : 1.f) : 0.f;
        float below = t1 == mx ? (t2 == mx ? -INFINITY : t2) : t1;
        cnt2 = wp_block_sum(cnt2, red);
        below = wp_block_max(below, red);
        if (cnt2 < 2.f) w.smin = below;
    }

    // ---- 1. T0 = the k-th largest thread maximum ----
    const uint32_t mkey = wp_key(t1);
    uint32_t prefix = 0u;
    int need = k;
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        if (tid < 256) sm.hist[tid] = 0;
        __syncthreads();
        if (pass == 0 || (mkey >> (shift + 8)) == prefix) atomicAdd(&sm.hist[(mkey >> shift) & 255u], 1);
        __syncthreads();
        if (tid < 64) {                                                // lane l owns bins 255 - 4l .. 252 - 4l (descending keys)
            const int b0 = 255 - 4 * tid;
            const int h0 = sm.hist[b0], h1 = sm.hist[b0 - 1], h2 = sm.hist[b0 - 2], h3 = sm.hist[b0 - 3];
            const int sum4 = h0 + h1 + h2 + h3;
            int inc = sum4;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(inc, o, 64);
                if (tid >= o) inc += t;
            }
            int above = inc - sum4;
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
    const float T0 = wp_unkey(prefix);                                 // a thread maximum: finite or -inf, never NaN

    // ---- 2. candidates (>= T0, float compare: a NaN is none) into LDS, thread order ----
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < NPT; ++j) cnt += v[j] >= T0 ? 1 : 0;
    int inc = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    __syncthreads();
    if (lane == 63) sm.wtot[wave] = inc;
    __syncthreads();
    int off = inc - cnt, n = 0;
#pragma unroll
    for (int w2 = 0; w2 < WP_THREADS / 64; ++w2) {
        const int t = sm.wtot[w2];
        off += w2 < wave ? t : 0;
        n += t;
    }
    if (n > WS_CAP || n < k) return false;                             // block-uniform
    {
        int q = off;
#pragma unroll
        for (int j = 0; j < NPT; ++j)
            if (v[j] >= T0) { sm.key[q] = wp_key(v[j]); sm.idx[q] = tid + j * WP_THREADS; ++q; }
        if (tid == 0) sm.n_cand = n;
    }
    __syncthreads();
    if (tid < n) {
        const uint32_t kc = sm.key[tid];
        int gt = 0, ge = 0;
        for (int j = 0; j < n; ++j) {
            const uint32_t kj = sm.key[j];
            gt += kj > kc ? 1 : 0;
            ge += kj >= kc ? 1 : 0;
        }
        if (gt < k && k <= ge) sm.sel_key = kc;                        // every writer writes the same key
    }
    __syncthreads();
    w.kth = wp_unkey(sm.sel_key);

    // ---- 3. Z exactly as the bisection form sums it; the candidates' e to LDS ----
    float z = 0.f;
    {
        int q = off;
#pragma unroll
        for (int j = 0; j < NPT; ++j) {
            const bool cand = v[j] >= T0;
            const float e = (v[j] >= w.kth && v[j] > -INFINITY) ? __expf(v[j] - mx) : (v[j] == mx ? 1.f : 0.f);
            z += e;
            if (cand) sm.p[q++] = e;
        }
    }
    z = wp_block_sum(z, red);
    w.invZ = 1.0f / z;

    // ---- 4. TopP on the candidates ----
    w.v0 = 0.f;
    if (top_p < 1.0f) {
        const float cut = 1.0f - top_p;
        const float pc = tid < n ? sm.p[tid] * w.invZ : 0.f;
        __syncthreads();
        if (tid < n) sm.p[tid] = pc;
        __syncthreads();
        float v0c = 1.0f;                                              // the bisection's upper end: nothing exceeds the cut -> 1.0
        if (pc > 0.f) {
            float f = 0.f;
            for (int j = 0; j < n; ++j) { const float pj = sm.p[j]; f += pj <= pc ? pj : 0.f; }
            if (f > cut) v0c = pc;
        }
        w.v0 = -wp_block_max(-v0c, red);
    }
    return true;
}

template <class F>
__device__ WarpStats row_warp_stats_global(F sc, int V, int top_k, float top_p, int min_keep, float* red) {
    const int tid = threadIdx.x;
    WarpStats w;
    float mx = -INFINITY;
    for (int i = tid; i < V; i += WP_THREADS) mx = fmaxf(mx, sc(i));
    w.mx = mx = wp_block_max(mx, red);

    // TopK: kth = the k-th largest score, k = max(top_k, min_keep); off when top_k <= 0 or k >= V
    w.kth = -INFINITY;
    const int k = top_k > 0 ? (top_k > min_keep ? top_k : min_keep) : 0;
    if (k > 0 && k < V) {
        uint32_t lo = 0u, hi = wp_key(mx) + 1u;          // count(key >= lo) >= k > count(key >= hi)
        while (hi - lo > 1u) {
            const uint32_t mid = lo + ((hi - lo) >> 1);
            float c = 0.f;
            for (int i = tid; i < V; i += WP_THREADS) c += wp_key(sc(i)) >= mid ? 1.f : 0.f;
            c = wp_block_sum(c, red);                     // exact: counts < 2^24
            if (c >= (float)k) lo = mid; else hi = mid;
        }
        w.kth = wp_unkey(lo);
    }
    float z = 0.f;
    for (int i = tid; i < V; i += WP_THREADS) { const float s = sc(i); z += s >= w.kth ? __expf(s - mx) : 0.f; }
    z = wp_block_sum(z, red);
    w.invZ = 1.0f / z;

    // TopP: ascending sort, drop while cumulative mass <= 1 - top_p  ==  keep p >= v0
    w.v0 = 0.f;
    if (top_p < 1.0f) {
        uint32_t lo = 0u, hi = 0x3f800000u;               // (lo, hi]
        const float cut = 1.0f - top_p;
        while (hi - lo > 1u) {
            const uint32_t mid = lo + ((hi - lo) >> 1);
            const float thr = __uint_as_float(mid);
            float f = 0.f;
            for (int i = tid; i < V; i += WP_THREADS) {
                const float s = sc(i);
                const float pr = s >= w.kth ? __expf(s - mx) * w.invZ : 0.f;
                f += pr <= thr ? pr : 0.f;
            }
            f = wp_block_sum(f, red);
            if (f > cut) hi = mid; else lo = mid;
        }
        w.v0 = __uint_as_float(hi);
    }
    // min_tokens_to_keep: 1 -> the maximum; 2 -> the second largest (== the maximum if it occurs twice)
    w.smin = mx;
    if (min_keep >= 2) {
        float cnt = 0.f, below = -INFINITY;
        for (int i = tid; i < V; i += WP_THREADS) {
            const float s = sc(i);
            if (s == mx) cnt += 1.f; else below = fmaxf(below, s);
        }
        cnt = wp_block_sum(cnt, red);
        below = wp_block_max(below, red);
        if (cnt < 2.f) w.smin = below;
    }
    return w;
}

// REGS = false keeps every pass on `sc` (beam-sample: its score functor is heavier and the register copy spills)
template <bool REGS = true, bool SELECT = true, class F>
__device__ WarpStats row_warp_stats(F sc, int V, int top_k, float top_p, int min_keep, float* red) {
    if (!REGS) {
        // beam-sample: one evaluation of the (heavy) score functor per element, everything else on registers -- the all-passes-on-`sc` form took
        // 317 us per decode step at 64 rows x 49157 columns (rocprof, BASELINE config 2 with num_beams 2: 17 % of the step)
        if (V <= 49 * WP_THREADS) {                                                                                 // StarVector: 49156 / 49157 columns
            const int k = top_k > min_keep ? top_k : min_keep;
            if (SELECT && top_k > 0 && k <= 256 && k < V) {
                WarpStats w;
                if (row_warp_stats_select<49>(sc, V, top_k, top_p, min_keep, red, w)) return w;
            }
            return row_warp_stats_regs<49, true>(sc, V, top_k, top_p, min_keep, red);
        }
        return row_warp_stats_global(sc, V, top_k, top_p, min_keep, red);
    }
    if (V <= 16 * WP_THREADS) return row_warp_stats_regs<16>(sc, V, top_k, top_p, min_keep, red);
    if (V <= 52 * WP_THREADS) return row_warp_stats_regs<52>(sc, V, top_k, top_p, min_keep, red);
    return row_warp_stats_global(sc, V, top_k, top_p, min_keep, red);
}

__device__ __forceinline__ uint64_t wp_splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

}  // namespace sv
