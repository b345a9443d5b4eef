// Attention output projection of the decode step WITHOUT split-K slabs (gfx950), and the LayerNorm fold that lets its consumer
// skip the row-update launch.
//
// The slab pipeline runs 7 launches per layer; two of them are row updates (split-K slab sum + bias + residual + LayerNorm ->
// the next GEMM's activation operand), 4.7 us each in situ for ~0.5 MB of traffic: launch seams, not work.  The one after the
// attention output projection goes away like this (gpt_bigcode/modeling_gpt_bigcode.py:694-755, ln_2 + mlp.c_fc):
//
//   * gemm_cols_resid_kernel: a block owns `cpb` <= 16 output columns over the WHOLE K (K = hidden: 128 KiB of activations per
//     block come out of L2 next to 32 KiB of weights -- affordable for this projection, not for the K = 4 * hidden down
//     projection, which keeps its slabs), so no cross-block reduction exists and the epilogue finishes the op in place:
//     h = bf16(h + bf16(x W^T + b)) in fragment order.
//   * the consumer c_fc reads the RAW residual stream as its MFMA operand; ln_2 is applied algebraically in its epilogue:
//         LN(h) W^T + bias = rstd * (h W'^T - mean * c1) + c2,   W' = bf16(W * gamma),  c1[n] = sum_k W'[n][k],
//         c2[n] = sum_k beta[k] W[n][k] + bias[n]
//     (fold_prepare_kernel builds W', c1, c2 once per engine).  The row statistics (sum, sum of squares) are accumulated from
//     the activation fragments on their way to the MFMA inside every c_fc block (a few packed VALU operations per fragment under
//     the weight stream; first version: per-block partials left by the producer, 64 KB re-read by every consumer block -- +1.4 us).
//     Nothing is normalised or rewritten: round 2's "LayerNorm inside the GEMM block" had to rewrite the operand before the
//     MFMA (two passes + packing), cost 2.5 us of VALU per block and lost.
//
// Cast points: the reference rounds LN(h) to bf16 before the GEMM; here the normalised activations are never materialised (the
// products (h_k - mean) * gamma_k * W[n][k] are formed from the bf16-rounded W' instead): same error budget, different
// roundings -- compared against the oracle with the same tolerances as every other kernel (tests/test_gpu_e2e.py).
#include "kernels.h"

namespace sv {

// ------------------------------------------------------------------------------------------------
// gemm_cols_resid_kernel<WAVES, G>
//   grid (ceil(N / cpb), MT), block WAVES * 64.  K is cut into chunks of 32 (one v_mfma_f32_16x16x32_bf16 k-step); wave w owns
//   the chunks [w * cpw, (w + 1) * cpw).  Lane (r = l & 15, g = l >> 4):
//     A (weights):      W[j * cpb + r][32 c + 8 g .. + 8]   read straight from the 32-column fragment image (the lanes of one
//                       column are 16-byte pieces of full 128-byte lines); lanes with r >= cpb hold zero
//     B (activations):  x[16 hb + r][32 c + 8 g .. + 8],  hb = 0, 1
//     D:                out[col 4 g + reg][row 16 hb + r] in acc[hb][reg]
//   W and x stream in groups of G chunks, two groups in flight.
// ------------------------------------------------------------------------------------------------
struct ColsKernarg { const bf16_t* Wp; const bf16_t* xp; bf16_t* h_xp; const bf16_t* bias; int K, N, cpb, out_KS; ColsArgs p; };

// CT = 16-column MFMA tiles per block (1: cpb <= 16, 2: cpb <= 32).  18 columns per block make N = 4608 exactly 256 blocks (one
// round on 256 CUs; 16 columns are 288 blocks = a second round for 32 of them).
template <int WAVES, int G, int CT>
__global__ __launch_bounds__(WAVES * 64) void gemm_cols_resid_kernel(const bf16_t* Wp_, const bf16_t* xp_, bf16_t* h_xp_, const bf16_t* bias_,
                                                                     int K_, int N_, int cpb_, int out_KS_, ColsArgs p_unused) {
    constexpr int LDT = 16 * CT + 1;
    extern __shared__ __attribute__((aligned(16))) char dg_smem[];
    float* red = reinterpret_cast<float*>(dg_smem);                     // [WAVES][CT][512]
    float* tile = red + WAVES * CT * 512;                                // [32][LDT]: finished x W^T
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int j = blockIdx.x, mt = blockIdx.y;
    const int KS = K_ >> 4, C = K_ >> 5;
    const int cpw = (C + WAVES - 1) / WAVES;
    const int c0 = wave * cpw;
    int nc = C - c0;
    nc = nc > cpw ? cpw : nc;
    nc = nc < 0 ? 0 : nc;
    bool wvalid[CT];
    const u32x4* wbase[CT];
#pragma unroll
    for (int t = 0; t < CT; ++t) {
        const int n = j * cpb_ + 16 * t + r;
        wvalid[t] = 16 * t + r < cpb_ && n < N_;
        const int nn = wvalid[t] ? n : 0;
        wbase[t] = reinterpret_cast<const u32x4*>(Wp_) + ((size_t)(nn >> 5) * KS + (g >> 1)) * 64 + (nn & 31) + 32 * (g & 1) + (size_t)c0 * 128;
    }
    const u32x4* xbase = reinterpret_cast<const u32x4*>(xp_) +
                         ((size_t)mt * KS + (g >> 1)) * 64 + r + 32 * (g & 1) + (size_t)c0 * 128;

    f32x4 acc[CT][2];
#pragma unroll
    for (int t = 0; t < CT; ++t) { acc[t][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[t][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    struct Grp { u32x4 w[CT][G]; u32x4 x[G][2]; };
    Grp ga, gb;
    auto load = [&](Grp& q, int ci) {
#pragma unroll
        for (int u = 0; u < G; ++u) {
            if (ci + u < nc) {                                   // wave-uniform
#pragma unroll
                for (int t = 0; t < CT; ++t) {
                    q.w[t][u] = zero4;
                    if (wvalid[t]) q.w[t][u] = __builtin_nontemporal_load(wbase[t] + (size_t)(ci + u) * 128);
                }
                q.x[u][0] = xbase[(size_t)(ci + u) * 128];
                q.x[u][1] = xbase[(size_t)(ci + u) * 128 + 16];
            }
        }
    };
    auto compute = [&](Grp& q, int ci) {
#pragma unroll
        for (int u = 0; u < G; ++u) {
            if (ci + u < nc) {
#pragma unroll
                for (int t = 0; t < CT; ++t) {
                    acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag4(q.w[t][u]), as_frag4(q.x[u][0]), acc[t][0], 0, 0, 0);
                    acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag4(q.w[t][u]), as_frag4(q.x[u][1]), acc[t][1], 0, 0, 0);
                }
            }
        }
    };
    load(ga, 0);
    if (G < nc) load(gb, G);

    // epilogue operands (they depend on nothing).  cpb a multiple of 8: thread t < 32 * cpb / 8 finishes 8 consecutive columns of
    // one row = one 16-byte piece of the fragment-order residual stream -> 16-byte loads / stores; other cpb: one thread per element.
    const bool vec = (cpb_ & 7) == 0 && (N_ & 7) == 0;
    const int v_row = tid & 31, v_ch = tid >> 5;                       // vec: (row, 8-column chunk of the block)
    const int v_n = j * cpb_ + 8 * v_ch;
    const bool v_on = vec && tid < 4 * cpb_ && v_n < N_;               // 32 * (cpb / 8) threads
    uint4 v_res = make_uint4(0u, 0u, 0u, 0u), v_bias = make_uint4(0u, 0u, 0u, 0u);
    size_t v_idx = 0;
    const int e_row = tid / cpb_, e_cr = tid - e_row * cpb_;
    const int e_n = j * cpb_ + e_cr;
    const bool e_on = !vec && tid < 32 * cpb_ && e_n < N_;
    float e_bias = 0.f, e_res = 0.f;
    size_t e_idx = 0;
    if (v_on) {
        if (bias_) v_bias = *reinterpret_cast<const uint4*>(bias_ + v_n);
        v_idx = xp_index(mt, out_KS_, v_row, v_n);
        v_res = *reinterpret_cast<const uint4*>(h_xp_ + v_idx);
    }
    if (e_on) {
        if (bias_) e_bias = bf2f(bias_[e_n]);
        e_idx = xp_index(mt, out_KS_, e_row, e_n);
        e_res = bf2f(h_xp_[e_idx]);
    }
    for (int ci = 0; ci < nc; ci += 2 * G) {
        compute(ga, ci);
        if (ci + 2 * G < nc) load(ga, ci + 2 * G);
        if (ci + G < nc) compute(gb, ci + G);
        if (ci + 3 * G < nc) load(gb, ci + 3 * G);
    }

    if (mt == 0) {
        // ColsArgs::poison: the activation buffer of the launch BEHIND this one (mlp_fused_kernel) gets its "not written yet" pattern,
        // every block its share; the arguments are read late, off the critical path
        const ColsArgs pl = sv_late_args<ColsArgs>(offsetof(ColsKernarg, p));
        if (pl.poison) {
            const unsigned share = ((pl.poison_bytes / 16 + gridDim.x - 1) / gridDim.x) * 16;
            const unsigned lo = j * share, hi = min(lo + share, pl.poison_bytes);
            const u32x4 ff = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
            for (unsigned off = lo + tid * 16; off < hi; off += WAVES * 64 * 16)
                *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(pl.poison) + off) = ff;
        }
        // ColsArgs::poison2: the polled LayerNorm-output buffer of the NEXT layer's fused row-update + c_attn launch -- touched with write-through (sc1)
        // stores only, like the attention launch's and the lm_head launch's pattern stores (rowops.hip, rowln_cattn_kernel): no XCD's L2 may keep a line
        // of it that memory does not have
        if (pl.poison2) {
            const __amdgpu_buffer_rsrc_t rsp = __builtin_amdgcn_make_buffer_rsrc(pl.poison2, 0, pl.poison2_bytes, 0x00020000);
            const unsigned share = ((pl.poison2_bytes / 16 + gridDim.x - 1) / gridDim.x) * 16;
            const unsigned lo = j * share, hi = min(lo + share, pl.poison2_bytes);
            for (unsigned off = lo + tid * 16; off < hi; off += WAVES * 64 * 16)
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}, rsp, (int)off, 0, 16);      // sc1
        }
    }
    // ---- K reduction across the waves (wave order), then one thread per output element / 8-column piece ----
    {
        float* my = red + wave * (CT * 512);
#pragma unroll
        for (int t = 0; t < CT; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                my[t * 512 + q * 64 + lane] = acc[t][0][q];
                my[t * 512 + (4 + q) * 64 + lane] = acc[t][1][q];
            }
    }
    __syncthreads();
    for (int idx = tid; idx < CT * 512; idx += WAVES * 64) {
        float s = red[idx];
#pragma unroll
        for (int w = 1; w < WAVES; ++w) s += red[w * (CT * 512) + idx];
        const int t = idx >> 9, hb = (idx >> 8) & 1, q = (idx >> 6) & 3, ln = idx & 63;
        tile[(16 * hb + (ln & 15)) * LDT + 16 * t + 4 * (ln >> 4) + q] = s;
    }
    __syncthreads();
    if (vec) {
        if (v_on) {
            float rr[8], bb[8], hn[8];
            unpack8(v_res, rr);
            unpack8(v_bias, bb);
#pragma unroll
            for (int e = 0; e < 8; ++e) hn[e] = bfround(rr[e] + bfround(tile[v_row * LDT + 8 * v_ch + e] + bb[e]));      // h = bf(h + bf(x W^T + b))
            *reinterpret_cast<uint4*>(h_xp_ + v_idx) = pack8(hn);
        }
        return;
    }
    if (e_on) h_xp_[e_idx] = f2bf(bfround(e_res + bfround(tile[e_row * LDT + e_cr] + e_bias)));
}

static size_t cols_smem(int waves, int ct) { return (size_t)waves * ct * 512 * 4 + (32 * (16 * ct + 1) + 16) * 4 + 64; }

// Every block re-reads its row tile's whole activation operand (32 x K bf16) from L2 next to cpb x K weights from HBM, so narrow
// blocks multiply L2 traffic: 4 columns per block at K = 4608 is 340 MB of L2 reads for 42 MB of weights (measured: +18 us per
// layer on StarVector-8B, profiles/fold6_r03_8b_ab.log).  K <= 2048: blocks = ceil(N / cpb) closest to a multiple of the 256 CUs
// (the activations are 128 KiB, the fill matters more), cpb a power of two <= 16; above that the (cpb <= 32) with the least bytes
// through the busiest CU: rounds of 256 blocks x (cpb + 32 rows) -- N = 4608: 18 columns = exactly 256 blocks.
int cols_pick_cpb(int N, int K) {
    if (K > 2048) {
        int best = 16;
        long best_cost = -1;
        for (int cpb = 8; cpb <= 32; ++cpb) {
            const long nb = (N + cpb - 1) / cpb, rounds = (nb + 255) / 256;
            const long cost = rounds * (cpb + 32);
            if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = cpb; }
        }
        return best;
    }
    int best = 8;
    double best_fill = 0.0;
    for (int cpb = 16; cpb >= 4; cpb >>= 1) {
        const int nb = (N + cpb - 1) / cpb;
        const int rounds = (nb + 255) / 256;
        const double fill = (double)nb / (rounds * 256.0);
        if (fill > best_fill + 1e-9) { best_fill = fill; best = cpb; }
    }
    return best;
}

int launch_gemm_cols(const ColsArgs& a, hipStream_t st) {
    if (a.K % 32 || a.cpb < 1 || a.cpb > 32 || !a.h_xp) return -1;
    dim3 grid((a.N + a.cpb - 1) / a.cpb, a.MT);
    if (a.cpb > 16)
        gemm_cols_resid_kernel<16, 2, 2><<<grid, 16 * 64, cols_smem(16, 2), st>>>(a.Wp, a.xp, a.h_xp, a.bias, a.K, a.N, a.cpb, a.out_KS, a);
    else
        gemm_cols_resid_kernel<16, 2, 1><<<grid, 16 * 64, cols_smem(16, 1), st>>>(a.Wp, a.xp, a.h_xp, a.bias, a.K, a.N, a.cpb, a.out_KS, a);
    return 0;
}

int init_cols_kernels() {
    int r = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_cols_resid_kernel<16, 2, 1>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)cols_smem(16, 1));
    if (!r) r = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_cols_resid_kernel<16, 2, 2>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)cols_smem(16, 2));
    return r;
}

// ------------------------------------------------------------------------------------------------
// LayerNorm fold of a Linear (see the header): W' = bf16(W * gamma) in the same fragment order, c1[n] = sum_k W'[n][k],
// c2[n] = sum_k beta[k] * W[n][k] + bias[n].  One wave per 32-column tile; lane (n = l & 31, half = l >> 5) walks its 8 k of every
// k-step in ascending order, the two halves are combined at the end (fixed order: deterministic).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void fold_prepare_kernel(const bf16_t* __restrict__ Wp, const bf16_t* __restrict__ gamma,
                                                          const bf16_t* __restrict__ beta, const bf16_t* __restrict__ bias,
                                                          bf16_t* __restrict__ Wf, float* __restrict__ c1, float* __restrict__ c2, int N, int K) {
    const int nt = blockIdx.x, lane = threadIdx.x;
    const int KS = K >> 4;
    const int n = nt * 32 + (lane & 31), half = lane >> 5;
    float s1 = 0.f, s2 = 0.f;
    for (int ks = 0; ks < KS; ++ks) {
        const size_t off = (((size_t)nt * KS + ks) * 64 + lane) * 8;
        const uint4 wq = *reinterpret_cast<const uint4*>(Wp + off);
        const int k0 = ks * 16 + half * 8;
        const uint4 gq = *reinterpret_cast<const uint4*>(gamma + k0);
        const uint4 bq = *reinterpret_cast<const uint4*>(beta + k0);
        float w[8], gg[8], bb[8], o[8];
        unpack8(wq, w); unpack8(gq, gg); unpack8(bq, bb);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            o[e] = bfround(w[e] * gg[e]);
            s1 += o[e];
            s2 += bb[e] * w[e];
        }
        *reinterpret_cast<uint4*>(Wf + off) = pack8(o);
    }
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 32, 64);
    if (half == 0) {
        c1[n] = n < N ? s1 : 0.f;
        c2[n] = n < N ? s2 + (bias ? bf2f(bias[n]) : 0.f) : 0.f;
    }
}
void launch_fold_prepare(const bf16_t* Wp, const bf16_t* gamma, const bf16_t* beta, const bf16_t* bias, bf16_t* Wf, float* c1, float* c2,
                         int N, int Npad, int K, hipStream_t st) {
    fold_prepare_kernel<<<Npad / 32, 64, 0, st>>>(Wp, gamma, beta, bias, Wf, c1, c2, N, K);
}

}  // namespace sv
