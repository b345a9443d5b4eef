// Decode-step weight-streaming GEMMs without split-K hand-offs (gfx950).
//
// A decode step multiplies <= 32 activation rows by every decoder weight once; the step is a chain of
// all-to-all seams (every output needs the whole K of every input row), and each seam costs a dependent
// launch (~4-5 us on this part) whatever it computes.  Round 1 ran 7 launches per layer: the GEMMs whose
// output is only 2048-2304 columns wide were split 4-way along K across workgroups (256 CUs must stream to
// reach HBM bandwidth: one CU sustains ~25 GB/s), which left fp32 slabs behind, and a separate per-row kernel
// summed the slabs, added bias + residual and applied the next LayerNorm.  Here every GEMM block owns FULL K:
//
//   * gemm_cols_kernel: a block owns `cpb` <= 16 output columns (8-9 for the 2048/2304-wide projections ->
//     256 blocks) over the whole K, split across its waves.  v_mfma_f32_16x16x32_bf16 with A = the weight rows
//     of the block's columns (read straight from the existing 32-column fragment image: the lanes of one
//     column are 16-byte pieces of full 128-byte lines), B = the activation rows in two halves of 16.  No
//     cross-block reduction exists, so the epilogue finishes the op: bias -> row-major q|k|v (c_attn), or
//     bias + residual -> the residual stream (attention c_proj, MLP c_proj).
//   * Because a block reads ALL of its 32 activation rows anyway, it can compute their LayerNorm statistics
//     itself (two-pass, fp32, like the reference's nn.LayerNorm): the LayerNorm that used to be a launch of its
//     own is a prologue of the consumer (c_attn: ln_1; c_fc: ln_2; lm_head: ln_f), with no hand-off.
//   * gemm_skinny_ln_kernel: the 32-column-tile kernel (v_mfma_f32_32x32x16_bf16) for the wide outputs
//     (c_fc: 8192 columns, lm_head: 49156), with the same in-block LayerNorm prologue.
//
// Decode step = embed + 24 x (c_attn[LN1] . attention . c_proj[+res] . c_fc[LN2,GELU] . c_proj[+res]) + lm_head[ln_f]
//             + argmax + finish = 124 launches (slab pipeline: 172), no slabs, no tickets, bitwise deterministic.
//
// MEASURED (MI355X, StarVector-1B, batch 32; profiles/decode_gemm_r02_fullk_vs_slabs.log, bench_r02_fullk_pipeline_n1.json):
// parity-green (the whole GPU suite passes on it) but SLOWER than the slab pipeline, 1453 vs 1336 us per decode step, so it is
// opt-in (SV_DECODE_PIPE=cols) and the slab pipeline stays the default.  Why: (1) the LayerNorm arithmetic repeated in EVERY
// block is VALU-bound -- 32 rows x K elements x ~11 lane-ops ~ 2.5 us per block (c_attn 4.3 -> 10.4 us, c_fc 6.5 -> 11.1 us,
// lm_head 38 -> 66 us with six blocks per CU), more than the 5.4 us launch of the per-row kernel it replaces saves; (2) a
// full-K block of the K = 8192 down projection pulls all 512 KB of activations through its CU's load path (12.0 vs 6.7 us).
// The 48 launches saved are worth ~125 us per step, the slower GEMMs cost ~380 us.
// Reference arithmetic: gpt_bigcode/modeling_gpt_bigcode.py:694-755 (block), :645-660 (MLP), :1114,1258 (ln_f, lm_head).
#include "kernels.h"

namespace sv {

__device__ __forceinline__ uint32_t dg_cvt_pk_bf16(float lo, float hi) { return pack2bf(lo, hi); }   // common.h (never inline asm: hazards)

// 8 bf16 in a 16-byte register group -> sum, or sum of squared deviations
__device__ __forceinline__ float dg_sum8(const u32x4& v) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) s += __uint_as_float(v[w] << 16) + __uint_as_float(v[w] & 0xffff0000u);
    return s;
}
__device__ __forceinline__ float dg_sqdev8(const u32x4& v, float mean) {
    float q = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const float a = __uint_as_float(v[w] << 16) - mean, b = __uint_as_float(v[w] & 0xffff0000u) - mean;
        q += a * a + b * b;
    }
    return q;
}
// y = (x - mean) * rstd * gamma + beta, rounded to bf16 (the reference's LayerNorm output dtype)
__device__ __forceinline__ u32x4 dg_normalize8(const u32x4& x, float mean, float rstd, const u32x4& g, const u32x4& b) {
    u32x4 y;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const float x0 = __uint_as_float(x[w] << 16), x1 = __uint_as_float(x[w] & 0xffff0000u);
        const float g0 = __uint_as_float(g[w] << 16), g1 = __uint_as_float(g[w] & 0xffff0000u);
        const float b0 = __uint_as_float(b[w] << 16), b1 = __uint_as_float(b[w] & 0xffff0000u);
        y[w] = dg_cvt_pk_bf16((x0 - mean) * rstd * g0 + b0, (x1 - mean) * rstd * g1 + b1);
    }
    return y;
}

// ------------------------------------------------------------------------------------------------
// gemm_cols_kernel<WAVES, G, LN, MAXC>
//   grid (ceil(N / cpb), MT), block WAVES * 64.  K is cut into chunks of 32 (one MFMA k-step); wave w owns the
//   chunks [w * cpw, (w + 1) * cpw).  Lane (r = l & 15, g = l >> 4):
//     A (weights):      W[j * cpb + r][32 c + 8 g .. + 8]   (lanes with r >= cpb load nothing and hold zero)
//     B (activations):  x[16 hb + r][32 c + 8 g .. + 8],  hb = 0, 1
//     D:                out[col 4 g + reg][row 16 hb + r] in acc[hb][reg]
//   LN = false: W and x stream in groups of G chunks, two groups in flight.
//   LN = true : the wave's x chunks (<= MAXC) stay in registers: row sums -> LDS -> mean, squared deviations -> LDS
//               -> rstd (two block barriers), normalise in place with gamma / beta staged in LDS, then the same MFMAs.
// ------------------------------------------------------------------------------------------------
template <int WAVES, int G, bool LN, int MAXC>
__global__ __launch_bounds__(WAVES * 64) void gemm_cols_kernel(ColsArgs p) {
    extern __shared__ __attribute__((aligned(16))) char dg_smem[];
    float* red = reinterpret_cast<float*>(dg_smem);                     // [WAVES][512]
    float* tile = red + WAVES * 512;                                     // [32][17]
    float* st_s = tile + 32 * 17 + 16;                                   // [2][WAVES][32]   (LN)
    bf16_t* gb_s = reinterpret_cast<bf16_t*>(st_s + 2 * WAVES * 32);     // gamma[K] | beta[K] (LN)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int j = blockIdx.x, mt = blockIdx.y;
    const int KS = p.K >> 4, C = p.K >> 5;
    const int cpw = (C + WAVES - 1) / WAVES;
    const int c0 = wave * cpw;
    int nc = C - c0;
    nc = nc > cpw ? cpw : nc;
    nc = nc < 0 ? 0 : nc;
    const int n = j * p.cpb + r;
    const bool wvalid = r < p.cpb && n < p.N;
    const u32x4* wbase = reinterpret_cast<const u32x4*>(p.Wp) +
                         ((size_t)(n >> 5) * KS + (g >> 1)) * 64 + (n & 31) + 32 * (g & 1) + (size_t)c0 * 128;
    const u32x4* xbase = reinterpret_cast<const u32x4*>(p.xp) +
                         ((size_t)mt * KS + (g >> 1)) * 64 + r + 32 * (g & 1) + (size_t)c0 * 128;

    // epilogue operands are requested first (they depend on nothing): thread e finishes (row, column) = (e / cpb, e % cpb)
    const int e_row = tid / p.cpb, e_cr = tid - e_row * p.cpb;
    const int e_n = j * p.cpb + e_cr;
    const bool e_on = tid < 32 * p.cpb && e_n < p.N;
    float e_bias = 0.f, e_res = 0.f;
    size_t e_idx = 0;
    if (e_on) {
        if (p.bias) e_bias = bf2f(p.bias[e_n]);
        if (p.out_mode == CO_RESID_XP) {
            e_idx = xp_index(mt, p.out_KS, e_row, e_n);
            e_res = bf2f(p.h_xp[e_idx]);
        }
    }

    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    const u32x4 zero4 = {0u, 0u, 0u, 0u};

    if constexpr (LN) {
        // ---- everything up front: the wave's activation chunks, the first weight chunks, gamma / beta -> LDS ----
        constexpr int WG = MAXC < 6 ? MAXC : 6;             // weight chunks in flight per wave
        u32x4 xr[MAXC][2];
        u32x4 wr[WG];
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
            if (c < nc) {
                xr[c][0] = xbase[(size_t)c * 128];
                xr[c][1] = xbase[(size_t)c * 128 + 16];
            }
#pragma unroll
        for (int c = 0; c < WG; ++c) {
            wr[c] = zero4;
            if (c < nc && wvalid) wr[c] = __builtin_nontemporal_load(wbase + (size_t)c * 128);
        }
        for (int i = tid; i < (p.K >> 2); i += WAVES * 64) {          // K/8 16-byte pieces of gamma, then of beta
            const int half = i >= (p.K >> 3);
            const int cc = half ? i - (p.K >> 3) : i;
            *reinterpret_cast<uint4*>(gb_s + (size_t)half * p.K + cc * 8) =
                *reinterpret_cast<const uint4*>((half ? p.ln_b : p.ln_g) + cc * 8);
        }
        // ---- row statistics, two-pass (nn.LayerNorm): lane (r, g) holds 8 features of rows r and 16 + r per chunk ----
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
            if (c < nc) { s0 += dg_sum8(xr[c][0]); s1 += dg_sum8(xr[c][1]); }
        s0 += __shfl_xor(s0, 16, 64); s0 += __shfl_xor(s0, 32, 64);
        s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
        if (g == 0) { st_s[wave * 32 + r] = s0; st_s[wave * 32 + 16 + r] = s1; }
        // (opaque to the compiler: otherwise it keeps the 8 unpacked floats of every chunk alive across the passes,
        //  three times the registers of the packed form)
#pragma unroll
        for (int c = 0; c < MAXC; ++c) asm volatile("" : "+v"(xr[c][0]), "+v"(xr[c][1]));
        __syncthreads();
        float m0 = 0.f, m1 = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) { m0 += st_s[w * 32 + r]; m1 += st_s[w * 32 + 16 + r]; }
        const float invK = 1.0f / (float)p.K;
        m0 *= invK; m1 *= invK;
        float q0 = 0.f, q1 = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
            if (c < nc) { q0 += dg_sqdev8(xr[c][0], m0); q1 += dg_sqdev8(xr[c][1], m1); }
        q0 += __shfl_xor(q0, 16, 64); q0 += __shfl_xor(q0, 32, 64);
        q1 += __shfl_xor(q1, 16, 64); q1 += __shfl_xor(q1, 32, 64);
        float* st_q = st_s + WAVES * 32;
        if (g == 0) { st_q[wave * 32 + r] = q0; st_q[wave * 32 + 16 + r] = q1; }
#pragma unroll
        for (int c = 0; c < MAXC; ++c) asm volatile("" : "+v"(xr[c][0]), "+v"(xr[c][1]));
        __syncthreads();
        float v0 = 0.f, v1 = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) { v0 += st_q[w * 32 + r]; v1 += st_q[w * 32 + 16 + r]; }
        const float rs0 = rsqrtf(v0 * invK + p.ln_eps), rs1 = rsqrtf(v1 * invK + p.ln_eps);
        // ---- normalise + MFMA; the weight ring keeps WG chunks in flight.  The scheduling fence after every chunk keeps
        //      the gamma / beta reads of later chunks from being hoisted (they would cost 8 VGPRs per chunk) ----
#pragma unroll
        for (int c = 0; c < MAXC; ++c) {
            if (c < nc) {
                const int k0 = (c0 + c) * 32 + g * 8;
                const u32x4 gv = *reinterpret_cast<const u32x4*>(gb_s + k0);
                const u32x4 bv = *reinterpret_cast<const u32x4*>(gb_s + p.K + k0);
                const u32x4 y0 = dg_normalize8(xr[c][0], m0, rs0, gv, bv);
                const u32x4 y1 = dg_normalize8(xr[c][1], m1, rs1, gv, bv);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag4(wr[c % WG]), as_frag4(y0), acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag4(wr[c % WG]), as_frag4(y1), acc1, 0, 0, 0);
                if (c + WG < MAXC && c + WG < nc) {
                    wr[c % WG] = zero4;
                    if (wvalid) wr[c % WG] = __builtin_nontemporal_load(wbase + (size_t)(c + WG) * 128);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        struct Grp { u32x4 w[G]; u32x4 x[G][2]; };
        Grp ga, gb;
        auto load = [&](Grp& q, int ci) {
#pragma unroll
            for (int u = 0; u < G; ++u) {
                if (ci + u < nc) {                                   // wave-uniform
                    q.w[u] = zero4;
                    if (wvalid) q.w[u] = __builtin_nontemporal_load(wbase + (size_t)(ci + u) * 128);
                    q.x[u][0] = xbase[(size_t)(ci + u) * 128];
                    q.x[u][1] = xbase[(size_t)(ci + u) * 128 + 16];
                }
            }
        };
        auto compute = [&](Grp& q, int ci) {
#pragma unroll
            for (int u = 0; u < G; ++u) {
                if (ci + u < nc) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag4(q.w[u]), as_frag4(q.x[u][0]), acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag4(q.w[u]), as_frag4(q.x[u][1]), acc1, 0, 0, 0);
                }
            }
        };
        load(ga, 0);
        if (G < nc) load(gb, G);
        for (int ci = 0; ci < nc; ci += 2 * G) {
            compute(ga, ci);
            if (ci + 2 * G < nc) load(ga, ci + 2 * G);
            if (ci + G < nc) compute(gb, ci + G);
            if (ci + 3 * G < nc) load(gb, ci + 3 * G);
        }
    }

    // ---- K reduction across the waves (wave order), then one thread per output element ----
    {
        float* my = red + wave * 512;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            my[q * 64 + lane] = acc0[q];
            my[(4 + q) * 64 + lane] = acc1[q];
        }
    }
    __syncthreads();
    if (tid < 512) {
        float s = red[tid];
#pragma unroll
        for (int w = 1; w < WAVES; ++w) s += red[w * 512 + tid];
        const int hb = tid >> 8, q = (tid >> 6) & 3, ln = tid & 63;
        tile[(16 * hb + (ln & 15)) * 17 + 4 * (ln >> 4) + q] = s;
    }
    __syncthreads();
    if (!e_on) return;
    const float v = tile[e_row * 17 + e_cr] + e_bias;
    if (p.out_mode == CO_ROWMAJOR) {
        p.out_rm[((size_t)mt * 32 + e_row) * p.ld_rm + e_n] = f2bf(v);
    } else if (p.out_mode == CO_RESID_XP) {
        p.h_xp[e_idx] = f2bf(e_res + bfround(v));                   // h = bf(h + bf(x W^T + b))
    } else {
        p.out_f32[((size_t)mt * 32 + e_row) * p.ldo + e_n] = v;
    }
}

static size_t cols_smem(int waves, int K, bool ln) {
    return (size_t)waves * 512 * 4 + (32 * 17 + 16) * 4 + (size_t)2 * waves * 32 * 4 + (ln ? (size_t)K * 4 : 0) + 64;
}

int cols_pick_cpb(int N) {
    // blocks = ceil(N / cpb) close to a multiple of 256 CUs with cpb <= 16
    const int rounds = (N + 256 * 16 - 1) / (256 * 16);
    int cpb = (N + 256 * rounds - 1) / (256 * rounds);
    if (cpb < 1) cpb = 1;
    if (cpb > 16) cpb = 16;
    return cpb;
}

template <int WAVES, int G, bool LN, int MAXC>
static void launch_cols_t(const ColsArgs& a, hipStream_t st) {
    dim3 grid((a.N + a.cpb - 1) / a.cpb, a.MT);
    gemm_cols_kernel<WAVES, G, LN, MAXC><<<grid, WAVES * 64, cols_smem(WAVES, a.K, LN), st>>>(a);
}

int launch_gemm_cols(const ColsArgs& a, hipStream_t st) {
    if (a.K % 32 || a.cpb < 1 || a.cpb > 16) return -1;
    const int C = a.K / 32;
    static const int force_w = getenv("SV_COLS_WAVES") ? atoi(getenv("SV_COLS_WAVES")) : 0;
    if (a.ln_g) {
        // the wave's activation chunks stay in registers: 16 waves up to K = 2048 (4 chunks each), 8 waves up to K = 5120.
        // OPEN ISSUE (profiles/cols_ln_first_launch_r02.log, tools/diag/test_diag_cols3.py): on about half of the MI355X parts
        // of the pool the first launch of this LayerNorm-prologue GEMM on NEW data, in a process that has run other kernels,
        // returns rows 16..31 of a few column blocks 2e-3 .. 8e-3 off; the next launch on the same data is exact.  Same with
        // 8 or 16 waves, with launches serialised, with the inputs resident and check-summed first, with the statistics
        // buffers zeroed or doubly fenced -- the cause is not found.  The pipeline that uses it is opt-in (SV_DECODE_PIPE=cols,
        // measured slower than the default), so the default path is not affected.
        if ((C + 15) / 16 <= 4 && force_w != 8) launch_cols_t<16, 2, true, 4>(a, st);
        else if ((C + 7) / 8 <= 8) launch_cols_t<8, 2, true, 8>(a, st);
        else if ((C + 7) / 8 <= 20) launch_cols_t<8, 2, true, 20>(a, st);
        else return -1;
        return 0;
    }
    if (force_w == 8) launch_cols_t<8, 4, false, 1>(a, st);
    else launch_cols_t<16, 2, false, 1>(a, st);
    return 0;
}

// ------------------------------------------------------------------------------------------------
// gemm_skinny_ln_kernel<WAVES, MAXKS>: y[32][N] = act(LN(h) . W^T + bias) on 32-column tiles, full K per block.
//   grid (Npad / 32, MT), block WAVES * 64; wave w owns k-steps [w * ksw, (w + 1) * ksw), ksw = K / 16 / WAVES <= MAXKS.
//   Lane (m = l & 31, half = l >> 5): A = W[nt * 32 + m][16 ks + 8 half ..] (one contiguous 1 KiB per wave load),
//   B = h[mt * 32 + m][16 ks + 8 half ..].  LayerNorm prologue as in gemm_cols_kernel (statistics of all 32 rows are
//   complete inside the block).  Epilogue shared by all waves: bias + activation -> fragment-order bf16 (c_fc), or
//   fp32 logits rounded to bf16 values (lm_head; HF casts bf16 logits to float32 before argmax).
// ------------------------------------------------------------------------------------------------
template <int WAVES, int MAXKS, bool EXACT>
__global__ __launch_bounds__(WAVES * 64) void gemm_skinny_ln_kernel(SkinnyLnArgs p) {
    constexpr bool LN = true;
    extern __shared__ __attribute__((aligned(16))) char dg_smem[];
    float (*red)[16][64] = reinterpret_cast<float (*)[16][64]>(dg_smem);               // [WAVES][16][64]
    float* st_s = reinterpret_cast<float*>(dg_smem + (size_t)WAVES * 16 * 64 * 4);      // [2][WAVES][32]
    bf16_t* gb_s = reinterpret_cast<bf16_t*>(st_s + 2 * WAVES * 32);                    // gamma[K] | beta[K]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nt = blockIdx.x, mt = blockIdx.y;
    const int KS = p.K >> 4;
    const int ksw = EXACT ? MAXKS : KS / WAVES;       // EXACT: the trip counts are compile-time constants (no guards)
    const int ks0 = wave * ksw;
    const int m = lane & 31, half = lane >> 5;
    const u32x4* wptr = reinterpret_cast<const u32x4*>(p.Wp) + ((size_t)nt * KS + ks0) * 64 + lane;
    const u32x4* xptr = reinterpret_cast<const u32x4*>(p.xp) + ((size_t)mt * KS + ks0) * 64 + lane;

    constexpr int WG = MAXKS < 8 ? MAXKS : 8;            // weight k-steps in flight per wave (8 KiB)
    u32x4 xr[MAXKS];
    u32x4 wr[WG];
#pragma unroll
    for (int c = 0; c < MAXKS; ++c)
        if (c < ksw) xr[c] = xptr[(size_t)c * 64];
#pragma unroll
    for (int c = 0; c < WG; ++c)
        if (c < ksw) wr[c] = __builtin_nontemporal_load(wptr + (size_t)c * 64);
    constexpr int RPW = 16 / WAVES;                      // accumulator rows finished per wave (WAVES in {2, 4, 8, 16})
    const int r0 = wave * RPW;
    const int n0 = nt * 32 + 8 * (r0 >> 2) + 4 * half + (r0 & 3);
    // accumulator row r0 + i <-> column n0 + i (+4 when i >= 4: rows r and r + 4 sit 8 columns apart)
    float bias_d[RPW];
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const int col = n0 + i + (i >> 2) * 4;
        bias_d[i] = 0.f;
        if (p.out_mode == SK_OUT_PACKED_ACT && p.bias && col < p.N) bias_d[i] = bf2f(p.bias[col]);
    }
    float mean = 0.f, rstd = 1.f;
    if constexpr (LN) {
        for (int i = tid; i < (p.K >> 2); i += WAVES * 64) {
            const int hf = i >= (p.K >> 3);
            const int cc = hf ? i - (p.K >> 3) : i;
            *reinterpret_cast<uint4*>(gb_s + (size_t)hf * p.K + cc * 8) =
                *reinterpret_cast<const uint4*>((hf ? p.ln_b : p.ln_g) + cc * 8);
        }
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < MAXKS; ++c)
            if (c < ksw) s += dg_sum8(xr[c]);
        s += __shfl_xor(s, 32, 64);
        if (half == 0) st_s[wave * 32 + m] = s;
#pragma unroll
        for (int c = 0; c < MAXKS; ++c)
            if (c < ksw) asm volatile("" : "+v"(xr[c]));      // keep the packed form only (see gemm_cols_kernel)
        __syncthreads();
#pragma unroll
        for (int w = 0; w < WAVES; ++w) mean += st_s[w * 32 + m];
        const float invK = 1.0f / (float)p.K;
        mean *= invK;
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < MAXKS; ++c)
            if (c < ksw) q += dg_sqdev8(xr[c], mean);
        q += __shfl_xor(q, 32, 64);
        float* st_q = st_s + WAVES * 32;
        if (half == 0) st_q[wave * 32 + m] = q;
#pragma unroll
        for (int c = 0; c < MAXKS; ++c)
            if (c < ksw) asm volatile("" : "+v"(xr[c]));
        __syncthreads();
        float var = 0.f;
#pragma unroll
        for (int w = 0; w < WAVES; ++w) var += st_q[w * 32 + m];
        rstd = rsqrtf(var * invK + p.ln_eps);
    }

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // normalise + MFMA, two k-steps per scheduling region (the fence keeps the gamma / beta reads of later k-steps from
    // being hoisted: 8 VGPRs each); the weight ring keeps WG k-steps in flight
#pragma unroll
    for (int c = 0; c < MAXKS; c += 2) {
        if (c < ksw) {
            u32x4 y0 = xr[c], y1 = xr[c + 1 < MAXKS ? c + 1 : c];
            const bool two = c + 1 < MAXKS && c + 1 < ksw;
            if constexpr (LN) {
                const int k0 = (ks0 + c) * 16 + half * 8;
                y0 = dg_normalize8(y0, mean, rstd, *reinterpret_cast<const u32x4*>(gb_s + k0),
                                   *reinterpret_cast<const u32x4*>(gb_s + p.K + k0));
                if (two)
                    y1 = dg_normalize8(y1, mean, rstd, *reinterpret_cast<const u32x4*>(gb_s + k0 + 16),
                                       *reinterpret_cast<const u32x4*>(gb_s + p.K + k0 + 16));
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag4(wr[c % WG]), as_frag4(y0), acc, 0, 0, 0);
            if (c + WG < MAXKS && c + WG < ksw) wr[c % WG] = __builtin_nontemporal_load(wptr + (size_t)(c + WG) * 64);
            if (two) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag4(wr[(c + 1) % WG]), as_frag4(y1), acc, 0, 0, 0);
                if (c + 1 + WG < MAXKS && c + 1 + WG < ksw)
                    wr[(c + 1) % WG] = __builtin_nontemporal_load(wptr + (size_t)(c + 1 + WG) * 64);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }

    // ---- K reduction across the waves (wave order); every wave finishes RPW accumulator rows ----
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
    __syncthreads();
    float v[RPW];
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        float t = red[0][r0 + i][lane];
#pragma unroll
        for (int w = 1; w < WAVES; ++w) t += red[w][r0 + i][lane];
        v[i] = t;
    }
    if (p.out_mode == SK_OUT_F32) {
        float* dst = p.out_f32 + ((size_t)mt * 32 + m) * p.ldo + n0;
#pragma unroll
        for (int i = 0; i < RPW; ++i) dst[i + (i >> 2) * 4] = p.round_bf16 ? bfround(v[i]) : v[i];
    } else {       // SK_OUT_PACKED_ACT
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            float x = 0.f;
            if (n0 + i + (i >> 2) * 4 < p.N) {
                x = bfround(v[i] + bias_d[i]);
                if (p.act != ACT_NONE) x = sv_act(x, p.act);
            }
            v[i] = x;
        }
        bf16_t* dst = p.out_xp + xp_index(mt, p.out_KS, m, n0);
        if constexpr (RPW == 2) {
            *reinterpret_cast<uint32_t*>(dst) = pack2bf(v[0], v[1]);
        } else if constexpr (RPW == 4) {
            uint2 o; o.x = pack2bf(v[0], v[1]); o.y = pack2bf(v[2], v[3]);
            *reinterpret_cast<uint2*>(dst) = o;
        } else {
#pragma unroll
            for (int i = 0; i < RPW; ++i) p.out_xp[xp_index(mt, p.out_KS, m, n0 + i + (i >> 2) * 4)] = f2bf(v[i]);
        }
    }
}

static size_t skinny_ln_smem(int waves, int K) {
    return (size_t)waves * 16 * 64 * 4 + (size_t)2 * waves * 32 * 4 + (size_t)K * 4 + 64;
}

template <int WAVES, int MAXKS, bool EXACT>
static void launch_skinny_ln_t(const SkinnyLnArgs& a, hipStream_t st) {
    dim3 grid(a.Npad / 32, a.MT);
    gemm_skinny_ln_kernel<WAVES, MAXKS, EXACT><<<grid, WAVES * 64, skinny_ln_smem(WAVES, a.K), st>>>(a);
}

int launch_gemm_skinny_ln(const SkinnyLnArgs& a, hipStream_t st) {
    const int KS = a.K / 16;
    if (a.K % 16 || a.Npad % 32) return -1;
    if (a.out_mode != SK_OUT_F32 && a.out_mode != SK_OUT_PACKED_ACT) return -1;
    if (!a.ln_g || !a.ln_b) return -1;                 // the LayerNorm prologue is what this kernel is for
    if (KS == 8 * 16) launch_skinny_ln_t<8, 16, true>(a, st);              // K = 2048 (StarVector-1B)
    else if (KS == 8 * 36) launch_skinny_ln_t<8, 36, true>(a, st);         // K = 4608 (StarVector-8B)
    else if (KS % 8 == 0 && KS / 8 <= 8) launch_skinny_ln_t<8, 8, false>(a, st);
    else if (KS % 8 == 0 && KS / 8 <= 36) launch_skinny_ln_t<8, 36, false>(a, st);
    else if (KS % 2 == 0 && KS / 2 <= 8) launch_skinny_ln_t<2, 8, false>(a, st);
    else return -1;
    return 0;
}

int init_decode_gemm_kernels() {
    // dynamic LDS above 64 KiB needs an opt-in (K = 4608 / 8192 gamma-beta staging on top of the reduction buffers)
    int r = 0;
    auto set = [&](const void* f) {
        if (!r) r = (int)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    };
    set(reinterpret_cast<const void*>(&gemm_cols_kernel<16, 2, true, 4>));
    set(reinterpret_cast<const void*>(&gemm_cols_kernel<8, 2, true, 8>));
    set(reinterpret_cast<const void*>(&gemm_cols_kernel<8, 2, true, 20>));
    set(reinterpret_cast<const void*>(&gemm_cols_kernel<16, 2, false, 1>));
    set(reinterpret_cast<const void*>(&gemm_cols_kernel<8, 4, false, 1>));
    set(reinterpret_cast<const void*>(&gemm_skinny_ln_kernel<8, 16, true>));
    set(reinterpret_cast<const void*>(&gemm_skinny_ln_kernel<8, 36, true>));
    set(reinterpret_cast<const void*>(&gemm_skinny_ln_kernel<8, 8, false>));
    set(reinterpret_cast<const void*>(&gemm_skinny_ln_kernel<8, 36, false>));
    set(reinterpret_cast<const void*>(&gemm_skinny_ln_kernel<2, 8, false>));
    return r;
}

}  // namespace sv
