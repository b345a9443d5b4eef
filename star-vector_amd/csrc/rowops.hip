// Row-wise (HBM/L2-bound) kernels of the path: LayerNorm flavours, embeddings, im2col, adapter norms.
// All loads/stores are 16-byte bf16x8 vectors; statistics in fp32 with wave64 shuffle reductions.
#include <cstddef>
#include <cstdio>
#include <cstring>
#include "kernels.h"

namespace sv {

// ------------------------------------------------------------------------------------------------
// LayerNorm over the last dim, one wave per row (clip_model.py:117-124, gpt_bigcode LN eps 1e-5)
// PACKED=false: y[M][D] row-major.  PACKED=true: y in skinny "xp" fragment order.
// ------------------------------------------------------------------------------------------------
template <bool PACKED>
__global__ __launch_bounds__(256) void layernorm_rows_kernel(const bf16_t* __restrict__ x, int ldx,
                                                             const bf16_t* __restrict__ g,
                                                             const bf16_t* __restrict__ b,
                                                             bf16_t* __restrict__ y, int ldy, int M, int D,
                                                             float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const bf16_t* xr = x + (size_t)row * ldx;
    const int NC = D >> 3;
    const int KS = D >> 4;
    constexpr int LN_MAXC = 9;                 // 16-byte chunks per lane kept in registers: D <= 4608 reads the row ONCE
    if (NC <= 64 * LN_MAXC) {
        // the row, gamma and beta are requested up front; statistics and the output come from registers (the three-pass
        // version re-read the row twice: 25 us for the 8288 x 2048 prefill rows, this one is bandwidth-bound)
        uint4 xq[LN_MAXC];
#pragma unroll
        for (int i = 0; i < LN_MAXC; ++i) {
            const int c = lane + i * 64;
            if (c < NC) xq[i] = *reinterpret_cast<const uint4*>(xr + c * 8);
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXC; ++i)
            if (lane + i * 64 < NC) {
                float f[8];
                unpack8(xq[i], f);
#pragma unroll
                for (int e = 0; e < 8; ++e) s += f[e];
            }
        const float mean = wave_sum(s) / (float)D;
        // (opaque: otherwise the compiler keeps the 8 unpacked floats of every chunk alive across the passes -- 3x the registers)
#pragma unroll
        for (int i = 0; i < LN_MAXC; ++i) asm volatile("" : "+v"(xq[i].x), "+v"(xq[i].y), "+v"(xq[i].z), "+v"(xq[i].w));
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < LN_MAXC; ++i)
            if (lane + i * 64 < NC) {
                float f[8];
                unpack8(xq[i], f);
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = f[e] - mean; q += d * d; }
            }
        const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
        for (int i = 0; i < LN_MAXC; ++i) asm volatile("" : "+v"(xq[i].x), "+v"(xq[i].y), "+v"(xq[i].z), "+v"(xq[i].w));
#pragma unroll
        for (int i = 0; i < LN_MAXC; ++i) {
            const int c = lane + i * 64;
            if (c < NC) {
                float f[8], gg[8], bb[8];
                unpack8(xq[i], f);
                unpack8(*reinterpret_cast<const uint4*>(g + c * 8), gg);
                unpack8(*reinterpret_cast<const uint4*>(b + c * 8), bb);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = (f[e] - mean) * rstd * gg[e] + bb[e];
                if (PACKED)
                    *reinterpret_cast<uint4*>(y + xp_index(row >> 5, KS, row & 31, c * 8)) = pack8(f);
                else
                    *reinterpret_cast<uint4*>(y + (size_t)row * ldy + c * 8) = pack8(f);
            }
        }
        return;
    }
    float s = 0.f;
    for (int c = lane; c < NC; c += 64) {
        float f[8];
        unpack8(*reinterpret_cast<const uint4*>(xr + c * 8), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += f[e];
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
    for (int c = lane; c < NC; c += 64) {
        float f[8];
        unpack8(*reinterpret_cast<const uint4*>(xr + c * 8), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) { float d = f[e] - mean; q += d * d; }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
    for (int c = lane; c < NC; c += 64) {
        float f[8], gg[8], bb[8];
        unpack8(*reinterpret_cast<const uint4*>(xr + c * 8), f);
        unpack8(*reinterpret_cast<const uint4*>(g + c * 8), gg);
        unpack8(*reinterpret_cast<const uint4*>(b + c * 8), bb);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = (f[e] - mean) * rstd * gg[e] + bb[e];
        if (PACKED)
            *reinterpret_cast<uint4*>(y + xp_index(row >> 5, KS, row & 31, c * 8)) = pack8(f);
        else
            *reinterpret_cast<uint4*>(y + (size_t)row * ldy + c * 8) = pack8(f);
    }
}

void launch_layernorm_rows(const bf16_t* x, int ldx, const bf16_t* g, const bf16_t* b, bf16_t* y, int ldy,
                           int M, int D, float eps, hipStream_t st) {
    layernorm_rows_kernel<false><<<(M + 3) / 4, 256, 0, st>>>(x, ldx, g, b, y, ldy, M, D, eps);
}
void launch_layernorm_rows_packed(const bf16_t* x, int ldx, const bf16_t* g, const bf16_t* b, bf16_t* yp,
                                  int M, int D, float eps, hipStream_t st) {
    layernorm_rows_kernel<true><<<(M + 3) / 4, 256, 0, st>>>(x, ldx, g, b, yp, 0, M, D, eps);
}

// ------------------------------------------------------------------------------------------------
// decode row update + LayerNorm (one block per row):
//   embedding mode : h = bf(wte[tok] + wpe[pos])                           (gpt_bigcode :1060-1063)
//   residual mode  : h = bf(h + bf(sum_s ws[s][row][:] + bias))            (block residual adds)
//   then           : xp = LN(h)   written in fragment order for the next skinny GEMM
// the split-K slabs are summed in slab order -> bitwise deterministic.
// ------------------------------------------------------------------------------------------------
// (leading scalar parameters: preloaded into SGPRs with the dispatch, see gemm_skinny_kernel)
struct RowUpdateKernarg { const float* ws; const bf16_t* bias; bf16_t* h; const bf16_t* g; const bf16_t* b;
                          int splitk, ldws, rows_ws, ldh, D, M; RowUpdateArgs p; };               // the kernarg segment
template <int THREADS>
__global__ __launch_bounds__(THREADS) void row_update_ln_kernel(const float* ws_, const bf16_t* bias_, bf16_t* h_, const bf16_t* g_,
                                                            const bf16_t* b_, int splitk_, int ldws_, int rows_ws_, int ldh_, int D_,
                                                            int M_, RowUpdateArgs p_unused) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* hrow = reinterpret_cast<float*>(smem_raw);          // [D]
    constexpr int NW = THREADS / 64;
    __shared__ float redbuf[2 * NW];
    const int row = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int D = D_, NC = D >> 3;
    // the residual row: row-major [M][ldh], or (ldh == 0) fragment order -- the 8 columns of a chunk are contiguous in both
    auto hchunk = [&](int c) -> bf16_t* {
        return ldh_ ? h_ + (size_t)row * ldh_ + c * 8 : h_ + xp_index(row >> 5, D >> 4, row & 31, c * 8);
    };

    // gamma / beta do not depend on anything this kernel computes: request them first, so that the last phase does not
    // start with a global round trip (the kernel is a chain of latencies: slabs -> mean -> variance -> normalise)
    constexpr int RU_PRE = 1024 / THREADS;          // chunks of 8 columns per thread held in registers: D <= 8192
    const bool pre = NC <= RU_PRE * THREADS;
    uint4 gpre[RU_PRE], bpre[RU_PRE];
    if (pre) {
#pragma unroll
        for (int i = 0; i < RU_PRE; ++i) {
            const int c = tid + i * THREADS;
            if (c < NC) {
                gpre[i] = *reinterpret_cast<const uint4*>(g_ + c * 8);
                bpre[i] = *reinterpret_cast<const uint4*>(b_ + c * 8);
            }
        }
    }

    // the rest of the arguments (common.h sv_late_args): the embedding mode needs its tables at once, the slab mode only reads
    // eps / the output pointer at the end -> after its loads are in flight
    RowUpdateArgs p;
    if (ws_ == nullptr) p = sv_late_args<RowUpdateArgs>(offsetof(RowUpdateKernarg, p));
    float s = 0.f;
    for (int c = tid; c < NC; c += THREADS) {
        float f[8];
        if (ws_ == nullptr) {
            const int tok = p.tokens[row], pos = p.positions[row];
            float a[8], w[8];
            unpack8(*reinterpret_cast<const uint4*>(p.wte + (size_t)tok * D + c * 8), a);
            if (p.wpe) {
                unpack8(*reinterpret_cast<const uint4*>(p.wpe + (size_t)pos * D + c * 8), w);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = bfround(a[e] + w[e]);
            } else {                    // rotary models: no learned position table
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = a[e];
            }
        } else {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = 0.f;
            // bias / residual and up to 4 slabs are requested together, then summed in slab order (a load per iteration,
            // each waited for, is one L2 round trip per slab)
            const uint4 bq = *reinterpret_cast<const uint4*>(bias_ + c * 8);
            const uint4 hq = *reinterpret_cast<const uint4*>(hchunk(c));
            for (int base = 0; base < splitk_; base += 4) {
                float4 a[4], b4[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int sp = base + j < splitk_ ? base + j : splitk_ - 1;
                    const float* src = ws_ + ((size_t)sp * rows_ws_ + row) * ldws_ + c * 8;
                    a[j] = *reinterpret_cast<const float4*>(src);
                    b4[j] = *reinterpret_cast<const float4*>(src + 4);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (base + j < splitk_) {
                        v[0] += a[j].x; v[1] += a[j].y; v[2] += a[j].z; v[3] += a[j].w;
                        v[4] += b4[j].x; v[5] += b4[j].y; v[6] += b4[j].z; v[7] += b4[j].w;
                    }
                }
            }
            float bb[8], hh[8];
            unpack8(bq, bb);
            unpack8(hq, hh);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = bfround(hh[e] + bfround(v[e] + bb[e]));
        }
        *reinterpret_cast<uint4*>(hchunk(c)) = pack8(f);
#pragma unroll
        for (int e = 0; e < 8; ++e) { hrow[c * 8 + e] = f[e]; s += f[e]; }
    }
    if (ws_ != nullptr) p = sv_late_args<RowUpdateArgs>(offsetof(RowUpdateKernarg, p));
    s = wave_sum(s);
    if (lane == 0) redbuf[wave] = s;
    __syncthreads();
    float ssum = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) ssum += redbuf[w];
    const float mean = ssum / (float)D;
    float q = 0.f;
    for (int c = tid; c < NC; c += THREADS) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { float d = hrow[c * 8 + e] - mean; q += d * d; }
    }
    q = wave_sum(q);
    if (lane == 0) redbuf[NW + wave] = q;
    __syncthreads();
    float qsum = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) qsum += redbuf[NW + w];
    const float rstd = rsqrtf(qsum / (float)D + p.eps);
    const int KS = D >> 4;
    if (pre) {
#pragma unroll
        for (int i = 0; i < RU_PRE; ++i) {
            const int c = tid + i * THREADS;
            if (c < NC) {
                float f[8], gg[8], bb[8];
                unpack8(gpre[i], gg);
                unpack8(bpre[i], bb);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = (hrow[c * 8 + e] - mean) * rstd * gg[e] + bb[e];
                *reinterpret_cast<uint4*>(p.xp_out + xp_index(row >> 5, KS, row & 31, c * 8)) = pack8(f);
            }
        }
        return;
    }
    for (int c = tid; c < NC; c += THREADS) {
        float f[8], gg[8], bb[8];
        unpack8(*reinterpret_cast<const uint4*>(g_ + c * 8), gg);
        unpack8(*reinterpret_cast<const uint4*>(b_ + c * 8), bb);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = (hrow[c * 8 + e] - mean) * rstd * gg[e] + bb[e];
        *reinterpret_cast<uint4*>(p.xp_out + xp_index(row >> 5, KS, row & 31, c * 8)) = pack8(f);
    }
}

void launch_row_update_ln(const RowUpdateArgs& a, hipStream_t st) {
    // one 8-column chunk per thread where the row is wide (StarVector-8B, D = 4608: 576 chunks): with 256 threads the row took
    // three serial load -> sum -> store rounds (7.2 us per launch against 4.7 at D = 2048)
    if (a.D > 2048)
        row_update_ln_kernel<1024><<<a.M, 1024, a.D * sizeof(float), st>>>(a.ws, a.bias, a.h, a.g, a.b, a.splitk, a.ldws, a.rows_ws, a.ldh, a.D, a.M, a);
    else
        row_update_ln_kernel<256><<<a.M, 256, a.D * sizeof(float), st>>>(a.ws, a.bias, a.h, a.g, a.b, a.splitk, a.ldws, a.rows_ws, a.ldh, a.D, a.M, a);
}

// ------------------------------------------------------------------------------------------------
// rowln_cattn_kernel (round 5): the row update AND the c_attn projection that consumes it, as ONE launch.
//   Two launches today: row_update_ln_kernel (32 blocks: slabs + bias + residual -> h, LayerNorm -> xp; 4.95 us, a chain of latencies
//   that touches ~1 MB) and gemm_skinny_kernel<8, false> (288 blocks: 9.4 MB of c_attn weights -> 4 fp32 slabs; 5.2 us of which ~1.3 us
//   stream) with a kernel boundary between them: 10.2 us per layer, 24 % of the decode step, nearly all of it latency.
//   What round 4 learned about hand-offs inside a launch (gemm.hip, mlp_fused_kernel): the price sits in the CONSUMER CU's own memory
//   queue, and polls must not queue behind weight requests.  This pair is the favourable case: a c_attn block's WHOLE weight share is
//   4 KiB per wave (K slice 512 = 4 k-steps per wave), so it is requested at t = 0, has landed long before the row update publishes,
//   and from then on the consumer's queue holds nothing but its polls; the producers are 32 blocks that start first.
//   Grid = 32 row blocks + (N / 32) x splitk GEMM blocks, 512 threads, all co-resident (two blocks per CU: <= 128 VGPRs, 34 KiB LDS).
//   Row role   blocks 0 .. 31 (waves 4-7 leave at once): row_update_ln_kernel<256>, value for value; the LayerNorm output goes out with
//              16-byte sc1 (write-through) stores.
//   GEMM role  block 32 + L: the (tile, K slice) of gemm_skinny_kernel's XCD-aware assignment for block L; weights -> registers, then the
//              wave polls the 4 KiB of activations of ITS k-steps with sc1 loads and recognises "not written yet" by the data: the buffer
//              is pre-filled with the bf16 pair 0xFFFF'FFFF, which no finite LayerNorm output produces -- by the attention launch of the
//              layer before (AttnDecodeArgs::poison2), for layer 0 by the lm_head launch of the step / prompt pass before
//              (SkinnyArgs::poison).  ONLY write-through stores and L1-bypassing loads ever touch this buffer: the first wiring armed it
//              with a memset node and shared it with the prompt pass's plain stores, and a stale line in one XCD's L2 turned up as NaN
//              logits in bench.py's third call.  Bounded by the wall
//              clock (give-up code 4 in *err: never a hang).  Then gemm_skinny_kernel<8, false>'s 4 MFMAs, LDS reduction in wave order
//              and slab store: bit-identical slabs.
//   Wide rows (StarVector-8B's 7-launch layer, hidden 4608: rowln_cattn_kernel<9, true>): the 1024-thread row update on a block's 512 threads
//              (rowln_wide_role), nine k-steps of weights per GEMM wave, 736 blocks at three per CU.  The 1B timing does NOT carry over: 52 MB
//              of weight requests from t = 0 are 8+ us of HBM stream and stretch the row role past 6.5 us -- there nobody polls before its own
//              weights have landed (the stream is the timer) and the activations are fetched in two batches behind them.  17.3 us per launch
//              against 18.5 for the pair (-1 ... -1.5 % per step on config 4); profiles/rowln_cattn_r05_ab.log, section 8.
//   Needs every block resident at once: on for an engine that owns its GPU (sv_config.exclusive_device), like the fused MLP launch.
//   (The row blocks are the FIRST blocks of the grid and wait for nothing, so a GEMM block never waits for a block dispatched behind it;
//    the all-resident rule is kept anyway -- dispatch order is observed behaviour, not a documented guarantee.)
// ------------------------------------------------------------------------------------------------
// A LayerNorm output that is NaN may carry ANY payload -- AMD hardware propagates the input's -- including the bf16 pair 0xFFFF'FFFF, which the
// GEMM role of the fused launch reads as "not written yet": every c_attn wave of the layer would then spin to its 5 ms budget and the call
// would fail with code 4 (blocks not resident) instead of code 1 (non-finite logits).  The row role therefore publishes that one pattern as
// the canonical quiet NaN pair: still NaN (the logits check reports it), never mistaken for the pattern.
__device__ __forceinline__ uint32_t sv_not_pattern(uint32_t w) { return w == 0xffffffffu ? 0x7fc07fc0u : w; }

// The row role for wide rows (2048 < D <= 8192): row_update_ln_kernel<1024>, value for value, on the 512 threads of a rowln_cattn block.
// That kernel gives thread v < 1024 the chunk v (8 columns) and reduces (sum, then sum of squared deviations) by a 64-lane shuffle
// per wave and a sum over the 16 waves in wave order.  Here thread t takes the chunks t (virtual wave t >> 6) and 512 + t (virtual
// wave 8 + (t >> 6)), each virtual wave is reduced by the same shuffle, and the 16 partials are added in the same order.
__device__ __forceinline__ void rowln_wide_role(const float* ws_, const bf16_t* bias_, bf16_t* h_, int splitk_ru_, int ldws_, int rows_ws_, int KS_,
                                                int row, char* smem, const RowCattnArgs p) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* hrow = reinterpret_cast<float*>(smem);                                 // [D] (D <= 8192 - 64 floats: launcher)
    float* redbuf = hrow + ((KS_ << 4) + 63) / 64 * 64;                           // [32]
    const int D = KS_ << 4, NC = D >> 3;
    float part[2] = {0.f, 0.f};
    uint4 gq[2], bq2[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = tid + i * 512;
        if (c < NC) {
            gq[i] = *reinterpret_cast<const uint4*>(p.g + c * 8);
            bq2[i] = *reinterpret_cast<const uint4*>(p.b + c * 8);
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = tid + i * 512;
        if (c >= NC) continue;
        bf16_t* hc = p.ldh ? h_ + (size_t)row * p.ldh + c * 8 : h_ + xp_index(row >> 5, KS_, row & 31, c * 8);
        float f[8];
        if (ws_ == nullptr) {                                                     // embedding mode: h = bf(wte[tok] + wpe[pos])
            const int tok = p.tokens[row], pos = p.positions[row];
            float a[8], w[8];
            unpack8(*reinterpret_cast<const uint4*>(p.wte + (size_t)tok * D + c * 8), a);
            if (p.wpe) {
                unpack8(*reinterpret_cast<const uint4*>(p.wpe + (size_t)pos * D + c * 8), w);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = bfround(a[e] + w[e]);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = a[e];
            }
        } else {                                                                  // h = bf(h + bf(sum of the slabs in slab order + bias))
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = 0.f;
            const uint4 bq = *reinterpret_cast<const uint4*>(bias_ + c * 8);
            const uint4 hq = *reinterpret_cast<const uint4*>(hc);
            float4 sa[4], sb[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int sp = j < splitk_ru_ ? j : splitk_ru_ - 1;
                const float* src = ws_ + ((size_t)sp * rows_ws_ + row) * ldws_ + c * 8;
                sa[j] = *reinterpret_cast<const float4*>(src);
                sb[j] = *reinterpret_cast<const float4*>(src + 4);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {                                         // slab order (launcher: at most 4 slabs)
                if (j < splitk_ru_) {
                    v[0] += sa[j].x; v[1] += sa[j].y; v[2] += sa[j].z; v[3] += sa[j].w;
                    v[4] += sb[j].x; v[5] += sb[j].y; v[6] += sb[j].z; v[7] += sb[j].w;
                }
            }
            float bb[8], hh[8];
            unpack8(bq, bb);
            unpack8(hq, hh);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = bfround(hh[e] + bfround(v[e] + bb[e]));
        }
        *reinterpret_cast<uint4*>(hc) = pack8(f);
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { hrow[c * 8 + e] = f[e]; s += f[e]; }
        part[i] = s;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {                                                 // virtual waves wave and 8 + wave
        const float t = wave_sum(part[i]);
        if (lane == 0) redbuf[8 * i + wave] = t;
    }
    __syncthreads();
    float ssum = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) ssum += redbuf[w];
    const float mean = ssum / (float)D;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = tid + i * 512;
        float q = 0.f;
        if (c < NC) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = hrow[c * 8 + e] - mean; q += d * d; }
        }
        q = wave_sum(q);
        if (lane == 0) redbuf[16 + 8 * i + wave] = q;
    }
    __syncthreads();
    float qsum = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) qsum += redbuf[16 + w];
    const float rstd = rsqrtf(qsum / (float)D + p.eps);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.xp_out, 0, (unsigned)((size_t)KS_ * 1024), 0x00020000);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = tid + i * 512;
        if (c < NC) {
            float gg[8], bb[8], o[8];
            unpack8(gq[i], gg);
            unpack8(bq2[i], bb);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (hrow[c * 8 + e] - mean) * rstd * gg[e] + bb[e];
            const uint4 ov = pack8(o);
            u32x4 q4 = {sv_not_pattern(ov.x), sv_not_pattern(ov.y), sv_not_pattern(ov.z), sv_not_pattern(ov.w)};
            __builtin_amdgcn_raw_buffer_store_b128(q4, rs, (int)(xp_index(0, KS_, row & 31, c * 8) * 2), 0, 16);      // sc1: write-through
        }
    }
}

struct RowCattnKernarg { const float* ws; const bf16_t* bias; bf16_t* h; const bf16_t* Wp; int splitk_ru, ldws, rows_ws, M, KS, ks_per_split, n_tiles, S;
                         RowCattnArgs p; };
// KPW: k-steps per GEMM wave (= K / 16 / splitk / 8: 4 at StarVector-1B, 9 at StarVector-8B -- the wave's whole weight share in 16 / 36
// registers).  WIDE: the row role of row_update_ln_kernel<1024> (D > 2048: one 8-column chunk per thread of 1024) run by the 512 threads of
// a block -- thread t takes the chunks t and 512 + t and the block reduces over the SAME 16 groups of 64 chunks in the same order, so the
// statistics are the 1024-thread kernel's bit for bit (rowln_wide_role below); otherwise row_update_ln_kernel<256> on waves 0-3.
template <int KPW, bool WIDE>
__global__ __launch_bounds__(512, WIDE ? 6 : 4) void rowln_cattn_kernel(const float* ws_, const bf16_t* bias_, bf16_t* h_, const bf16_t* Wp_, int splitk_ru_,
                                                             int ldws_, int rows_ws_, int M_, int KS_, int ks_per_split_, int n_tiles_, int S_,
                                                             RowCattnArgs p_unused) {
    extern __shared__ __attribute__((aligned(16))) char rc_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int ROWB = 32;                                 // row-role blocks (a multiple of 8: the GEMM blocks keep their XCDs)
    if (blockIdx.x < ROWB) {
        // ------------------------------------------------ row role ------------------------------------------------
        const int row = blockIdx.x;
        if (row >= M_) return;
        if constexpr (WIDE) {
            rowln_wide_role(ws_, bias_, h_, splitk_ru_, ldws_, rows_ws_, KS_, row, rc_smem,
                            sv_late_args<RowCattnArgs>(offsetof(RowCattnKernarg, p)));
            return;
        }
        if (wave >= 4) return;
        constexpr int THREADS = 256, NW = 4;
        float* hrow = reinterpret_cast<float*>(rc_smem);                          // [D]
        float* redbuf = hrow + 2048;                                              // [2 * NW]
        const int KSD = KS_, D = KS_ << 4, NC = D >> 3;                           // launcher: K of the projection == D <= 2048 -> one 8-column chunk per thread
        const int c = tid;
        const bool on = c < NC;
        bf16_t* hc = h_ + xp_index(row >> 5, KSD, row & 31, c * 8);               // the residual stream lives in fragment order
        uint4 gq = make_uint4(0, 0, 0, 0), bq2 = gq, bq = gq, hq = gq;
        float4 sa[4], sb[4];
        // what the leading (preloaded) arguments reach goes out first: bias, residual and up to 4 slabs; then the late arguments
        if (on && ws_ != nullptr) {
            bq = *reinterpret_cast<const uint4*>(bias_ + c * 8);
            hq = *reinterpret_cast<const uint4*>(hc);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int sp = j < splitk_ru_ ? j : splitk_ru_ - 1;
                const float* src = ws_ + ((size_t)sp * rows_ws_ + row) * ldws_ + c * 8;
                sa[j] = *reinterpret_cast<const float4*>(src);
                sb[j] = *reinterpret_cast<const float4*>(src + 4);
            }
        }
        const RowCattnArgs p = sv_late_args<RowCattnArgs>(offsetof(RowCattnKernarg, p));
        float f[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = 0.f;
        if (on) {
            gq = *reinterpret_cast<const uint4*>(p.g + c * 8);
            bq2 = *reinterpret_cast<const uint4*>(p.b + c * 8);
            if (ws_ == nullptr) {                                                 // embedding mode: h = bf(wte[tok] + wpe[pos])
                const int tok = p.tokens[row], pos = p.positions[row];
                float a[8], w[8];
                unpack8(*reinterpret_cast<const uint4*>(p.wte + (size_t)tok * D + c * 8), a);
                if (p.wpe) {
                    unpack8(*reinterpret_cast<const uint4*>(p.wpe + (size_t)pos * D + c * 8), w);
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = bfround(a[e] + w[e]);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] = a[e];
                }
            } else {                                                              // h = bf(h + bf(sum of the slabs in slab order + bias))
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {                                     // slab order (launcher: at most 4 slabs)
                    if (j < splitk_ru_) {
                        v[0] += sa[j].x; v[1] += sa[j].y; v[2] += sa[j].z; v[3] += sa[j].w;
                        v[4] += sb[j].x; v[5] += sb[j].y; v[6] += sb[j].z; v[7] += sb[j].w;
                    }
                }
                float bb[8], hh[8];
                unpack8(bq, bb);
                unpack8(hq, hh);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = bfround(hh[e] + bfround(v[e] + bb[e]));
            }
            *reinterpret_cast<uint4*>(hc) = pack8(f);
        }
        float s = 0.f;
        if (on) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { hrow[c * 8 + e] = f[e]; s += f[e]; }
        }
        s = wave_sum(s);
        if (lane == 0) redbuf[wave] = s;
        __syncthreads();
        float ssum = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) ssum += redbuf[w];
        const float mean = ssum / (float)D;
        float q = 0.f;
        if (on) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = hrow[c * 8 + e] - mean; q += d * d; }
        }
        q = wave_sum(q);
        if (lane == 0) redbuf[NW + wave] = q;
        __syncthreads();
        float qsum = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) qsum += redbuf[NW + w];
        const float rstd = rsqrtf(qsum / (float)D + p.eps);
        if (on) {
            float gg[8], bb[8], o[8];
            unpack8(gq, gg);
            unpack8(bq2, bb);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (hrow[c * 8 + e] - mean) * rstd * gg[e] + bb[e];
            const uint4 ov = pack8(o);
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.xp_out, 0, (unsigned)((size_t)KSD * 1024), 0x00020000);
            u32x4 q4 = {sv_not_pattern(ov.x), sv_not_pattern(ov.y), sv_not_pattern(ov.z), sv_not_pattern(ov.w)};
            __builtin_amdgcn_raw_buffer_store_b128(q4, rs, (int)(xp_index(0, KSD, row & 31, c * 8) * 2), 0, 16);      // sc1: write-through
        }
        return;
    }
    // ------------------------------------------------ GEMM role ------------------------------------------------
    constexpr int WAVES = 8, RPW = 2;                        // KPW k-steps per wave (launcher-checked): the whole share fits 4 * KPW registers
    float (*red)[16][64] = reinterpret_cast<float (*)[16][64]>(rc_smem);          // [WAVES][16][64]
    const long long t_start = wall_clock64();
    const int L = blockIdx.x - ROWB;
    const int xcd = L & 7, ii = L >> 3;
    const int tpg = (n_tiles_ * S_) >> 3;                    // tiles per XCD (gemm_skinny_kernel, flags & 1)
    const int split = xcd % S_;
    const int nt = (xcd / S_) * tpg + ii;
    const int m = lane & 31, half = lane >> 5;
    const int ks0 = split * ks_per_split_ + wave * KPW;
    const u32x4* wptr = reinterpret_cast<const u32x4*>(Wp_) + ((size_t)nt * KS_ + ks0) * 64 + lane;
    u32x4 w[KPW];
    if constexpr (WIDE) {
        // StarVector-8B: the projection's weights are 52 MB -- 8 us of HBM stream that the launch cannot go below, and a memory system
        // saturated from t = 0 stretches every dependent access of the 16 row blocks by the drain time of the CUs' miss queues (measured:
        // the row role published after > 6.5 us instead of 3.6, and polls that fail add their traffic on top: 4037 vs 3969 us per step).
        // So the row blocks' loads go FIRST -- the GEMM blocks of the first resident round hold their weight requests back for p.delay
        // ticks -- and nobody polls before its weights have landed (by then the LayerNorm output has long been published): the weight
        // stream itself is the timer.
        const RowCattnArgs pw = sv_late_args<RowCattnArgs>(offsetof(RowCattnKernarg, p));
        if ((int)blockIdx.x < pw.first_round)
            while (wall_clock64() - t_start < (long long)pw.delay) __builtin_amdgcn_s_sleep(4);
    }
#pragma unroll
    for (int u = 0; u < KPW; ++u) w[u] = __builtin_nontemporal_load(wptr + (size_t)u * 64);          // streamed once, depends on nothing
    const RowCattnArgs p = sv_late_args<RowCattnArgs>(offsetof(RowCattnKernarg, p));
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(p.xp_out, 0, (unsigned)((size_t)KS_ * 1024), 0x00020000);
    // 2304 waves polling 4 KiB each would put ~9 MB per poll round on the L2s while the 32 row blocks are still loading (first form,
    // measured: +1 us per layer over the two launches).  So nobody polls before the row update can possibly have published (its loads
    // alone take ~2 us): the first poll goes out p.delay x 10 ns after the block started.  Measured (profiles/rowln_cattn_r05_ab.log): a
    // poll that finds the pattern costs ~1.2 us whatever happens next (a full round trip before the retry), a poll that comes late
    // costs its lateness: 1043 us per step with the first poll at <= 3.2 us, 1015 at 3.6 .. 3.9 us, +5 us per step for every 0.2 us
    // after that -- so the poll is TIMED to land just behind the publish.  A wave watches ONE k-step (1 KiB) -- a row is published by one
    // store instruction of its block, so its k-steps turn up together -- and only then fetches the other three, re-checking them
    // (all four k-steps per poll: 1032 vs 1013 us per step at the same timing).
    // (wall clock, 100 MHz: the row update's latency is memory latency, not shader clocks -- an s_sleep count would drift with DVFS)
    // (blocks beyond the first resident round -- StarVector-8B: 704 GEMM blocks on 512 slots -- start when the data has long been there: no wait)
    if constexpr (WIDE) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                          // the weights are here: now look for the activations
        // (requesting the first batch of activations BEHIND the weights at t = 0 -- a wave's loads return in order -- was tried: the batch is
        //  read out of L2 long before the row blocks publish, every wave then pays the retry anyway, and the 20 registers held across the
        //  stream spill: 4298 vs 3909 us per step)
    } else {
        if ((int)blockIdx.x < p.first_round)
            while (wall_clock64() - t_start < (long long)p.delay) __builtin_amdgcn_s_sleep(4);
    }
    // XB: activation k-steps requested per batch.  The narrow form keeps all 4 in flight; the wide form takes its 9 as 5 + 4 (20
    // registers instead of 36: with the 36 weight registers the kernel stays at <= 80 VGPRs, i.e. THREE blocks per CU -- all 736 blocks
    // resident at once; at two per CU the last 224 blocks only start when the first ones leave and the launch streams its weights in two
    // rounds: 17.7 us against ... measured below) -- the weights have landed by then, so a batch costs one L2 round trip.
#ifndef SV_RC_XB
#define SV_RC_XB 5           // (build-time A/B: -DSV_RC_XB=3 = three batches of three)
#endif
    constexpr int XB = WIDE ? SV_RC_XB : KPW;
    u32x4 x[XB];
    int gave_up = 1;
    auto patt = [&](const u32x4& v) { return m < M_ && (v[0] == 0xffffffffu || v[1] == 0xffffffffu || v[2] == 0xffffffffu || v[3] == 0xffffffffu); };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int it = 0;; ++it) {
        x[0] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, ks0 * 1024 + lane * 16, 0, 16);                 // sc1: L1 bypass
        if (!__any(patt(x[0]))) {
#pragma unroll
            for (int u = 1; u < XB; ++u) x[u] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (ks0 + u) * 1024 + lane * 16, 0, 16);
            bool bad = false;
#pragma unroll
            for (int u = 1; u < XB; ++u) bad = bad || patt(x[u]);      // rows >= M of the tile are never written: only live rows are examined
            if (!__any(bad)) { gave_up = 0; break; }
        }
        if ((it & 7) == 7 && (wall_clock64() - t_start > (long long)p.spin_ticks ||
                              __hip_atomic_load(p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) break;
        __builtin_amdgcn_s_sleep(8);
    }
#pragma unroll
    for (int u = 0; u < XB; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag4(w[u]), as_frag4(x[u]), acc, 0, 0, 0);
    if constexpr (XB < KPW) {
        // the remaining batches, in k order (the accumulation order of gemm_skinny_kernel: one chain over the wave's k-steps); every value
        // that reaches an MFMA has been examined for the pattern
#pragma unroll
        for (int b0 = XB; b0 < KPW; b0 += XB) {
            for (int it = 0; !gave_up; ++it) {
#pragma unroll
                for (int u = 0; u < XB; ++u)
                    if (b0 + u < KPW) x[u] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (ks0 + b0 + u) * 1024 + lane * 16, 0, 16);
                bool bad = false;
#pragma unroll
                for (int u = 0; u < XB; ++u)
                    if (b0 + u < KPW) bad = bad || patt(x[u]);
                if (!__any(bad)) break;
                if ((it & 7) == 7 && (wall_clock64() - t_start > (long long)p.spin_ticks ||
                                      __hip_atomic_load(p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) { gave_up = 1; break; }
                __builtin_amdgcn_s_sleep(8);
            }
#pragma unroll
            for (int u = 0; u < XB; ++u)
                if (b0 + u < KPW) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag4(w[b0 + u]), as_frag4(x[u]), acc, 0, 0, 0);
        }
    }
    if (gave_up && lane == 0) {
        const int seen = atomicCAS(p.err, 0, 4);       // the step's result is void; the FIRST code raised survives (0 -> 4 only)
        if (p.dbg && seen == 0) {                // the first wave to give up leaves a note (read by the host when it reports the failure)
            p.dbg[0] = blockIdx.x; p.dbg[1] = wave; p.dbg[2] = ks0; p.dbg[3] = wall_clock64() - t_start;
            p.dbg[4] = ((long long)x[0][0] << 32) | x[0][1]; p.dbg[5] = p.layer; p.dbg[6] = M_; p.dbg[7] = 1;
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
    __syncthreads();
    float v[RPW];
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const int r = wave * RPW + i;
        float t = red[0][r][lane];
#pragma unroll
        for (int q = 1; q < WAVES; ++q) t += red[q][r][lane];
        v[i] = t;
    }
    {
        const int r = wave * RPW;
        const int n0 = nt * 32 + 8 * (r >> 2) + 4 * half + (r & 3);
        *reinterpret_cast<float2*>(p.ws_out + ((size_t)split * 32 + m) * p.ldws_out + n0) = make_float2(v[0], v[1]);
    }
}

// shape rule of the launch (host arithmetic; sv_create): D = hidden = K of the projection, Npad x K weight, split-K `splitk` slabs, the row
// update in front sums `splitk_ru` slabs; blocks = 32 + (Npad / 32) * splitk.  k-steps per wave 4 (StarVector-1B) or 9 (StarVector-8B).
// D <= 2048: all blocks resident at once (two per CU).  Wider rows (StarVector-8B: 32 + 704 blocks on 512 slots) rely on what the narrow
// form only has as a second line of defence: the row blocks are the FIRST blocks of the grid and wait for nothing, blocks are dispatched in
// index order, so a GEMM block never waits for a block dispatched behind it (and every wait is bounded by the wall clock anyway).
static int rowln_kpw(int K, int splitk) {
    const int KS = K / 16;
    if (splitk < 1 || 8 % splitk || KS % (splitk * 8)) return 0;
    const int kpw = KS / splitk / 8;
    return (kpw == 4 || kpw == 9) ? kpw : 0;
}
bool rowln_cattn_fits(int D, int Npad, int K, int splitk, int splitk_ru, int num_cus) {
    const int NT = Npad / 32;
    if (D > 8128 || (D & 15) || K != D || !rowln_kpw(K, splitk)) return false;
    if ((NT * splitk) % 8 || NT % (8 / splitk) || splitk_ru < 1 || splitk_ru > 4) return false;
    return D > 2048 || 32 + NT * splitk <= 2 * num_cus;
}
// Blocks of the launch a CU really holds at once: the narrow form needs ALL its blocks resident (two per CU), the wide form's first round is
// three per CU -- both are what __launch_bounds__ asks of the compiler, not something the host arithmetic above can see.  Asked of the runtime once per
// form on the engine's device (sv_create); a compiler that spills or drops occupancy turns the fused launch off instead of leaving it to the 5 ms give-up.
int rowln_cattn_blocks_per_cu(bool wide) {
    static int occ[2] = {-1, -1};
    int& o = occ[wide ? 1 : 0];
    if (o < 0) {
        int n = 0;
        const size_t smem = (size_t)8 * 16 * 64 * 4 + 64;
        const hipError_t r = wide ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, rowln_cattn_kernel<9, true>, 512, smem)
                                  : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, rowln_cattn_kernel<4, false>, 512, smem);
        o = r == hipSuccess ? n : 0;
    }
    return o;
}
bool rowln_cattn_resident(bool wide) {
    const int have = rowln_cattn_blocks_per_cu(wide), want = wide ? 3 : 2;
    if (have >= want) return true;
    fprintf(stderr, "[starvector_amd] rowln_cattn_kernel<%s>: %d block(s) per CU resident, the fused row update + c_attn launch needs %d -- off for this engine\n",
            wide ? "9, true" : "4, false", have, want);
    return false;
}
int launch_rowln_cattn(const RowUpdateArgs& ru, const SkinnyArgs& sk, int* err, int spin_ticks, hipStream_t st, int delay, long long* dbg, int layer, int num_cus) {
    const int KS = sk.K / 16, NT = sk.Npad / 32;
    const bool wide = ru.D > 2048;
    if (ru.D > 8128 || (ru.D & 15) || (!wide && ru.ldh != 0) || ru.M < 1 || ru.M > 32 || sk.MT != 1 || sk.Wq || sk.out_mode != SK_OUT_PARTIAL) return -1;
    if (ru.ws && (ru.splitk < 1 || ru.splitk > 4)) return -1;              // the row role sums at most 4 slabs
    const int kpw = rowln_kpw(sk.K, sk.splitk);
    if (!kpw || (wide && kpw != 9) || (!wide && kpw != 4)) return -1;      // the two instantiations
    if ((NT * sk.splitk) % 8 || NT % (8 / sk.splitk) || sk.K != ru.D || sk.xp != ru.xp_out || !err) return -1;
    RowCattnArgs a;
    memset(&a, 0, sizeof(a));
    a.g = ru.g; a.b = ru.b; a.eps = ru.eps; a.D = ru.D; a.wte = ru.wte; a.wpe = ru.wpe; a.tokens = ru.tokens; a.positions = ru.positions;
    a.xp_out = ru.xp_out; a.ws_out = sk.ws; a.ldws_out = sk.ldws; a.err = err; a.spin_ticks = spin_ticks; a.delay = delay; a.dbg = dbg; a.layer = layer;
    a.first_round = (wide ? 3 : 2) * (num_cus > 0 ? num_cus : 256); a.ldh = ru.ldh;
    const size_t smem = (size_t)8 * 16 * 64 * 4 + 64;
    const int grid = 32 + NT * sk.splitk;
    if (wide)
        rowln_cattn_kernel<9, true><<<grid, 512, smem, st>>>(ru.ws, ru.bias, ru.h, sk.Wp, ru.splitk, ru.ldws, ru.rows_ws, ru.M, KS, KS / sk.splitk, NT, sk.splitk, a);
    else
        rowln_cattn_kernel<4, false><<<grid, 512, smem, st>>>(ru.ws, ru.bias, ru.h, sk.Wp, ru.splitk, ru.ldws, ru.rows_ws, ru.M, KS, KS / sk.splitk, NT, sk.splitk, a);
    return 0;
}

// ------------------------------------------------------------------------------------------------
// patch embedding operand: image [B][3][S][S] -> patches [B*NP][Kpad]  (k = c*P*P + ky*P + kx)
// (clip_model.py:174,182: Conv2d(3->width, k=P, s=P, bias=False) == GEMM over flattened patches)
// ------------------------------------------------------------------------------------------------
__global__ void im2col_kernel(const bf16_t* __restrict__ img, bf16_t* __restrict__ out, int B, int S, int P,
                              int Kpad) {
    const int G = S / P, NP = G * G, K = 3 * P * P;
    const int chunks = Kpad >> 3;
    const size_t total = (size_t)B * NP * chunks;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int ch = (int)(i % chunks);
        const size_t rowi = i / chunks;
        const int pidx = (int)(rowi % NP), b = (int)(rowi / NP);
        const int py = pidx / G, px = pidx % G;
        uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = ch * 8 + e;
            bf16_t v = 0;
            if (k < K) {
                const int c = k / (P * P), rem = k % (P * P);
                const int ky = rem / P, kx = rem % P;
                v = img[(((size_t)b * 3 + c) * S + (py * P + ky)) * S + px * P + kx];
            }
            w[e >> 1] |= (uint32_t)v << ((e & 1) * 16);
        }
        *reinterpret_cast<uint4*>(out + rowi * Kpad + ch * 8) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}
void launch_im2col(const bf16_t* img, bf16_t* out, int B, int img_size, int patch, int Kpad, hipStream_t st) {
    const int G = img_size / patch;
    size_t total = (size_t)B * G * G * (Kpad / 8);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 16384) blocks = 16384;
    im2col_kernel<<<blocks, 256, 0, st>>>(img, out, B, img_size, patch, Kpad);
}

// ------------------------------------------------------------------------------------------------
// ViT token assembly + ln_pre (clip_model.py:185-187): x[b][0] = cls, x[b][1+p] = patch_out ;
// x = bf(x + pos) ; x = LN(x).  One wave per token row.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vit_embed_lnpre_kernel(const bf16_t* __restrict__ patch_out, int ldp,
                                                              const bf16_t* __restrict__ cls,
                                                              const bf16_t* __restrict__ pos,
                                                              const bf16_t* __restrict__ g,
                                                              const bf16_t* __restrict__ b,
                                                              bf16_t* __restrict__ x, int B, int NP, int Dv,
                                                              float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int T = NP + 1;
    if (row >= B * T) return;
    const int bi = row / T, tok = row % T;
    const bf16_t* src = tok == 0 ? cls : patch_out + ((size_t)bi * NP + tok - 1) * ldp;
    const bf16_t* pr = pos + (size_t)tok * Dv;
    const int NC = Dv >> 3;
    // Dv <= 8*64*4 = 2048 -> at most 4 chunks per lane kept in registers
    float v[4][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + i * 64;
        if (c < NC) {
            float a[8], pp[8];
            unpack8(*reinterpret_cast<const uint4*>(src + c * 8), a);
            unpack8(*reinterpret_cast<const uint4*>(pr + c * 8), pp);
#pragma unroll
            for (int e = 0; e < 8; ++e) { v[i][e] = bfround(a[e] + pp[e]); s += v[i][e]; }
        }
    }
    const float mean = wave_sum(s) / (float)Dv;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (lane + i * 64 < NC) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { float d = v[i][e] - mean; q += d * d; }
        }
    }
    const float rstd = rsqrtf(wave_sum(q) / (float)Dv + eps);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = lane + i * 64;
        if (c < NC) {
            float gg[8], bb[8], f[8];
            unpack8(*reinterpret_cast<const uint4*>(g + c * 8), gg);
            unpack8(*reinterpret_cast<const uint4*>(b + c * 8), bb);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = (v[i][e] - mean) * rstd * gg[e] + bb[e];
            *reinterpret_cast<uint4*>(x + (size_t)row * Dv + c * 8) = pack8(f);
        }
    }
}
void launch_vit_embed_lnpre(const bf16_t* patch_out, int ldp, const bf16_t* cls, const bf16_t* pos,
                            const bf16_t* g, const bf16_t* b, bf16_t* x, int B, int NP, int Dv, float eps,
                            hipStream_t st) {
    int rows = B * (NP + 1);
    vit_embed_lnpre_kernel<<<(rows + 3) / 4, 256, 0, st>>>(patch_out, ldp, cls, pos, g, b, x, B, NP, Dv, eps);
}

// ------------------------------------------------------------------------------------------------
// decoder prefill input: h = bf(inputs_embeds + wpe[0..S0-1])   (gpt_bigcode :980-985,1060-1063)
// ------------------------------------------------------------------------------------------------
__global__ void dec_embed_kernel(const bf16_t* __restrict__ emb, const bf16_t* __restrict__ wpe,
                                 bf16_t* __restrict__ h, int B, int S0, int D) {
    const int NC = D >> 3;
    const size_t total = (size_t)B * S0 * NC;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % NC);
        const size_t row = i / NC;
        const int t = (int)(row % S0);
        float a[8], w[8];
        unpack8(*reinterpret_cast<const uint4*>(emb + row * D + c * 8), a);
        unpack8(*reinterpret_cast<const uint4*>(wpe + (size_t)t * D + c * 8), w);
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] += w[e];
        *reinterpret_cast<uint4*>(h + row * D + c * 8) = pack8(a);
    }
}
void launch_dec_embed(const bf16_t* emb, const bf16_t* wpe, bf16_t* h, int B, int S0, int D, hipStream_t st) {
    size_t total = (size_t)B * S0 * (D / 8);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 16384) blocks = 16384;
    dec_embed_kernel<<<blocks, 256, 0, st>>>(emb, wpe, h, B, S0, D);
}

// wte lookup (starvector_v1.py:16-18): out[i][:] = table[ids[i]][:]
__global__ void gather_rows_kernel(const bf16_t* __restrict__ table, const int64_t* __restrict__ ids,
                                   bf16_t* __restrict__ out, int n, int D, int per_seq, size_t seq_stride) {
    const int NC = D >> 3;
    const size_t total = (size_t)n * NC;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % NC);
        const size_t r = i / NC;
        // row r = (sequence r / per_seq, position r % per_seq): sequences seq_stride elements apart (= per_seq * D: back to back)
        *reinterpret_cast<uint4*>(out + (r / per_seq) * seq_stride + (r % per_seq) * (size_t)D + c * 8) =
            *reinterpret_cast<const uint4*>(table + (size_t)ids[r] * D + c * 8);
    }
}
void launch_gather_rows(const bf16_t* table, const int64_t* ids, bf16_t* out, int n, int D, hipStream_t st, int per_seq, size_t seq_stride) {
    if (per_seq < 1) { per_seq = n > 0 ? n : 1; seq_stride = (size_t)per_seq * D; }
    size_t total = (size_t)n * (D / 8);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    gather_rows_kernel<<<blocks, 256, 0, st>>>(table, ids, out, n, D, per_seq, seq_stride);
}

// last prompt row of every sequence (only that row feeds ln_f + lm_head in prefill)
__global__ void gather_last_rows_kernel(const bf16_t* __restrict__ h, bf16_t* __restrict__ out, int B, int S0,
                                        int D) {
    const int NC = D >> 3;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B * NC; i += gridDim.x * blockDim.x) {
        const int c = i % NC, b = i / NC;
        *reinterpret_cast<uint4*>(out + (size_t)b * D + c * 8) =
            *reinterpret_cast<const uint4*>(h + ((size_t)b * S0 + S0 - 1) * D + c * 8);
    }
}
// the last n_keep rows of every sequence, packed [B * n_keep][D] (scoring forward: logits of the kept positions only)
__global__ void gather_tail_rows_kernel(const bf16_t* __restrict__ h, bf16_t* __restrict__ out, int B, int S0, int n_keep,
                                        int D) {
    const int NC = D >> 3;
    const size_t total = (size_t)B * n_keep * NC;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % NC);
        const size_t r = i / NC;
        const int j = (int)(r % n_keep), b = (int)(r / n_keep);
        *reinterpret_cast<uint4*>(out + r * D + c * 8) =
            *reinterpret_cast<const uint4*>(h + ((size_t)b * S0 + S0 - n_keep + j) * D + c * 8);
    }
}
void launch_gather_tail_rows(const bf16_t* h, bf16_t* out, int B, int S0, int n_keep, int D, hipStream_t st) {
    size_t total = (size_t)B * n_keep * (D / 8);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 16384) blocks = 16384;
    gather_tail_rows_kernel<<<blocks, 256, 0, st>>>(h, out, B, S0, n_keep, D);
}
void launch_gather_last_rows(const bf16_t* h, bf16_t* out, int B, int S0, int D, hipStream_t st) {
    int total = B * (D / 8);
    gather_last_rows_kernel<<<(total + 255) / 256, 256, 0, st>>>(h, out, B, S0, D);
}

// ------------------------------------------------------------------------------------------------
// adapter norm: nn.LayerNorm([Q, D]) over the joint plane, affine [Q][D] (adapter.py:25-26,38)
// one 1024-thread block per sample, two-pass statistics (mean, then centred second moment)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void plane_layernorm_kernel(const bf16_t* __restrict__ x,
                                                               const bf16_t* __restrict__ g,
                                                               const bf16_t* __restrict__ b,
                                                               bf16_t* __restrict__ y, int QD, float eps, size_t y_batch_stride) {
    __shared__ float red[16];
    __shared__ float stat[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bf16_t* xs = x + (size_t)blockIdx.x * QD;
    bf16_t* ys = y + (size_t)blockIdx.x * y_batch_stride;         // (= QD: planes back to back; larger: straight into a prefill buffer)
    const int NC = QD >> 3;
    float s = 0.f;
    for (int c = tid; c < NC; c += 1024) {
        float f[8];
        unpack8(*reinterpret_cast<const uint4*>(xs + (size_t)c * 8), f);
        float t = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) t += f[e];
        s += t;
    }
    s = wave_sum(s);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    if (tid == 0) {
        float t = 0.f;
        for (int w = 0; w < 16; ++w) t += red[w];
        stat[0] = t / (float)QD;
    }
    __syncthreads();
    const float mean = stat[0];
    float q = 0.f;
    for (int c = tid; c < NC; c += 1024) {
        float f[8];
        unpack8(*reinterpret_cast<const uint4*>(xs + (size_t)c * 8), f);
        float t = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { float d = f[e] - mean; t += d * d; }
        q += t;
    }
    q = wave_sum(q);
    __syncthreads();
    if (lane == 0) red[wave] = q;
    __syncthreads();
    if (tid == 0) {
        float t = 0.f;
        for (int w = 0; w < 16; ++w) t += red[w];
        stat[1] = rsqrtf(t / (float)QD + eps);
    }
    __syncthreads();
    const float rstd = stat[1];
    for (int c = tid; c < NC; c += 1024) {
        float f[8], gg[8], bb[8];
        unpack8(*reinterpret_cast<const uint4*>(xs + (size_t)c * 8), f);
        unpack8(*reinterpret_cast<const uint4*>(g + (size_t)c * 8), gg);
        unpack8(*reinterpret_cast<const uint4*>(b + (size_t)c * 8), bb);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = (f[e] - mean) * rstd * gg[e] + bb[e];
        *reinterpret_cast<uint4*>(ys + (size_t)c * 8) = pack8(f);
    }
}
void launch_plane_layernorm(const bf16_t* x, const bf16_t* g, const bf16_t* b, bf16_t* y, int B, int QD,
                            float eps, hipStream_t st, size_t y_batch_stride) {
    plane_layernorm_kernel<<<B, 1024, 0, st>>>(x, g, b, y, QD, eps, y_batch_stride ? y_batch_stride : (size_t)QD);
}

// adapter_norm="batch_norm": nn.BatchNorm1d(Q) in eval = per-token affine from running statistics
// (adapter.py:27-28)
__global__ void token_batchnorm_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                       const bf16_t* __restrict__ b, const bf16_t* __restrict__ rm,
                                       const bf16_t* __restrict__ rv, bf16_t* __restrict__ y, int B, int Q,
                                       int D, float eps, size_t y_batch_stride) {
    const int NC = D >> 3;
    const size_t total = (size_t)B * Q * NC;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const size_t row = i / NC;
        const int c = (int)(i % NC);
        const int qi = (int)(row % Q);
        const float mean = bf2f(rm[qi]);
        const float inv = 1.0f / sqrtf(bf2f(rv[qi]) + eps);
        const float ww = bf2f(w[qi]), bb = bf2f(b[qi]);
        float f[8];
        unpack8(*reinterpret_cast<const uint4*>(x + row * D + c * 8), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = (f[e] - mean) * inv * ww + bb;
        *reinterpret_cast<uint4*>(y + (row / Q) * y_batch_stride + (size_t)qi * D + c * 8) = pack8(f);
    }
}
void launch_token_batchnorm(const bf16_t* x, const bf16_t* w, const bf16_t* b, const bf16_t* rm,
                            const bf16_t* rv, bf16_t* y, int B, int Q, int D, float eps, hipStream_t st, size_t y_batch_stride) {
    size_t total = (size_t)B * Q * (D / 8);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 16384) blocks = 16384;
    token_batchnorm_kernel<<<blocks, 256, 0, st>>>(x, w, b, rm, rv, y, B, Q, D, eps, y_batch_stride ? y_batch_stride : (size_t)Q * D);
}

}  // namespace sv
