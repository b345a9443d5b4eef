// bf16 MFMA GEMMs for gfx950 (CDNA4).
//
//  * gemm_bf16_kernel   : big-M tile GEMM (ViT / adapter / decoder prefill).  128x128x64 block tile,
//                         4 waves (2x2), each wave 64x64 = 2x2 v_mfma_f32_32x32x16_bf16 accumulators,
//                         operands staged HBM->LDS with global_load_lds (16 B/lane, 1 KiB/wave-instr),
//                         double-buffered, one barrier per K-tile.  The weight arrives pre-packed in
//                         MFMA fragment order (linear, conflict-free LDS image); the activation tile is
//                         row-major with an XOR swizzle applied on the SOURCE address (LDS-DMA destinations
//                         are lane-linear) and the matching XOR on the ds_read_b128.
//  * gemm_skinny_kernel : M<=32 weight-streaming GEMM for the autoregressive step (HBM-bound): weights
//                         and activations both in fragment order, loaded straight to VGPRs as fully
//                         coalesced 1 KiB wave loads, K split across the waves of a block (LDS reduce)
//                         and optionally across blocks (fp32 slabs, reduced in a fixed order by the
//                         consumer kernel => deterministic).
//
// Orientation: MFMA A operand = weight fragment (i = output column n), B operand = activation
// fragment (j = row m).  D[i=n][j=m]: lane l holds m = l&31 and n = 8*(r>>2) + 4*(l>>5) + (r&3),
// i.e. 4 consecutive output columns per register group -> 8-byte bf16 stores per row.
#include <cstdlib>
#include <cstring>
#include <cstdio>
#include <map>
#include <mutex>
#include <tuple>
#include <type_traits>
#include "kernels.h"

namespace sv {

// ------------------------------------------------------------------------------------------------
// weight packing
// ------------------------------------------------------------------------------------------------
template <typename SrcT>
__global__ void pack_weight_kernel(const SrcT* __restrict__ W, bf16_t* __restrict__ Wp, int N, int K,
                                   int Npad, int Kpad) {
    const int KS = Kpad >> 4;
    const size_t total = (size_t)(Npad >> 5) * KS * 64;
    for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < total;
         c += (size_t)gridDim.x * blockDim.x) {
        int lane = (int)(c & 63);
        size_t t = c >> 6;
        int ks = (int)(t % KS);
        int nt = (int)(t / KS);
        int n = nt * 32 + (lane & 31);
        int k0 = ks * 16 + (lane >> 5) * 8;
        float f[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            int k = k0 + e;
            float v = 0.f;
            if (n < N && k < K) {
                if constexpr (sizeof(SrcT) == 4) v = (float)W[(size_t)n * K + k];
                else v = bf2f((bf16_t)W[(size_t)n * K + k]);
            }
            f[e] = v;
        }
        *reinterpret_cast<uint4*>(Wp + c * 8) = pack8(f);
    }
}

void launch_pack_weight(const void* src, int src_is_f32, bf16_t* dst, int N, int K, int Npad, int Kpad,
                        hipStream_t st) {
    size_t total = (size_t)(Npad / 32) * (Kpad / 16) * 64;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 65535) blocks = 65535;
    if (src_is_f32)
        pack_weight_kernel<float><<<blocks, 256, 0, st>>>((const float*)src, dst, N, K, Npad, Kpad);
    else
        pack_weight_kernel<bf16_t><<<blocks, 256, 0, st>>>((const bf16_t*)src, dst, N, K, Npad, Kpad);
}

template <typename SrcT>
__global__ void convert_bf16_kernel(const SrcT* __restrict__ s, bf16_t* __restrict__ d, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        if constexpr (sizeof(SrcT) == 4) d[i] = f2bf((float)s[i]);
        else d[i] = (bf16_t)s[i];
    }
}
void launch_convert_to_bf16(const void* src, int src_is_f32, bf16_t* dst, size_t n, hipStream_t st) {
    int blocks = (int)((n + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    if (src_is_f32) convert_bf16_kernel<float><<<blocks, 256, 0, st>>>((const float*)src, dst, n);
    else convert_bf16_kernel<bf16_t><<<blocks, 256, 0, st>>>((const bf16_t*)src, dst, n);
}

// ------------------------------------------------------------------------------------------------
// fp8 (OCP e4m3) weight-only quantisation at load time (BASELINE config 5: fp8 weights).  One scale per output row.
// ------------------------------------------------------------------------------------------------
template <typename SrcT>
__device__ __forceinline__ float ld_as_f32(const SrcT* p) {
    if constexpr (sizeof(SrcT) == 4) return (float)*p;
    else return bf2f((bf16_t)*p);
}
template <typename SrcT>
__global__ __launch_bounds__(256) void fp8_row_scale_kernel(const SrcT* __restrict__ src, float* __restrict__ scale, int N,
                                                            int K, int Npad) {
    __shared__ float red[4];
    const int n = blockIdx.x, tid = threadIdx.x;
    float m = 0.f;
    if (n < N)
        for (int k = tid; k < K; k += 256) m = fmaxf(m, fabsf(ld_as_f32(src + (size_t)n * K + k)));
    m = wave_max(m);
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    if (tid == 0) {
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        scale[n] = m > 0.f ? m / 448.0f : 1.0f;             // e4m3 max normal = 448
    }
}
// one thread = one lane's 16 fp8 bytes (two k-steps) of one (n-tile, k-step pair)
template <typename SrcT>
__global__ void pack_weight_fp8_kernel(const SrcT* __restrict__ src, const float* __restrict__ scale, bf16_t* __restrict__ dst_bf16,
                                       uint8_t* __restrict__ dst_q, int N, int K, int Npad, int Kpad) {
    const int KS = Kpad >> 4, KS2 = KS >> 1;
    const size_t total = (size_t)(Npad >> 5) * KS2 * 64;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        const size_t t = i >> 6;
        const int ks2 = (int)(t % KS2), nt = (int)(t / KS2);
        const int n = nt * 32 + (lane & 31);
        const float sc = n < N ? scale[n] : 1.0f;
        uint32_t words[4];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int ks = 2 * ks2 + h;
            float f[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = 16 * ks + 8 * (lane >> 5) + e;
                f[e] = (n < N && k < K) ? ld_as_f32(src + (size_t)n * K + k) / sc : 0.f;
            }
            int w0 = 0, w1 = 0;
            w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], w0, false);
            w0 = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], w0, true);
            w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], w1, false);
            w1 = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], w1, true);
            words[2 * h] = (uint32_t)w0;
            words[2 * h + 1] = (uint32_t)w1;
            // the same values as bf16 (exact: 3 mantissa bits) for the big-M kernels
            float q[8];
            const f32x2 a0 = __builtin_amdgcn_cvt_pk_f32_fp8(w0, false), a1 = __builtin_amdgcn_cvt_pk_f32_fp8(w0, true);
            const f32x2 a2 = __builtin_amdgcn_cvt_pk_f32_fp8(w1, false), a3 = __builtin_amdgcn_cvt_pk_f32_fp8(w1, true);
            q[0] = a0[0]; q[1] = a0[1]; q[2] = a1[0]; q[3] = a1[1]; q[4] = a2[0]; q[5] = a2[1]; q[6] = a3[0]; q[7] = a3[1];
            *reinterpret_cast<uint4*>(dst_bf16 + (((size_t)nt * KS + ks) * 64 + lane) * 8) = pack8(q);
        }
        *reinterpret_cast<uint4*>(dst_q + (((size_t)nt * KS2 + ks2) * 64 + lane) * 16) = make_uint4(words[0], words[1], words[2], words[3]);
    }
}
void launch_pack_weight_fp8(const void* src, int src_is_f32, bf16_t* dst_bf16, uint8_t* dst_q, float* scale, int N, int K,
                            int Npad, int Kpad, hipStream_t st) {
    const size_t total = (size_t)(Npad / 32) * (Kpad / 32) * 64;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 16384) blocks = 16384;
    if (src_is_f32) {
        fp8_row_scale_kernel<float><<<Npad, 256, 0, st>>>((const float*)src, scale, N, K, Npad);
        pack_weight_fp8_kernel<float><<<blocks, 256, 0, st>>>((const float*)src, scale, dst_bf16, dst_q, N, K, Npad, Kpad);
    } else {
        fp8_row_scale_kernel<bf16_t><<<Npad, 256, 0, st>>>((const bf16_t*)src, scale, N, K, Npad);
        pack_weight_fp8_kernel<bf16_t><<<blocks, 256, 0, st>>>((const bf16_t*)src, scale, dst_bf16, dst_q, N, K, Npad, Kpad);
    }
}

// ------------------------------------------------------------------------------------------------
// big-M GEMM
// ------------------------------------------------------------------------------------------------
#define GB_M 128
#define GB_N 128
#define GB_K 64
#define GB_BUF 32768   // per stage: 16 KiB activations + 16 KiB weights

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void lds_dma16(const void* g, char* lds_wave_base) {
    // 16 B per lane; LDS destination = wave-uniform base + lane*16 (hardware), source is per lane
    __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)lds_wave_base, 16, 0, 0);
}

// ------------------------------------------------------------------------------------------------
// Epilogue through LDS (bf16 outputs).  The MFMA accumulator layout is "one output ROW per lane": stored straight from
// registers, a wave instruction writes 8 bytes into each of 32 different rows (32 cache lines, 16 B used of each), and the
// residual is read the same way -- the store tail is issue-bound, not bandwidth-bound (cdna guide T21).  Instead every wave
// parks its finished values (bias, activation, first rounding already applied: bf16, exact) in its own LDS strip of
// 64 rows x 64 columns (row stride 144 B), then reads the strip back ROW-CONTIGUOUS: lane -> (row it*8 + l>>3, 16-byte
// chunk l&7), so a wave instruction covers 8 rows x 128 contiguous bytes: full lines for the residual loads and the stores.
// Same arithmetic in the same order as the register epilogue (the residual add happens on the bf16-rounded value, as
// before), so results are bit-identical.
// ------------------------------------------------------------------------------------------------
#define EPI_ROW_BYTES 144
__device__ __forceinline__ void epi_park4(char* strip, int row, int col, const float v[4]) {
    uint2 o;
    o.x = pack2bf(v[0], v[1]);
    o.y = pack2bf(v[2], v[3]);
    *reinterpret_cast<uint2*>(strip + row * EPI_ROW_BYTES + col * 2) = o;
}
// strip: this wave's 64 x 64 values; (gm0, gn0): global coordinates of the strip's corner
__device__ __forceinline__ void epi_flush_strip(const char* strip, const GemmArgs& p, int gm0, int gn0, int lane) {
    const int rsub = lane >> 3, chunk = lane & 7;
    const int n = gn0 + chunk * 8;
    const bool nok = n < p.N;                               // N % 8 == 0 on this path (checked by the launcher)
    uint4 rq[8];
    if (p.R) {
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int m = gm0 + it * 8 + rsub;
            rq[it] = (nok && m < p.M) ? *reinterpret_cast<const uint4*>(p.R + (size_t)m * p.ldr + n) : make_uint4(0u, 0u, 0u, 0u);
        }
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int r = it * 8 + rsub;
        const int m = gm0 + r;
        uint4 v = *reinterpret_cast<const uint4*>(strip + r * EPI_ROW_BYTES + chunk * 16);
        if (p.R) {
            float a[8], b[8];
            unpack8(v, a);
            unpack8(rq[it], b);
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] += b[e];         // bf(x W^T + b) + residual, rounded by the store below
            v = pack8(a);
        }
        if (nok && m < p.M) *reinterpret_cast<uint4*>((bf16_t*)p.C + (size_t)m * p.ldc + n) = v;
    }
}

// first row of row tile tm (tile height T = 128 / 256).  With a sequence structure (GemmArgs::seq_rows) the tiles cover the first
// 256 * (seq_rows / 256) rows of every sequence and never straddle two sequences; the rows left over go through gemm_tailk_kernel.
__device__ __forceinline__ int tile_first_row(const GemmArgs& p, int tm, int T) {
    if (p.seq_rows == 0) return tm * T;
    const int per = (p.seq_rows >> 8) * (256 / T);
    const int s = tm / per;
    return s * p.seq_rows + (tm - s * per) * T;
}

template <typename OutT>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware tile order (as in the 256^2 kernel below): workgroups are dealt round-robin to the 8 XCDs in launch order, each
    // XCD has its own L2.  In launch order every XCD walks the WHOLE activation matrix and every weight tile (rocprofv3
    // FETCH_SIZE of this kernel: 273 MB per launch against ~80 MB of operands).  Instead each XCD gets a contiguous run of
    // tile ids and walks it in bands of 8 m-tiles, m fastest: the 64 tiles resident on an XCD (32 CUs x 2 blocks) form an
    // 8 x 8 patch that shares 8 activation and 8 weight tiles through that XCD's L2.
    int tm, tn;
    {
        const int tiles_n = gridDim.x, tiles_m = gridDim.y, T = tiles_m * tiles_n;
        const int id = blockIdx.y * tiles_n + blockIdx.x;
        const int q = T >> 3, rem = T & 7, xcd = id & 7, loc = id >> 3;
        const int wg = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + loc;
        const int band = wg / (8 * tiles_n), r_in = wg - band * 8 * tiles_n;
        const int band_rows = min(8, tiles_m - band * 8);
        tm = band * 8 + r_in % band_rows;
        tn = r_in / band_rows;
    }
    const int m0 = tile_first_row(p, tm, GB_M);
    const int n0 = tn * GB_N;
    const int KS = p.K >> 4;
    const int KT = p.K / GB_K;
    const int NT_total = (p.N + 31) >> 5;   // packed n-tiles available (Npad/32 >= this)

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // per-lane constants of the activation staging: 4 wave-instructions of 8 rows x 128 B each
    // lane -> (row within the 8-row group, physical 16-B chunk); source chunk = p ^ f(row)
    const int srow = lane >> 3, schunk = lane & 7;

    auto stage = [&](int kt, int buf) {
        char* base = smem + buf * GB_BUF;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ii = wave * 4 + i;                 // 0..15 : rows 8*ii .. 8*ii+7
            const int r = ii * 8 + srow;
            const int c = schunk ^ ((r >> 1) & 7);
            int grow = m0 + r;
            grow = grow < p.M ? grow : p.M - 1;
            const bf16_t* g = p.A + (size_t)grow * p.lda + kt * GB_K + c * 8;
            lds_dma16(g, base + ii * 1024);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ci = wave * 4 + i;                 // 0..15 : (n-tile t, k-step s)
            const int t = ci >> 2, s = ci & 3;
            int nt = (n0 >> 5) + t;
            nt = nt < NT_total ? nt : NT_total - 1;
            const bf16_t* g = p.Wp + (((size_t)nt * KS + kt * 4 + s) * 64 + lane) * 8;
            lds_dma16(g, base + 16384 + ci * 1024);
        }
    };

    stage(0, 0);
    for (int kt = 0; kt < KT; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < KT) stage(kt + 1, (kt + 1) & 1);
        const char* base = smem + (kt & 1) * GB_BUF;
        // all 16 fragments of the K-tile are requested before the first MFMA (64 VGPRs): the LDS latency of
        // k-step s+1 hides behind the MFMAs of k-step s instead of stalling every second instruction
        bf16x8 wf[4][2], xf[4][2];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
                wf[s][nt] = *reinterpret_cast<const bf16x8*>(base + 16384 + (((wn * 2 + nt) * 4 + s) * 1024) + lane * 16);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int r = wm * 64 + mt * 32 + (lane & 31);
                const int c = (2 * s + (lane >> 5)) ^ ((r >> 1) & 7);
                xf[s][mt] = *reinterpret_cast<const bf16x8*>(base + r * 128 + c * 16);
            }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[s][nt], xf[s][mt], acc[nt][mt], 0, 0, 0);
    }

    const int half = lane >> 5;
    uint2 bq[2][4];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const int n = n0 + wn * 64 + nt * 32 + rg * 8 + half * 4;
            bq[nt][rg] = (p.bias && n < p.N) ? *reinterpret_cast<const uint2*>(p.bias + n) : make_uint2(0u, 0u);
        }
    if constexpr (sizeof(OutT) == 2) {
        if ((p.N & 7) == 0 && (p.ldc & 7) == 0 && (!p.R || (p.ldr & 7) == 0)) {
            // ---- epilogue through LDS (see epi_flush_strip): park the wave's 64 x 64 values as bf16, flush row-contiguous
            __syncthreads();                                   // every wave has read its last K-tile fragments
            char* strip = smem + wave * (64 * EPI_ROW_BYTES);
            // the activation is resolved ONCE, outside the 64 unrolled values: with a run-time switch inside, every value
            // carries all three exp/divide sequences and the epilogue no longer fits the instruction cache
            auto park = [&](auto tag) {
                constexpr int ACT = decltype(tag)::value;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                        for (int rg = 0; rg < 4; ++rg) {
                            const int n = n0 + wn * 64 + nt * 32 + rg * 8 + half * 4;
                            const float bj[4] = {__uint_as_float(bq[nt][rg].x << 16), __uint_as_float(bq[nt][rg].x & 0xffff0000u),
                                                 __uint_as_float(bq[nt][rg].y << 16), __uint_as_float(bq[nt][rg].y & 0xffff0000u)};
                            float cs[4] = {1.f, 1.f, 1.f, 1.f};
                            if (p.cscale && n < p.N) {
                                const float4 c4 = *reinterpret_cast<const float4*>(p.cscale + n);
                                cs[0] = c4.x; cs[1] = c4.y; cs[2] = c4.z; cs[3] = c4.w;
                            }
                            float v[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                float x = acc[nt][mt][rg * 4 + e] * cs[e] + bj[e];
                                if constexpr (ACT != ACT_NONE) x = sv_act(bfround(x), ACT);
                                v[e] = x;
                            }
                            epi_park4(strip, mt * 32 + (lane & 31), nt * 32 + rg * 8 + half * 4, v);
                        }
            };
            switch (p.act) {
                case ACT_QUICKGELU: park(std::integral_constant<int, ACT_QUICKGELU>{}); break;
                case ACT_SWISH: park(std::integral_constant<int, ACT_SWISH>{}); break;
                case ACT_GELU_TANH: park(std::integral_constant<int, ACT_GELU_TANH>{}); break;
                default: park(std::integral_constant<int, ACT_NONE>{}); break;
            }
            epi_flush_strip(strip, p, m0 + wm * 64, n0 + wn * 64, lane);      // the strip is this wave's own: no barrier
            return;
        }
    }
    // register epilogue (fp32 outputs, odd shapes): bias -> round to bf16 (the reference's Linear output) -> activation
    // -> (+ residual); the residual of a whole m-tile is requested before the first value is needed
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int m = m0 + wm * 64 + mt * 32 + (lane & 31);
        const bool mok = m < p.M;
        uint2 rq[2][4];
        if (p.R) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const int n = n0 + wn * 64 + nt * 32 + rg * 8 + half * 4;
                    rq[nt][rg] = (mok && n < p.N) ? *reinterpret_cast<const uint2*>(p.R + (size_t)m * p.ldr + n) : make_uint2(0u, 0u);
                }
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int n = n0 + wn * 64 + nt * 32 + rg * 8 + half * 4;
                const float bj[4] = {__uint_as_float(bq[nt][rg].x << 16), __uint_as_float(bq[nt][rg].x & 0xffff0000u),
                                     __uint_as_float(bq[nt][rg].y << 16), __uint_as_float(bq[nt][rg].y & 0xffff0000u)};
                const float4 c4 = (p.cscale && n < p.N) ? *reinterpret_cast<const float4*>(p.cscale + n) : make_float4(1.f, 1.f, 1.f, 1.f);
                const float cs[4] = {c4.x, c4.y, c4.z, c4.w};
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float x = acc[nt][mt][rg * 4 + e] * cs[e] + bj[e];
                    if (p.act != ACT_NONE) x = sv_act(bfround(x), p.act);
                    v[e] = x;
                }
                if (p.R) {
                    v[0] = bfround(v[0]) + __uint_as_float(rq[nt][rg].x << 16);
                    v[1] = bfround(v[1]) + __uint_as_float(rq[nt][rg].x & 0xffff0000u);
                    v[2] = bfround(v[2]) + __uint_as_float(rq[nt][rg].y << 16);
                    v[3] = bfround(v[3]) + __uint_as_float(rq[nt][rg].y & 0xffff0000u);
                }
                if (!mok || n >= p.N) continue;       // N % 4 == 0
                if constexpr (sizeof(OutT) == 4) {
                    *reinterpret_cast<float4*>((float*)p.C + (size_t)m * p.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
                    uint2 o;
                    o.x = pack2bf(v[0], v[1]);
                    o.y = pack2bf(v[2], v[3]);
                    *reinterpret_cast<uint2*>((bf16_t*)p.C + (size_t)m * p.ldc + n) = o;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// big-M GEMM, 256x256x64 tile, 8 waves in two ping-pong groups (prefill / ViT when the tile count fills the chip)
//
//   The 128^2 kernel above parks every wave at `vmcnt(0)` + barrier once per K-tile and issues its LDS reads right
//   in front of the MFMAs that need them.  Here a K-tile is cut into 2 PHASES (two 64x32 quadrants of the wave's
//   128x64 output each: 16 MFMAs 32x32x16), and every phase is
//        [ds_read this phase's fragments | LDS-DMA two half-tiles | vmcnt(8) lgkmcnt(0)]  barrier
//        [16 MFMAs at raised priority]  barrier
//   Waves 0-3 and 4-7 (one of each per SIMD) run one barrier apart, so while one group feeds the MFMA pipe the other
//   issues its LDS / LDS-DMA traffic.  The LDS-DMA queue is never drained in steady state: `vmcnt(8)` leaves the
//   four half-tiles staged in this phase and the one before in flight across the barriers.
//
//   LDS (2 K-tile buffers x 64 KiB): [X0 | X1 | W0 | W1], 16 KiB each.  X-half i = rows {128*wr + 64*i + 0..63} of
//   both wave rows (128 rows x 64 k, XOR-swizzled 16-B chunks like the 128^2 kernel); W-half j = fragment
//   (2*wc + j) of the four wave columns (packed fragments, lane-linear).  Phase 1 = quadrants (0,0) (0,1), reads W0 W1;
//   phase 2 = (1,1) (1,0), reads X1 and X0 of the NEXT K-tile.
//
//   Staging (round 6; tools/diag/gemm_lab.hip is the A/B lab, profiles/gemm_lab_r06.log the numbers): there are only two K-tile
//   buffers, but a half-tile's slot is free as soon as its fragments sit in registers, so K-tile t+2 is staged INTO THE
//   BUFFER K-TILE t IS BEING COMPUTED FROM, slot by slot, one phase after the slot's last ds_read:
//        phase 1 of t:  X1(t+1)  X0(t+2)          phase 2 of t:  W0(t+2)  W1(t+2)
//   Every half-tile is issued three phases (1.5 K-tiles) before its first read and has two phases to land before the
//   wait that retires it.  Round 2-5's loop (4 phases of 8 MFMAs, one half-tile per phase ONE K-tile ahead, vmcnt(2) after
//   every phase, wave-uniform `more ?` branches at every phase) measured 1195-1258 TF at 8192^3 on uniform random operands
//   where this one does 1315 (zero-filled: 1453 -> 1768); the same loop without LDS-DMA runs at 2190 TF and the LDS-DMA
//   stream alone at the equivalent of 2050 TF (1.05 us per K-tile): the loop is now within 16 % of max(MFMA, LDS-DMA).
//
//   Hazards (A = waves 0-3, B = waves 4-7, one barrier late; a phase's first barrier = B1, second = B2):
//     WAR  the ds_reads of a phase are waited for (`lgkmcnt(0)`) BEFORE its B1; the slot is restaged in the NEXT phase,
//          i.e. after B2 of the reading phase for group A and after the following B1 for group B: both groups' reads are
//          complete (B's reads of phase p complete before the barrier A meets as B2 of p).
//     RAW  LDS-DMA data is ordered for a ds_read only by the issuing wave's vmcnt followed by a barrier the reader has
//          passed.  The wait sits before B1 of phase p (every wave, both groups) and the data is read in phase p+1: group A's
//          reads of p+1 come after its B2 of p, which B meets as its B1 of p -- B has waited.  vmcnt(8) after the phase's own
//          two half-tiles retires everything issued two or more phases ago = exactly what phase p+1 reads.
// ------------------------------------------------------------------------------------------------
#define G2_T 256
#define G2_HALF 16384
#define G2_BUF 65536

#define G2_BARRIER() asm volatile("s_barrier" ::: "memory")

template <typename OutT>
__global__ __launch_bounds__(512) void gemm256_kernel(GemmArgs p, int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const long long t_start = p.trace ? wall_clock64() : 0;

    // XCD-aware tile order: blocks are dealt round-robin to the 8 XCDs; give each XCD a contiguous run of tile ids and
    // walk the tile grid in bands of 4 m-tiles so that one XCD round (32 CUs) covers a 4 x 8 patch sharing L2 lines
    const int T = tiles_m * tiles_n;
    const int id = blockIdx.x;
    const int q = T >> 3, rem = T & 7, xcd = id & 7, loc = id >> 3;
    const int wg = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + loc;
    const int band = wg / (4 * tiles_n), r_in = wg - band * 4 * tiles_n;
    const int band_rows = min(4, tiles_m - band * 4);
    const int tm = band * 4 + r_in % band_rows, tn = r_in / band_rows;
    const int m0 = tile_first_row(p, tm, G2_T), n0 = tn * G2_T;
    const int KS = p.K >> 4;
    const int KT = p.K >> 6;
    const int NT_total = (p.N + 31) >> 5;

    f32x16 acc[2][4];                 // [n-tile j][m-tile 2*i + mt2]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // staging addresses.  X: piece g = 2*wave + q covers local rows 8g .. 8g+7 of a half (128 rows)
    const bf16_t* xsrc[2][2];          // [half i][piece]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) {
            const int g = wave * 2 + pc;
            const int r = g * 8 + (lane >> 3);                   // 0..127 inside the half
            int grow = m0 + (r >> 6) * 128 + i * 64 + (r & 63);
            grow = grow < p.M ? grow : p.M - 1;
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            xsrc[i][pc] = p.A + (size_t)grow * p.lda + c * 8;
        }
    const bf16_t* wsrc[2][2];          // [half j][piece]: fragment f = 2*wave + pc -> (wave column f>>2, k-step f&3)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) {
            const int f = wave * 2 + pc;
            int nt = (n0 >> 5) + (f >> 2) * 2 + j;
            nt = nt < NT_total ? nt : NT_total - 1;
            wsrc[j][pc] = p.Wp + (((size_t)nt * KS + (f & 3)) * 64 + lane) * 8;
        }
    auto stage = [&](int h, int kt, char* buf) {          // h: 0 X0, 1 W0, 2 W1, 3 X1
        if (h == 0 || h == 3) {
            const int i = h == 0 ? 0 : 1;
            char* dst = buf + i * G2_HALF + wave * 2048;
            lds_dma16(xsrc[i][0] + kt * 64, dst);
            lds_dma16(xsrc[i][1] + kt * 64, dst + 1024);
        } else {
            const int j = h - 1;
            char* dst = buf + 2 * G2_HALF + j * G2_HALF + wave * 2048;
            lds_dma16(wsrc[j][0] + (size_t)kt * 4 * 512, dst);
            lds_dma16(wsrc[j][1] + (size_t)kt * 4 * 512, dst + 1024);
        }
    };
    // fragment read offsets
    int xoff[2][4];                    // [mt2][k-step] byte offset inside an X half
#pragma unroll
    for (int mt2 = 0; mt2 < 2; ++mt2)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int r = wr * 64 + mt2 * 32 + (lane & 31);
            xoff[mt2][ks] = r * 128 + (((2 * ks + (lane >> 5)) ^ ((r >> 1) & 7)) << 4);
        }
    const int woff = wc * 4096 + lane * 16;            // + ks * 1024 inside a W half

    bf16x8 x0[2][4], x1[2][4], w0[4], w1[4];
    auto read_x = [&](bf16x8 (&x)[2][4], const char* half) {
#pragma unroll
        for (int mt2 = 0; mt2 < 2; ++mt2)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) x[mt2][ks] = *reinterpret_cast<const bf16x8*>(half + xoff[mt2][ks]);
    };
    auto read_w = [&](bf16x8 (&w)[4], const char* half) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) w[ks] = *reinterpret_cast<const bf16x8*>(half + woff + ks * 1024);
    };
    // two quadrants = 16 MFMAs between a barrier pair
    auto quads = [&](bf16x8 (&wa)[4], bf16x8 (&wb)[4], bf16x8 (&x)[2][4], int ja, int jb, int i) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int mt2 = 0; mt2 < 2; ++mt2)
                acc[ja][2 * i + mt2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[ks], x[mt2][ks], acc[ja][2 * i + mt2], 0, 0, 0);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int mt2 = 0; mt2 < 2; ++mt2)
                acc[jb][2 * i + mt2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb[ks], x[mt2][ks], acc[jb][2 * i + mt2], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };

    // prologue: K-tile 0 entirely and X0 W0 W1 of K-tile 1, in the loop's issue order (X1(t), X0(t+1) | W0(t+1), W1(t+1) | ...)
    stage(0, 0, smem); stage(1, 0, smem); stage(2, 0, smem); stage(3, 0, smem);
    stage(0, 1, smem + G2_BUF); stage(1, 1, smem + G2_BUF); stage(2, 1, smem + G2_BUF);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // X0 W0 W1 of K-tile 0 have landed; X1(0) X0(1) W0(1) W1(1) stay in flight
    G2_BARRIER();
    read_x(x0, smem);                                     // X0 of K-tile 0 (later tiles: read in phase 2 of the tile before)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    G2_BARRIER();                                         // every wave holds its X0(0): phase 1 of K-tile 0 restages that slot
    if (wr == 1) G2_BARRIER();                            // group B runs one barrier behind group A
    const long long t_pro = p.trace ? wall_clock64() : 0;

    // one K-tile = two phases; MODE 0: steady state, 1: second-to-last K-tile (only X1 of the last one is still to be staged), 2: last
    auto ktile = [&](int t, auto mode_tag) {
        constexpr int MODE = decltype(mode_tag)::value;
        char* buf = smem + (t & 1) * G2_BUF;
        char* nbuf = smem + ((t + 1) & 1) * G2_BUF;
        // phase 1: quadrants (0,0) (0,1)
        read_w(w0, buf + 2 * G2_HALF); read_w(w1, buf + 3 * G2_HALF);
        if constexpr (MODE <= 1) stage(3, t + 1, nbuf);                  // X1(t+1): slot last read in phase 2 of t-1
        if constexpr (MODE == 0) stage(0, t + 2, buf);                   // X0(t+2): slot last read in phase 2 of t-1
        if constexpr (MODE == 0) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        G2_BARRIER(); quads(w0, w1, x0, 0, 1, 0); G2_BARRIER();
        // phase 2: quadrants (1,1) (1,0)
        read_x(x1, buf + G2_HALF);
        if constexpr (MODE <= 1) read_x(x0, nbuf);                        // X0 of the next K-tile
        if constexpr (MODE == 0) { stage(1, t + 2, buf); stage(2, t + 2, buf); }      // W0 W1 of t+2: slots last read in phase 1 of t
        if constexpr (MODE == 0) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        G2_BARRIER(); quads(w1, w0, x1, 1, 0, 1); G2_BARRIER();
    };
    {
        int t = 0;
        for (; t < KT - 2; ++t) ktile(t, std::integral_constant<int, 0>{});
        ktile(t, std::integral_constant<int, 1>{}); ++t;
        ktile(t, std::integral_constant<int, 2>{});
    }
    if (wr == 0) G2_BARRIER();                            // both groups execute the same number of barriers
    const long long t_loop = p.trace ? wall_clock64() : 0;
    auto stamp = [&]() {          // tools/gemm_trace.py: {start, K-tile 0 staged, K loop done, epilogue stored (drained), tile m, tile n, wave}
        if (p.trace && lane == 0 && (wave == 0 || wave == 7)) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            long long* q = p.trace + ((size_t)blockIdx.x * 2 + (wave ? 1 : 0)) * 8;
            q[0] = t_start; q[1] = t_pro; q[2] = t_loop; q[3] = wall_clock64(); q[4] = tm; q[5] = tn; q[6] = wave;
        }
    };

    const int half = lane >> 5;
    uint2 bq[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const int n = n0 + wc * 64 + j * 32 + rg * 8 + half * 4;
            bq[j][rg] = (p.bias && n < p.N) ? *reinterpret_cast<const uint2*>(p.bias + n) : make_uint2(0u, 0u);
        }
    if constexpr (sizeof(OutT) == 2) {
        if ((p.N & 7) == 0 && (p.ldc & 7) == 0 && (!p.R || (p.ldr & 7) == 0)) {
            // ---- epilogue through LDS (see epi_flush_strip), two passes of 64 rows per wave (8 waves x 9 KiB strips) ----
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            G2_BARRIER();                                      // both groups are past their last LDS reads
            char* strip = smem + wave * (64 * EPI_ROW_BYTES);
            // activation resolved once, outside the 128 unrolled values (see the 128^2 kernel): the run-time switch inside
            // the unrolled epilogue cost ~100 us per tile here (c_fc + GELU on this kernel: 400 TF instead of ~1000)
            auto run = [&](auto tag) {
                constexpr int ACT = decltype(tag)::value;
#pragma unroll
                for (int hp = 0; hp < 2; ++hp) {
#pragma unroll
                    for (int mq = 0; mq < 2; ++mq) {
                        const int mt = hp * 2 + mq;
#pragma unroll
                        for (int j = 0; j < 2; ++j)
#pragma unroll
                            for (int rg = 0; rg < 4; ++rg) {
                                const int n = n0 + wc * 64 + j * 32 + rg * 8 + half * 4;
                                const float bj[4] = {__uint_as_float(bq[j][rg].x << 16), __uint_as_float(bq[j][rg].x & 0xffff0000u),
                                                     __uint_as_float(bq[j][rg].y << 16), __uint_as_float(bq[j][rg].y & 0xffff0000u)};
                                float cs[4] = {1.f, 1.f, 1.f, 1.f};
                                if (p.cscale && n < p.N) {
                                    const float4 c4 = *reinterpret_cast<const float4*>(p.cscale + n);
                                    cs[0] = c4.x; cs[1] = c4.y; cs[2] = c4.z; cs[3] = c4.w;
                                }
                                float v[4];
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    float x = acc[j][mt][rg * 4 + e] * cs[e] + bj[e];
                                    if constexpr (ACT != ACT_NONE) x = sv_act(bfround(x), ACT);
                                    v[e] = x;
                                }
                                epi_park4(strip, mq * 32 + (lane & 31), j * 32 + rg * 8 + half * 4, v);
                            }
                    }
                    // rows of pass hp: m0 + wr*128 + hp*64 + (mq*32 + lane&31)  (mt = 2*hp + mq  ->  (mt>>1)*64 + (mt&1)*32)
                    epi_flush_strip(strip, p, m0 + wr * 128 + hp * 64, n0 + wc * 64, lane);
                }
            };
            switch (p.act) {
                case ACT_QUICKGELU: run(std::integral_constant<int, ACT_QUICKGELU>{}); break;
                case ACT_SWISH: run(std::integral_constant<int, ACT_SWISH>{}); break;
                case ACT_GELU_TANH: run(std::integral_constant<int, ACT_GELU_TANH>{}); break;
                default: run(std::integral_constant<int, ACT_NONE>{}); break;
            }
            stamp();
            return;
        }
    }
    // register epilogue (fp32 outputs, odd shapes)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int m = m0 + wr * 128 + (mt >> 1) * 64 + (mt & 1) * 32 + (lane & 31);
        const bool mok = m < p.M;
        uint2 rq[2][4];
        if (p.R) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const int n = n0 + wc * 64 + j * 32 + rg * 8 + half * 4;
                    rq[j][rg] = (mok && n < p.N) ? *reinterpret_cast<const uint2*>(p.R + (size_t)m * p.ldr + n) : make_uint2(0u, 0u);
                }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int n = n0 + wc * 64 + j * 32 + rg * 8 + half * 4;
                const float bj[4] = {__uint_as_float(bq[j][rg].x << 16), __uint_as_float(bq[j][rg].x & 0xffff0000u),
                                     __uint_as_float(bq[j][rg].y << 16), __uint_as_float(bq[j][rg].y & 0xffff0000u)};
                const float4 c4 = (p.cscale && n < p.N) ? *reinterpret_cast<const float4*>(p.cscale + n) : make_float4(1.f, 1.f, 1.f, 1.f);
                const float cs[4] = {c4.x, c4.y, c4.z, c4.w};
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float x = acc[j][mt][rg * 4 + e] * cs[e] + bj[e];
                    if (p.act != ACT_NONE) x = sv_act(bfround(x), p.act);
                    v[e] = x;
                }
                if (p.R) {
                    v[0] = bfround(v[0]) + __uint_as_float(rq[j][rg].x << 16);
                    v[1] = bfround(v[1]) + __uint_as_float(rq[j][rg].x & 0xffff0000u);
                    v[2] = bfround(v[2]) + __uint_as_float(rq[j][rg].y << 16);
                    v[3] = bfround(v[3]) + __uint_as_float(rq[j][rg].y & 0xffff0000u);
                }
                if (!mok || n >= p.N) continue;       // N % 4 == 0
                if constexpr (sizeof(OutT) == 4) {
                    *reinterpret_cast<float4*>((float*)p.C + (size_t)m * p.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
                    uint2 o;
                    o.x = pack2bf(v[0], v[1]);
                    o.y = pack2bf(v[2], v[3]);
                    *reinterpret_cast<uint2*>((bf16_t*)p.C + (size_t)m * p.ldc + n) = o;
                }
            }
        }
    }
    stamp();
}

// row tiles of the 128^2 kernel's grid (sequence structure: two per 256 rows of every sequence)
static int tiles128_m(const GemmArgs& a) { return a.seq_rows ? (a.M / a.seq_rows) * (a.seq_rows / 256) * 2 : (a.M + GB_M - 1) / GB_M; }
static void launch_gemm256(const GemmArgs& a, hipStream_t st) {
    if (a.K < 128) {                     // the loop's prologue stages into K-tile 1: a single K-tile goes through the 128^2 kernel (same bits)
        dim3 grid((a.N + GB_N - 1) / GB_N, tiles128_m(a));
        if (a.out_f32) gemm_bf16_kernel<float><<<grid, 256, 2 * GB_BUF, st>>>(a);
        else gemm_bf16_kernel<bf16_t><<<grid, 256, 2 * GB_BUF, st>>>(a);
        return;
    }
    const int tiles_m = a.seq_rows ? (a.M / a.seq_rows) * (a.seq_rows / G2_T) : (a.M + G2_T - 1) / G2_T, tiles_n = (a.N + G2_T - 1) / G2_T;
    if (a.out_f32) gemm256_kernel<float><<<tiles_m * tiles_n, 512, 2 * G2_BUF, st>>>(a, tiles_m, tiles_n);
    else gemm256_kernel<bf16_t><<<tiles_m * tiles_n, 512, 2 * G2_BUF, st>>>(a, tiles_m, tiles_n);
}

// (Round 4 tried this tile as a PERSISTENT kernel with a stream-K remainder -- one block per CU, a split tile's accumulator handed from the
// block that owns its first K-tiles to the block that continues the K loop, so the bits stay those of this kernel.  The hand-off cost
// 3 + 6 us per block, but blocks at different K offsets stop sharing A / W panels in their XCD's L2 and the K loop went from 1.64 to
// 1.85-2.29 us per K-tile: slower than the tuned forms on every prefill shape.  profiles/gemm_trace_r04.log; the kernel is in git history.)

// ------------------------------------------------------------------------------------------------
// tail rows of a big-M GEMM.  32 x 259 prompt rows are 32 full 256-row tiles plus 96 rows; run as a 33rd tile row
// those 96 rows cost every GEMM one more (nearly empty) round of the chip.  They go through this kernel instead: one
// wave per 32 x 32 output tile walking the whole K, operands straight from global memory (weight fragments are 1 KiB
// contiguous; the activation fragment is 16 B per lane from its own row), 8 k-steps of loads in flight.
// Same MFMA, same operand roles and the same ascending k order as the tile kernels, so a row's result does not depend
// on which kernel computed it (bit-identical: batch composition cannot change a token).
// ------------------------------------------------------------------------------------------------
#define GT_D 7        // chunks of 4 k-steps in flight: 7 * 8 = 56 loads (the vmcnt counter holds 63)
__global__ __launch_bounds__(64) void gemm_tail_kernel(GemmArgs p) {
    const int lane = threadIdx.x;
    const int nt = blockIdx.x, mt = blockIdx.y;
    const int KS = p.K >> 4;
    const int NCH = KS >> 2;                       // K % 64 == 0
    int row = mt * 32 + (lane & 31);
    const bool rok = row < p.M;
    row = rok ? row : p.M - 1;
    const bf16_t* xrow = p.A + (size_t)row * p.lda + 8 * (lane >> 5);
    const bf16_t* wfr = p.Wp + ((size_t)nt * KS * 64 + lane) * 8;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // a ring of GT_D chunks, statically indexed (fully unrolled): chunk c lives in slot c % GT_D.  No per-element
    // conditions: hipcc answers those with a branch and a vmcnt(0) around every load.
    bf16x8 w[GT_D][4], x[GT_D][4];
    auto load = [&](int d, int c) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            w[d][s] = *reinterpret_cast<const bf16x8*>(wfr + (size_t)(4 * c + s) * 512);
            x[d][s] = *reinterpret_cast<const bf16x8*>(xrow + (4 * c + s) * 16);
        }
    };
    auto mma = [&](int d) {
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[d][s], x[d][s], acc, 0, 0, 0);
    };
#pragma unroll
    for (int d = 0; d < GT_D; ++d)
        if (d < NCH) load(d, d);
    int c = 0;
    for (; c + 2 * GT_D <= NCH; c += GT_D) {       // steady state: every slot is refilled right after it is consumed
#pragma unroll
        for (int d = 0; d < GT_D; ++d) { mma(d); load(d, c + GT_D + d); }
    }
#pragma unroll
    for (int d = 0; d < GT_D; ++d) {               // drain (chunk-level, wave-uniform conditions only)
        if (c + d < NCH) mma(d);
        if (c + GT_D + d < NCH) load(d, c + GT_D + d);
    }
#pragma unroll
    for (int d = 0; d < GT_D; ++d)
        if (c + GT_D + d < NCH) mma(d);
    if (!rok) return;
    const int half = lane >> 5;
    const int m = row;
    // bias and residual of the lane's 16 outputs are requested together (one wave per CU: nothing else hides a load)
    uint2 bq[4], rq[4];
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
        const int n = nt * 32 + rg * 8 + half * 4;
        bq[rg] = (p.bias && n < p.N) ? *reinterpret_cast<const uint2*>(p.bias + n) : make_uint2(0u, 0u);
        rq[rg] = (p.R && n < p.N) ? *reinterpret_cast<const uint2*>(p.R + (size_t)m * p.ldr + n) : make_uint2(0u, 0u);
    }
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
        const int n = nt * 32 + rg * 8 + half * 4;
        if (n >= p.N) continue;       // N % 4 == 0
        const float bj[4] = {__uint_as_float(bq[rg].x << 16), __uint_as_float(bq[rg].x & 0xffff0000u),
                             __uint_as_float(bq[rg].y << 16), __uint_as_float(bq[rg].y & 0xffff0000u)};
        const float4 c4 = p.cscale ? *reinterpret_cast<const float4*>(p.cscale + n) : make_float4(1.f, 1.f, 1.f, 1.f);
        const float cs[4] = {c4.x, c4.y, c4.z, c4.w};
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float x = acc[rg * 4 + e] * cs[e] + bj[e];
            if (p.act != ACT_NONE) x = sv_act(bfround(x), p.act);
            v[e] = x;
        }
        if (p.R) {
            v[0] = bfround(v[0]) + __uint_as_float(rq[rg].x << 16);
            v[1] = bfround(v[1]) + __uint_as_float(rq[rg].x & 0xffff0000u);
            v[2] = bfround(v[2]) + __uint_as_float(rq[rg].y << 16);
            v[3] = bfround(v[3]) + __uint_as_float(rq[rg].y & 0xffff0000u);
        }
        if (p.out_f32) {
            *reinterpret_cast<float4*>((float*)p.C + (size_t)m * p.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
            uint2 o;
            o.x = pack2bf(v[0], v[1]);
            o.y = pack2bf(v[2], v[3]);
            *reinterpret_cast<uint2*>((bf16_t*)p.C + (size_t)m * p.ldc + n) = o;
        }
    }
}

// (Round 5 built the four-waves-per-row-tile form of this kernel -- activation fragments once per block through LDS-DMA, a 48-deep weight ring per
// wave, every wait by hand: bit-identical and slower on 7 of 8 prefill / ViT shapes, see profiles/gemm_tail4_r05_ab.log; removed.  The test that
// put it next to this kernel and the 256^2 kernel stays: forms 2 / 1 of sv_debug_set_gemm_form.)
// the remainder rows [tail x N] of a peeled GEMM (t = the GemmArgs of those rows)
static void launch_gemm_tail(const GemmArgs& t, hipStream_t st) {
    gemm_tail_kernel<<<dim3((t.N + 31) / 32, (t.M + 31) / 32), 64, 0, st>>>(t);
}

// ------------------------------------------------------------------------------------------------
// The rows a SEQUENCE leaves over its 256-row tiles (round 6).  A prompt of 259 rows (257 visual + 2) is one tile + 3 rows, a ViT
// image of 257 tokens one tile + 1 row: at batch 32 that is 96 / 32 remainder rows per GEMM, and through gemm_tail_kernel -- one
// wave per 32 x 32 tile walking the WHOLE K, because its bits had to equal the tile kernels' -- they cost 10 / 18 / 33 us (c_proj /
// c_fc / down projection: a 512-MFMA dependent chain fed by one wave's 56 loads in flight) and 1.9 ms of a 26 ms time to first token.
// Here the remainder is defined per SEQUENCE instead of per batch (GemmArgs::seq_rows: the last S % 256 rows of every sequence,
// whatever the batch), which frees the summation order: a block of 8 waves per 32 x 32 tile, wave w walks chunks
// [w * NCH / 8, (w + 1) * NCH / 8) of 64 k (a function of K alone), the eight partial tiles meet in LDS and are summed in wave
// order.  Which kernel computes a row -- and in what order its k are summed -- depends on the row's position in its sequence only:
// a sequence's tokens stay independent of the batch it shares (the property the tail kernel's bit-identity bought, kept by
// construction instead).  The pruned last prompt layer (one row per sequence, GemmArgs::splitk_rows) takes the same kernel when its
// rows are such remainder rows.
// ------------------------------------------------------------------------------------------------
#define GTK_WAVES 8
template <int D>
__global__ __launch_bounds__(GTK_WAVES * 64) void gemm_tailk_kernel(GemmArgs p) {
    __shared__ float red[GTK_WAVES][16][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nt = blockIdx.x, mt = blockIdx.y;
    const int KS = p.K >> 4;
    const int NCH = KS >> 2;                       // K % 64 == 0
    const int c0 = wave * NCH / GTK_WAVES, n = (wave + 1) * NCH / GTK_WAVES - c0;
    int i = mt * 32 + (lane & 31);
    const bool rok = i < p.M;
    i = rok ? i : p.M - 1;
    const int row = p.seq_tail ? (i / p.seq_tail) * p.seq_rows + (p.seq_rows - p.seq_tail) + i % p.seq_tail : i;
    const bf16_t* xrow = p.A + (size_t)row * p.lda + 8 * (lane >> 5) + c0 * 64;
    const bf16_t* wfr = p.Wp + (((size_t)nt * KS + c0 * 4) * 64 + lane) * 8;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // a ring of D chunks of 4 k-steps, statically indexed (see gemm_tail_kernel)
    bf16x8 w[D][4], x[D][4];
    auto load = [&](int d, int c) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            w[d][s] = *reinterpret_cast<const bf16x8*>(wfr + (size_t)(4 * c + s) * 512);
            x[d][s] = *reinterpret_cast<const bf16x8*>(xrow + (4 * c + s) * 16);
        }
    };
    auto mma = [&](int d) {
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[d][s], x[d][s], acc, 0, 0, 0);
    };
#pragma unroll
    for (int d = 0; d < D; ++d)
        if (d < n) load(d, d);
    int c = 0;
    for (; c + 2 * D <= n; c += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) { mma(d); load(d, c + D + d); }
    }
#pragma unroll
    for (int d = 0; d < D; ++d) {
        if (c + d < n) mma(d);
        if (c + D + d < n) load(d, c + D + d);
    }
#pragma unroll
    for (int d = 0; d < D; ++d)
        if (c + D + d < n) mma(d);
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
    __syncthreads();
    if (wave >= 4 || !rok) return;
    // waves 0..3 finish register group rg = wave of every lane: 4 consecutive columns of the lane's row, partial tiles summed in wave order
    const int rg = wave, half = lane >> 5, m = row;
    const int ncol = nt * 32 + rg * 8 + half * 4;
    if (ncol >= p.N) return;              // N % 4 == 0
    const uint2 bq = p.bias ? *reinterpret_cast<const uint2*>(p.bias + ncol) : make_uint2(0u, 0u);
    const uint2 rq = p.R ? *reinterpret_cast<const uint2*>(p.R + (size_t)m * p.ldr + ncol) : make_uint2(0u, 0u);
    const float4 c4 = p.cscale ? *reinterpret_cast<const float4*>(p.cscale + ncol) : make_float4(1.f, 1.f, 1.f, 1.f);
    const float cs[4] = {c4.x, c4.y, c4.z, c4.w};
    const float bj[4] = {__uint_as_float(bq.x << 16), __uint_as_float(bq.x & 0xffff0000u), __uint_as_float(bq.y << 16), __uint_as_float(bq.y & 0xffff0000u)};
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float t = red[0][rg * 4 + e][lane];
#pragma unroll
        for (int ww = 1; ww < GTK_WAVES; ++ww) t += red[ww][rg * 4 + e][lane];
        float xv = t * cs[e] + bj[e];
        if (p.act != ACT_NONE) xv = sv_act(bfround(xv), p.act);
        v[e] = xv;
    }
    if (p.R) {
        v[0] = bfround(v[0]) + __uint_as_float(rq.x << 16);
        v[1] = bfround(v[1]) + __uint_as_float(rq.x & 0xffff0000u);
        v[2] = bfround(v[2]) + __uint_as_float(rq.y << 16);
        v[3] = bfround(v[3]) + __uint_as_float(rq.y & 0xffff0000u);
    }
    if (p.out_f32) {
        *reinterpret_cast<float4*>((float*)p.C + (size_t)m * p.ldc + ncol) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
        uint2 o;
        o.x = pack2bf(v[0], v[1]);
        o.y = pack2bf(v[2], v[3]);
        *reinterpret_cast<uint2*>((bf16_t*)p.C + (size_t)m * p.ldc + ncol) = o;
    }
}
// t: the GemmArgs of the remainder rows (M = their number; seq_tail / seq_rows = the row map, or compact rows)
static void launch_gemm_tailk(const GemmArgs& t, hipStream_t st) {
    // ring depth 3 / 4 / 6 chunks measured equal at every prefill shape (profiles/gemm_seq_remainder_r06.log): the launch is bound by its L2 traffic
    gemm_tailk_kernel<4><<<dim3((t.N + 31) / 32, (t.M + 31) / 32), GTK_WAVES * 64, 0, st>>>(t);
}

// Pick the tile kernel, and decide whether to peel a small row remainder, by a cost model fitted to measurements at
// the prefill / ViT shapes (tools/bench_gemm.py).  A "round" is one wave of tiles over the chip (128^2: 2 blocks per
// CU, 256^2: 1 block per CU); the last round costs as much as a full one, which is the whole reason for peeling.
static double tiles_us(int M, int N, int K, int act, bool* use256) {
    const int cus = 256;
    const double k = (double)K;
    const long t128 = (long)((M + 127) / 128) * ((N + 127) / 128);
    const long t256 = (long)((M + 255) / 256) * ((N + 255) / 256);
    // 128^2 (2 blocks per CU smooth the schedule): between the fractional and the whole number of rounds
    const double f128 = (double)t128 / (2 * cus), c128 = (double)((t128 + 2 * cus - 1) / (2 * cus));
    const double r128 = f128 < 1.0 ? 1.0 : 0.5 * (f128 + c128);
    const double r256 = (double)((t256 + cus - 1) / cus);
    const double us128 = r128 * 0.0195 * k * (act ? 1.12 : 1.0) + 10.0;
    const double us256 = r256 * (0.0243 * k + 33.0);
    // the fit comes from chip-filling grids: a 256^2 grid that leaves CUs idle (few requests) measured slower than 128^2
    const bool u = us256 < us128 && t256 >= 200;
    if (use256) *use256 = u;
    return u ? us256 : us128;
}
static double tail_us(int tail, int N, int K) {
    return 7.0 + 4.2e-5 * (double)((N + 31) / 32) * (double)((tail + 31) / 32) * (double)K;
}

static void launch_gemm_tiles(const GemmArgs& a, hipStream_t st, bool force128 = false) {
    dim3 grid((a.N + GB_N - 1) / GB_N, tiles128_m(a));
    bool use256 = false;
    (void)tiles_us(a.seq_rows ? (int)grid.y * GB_M : a.M, a.N, a.K, a.act, &use256);
    if (!force128 && use256) { launch_gemm256(a, st); return; }
    if (a.out_f32) gemm_bf16_kernel<float><<<grid, 256, 2 * GB_BUF, st>>>(a);
    else gemm_bf16_kernel<bf16_t><<<grid, 256, 2 * GB_BUF, st>>>(a);
}

// The dispatch decision for an M x N x K big-M GEMM as plain host arithmetic (exported for the CPU tests through
// sv_debug_gemm_plan): peel the row remainder or not, how the remainder runs, and which tile kernel the main part takes.
GemmPlan gemm_plan(int M, int N, int K, int act, int tail_on) {
    GemmPlan pl;
    const int tail = M % 256, main_rows = M - tail;
    bool peel = tail_on && tail > 0 && tail <= 96 && main_rows >= 2048;
    // two ways to do the remainder: one wave per 32x32 tile (good for few column tiles / long K), or one row of 128^2
    // tiles (LDS-shared operands: good for wide N, but a single block per 128 columns walks the whole K alone)
    const double t_wave = tail_us(tail, N, K), t_tile = 10.0 + 0.03 * (double)K;   // measured: a lone K=2048 tile row ~70 us
    const bool tail_by_tiles = t_tile < t_wave;
    if (peel && tail_on == 1)
        peel = tiles_us(M, N, K, act, nullptr) - tiles_us(main_rows, N, K, act, nullptr) > (tail_by_tiles ? t_tile : t_wave);
    pl.peel = peel ? 1 : 0;
    pl.tail_rows = peel ? tail : 0;
    pl.tail_by_tiles = peel && tail_by_tiles ? 1 : 0;
    bool use256 = false;
    pl.est_us = tiles_us(peel ? main_rows : M, N, K, act, &use256) + (peel ? (tail_by_tiles ? t_tile : t_wave) : 0.0);
    pl.main_256 = use256 ? 1 : 0;
    return pl;
}

// One launch of the chosen configuration: tile kernel (kernel = 0: 128^2, 1: 256^2) and whether the row remainder over a
// multiple of 256 is peeled into the tail kernel.  Every configuration computes the same bits (same MFMA, operand roles
// and ascending-k order: tests/test_gpu_ops.py::test_linear_big_m_kernels_agree_bitwise), so the choice is speed only.
static void launch_gemm_config(const GemmArgs& a, hipStream_t st, int kernel256, bool peel, bool tail_by_tiles) {
    auto tiles = [&](const GemmArgs& g, bool force128) {
        dim3 grid((g.N + GB_N - 1) / GB_N, tiles128_m(g));
        if (kernel256 && !force128) { launch_gemm256(g, st); return; }
        if (g.out_f32) gemm_bf16_kernel<float><<<grid, 256, 2 * GB_BUF, st>>>(g);
        else gemm_bf16_kernel<bf16_t><<<grid, 256, 2 * GB_BUF, st>>>(g);
    };
    if (a.seq_rows) {                    // (sanitised by the caller: the rule holds) tiles over the full 256-row tiles of every sequence + the split-K remainder
        tiles(a, false);
        GemmArgs t = a;
        t.seq_tail = seq_peel_rows(a.seq_rows);
        t.M = (a.M / a.seq_rows) * t.seq_tail;
        if (a.tail_mark) a.tail_mark(a.tail_ctx, st);
        launch_gemm_tailk(t, st);
        return;
    }
    const int tail = a.M % 256, main_rows = a.M - tail;
    if (peel && tail > 0 && main_rows > 0) {
        GemmArgs m = a;
        m.M = main_rows;
        tiles(m, false);
        GemmArgs t = a;
        t.A = a.A + (size_t)main_rows * a.lda;
        t.R = a.R ? a.R + (size_t)main_rows * a.ldr : nullptr;
        t.C = a.out_f32 ? (void*)((float*)a.C + (size_t)main_rows * a.ldc) : (void*)((bf16_t*)a.C + (size_t)main_rows * a.ldc);
        t.M = tail;
        if (a.tail_mark) a.tail_mark(a.tail_ctx, st);
        if (tail_by_tiles) tiles(t, true);
        else launch_gemm_tail(t, st);
        return;
    }
    tiles(a, false);
}

// "Measure, don't guess": the first time a big-M shape shows up (outside a stream capture) the candidate configurations are
// timed on the real operands into a scratch output (the residual operand may alias the real output, so the real one is not
// touched), and the fastest is remembered for the process.  The analytic model above (gemm_plan) stays the choice for small
// problems, inside captures and when SV_GEMM_AUTOTUNE=0; it is also what the CPU tests pin.
// The key buckets M (rows rounded up to 1024, plus whether a peelable remainder exists): variable prompt lengths / admit sizes
// of a serving process map to a bounded set of keys, so a live request stream does not keep re-tuning (and g_tune stays small).
struct TuneKey {
    int Mb, peelable, N, K, act, res, f32, fp8;
    bool operator<(const TuneKey& o) const {
        return std::tie(Mb, peelable, N, K, act, res, f32, fp8) < std::tie(o.Mb, o.peelable, o.N, o.K, o.act, o.res, o.f32, o.fp8);
    }
};
static std::mutex g_tune_mu;
static std::map<TuneKey, int> g_tune;          // bit 0: 256^2 kernel, bit 1: peel, bit 2: tail as a row of 128^2 tiles
struct TuneScratch { void* p = nullptr; size_t bytes = 0; };
static std::map<int, TuneScratch> g_tune_scratch;      // per device: grow-only scratch output of the timing runs (no hipFree per shape = no device sync)

static int autotune_gemm(const GemmArgs& a, hipStream_t st, const GemmPlan& model) {
    const int fallback = (model.main_256 ? 1 : 0) | (model.peel ? 2 : 0) | (model.tail_by_tiles ? 4 : 0);
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return fallback; }
    const size_t esz = a.out_f32 ? 4 : 2;
    const size_t need = (size_t)a.M * a.ldc * esz;
    int dev_id = 0;
    if (hipGetDevice(&dev_id) != hipSuccess) { (void)hipGetLastError(); return fallback; }
    TuneScratch& ts = g_tune_scratch[dev_id];           // (the caller holds g_tune_mu)
    if (need > ts.bytes) {
        void* bigger = nullptr;
        if (hipMalloc(&bigger, need + need / 2) != hipSuccess) { (void)hipGetLastError(); return fallback; }
        if (ts.p) (void)hipFree(ts.p);
        ts.p = bigger; ts.bytes = need + need / 2;
    }
    void* scratch = ts.p;
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return fallback;
    GemmArgs t = a;
    t.C = scratch;
    t.tail_mark = nullptr;                       // the timing runs are not part of anybody's profile
    const int tail = a.M % 256, main_rows = a.M - tail;
    const bool can_peel = !a.seq_rows && tail > 0 && tail <= 96 && main_rows >= 2048;
    const long t256 = (long)(a.seq_rows ? (a.M / a.seq_rows) * (a.seq_rows / 256) : (a.M + 255) / 256) * ((a.N + 255) / 256);
    int best = fallback;
    float best_ms = 1e30f;
    for (int k256 = 0; k256 < 2; ++k256) {
        if (k256 && t256 < 100) continue;               // a 256^2 grid that leaves most CUs idle is never the answer
        for (int peel = 0; peel < (can_peel ? 2 : 1); ++peel) {
            const int cfg = k256 | (peel ? 2 : 0) | (peel && model.tail_by_tiles ? 4 : 0);
            launch_gemm_config(t, st, k256, peel != 0, (cfg & 4) != 0);                 // warm-up
            (void)hipEventRecord(e0, st);
            for (int r = 0; r < 3; ++r) launch_gemm_config(t, st, k256, peel != 0, (cfg & 4) != 0);
            (void)hipEventRecord(e1, st);
            if (hipEventSynchronize(e1) != hipSuccess) { (void)hipGetLastError(); continue; }
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, e0, e1) == hipSuccess && ms < best_ms) { best_ms = ms; best = cfg; }
        }
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (getenv("SV_GEMM_AUTOTUNE_LOG"))
        fprintf(stderr, "[sv gemm autotune] M %d N %d K %d act %d res %d%s -> %s%s (%.1f us; model said %s%s)\n", a.M, a.N, a.K, a.act,
                a.R ? 1 : 0, a.seq_rows ? " seq" : "", (best & 1) ? "256^2" : "128^2", (best & 2) ? " + peeled tail" : "", best_ms * 1000.f / 3.f,
                (fallback & 1) ? "256^2" : "128^2", (fallback & 2) ? " + peeled tail" : "");
    return best;
}

// the per-sequence form: where the cost model peels the remainder of a reference batch of 32 sequences (measured, tools/tail_ab.py /
// profiles/gemm_seq_remainder_r06.log: the split-K remainder launch beats the one-wave-per-tile one where there WAS a remainder launch -- c_proj,
// down projection, the ViT's out / MLP projections, the adapter -- and costs a launch where the rows were not peeled -- c_attn, the ViT's in_proj)
bool gemm_seq_form(int S, int N, int K, int act) {
    if (S <= 0 || !seq_peel_rows(S)) return false;
    return gemm_plan(32 * S, N, K, act, 1).peel != 0;
}
// the sequence structure is used only where its rule holds and the rows are whole sequences
static GemmArgs seq_sanitised(const GemmArgs& a0) {
    GemmArgs a = a0;
    const bool form = a.seq_rows > 0 && gemm_seq_form(a.seq_rows, a.N, a.K, a.act);
    if (a.splitk_rows) { a.splitk_rows = form ? 1 : 0; a.seq_rows = 0; }       // compact last rows: the remainder kernel iff the full problem would use it for them
    else if (!form || a.M % a.seq_rows) a.seq_rows = 0;
    a.seq_tail = 0;
    return a;
}
static void launch_gemm_model(const GemmArgs& a, const GemmPlan& pl, hipStream_t st);
// every row through the one-wave-per-tile kernel (bit-identical to the tile kernels); with the sequence structure: the full 256-row tiles' rows of
// every sequence that way, the rows a sequence leaves over through the per-sequence split-K remainder kernel as in every other form
static void launch_gemm_rows_by_tail(const GemmArgs& a, hipStream_t st) {
    if (!a.seq_rows) { launch_gemm_tail(a, st); return; }
    const int nseq = a.M / a.seq_rows, main = 256 * (a.seq_rows / 256);
    for (int sq = 0; sq < nseq; ++sq) {
        GemmArgs m = a;
        const size_t r0 = (size_t)sq * a.seq_rows;
        m.seq_rows = 0;
        m.A = a.A + r0 * a.lda;
        m.R = a.R ? a.R + r0 * a.ldr : nullptr;
        m.C = a.out_f32 ? (void*)((float*)a.C + r0 * a.ldc) : (void*)((bf16_t*)a.C + r0 * a.ldc);
        m.M = main;
        m.tail_mark = nullptr;
        if (main > 0) launch_gemm_tail(m, st);
    }
    GemmArgs t = a;
    t.seq_tail = seq_peel_rows(a.seq_rows);
    t.M = nseq * t.seq_tail;
    if (a.tail_mark) a.tail_mark(a.tail_ctx, st);
    launch_gemm_tailk(t, st);
}
// the small-M decision, measured: bit 4 (16) = rows by the one-wave-per-tile kernel, 0 = the cost model's tiles
static int autotune_small(const GemmArgs& a, hipStream_t st, const GemmPlan& model) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return 0; }
    const size_t esz = a.out_f32 ? 4 : 2;
    const size_t need = (size_t)a.M * a.ldc * esz;
    int dev_id = 0;
    if (hipGetDevice(&dev_id) != hipSuccess) { (void)hipGetLastError(); return 0; }
    TuneScratch& ts = g_tune_scratch[dev_id];           // (the caller holds g_tune_mu)
    if (need > ts.bytes) {
        void* bigger = nullptr;
        if (hipMalloc(&bigger, need + need / 2) != hipSuccess) { (void)hipGetLastError(); return 0; }
        if (ts.p) (void)hipFree(ts.p);
        ts.p = bigger; ts.bytes = need + need / 2;
    }
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return 0;
    GemmArgs t = a;
    t.C = ts.p;
    t.tail_mark = nullptr;
    float ms[2] = {1e30f, 1e30f};
    for (int form = 0; form < 2; ++form) {
        auto run = [&]() { if (form) launch_gemm_rows_by_tail(t, st); else launch_gemm_model(t, model, st); };
        run();                                               // warm-up
        (void)hipEventRecord(e0, st);
        for (int r = 0; r < 3; ++r) run();
        (void)hipEventRecord(e1, st);
        if (hipEventSynchronize(e1) != hipSuccess) { (void)hipGetLastError(); continue; }
        float v = 0.f;
        if (hipEventElapsedTime(&v, e0, e1) == hipSuccess) ms[form] = v;
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    const int best = ms[1] < ms[0] ? 16 : 0;
    if (getenv("SV_GEMM_AUTOTUNE_LOG"))
        fprintf(stderr, "[sv gemm autotune] M %d N %d K %d act %d res %d%s -> %s (tiles %.1f us, one wave per tile %.1f us)\n", a.M, a.N, a.K, a.act, a.R ? 1 : 0,
                a.seq_rows ? " seq" : "", best ? "one wave per 32 x 32 tile" : "tiles", ms[0] * 1000.f / 3.f, ms[1] * 1000.f / 3.f);
    return best;
}

void launch_gemm_fixed(const GemmArgs& a0, int kernel256, int peel, hipStream_t st) {
    const GemmArgs a = seq_sanitised(a0);
    if (a.splitk_rows) { launch_gemm_tailk(a, st); return; }
    launch_gemm_config(a, st, kernel256 != 0, peel != 0, false);
}

static std::atomic<int> g_gemm_form{-1};        // test surface (sv_debug_set_gemm_form): -1 = tuned, 0 / 1 = one fixed form
void set_gemm_form(int form) { g_gemm_form = form; }

void launch_gemm(const GemmArgs& a0, hipStream_t st) {
    const GemmArgs a = seq_sanitised(a0);
    if (a.splitk_rows) { launch_gemm_tailk(a, st); return; }       // compact remainder rows (the pruned last prompt layer)
    const int fixed = g_gemm_form.load();
    // forms 0 / 1: one tile kernel, rows not peeled; 2: 256^2 tiles + the row remainder through the tail kernel -- the test surface that puts
    // every kernel next to the others
    if (fixed >= 2) { launch_gemm_fixed(a, 1, 1, st); return; }
    if (fixed >= 0) { launch_gemm_fixed(a, fixed, 0, st); return; }
    // a handful of rows (the pruned last layer of a prompt pass: one row per sequence): one wave per 32 x 32 tile with 56 loads in flight
    // beats a single row of 128^2 tiles that walk the whole K alone -- same bits either way
    if (a.M <= 96 && tail_us(a.M, a.N, a.K) < 10.0 + 0.03 * (double)a.K) { launch_gemm_tail(a, st); return; }
    const int Mt = a.seq_rows ? (a.M / a.seq_rows) * (a.seq_rows / 256) * 256 : a.M;       // rows the tile kernels cover
    const GemmPlan pl = gemm_plan(Mt, a.N, a.K, a.act, 1);
    static const bool tune_on = !(getenv("SV_GEMM_AUTOTUNE") && atoi(getenv("SV_GEMM_AUTOTUNE")) == 0);
    if (tune_on && a.M >= 1024 && (long)a.M * a.N >= (1L << 22)) {
        const int tail = a.M % 256;
        const TuneKey key{(a.M + 1023) / 1024, a.seq_rows ? 2 : (tail > 0 && tail <= 96 && a.M - tail >= 2048) ? 1 : 0, a.N, a.K, a.act, a.R ? 1 : 0,
                          a.out_f32, a.cscale ? 1 : 0};
        int cfg;
        {
            std::lock_guard<std::mutex> lk(g_tune_mu);        // held across the timing runs: one tuner at a time, one scratch
            auto it = g_tune.find(key);
            if (it != g_tune.end()) cfg = it->second;
            else { cfg = autotune_gemm(a, st, pl); g_tune[key] = cfg; }
        }
        launch_gemm_config(a, st, cfg & 1, (cfg & 2) != 0, (cfg & 4) != 0);
        return;
    }
    // A FEW HUNDRED rows (round 6, third session: the prompt pass of one to three requests -- what the reference's own callers run): 2 - 9 row tiles
    // of 128^2 are 16 - 54 blocks on 256 CUs, each walking the whole K alone (down projection at 259 rows: 108 us).  The one-wave-per-tile kernel --
    // same MFMA, operand roles and ascending k as the tile kernels, hence the same bits, which is why it may be chosen by speed alone and a row's
    // result still does not depend on the batch it arrives in -- spreads the same rows over M / 32 x N / 32 waves.  Measured once per shape like
    // the big-M forms (the weights' re-reads per row tile decide; no model is trusted here).
    if (tune_on && a.M > 96 && a.M < 1024 && (!a.seq_rows || a.M / a.seq_rows <= 4)) {
        const TuneKey key{-((a.M + 63) / 64), a.seq_rows ? 2 : 0, a.N, a.K, a.act, a.R ? 1 : 0, a.out_f32, a.cscale ? 1 : 0};
        int cfg;
        {
            std::lock_guard<std::mutex> lk(g_tune_mu);
            auto it = g_tune.find(key);
            if (it != g_tune.end()) cfg = it->second;
            else { cfg = autotune_small(a, st, pl); g_tune[key] = cfg; }
        }
        if (cfg & 16) { launch_gemm_rows_by_tail(a, st); return; }
    }
    launch_gemm_model(a, pl, st);
}

// the untuned choice: the cost model's tile kernel, its peel decision, the per-sequence remainder
static void launch_gemm_model(const GemmArgs& a, const GemmPlan& pl, hipStream_t st) {
    if (a.seq_rows) { launch_gemm_config(a, st, pl.main_256, false, false); return; }
    const int tail = a.M % 256, main_rows = a.M - tail;
    const bool peel = pl.peel != 0, tail_by_tiles = pl.tail_by_tiles != 0;
    if (peel) {
        GemmArgs m = a;
        m.M = main_rows;
        launch_gemm_tiles(m, st);
        GemmArgs t = a;
        t.A = a.A + (size_t)main_rows * a.lda;
        t.R = a.R ? a.R + (size_t)main_rows * a.ldr : nullptr;
        t.C = a.out_f32 ? (void*)((float*)a.C + (size_t)main_rows * a.ldc) : (void*)((bf16_t*)a.C + (size_t)main_rows * a.ldc);
        t.M = tail;
        if (a.tail_mark) a.tail_mark(a.tail_ctx, st);
        if (tail_by_tiles) launch_gemm_tiles(t, st, true);
        else launch_gemm_tail(t, st);
        return;
    }
    launch_gemm_tiles(a, st);
}

// ------------------------------------------------------------------------------------------------
// skinny GEMM (decode step)
//   out[m][n] = epi( sum_k x[m][k] W[n][k] ),  M <= 32 rows per tile, HBM-bound weight streaming.
//   * weights and activations both in MFMA fragment order: every wave load is one contiguous 1 KiB;
//   * K split across the waves of a block (LDS reduce in wave order; every wave finishes 16 / WAVES of the
//     accumulator rows, so the tail of the kernel is WAVES times shorter than a wave-0 epilogue) and, for
//     narrow outputs, across `splitk` blocks: each block writes an fp32 slab and the CONSUMER (decode
//     attention / row update) sums the slabs in slab order -- no hand-off inside the launch, bitwise deterministic;
//   * outputs: fp32 slabs (c_attn, both c_proj); bias + activation -> fragment-order bf16 (c_fc);
//     fp32 logits rounded to bf16 values (lm_head).
//   Round 1-2 variants that were measured and lost (LayerNorm prologue, ticket-merged split-K with fused residual
//   epilogues, the row update inside the consumer launch, two column tiles per wave at <= 32 rows, four register
//   chunks in flight, full-K blocks) are in git history and in profiles/SUMMARY_r02.md, not in the library.  (Column
//   tiles per block came back in round 3 for 33..64 rows, where the activation re-reads do bound the launch:
//   gemm_skinny_mt2_kernel.)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) { return pack2bf(lo, hi); }   // common.h: v_cvt_pk_bf16_f32

// 8 e4m3 bytes (one lane's share of a k-step) -> the bf16 MFMA operand.  gfx950 converts two fp8 to a packed bf16 pair in ONE
// instruction (v_cvt_scalef32_pk_bf16_fp8, scale 1.0: exact, every e4m3 value is a bf16 value); round 2 went through float
// (v_cvt_pk_f32_fp8 + shift / and-or per pair: 12 VALU instructions per k-step against these 4, next to 2 MFMAs of 8 passes).
typedef __bf16 sv_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u32x4 fp8x8_to_bf16x8(uint32_t lo, uint32_t hi) {
    const sv_bf16x2 a0 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(lo, 1.0f, false), a1 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(lo, 1.0f, true);
    const sv_bf16x2 a2 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(hi, 1.0f, false), a3 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(hi, 1.0f, true);
    union { sv_bf16x2 b; uint32_t u; } c0, c1, c2, c3;
    c0.b = a0; c1.b = a1; c2.b = a2; c3.b = a3;
    u32x4 wf = {c0.u, c1.u, c2.u, c3.u};
    return wf;
}

// one chunk of the weight / activation stream held in registers
template <int CH>
struct SkChunk {
    u32x4 w[CH];
    u32x4 x[CH];
};

template <int CH>
__device__ __forceinline__ void sk_load(SkChunk<CH>& c, const u32x4* wptr, const u32x4* xptr, int ks, int ks_end) {
#pragma unroll
    for (int u = 0; u < CH; ++u)
        if (ks + u < ks_end) c.w[u] = __builtin_nontemporal_load(wptr + (size_t)(ks + u) * 64);   // streamed once
#pragma unroll
    for (int u = 0; u < CH; ++u)
        if (ks + u < ks_end) c.x[u] = xptr[(size_t)(ks + u) * 64];                                 // wave-uniform guard
}

// the same without guards: the chunk lies inside the wave's k range.  The steady-state loops use this one on purpose: with the
// wave-uniform guards of sk_load every load sits behind a branch, the compiler's wait-count analysis merges "loaded" and "not
// loaded" paths and falls back to s_waitcnt vmcnt(0) in front of EVERY chunk's MFMAs -- the wave then waits for the loads it has
// just issued (two chunks "in flight" were one round trip per chunk pair; found in round 3 from the ISA).
template <int CH>
__device__ __forceinline__ void sk_load_full(SkChunk<CH>& c, const u32x4* wptr, const u32x4* xptr, int ks) {
#pragma unroll
    for (int u = 0; u < CH; ++u) c.w[u] = __builtin_nontemporal_load(wptr + (size_t)(ks + u) * 64);
#pragma unroll
    for (int u = 0; u < CH; ++u) c.x[u] = xptr[(size_t)(ks + u) * 64];
}

// the bias of this lane's RPW output columns, requested BEFORE the weight stream (the epilogue must not start with a round trip)
template <int RPW>
__device__ __forceinline__ void sk_bias(const SkinnyArgs& p, float* bias_d, int r0, int nt, int half, float* fc1 = nullptr) {
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        bias_d[i] = 0.f;
        if (fc1) fc1[i] = 0.f;
        const int r = r0 + i;
        const int n = nt * 32 + 8 * (r >> 2) + 4 * half + (r & 3);
        if (p.out_mode == SK_OUT_PACKED_ACT && n < p.N) {
            if (fc1 && p.fold_c1) { fc1[i] = p.fold_c1[n]; bias_d[i] = p.fold_c2[n]; }
            else if (p.bias) bias_d[i] = bf2f(p.bias[n]);
        }
    }
}
// The epilogue operands must have LANDED before the k loop starts: a load that stays pending across the loop makes the compiler's
// wait-count analysis give up on the loop (an unbounded distance to the oldest pending load) and put s_waitcnt vmcnt(0) at the
// top of every iteration.  An empty asm that "uses" the values forces the wait here, once, in the prologue.
template <int RPW>
__device__ __forceinline__ void sk_settle(float* a, float* b = nullptr) {
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        asm volatile("" : "+v"(a[i]));
        if (b) asm volatile("" : "+v"(b[i]));
    }
}
// shared tail of the skinny kernels: v[i] = the reduced accumulator row r0 + i of lane (m, half), i.e. output
// column n(r) = nt*32 + 8*(r >> 2) + 4*half + (r & 3) of row mt*32 + m; RPW consecutive rows r (RPW in {1, 2, 4, 8, 16})
// fold: LayerNorm applied algebraically (decode_cols.hip): x = rstd * (acc - mean * c1[n]) + c2[n]; fc1 / fc2 = this lane's RPW
// values of c1 / c2, (mean, rstd) = the statistics of row m
struct SkFold { bool on; float mean, rstd; };
template <int RPW>
__device__ __forceinline__ void sk_store(const SkinnyArgs& p, float* v, const float* bias_d, int r0, int nt, int mt, int split, int m,
                                         int half, const SkFold fold = SkFold{false, 0.f, 1.f}, const float* fc1 = nullptr) {
    constexpr int G = RPW >= 4 ? RPW / 4 : 1;          // groups of (up to) 4 consecutive columns
    constexpr int W = RPW >= 4 ? 4 : RPW;
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const int r = r0 + 4 * g;
        const int n0 = nt * 32 + 8 * (r >> 2) + 4 * half + (r & 3);
        float* vv = v + 4 * g;
        auto store_f32 = [&](float* dst) {
            if constexpr (W == 4) *reinterpret_cast<float4*>(dst) = make_float4(vv[0], vv[1], vv[2], vv[3]);
            else if constexpr (W == 2) *reinterpret_cast<float2*>(dst) = make_float2(vv[0], vv[1]);
            else dst[0] = vv[0];
        };
        // TWO separate stores on purpose: written as one store through `cond ? ws + .. : out_f32 + ..` (or an if / else that only
        // picks the pointer) hipcc 7.2 keeps the out_f32 base register for both arms and the slab store goes to a null pointer
        // (seen twice: the fp8 kernel in round 2, this helper in round 3 -- a GPU memory fault at 0x1000).
        if (p.out_mode == SK_OUT_PARTIAL) {
            store_f32(p.ws + ((size_t)split * p.MT * 32 + mt * 32 + m) * p.ldws + n0);
        } else if (p.out_mode == SK_OUT_F32) {
            if (p.round_bf16) {
#pragma unroll
                for (int i = 0; i < W; ++i) vv[i] = bfround(vv[i]);
            }
            store_f32(p.out_f32 + ((size_t)mt * 32 + m) * p.ldo + n0);
        } else {   // SK_OUT_PACKED_ACT
#pragma unroll
            for (int i = 0; i < W; ++i) {
                float x = 0.f;
                if (n0 + i < p.N) {
                    // (fold: bias_d carries c2 = sum_k beta_k W[n][k] + bias[n])
                    x = fold.on ? bfround(fold.rstd * (vv[i] - fold.mean * fc1[4 * g + i]) + bias_d[4 * g + i])
                                : bfround(vv[i] + bias_d[4 * g + i]);
                    if (p.act != ACT_NONE) x = sv_act(x, p.act);
                }
                vv[i] = x;
            }
            bf16_t* dst = p.out_xp + xp_index(mt, p.out_KS, m, n0);
            if constexpr (W == 4) {
                uint2 o; o.x = pack2bf(vv[0], vv[1]); o.y = pack2bf(vv[2], vv[3]);
                *reinterpret_cast<uint2*>(dst) = o;
            } else if constexpr (W == 2) {
                *reinterpret_cast<uint32_t*>(dst) = pack2bf(vv[0], vv[1]);
            } else {
                dst[0] = f2bf(vv[0]);
            }
        }
    }
}

// Leading scalar parameters = the few values the first address computation needs: they are PRELOADED into SGPRs by the command
// processor (build flag -amdgpu-kernarg-preload-count; only scalar / pointer parameters qualify, not a by-value struct), so the
// weight stream is requested without first waiting for a scalar load of the argument block (a cold K$ miss at every launch).
struct SkinnyKernarg { const void* W; const bf16_t* x; int KS; int ks_per_split; int flags; SkinnyArgs p; };      // the kernarg segment of the skinny kernels
// FOLD: the LayerNorm-folded c_fc (decode_cols.hip) -- its own instantiation, so that the statistics registers do not cost the
// other GEMMs their second block per CU (<= 128 VGPRs)
// HEAD: the lm_head launch of a step that folds the greedy selection into the epilogue and / or arms the polled buffer of the next
// step's fused row-update launch (SkinnyArgs::amax / ::poison) -- its own instantiation as well: carried as run-time branches by every
// skinny launch, the two cost StarVector-8B's 129 GEMM launches per step 0.2 us each (3982 vs 3950 us per step, same box: round 5)
template <int WAVES, bool FOLD = false, bool HEAD = false>
__global__ __launch_bounds__(WAVES * 64) void gemm_skinny_kernel(const bf16_t* Wp_, const bf16_t* xp_, int KS_, int ks_per_split_, int flags_, SkinnyArgs p_unused) {
    constexpr int CH = 4;                            // k-steps per register chunk (two chunks = 8 KiB of W in flight per wave)
    constexpr int NB = 2;
    extern __shared__ __attribute__((aligned(16))) char sk_smem[];
    float (*red)[16][64] = reinterpret_cast<float (*)[16][64]>(sk_smem);          // [WAVES][16][64]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int nt = blockIdx.x;
    int split = blockIdx.y;
    const int mt = blockIdx.z;
    if (flags_ & 1) {
        // XCD-aware (tile, K slice) assignment of a split-K launch (speed only, results identical): blocks are dealt round-robin to
        // the 8 XCDs, so in launch order every XCD meets all K slices and pulls the WHOLE activation matrix into its L2
        // (down projection: 8 x 512 KB).  Here XCD x works on K slice x % splitk only -> 1 / splitk of it (launcher: splitk | 8).
        const int S = gridDim.y, L = blockIdx.x + gridDim.x * blockIdx.y;
        const int xcd = L & 7, i = L >> 3;
        const int tpg = (gridDim.x * S) >> 3;             // tiles per XCD
        split = xcd % S;
        nt = (xcd / S) * tpg + i;
    }
    const int KS = KS_;                                    // k-steps of the whole K and of this block's split (host-computed:
    const int ks_per_split = ks_per_split_;                //  no integer division in front of the first load)
    const int ks_per_wave = ks_per_split / WAVES;
    const int ks0 = split * ks_per_split + wave * ks_per_wave;
    const int m = lane & 31;
    const int half = lane >> 5;

    const u32x4* wptr = reinterpret_cast<const u32x4*>(Wp_) + ((size_t)nt * KS + ks0) * 64 + lane;
    const u32x4* xptr = reinterpret_cast<const u32x4*>(xp_) + ((size_t)mt * KS + ks0) * 64 + lane;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    constexpr int RPW = 16 / WAVES;
    SkChunk<CH> ck[NB];
    SkinnyArgs p;
    float bias_d[RPW], fc1[RPW];
    // LayerNorm fold: the row statistics come out of the activation stream itself -- lane (m, half) sees 8 values of row m per
    // k-step on their way to the MFMA, so (sum, sum of squares) of the wave's K range are a few packed VALU operations per
    // fragment, hidden under the weight stream (nothing is normalised or rewritten: the statistics are only needed in the
    // epilogue).  No producer-side partials, no extra global reads.
    constexpr bool fold_on = FOLD;
    float fs1 = 0.f, fs2 = 0.f;
    auto fold_acc = [&](const u32x4& xv) {
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float a = __uint_as_float(xv[w] << 16), b = __uint_as_float(xv[w] & 0xffff0000u);
            fs1 += a + b;
            fs2 = fmaf(a, a, fmaf(b, b, fs2));
        }
    };
    auto late = [&]() {                                     // the stream is in flight: now the rest of the arguments + the bias
        p = sv_late_args<SkinnyArgs>(offsetof(SkinnyKernarg, p));
        sk_bias<RPW>(p, bias_d, wave * RPW, nt, half, FOLD ? fc1 : nullptr);
        if constexpr (HEAD) {
            if (p.out_mode == SK_OUT_F32 && p.poison) {      // SkinnyArgs::poison: the first blocks of the lm_head launch, fire and forget
                const unsigned off = ((blockIdx.x + gridDim.x * blockIdx.z) * (unsigned)(WAVES * 64) + (unsigned)tid) * 16u;
                if (off < p.poison_bytes) {
                    const __amdgpu_buffer_rsrc_t rsp = __builtin_amdgcn_make_buffer_rsrc(p.poison, 0, p.poison_bytes, 0x00020000);
                    __builtin_amdgcn_raw_buffer_store_b128(u32x4{0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}, rsp, (int)off, 0, 16);      // sc1: write-through
                }
            }
        }
        sk_settle<RPW>(bias_d, FOLD ? fc1 : nullptr);
    };
    auto guarded = [&](int ks) {                            // the ragged end of the range (and short ranges): guard per k-step
        for (; ks < ks_per_wave; ks += NB * CH) {
#pragma unroll
            for (int b = 0; b < NB; ++b) {
#pragma unroll
                for (int u = 0; u < CH; ++u)
                    if (ks + b * CH + u < ks_per_wave) {
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag4(ck[b].w[u]), as_frag4(ck[b].x[u]), acc, 0, 0, 0);
                        if constexpr (FOLD) fold_acc(ck[b].x[u]);
                    }
                if (ks + (b + NB) * CH < ks_per_wave) sk_load<CH>(ck[b], wptr, xptr, ks + (b + NB) * CH, ks_per_wave);
            }
        }
    };
    if (ks_per_wave >= 2 * NB * CH) {
        // long ranges (StarVector-8B: 36 .. 48 k-steps per wave): a branch-free steady state, so that the MFMAs of chunk b wait for
        // chunk b only (s_waitcnt vmcnt(8): the other chunk's loads stay in flight) -- same k order, same results
#pragma unroll
        for (int b = 0; b < NB; ++b) sk_load_full<CH>(ck[b], wptr, xptr, b * CH);
        late();
        int ks = 0;
        for (; ks + 2 * NB * CH <= ks_per_wave; ks += NB * CH) {
#pragma unroll
            for (int b = 0; b < NB; ++b) {
#pragma unroll
                for (int u = 0; u < CH; ++u) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag4(ck[b].w[u]), as_frag4(ck[b].x[u]), acc, 0, 0, 0);
                    if constexpr (FOLD) fold_acc(ck[b].x[u]);
                }
                __builtin_amdgcn_sched_barrier(0);          // keep "compute chunk b, refill chunk b" in this order: the scheduler
                sk_load_full<CH>(ck[b], wptr, xptr, ks + (b + NB) * CH);     // otherwise hoists the other chunk's MFMAs above the refill
                __builtin_amdgcn_sched_barrier(0);          // and the wave drains to 2 outstanding loads every iteration
            }
        }
        guarded(ks);
    } else {
#pragma unroll
        for (int b = 0; b < NB; ++b)
            if (b * CH < ks_per_wave) sk_load<CH>(ck[b], wptr, xptr, b * CH, ks_per_wave);
        late();
        guarded(0);
    }

    // ---- K reduction across the waves of the block (wave order), every wave finishes RPW accumulator rows ----
    float v[RPW];
    float2* fst_s = reinterpret_cast<float2*>(sk_smem + (size_t)WAVES * 16 * 64 * 4);      // [WAVES][32] partial row statistics
    if constexpr (FOLD) {
        fs1 += __shfl_xor(fs1, 32, 64);                          // the two 8-column halves of a k-step
        fs2 += __shfl_xor(fs2, 32, 64);
        if (half == 0) fst_s[wave * 32 + m] = make_float2(fs1, fs2);
    }
    if constexpr (WAVES > 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const int r = wave * RPW + i;
            float t = red[0][r][lane];
#pragma unroll
            for (int w = 1; w < WAVES; ++w) t += red[w][r][lane];
            v[i] = t;
        }
    } else {
        if (fold_on) __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = acc[i];
    }
    SkFold fold{false, 0.f, 1.f};
    if (fold_on) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int q = 0; q < WAVES; ++q) { const float2 t = fst_s[q * 32 + m]; s1 += t.x; s2 += t.y; }          // wave (= K) order
        const float invD = 1.0f / (float)p.fold_D;
        const float mean = s1 * invD;
        float var = s2 * invD - mean * mean;
        var = var > 0.f ? var : 0.f;
        fold = SkFold{true, mean, rsqrtf(var + p.fold_eps)};
    }
    sk_store<RPW>(p, v, bias_d, wave * RPW, nt, mt, split, m, half, fold, fc1);
    if constexpr (HEAD) {
        // Greedy selection folded into the lm_head launch: the logits this block has just rounded and stored are
        // still in registers -- lane (m, half) of wave w holds RPW columns of row m.  Best (value, lowest column) of the lane, of the
        // two halves (one shuffle), of the 8 waves (the statistics slots of the FOLD instantiation: unused here), then ONE atomic max
        // per row and block on a 64-bit key, fire and forget (no returned value, nothing waits for it).  (First form, measured: a
        // filter -- the row's maximum read when the block starts, atomics only where the block beats it -- put a global round trip in
        // front of every block's k loop: 1066 vs 1058 us per step, i.e. SLOWER than the separate argmax launch.)
        // argmax_kernel + the slice merge (a 6 us launch of the step) become a decode in finish_step_kernel.  Bit-identical selection:
        // same bf16-rounded values, columns < N only, NaN never wins, lowest index on ties (tests/test_gpu_ops.py).
        if (p.out_mode == SK_OUT_F32 && p.amax) {
            unsigned long long key = 0ull;
#pragma unroll
            for (int i = 0; i < RPW; ++i) {
                const int r = wave * RPW + i;
                const int n = nt * 32 + 8 * (r >> 2) + 4 * half + (r & 3);
                if (n < p.N) { const unsigned long long k = sv_amax_key(v[i], (unsigned)n); key = k > key ? k : key; }
            }
            { const unsigned long long o = __shfl_xor(key, 32, 64); key = o > key ? o : key; }
            unsigned long long* key_s = reinterpret_cast<unsigned long long*>(fst_s);            // [WAVES][32]
            if (half == 0) key_s[wave * 32 + m] = key;
            __syncthreads();
            if (wave == 0 && half == 0 && mt * 32 + m < p.amax_rows) {
#pragma unroll
                for (int q = 1; q < WAVES; ++q) { const unsigned long long o = key_s[q * 32 + m]; key = o > key ? o : key; }
                (void)__hip_atomic_fetch_max(p.amax + (size_t)(mt * 32 + m) * SV_AMAX_STRIDE, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// gemm_head_persist_kernel (round 6, fourth session): the lm_head of a <= 32-row decode step as ONE round of blocks that each walk several
// column tiles.  gemm_skinny_kernel<8, false, true> runs 1537 blocks of one 32-column tile (StarVector-1B: K = 2048, 16 k-steps per wave) --
// six rounds on 256 CUs, and in every block a wave's whole life is 16 weight loads between a ramp (its 16 KiB of the activations out of L2
// again: as many bytes as the weights) and a drain (LDS reduction, epilogue): 42.6 us for 201 MB = 4.7 TB/s.  Here
//   * a wave's share of the ACTIVATIONS (16 k-steps x 16 B per lane = 64 VGPRs) is loaded once per block and stays in registers: the L2 -> CU side
//     carries weights only;
//   * the weights roll through 16 registers per lane: the load of k-step u of the block's NEXT tile is issued right behind the MFMA that consumed
//     k-step u of this one -- 16 KiB per wave, 128 KiB per CU always in flight, across the tile boundaries (reduction and epilogue of tile i run
//     under the stream of tile i + 1);
//   * the ragged last tile (V = 49156: 4 valid columns) reads only those columns' pieces (the other lanes' addresses fold onto them, their
//     registers are zeroed behind the load: what the zero-padded image holds);
//   * the greedy selection's per-row key is kept in registers across the block's tiles: one atomic max per row and BLOCK.
// Same per-wave k ranges, same MFMA order, same cross-wave sum (wave order), same rounding and stores as the one-tile kernel: bit-identical logits
// and selection (tests/test_gpu_ops.py).  Scope (launcher): F32 mode, one row tile, bf16 weights, K / 16 == 128, split-K 1.
// ------------------------------------------------------------------------------------------------
//   MT = row tiles (1: <= 32 rows; 2: 33..64 rows -- both tiles' activation shares in registers, 128 VGPRs, every weight register feeds two MFMAs; the
//        two-row-tile kernels' order: bit-identical to gemm_skinny_mt2x_kernel)
struct HeadKernarg { const void* W; const bf16_t* x; int KS; int ks_per_split; int n_tiles; SkinnyArgs p; FinishArgs f; };      // the kernarg segment of gemm_head_persist_kernel
template <int MT>
__global__ __launch_bounds__(512) void gemm_head_persist_kernel(const bf16_t* Wp_, const bf16_t* xp_, int KS_, int ks_per_split_, int n_tiles_, SkinnyArgs p_unused,
                                                                 FinishArgs f_unused) {
    constexpr int WAVES = 8, RPW = 2, KSW = 16;            // k-steps per wave: KS_ == WAVES * KSW (launcher)
    extern __shared__ __attribute__((aligned(16))) char sk_smem[];
    float (*red)[WAVES][16][64] = reinterpret_cast<float (*)[WAVES][16][64]>(sk_smem);          // [MT][WAVES][16][64]
    unsigned long long* key_s = reinterpret_cast<unsigned long long*>(sk_smem + (size_t)MT * WAVES * 16 * 64 * 4);      // [WAVES][32]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 31, half = lane >> 5;
    const int KS = KS_, ks0 = wave * KSW, G = gridDim.x, n_tiles = n_tiles_;
    (void)ks_per_split_;

    u32x4 x[MT][KSW], w[KSW];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
        const u32x4* xptr = reinterpret_cast<const u32x4*>(xp_) + ((size_t)mi * KS + ks0) * 64 + lane;
#pragma unroll
        for (int u = 0; u < KSW; ++u) x[mi][u] = xptr[(size_t)u * 64];
    }
    const u32x4* wbase = reinterpret_cast<const u32x4*>(Wp_) + (size_t)ks0 * 64;
    int nt = blockIdx.x;
#pragma unroll
    for (int u = 0; u < KSW; ++u) w[u] = __builtin_nontemporal_load(wbase + ((size_t)nt * KS + u) * 64 + lane);

    SkinnyArgs p = sv_late_args<SkinnyArgs>(offsetof(SkinnyKernarg, p));
    if (p.poison) {                                          // SkinnyArgs::poison: fire and forget (gemm_skinny_kernel<.., HEAD>)
        const unsigned off = (blockIdx.x * (unsigned)(WAVES * 64) + (unsigned)tid) * 16u;
        if (off < p.poison_bytes) {
            const __amdgpu_buffer_rsrc_t rsp = __builtin_amdgcn_make_buffer_rsrc(p.poison, 0, p.poison_bytes, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b128(u32x4{0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}, rsp, (int)off, 0, 16);      // sc1: write-through
        }
    }
    const int ragged_cols = p.N - (n_tiles - 1) * 32;        // valid columns of the last tile (32: none missing)
    unsigned long long key = 0ull;
    const float bias0[RPW] = {0.f, 0.f};

    // one tile: 16 MFMAs; REFILL: each consumed register is re-requested for tile `nn` (lane address `ln`: folded for the ragged tile)
    auto tile = [&](auto refill_tag, int nn, int ln) {
        constexpr bool REFILL = decltype(refill_tag)::value;
        f32x16 acc[MT];
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][r] = 0.f;
        const u32x4* nsrc = wbase + (size_t)nn * KS * 64 + ln;
#pragma unroll
        for (int u = 0; u < KSW; ++u) {
#pragma unroll
            for (int mi = 0; mi < MT; ++mi)
                acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag4(w[u]), as_frag4(x[mi][u]), acc[mi], 0, 0, 0);
            if constexpr (REFILL) w[u] = __builtin_nontemporal_load(nsrc + (size_t)u * 64);
        }
        // K reduction across the waves (wave order), every wave finishes RPW accumulator rows; epilogue = sk_store's F32 mode
#pragma unroll
        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[mi][wave][r][lane] = acc[mi][r];
        __syncthreads();
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
            float v[RPW];
#pragma unroll
            for (int i = 0; i < RPW; ++i) {
                const int r = wave * RPW + i;
                float t = red[mi][0][r][lane];
#pragma unroll
                for (int q = 1; q < WAVES; ++q) t += red[mi][q][r][lane];
                v[i] = t;
            }
            sk_store<RPW>(p, v, bias0, wave * RPW, nt, mi, 0, m, half);
            if (MT == 1 && p.amax) {
#pragma unroll
                for (int i = 0; i < RPW; ++i) {
                    const int r = wave * RPW + i;
                    const int n = nt * 32 + 8 * (r >> 2) + 4 * half + (r & 3);
                    if (n < p.N) { const unsigned long long k = sv_amax_key(v[i], (unsigned)n); key = k > key ? k : key; }
                }
            }
        }
        __syncthreads();                                     // red is free for the next tile
    };
    for (; nt + G < n_tiles; nt += G) {
        const int nn = nt + G;
        const bool rag = nn == n_tiles - 1 && ragged_cols < 32;          // block-uniform
        const int ln = rag ? ((half << 5) | (m % ragged_cols)) : lane;
        tile(std::true_type{}, nn, ln);
        if (rag) {                                           // the lanes of the missing columns: zero, as the padded image has it (the loads land here)
            if (m >= ragged_cols) {
#pragma unroll
                for (int u = 0; u < KSW; ++u) w[u] = u32x4{0u, 0u, 0u, 0u};
            }
        }
    }
    tile(std::false_type{}, 0, 0);

    if (MT == 1 && p.amax) {
        { const unsigned long long o = __shfl_xor(key, 32, 64); key = o > key ? o : key; }
        if (half == 0) key_s[wave * 32 + m] = key;
        __syncthreads();
        if (wave == 0 && half == 0 && m < p.amax_rows) {
#pragma unroll
            for (int q = 1; q < WAVES; ++q) { const unsigned long long o = key_s[q * 32 + m]; key = o > key ? o : key; }
            (void)__hip_atomic_fetch_max(p.amax + (size_t)m * SV_AMAX_STRIDE, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (p.fin_cnt) {
            // The step's bookkeeping in this launch (SkinnyArgs::finish; fifth session of round 6): the key atomics of this block are acknowledged (vmcnt counts
            // them on gfx9), the block draws a ticket, and the last of the grid -- every row's key is final -- decodes the keys and does what finish_step_kernel does,
            // in its first wave (<= 32 rows).  The hand-off is the decode attention's: agent-scope atomics both ways, no fence, nobody waits.
            __shared__ int fin_last;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) fin_last = __hip_atomic_fetch_add(p.fin_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1 : 0;
            __syncthreads();
            if (fin_last && wave == 0) {
                if (lane == 0) __hip_atomic_store(p.fin_cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // re-arm
                const FinishArgs f = sv_late_args<FinishArgs>(offsetof(HeadKernarg, f));
                if (!*f.done) {
                    const int t = *f.step;
                    int still = 0;
                    if (lane < f.B) {
                        unsigned long long* slot = f.amax + (size_t)lane * SV_AMAX_STRIDE;
                        const int nxt = sv_amax_index(__hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                        __hip_atomic_store(slot, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        still = finish_step_row(f, lane, t, nxt);
                    }
                    const int any_unf = __ballot(still) != 0ull;
                    if (lane == 0) finish_step_call(f, t, any_unf);
                }
            }
        }
    }
}
// ------------------------------------------------------------------------------------------------
// gemm_skinny_tailsplit_kernel (round 6, fourth session): the column tiles of a whole-K skinny GEMM that do not fit the first round of blocks.
// StarVector-8B's c_fc at <= 32 rows is 576 one-tile blocks on 512 block slots (two 8-wave blocks per CU): the second "round" is 64 blocks that each stream
// 288 KiB alone (~30 GB/s per lone CU): ~10 of the launch's 36 us for 11 % of its bytes (rocprof by grid, profiles/rocprof_r06_8b_im2svg_by_grid.csv).
// Here the first 512 tiles stay one launch of the ordinary kernel and the T left-over tiles go to 4 T blocks of this one: block (tile, q) takes the q-th
// quarter of K (8 waves x KS / 32 k-steps, all requested up front), reduces across its waves as the ordinary kernel does, leaves its 32 x 32 fp32 partial in
// the engine's scratch through write-through stores and draws an arrival ticket; the last of the four sums the partials in q order and runs the epilogue
// (bias + activation -> packed activations).  The hand-off is the decode attention's (sc1 stores, every wave drains, one relaxed agent-scope ticket, sc1 loads
// in the last arriver, ticket re-armed): no fences, nobody waits.  The sum order of those tiles' columns is (quarter, wave) instead of wave: other roundings
// than the one-tile kernel for T / n_tiles of the columns, the same for every batch <= 32 (one kernel, one order: batch-independent).
// ------------------------------------------------------------------------------------------------
template <int KPW>                                           // k-steps per wave (KS / 32): 9 for StarVector-8B
__global__ __launch_bounds__(512) void gemm_skinny_tailsplit_kernel(const bf16_t* Wp_, const bf16_t* xp_, int KS_, int tile0_, int flags_, SkinnyArgs p_unused) {
    constexpr int WAVES = 8, RPW = 2;
    extern __shared__ __attribute__((aligned(16))) char sk_smem[];
    float (*red)[16][64] = reinterpret_cast<float (*)[16][64]>(sk_smem);          // [WAVES][16][64]
    __shared__ int last_s;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 31, half = lane >> 5;
    const int tt = blockIdx.x >> 2, q = blockIdx.x & 3;        // left-over tile, K quarter
    const int nt = tile0_ + tt, KS = KS_;
    const int ks0 = q * (KS >> 2) + wave * KPW;
    (void)flags_;
    const u32x4* wptr = reinterpret_cast<const u32x4*>(Wp_) + ((size_t)nt * KS + ks0) * 64 + lane;
    const u32x4* xptr = reinterpret_cast<const u32x4*>(xp_) + (size_t)ks0 * 64 + lane;
    u32x4 w[KPW], x[KPW];
#pragma unroll
    for (int u = 0; u < KPW; ++u) w[u] = __builtin_nontemporal_load(wptr + (size_t)u * 64);
#pragma unroll
    for (int u = 0; u < KPW; ++u) x[u] = xptr[(size_t)u * 64];
    SkinnyArgs p = sv_late_args<SkinnyArgs>(offsetof(SkinnyKernarg, p));
    float bias_d[RPW];
    sk_bias<RPW>(p, bias_d, wave * RPW, nt, half);
    sk_settle<RPW>(bias_d);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int u = 0; u < KPW; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag4(w[u]), as_frag4(x[u]), acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
    __syncthreads();
    float v[RPW];
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const int r = wave * RPW + i;
        float t = red[0][r][lane];
#pragma unroll
        for (int g = 1; g < WAVES; ++g) t += red[g][r][lane];
        v[i] = t;
    }
    // partial of (tile, quarter): [16 accumulator rows][64 lanes] floats, write-through; every wave drains, then one ticket per block
    const unsigned bytes = (unsigned)SV_TAIL_TILES * 4u * 16u * 64u * 4u;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(p.tail_ws, 0, bytes, 0x00020000);
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const int r = wave * RPW + i;
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[i]), rs, (int)(((((unsigned)tt * 4u + (unsigned)q) * 16u + (unsigned)r) * 64u + (unsigned)lane) * 4u), 0, 16);     // sc1
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const unsigned t = __hip_atomic_fetch_add(p.tail_cnt + tt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last_s = t == 3u ? 1 : 0;
    }
    __syncthreads();
    if (!last_s) return;
    float part[RPW][4];
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const int r = wave * RPW + i;
#pragma unroll
        for (int g = 0; g < 4; ++g)
            part[i][g] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)(((((unsigned)tt * 4u + (unsigned)g) * 16u + (unsigned)r) * 64u + (unsigned)lane) * 4u), 0, 16));   // sc1: L1 bypass
    }
#pragma unroll
    for (int i = 0; i < RPW; ++i) v[i] = ((part[i][0] + part[i][1]) + part[i][2]) + part[i][3];          // quarter order
    if (tid == 0) __hip_atomic_store(p.tail_cnt + tt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm
    sk_store<RPW>(p, v, bias_d, wave * RPW, nt, 0, 0, m, half);
}
std::atomic<int> g_tailsplit{1};
void set_tailsplit(int on) { g_tailsplit = on; }

std::atomic<int> g_head_persist{1};     // 1: the lm_head of a one-row-tile step through gemm_head_persist_kernel where it applies; 0: the one-tile kernel (A/B)
void set_head_persist(int on) { g_head_persist = on; }
// false: outside the kernel's scope
// G = blocks of the persistent lm_head launch for `a`; 0: outside the kernel's scope
static int head_persist_grid(const SkinnyArgs& a) {
    const char* ev = getenv("SV_HEAD_PERSIST");             // read per call (A/B in one process; a captured graph keeps its choice)
    const int env = ev ? atoi(ev) : -1;
    const int on = env >= 0 ? env : g_head_persist.load(std::memory_order_relaxed);
    if (!on) return 0;
    const int n_tiles = a.Npad / 32;
    if (a.out_mode != SK_OUT_F32 || a.MT < 1 || a.MT > 2 || a.Wq || a.splitk != 1 || a.K / 16 != 128 || n_tiles < 512 || a.fold_c1) return 0;
    if (a.MT == 2 && (a.amax || a.poison)) return 0;      // (the folded selection and the pattern stores belong to one-row-tile steps)
    if (a.N <= (n_tiles - 1) * 32 || a.N > n_tiles * 32) return 0;
    static int cus = 0;
    if (!cus) {
        int dev = 0; hipDeviceProp_t pr;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) return 0;
        cus = pr.multiProcessorCount;
    }
    const int G = cus < n_tiles ? cus : n_tiles;
    if (a.poison && (size_t)G * 512 * 16 < a.poison_bytes) return 0;      // the pattern stores are 16 bytes per thread of the grid
    return G;
}
// true: the lm_head launch of `a` runs the step's bookkeeping itself (SkinnyArgs::finish) -- the caller launches no finish_step_kernel
bool skinny_head_folds_finish(const SkinnyArgs& a) {
    return a.finish && a.fin_cnt && a.amax && a.MT == 1 && a.finish->B <= 32 && a.finish->amax == a.amax && head_persist_grid(a) > 0;
}
// false: outside the kernel's scope
static bool launch_head_persist(const SkinnyArgs& a_, hipStream_t st) {
    const int G = head_persist_grid(a_);
    if (!G) return false;
    SkinnyArgs a = a_;
    FinishArgs f;
    memset(&f, 0, sizeof(f));
    if (skinny_head_folds_finish(a)) f = *a.finish; else a.fin_cnt = nullptr;
    a.finish = nullptr;                                         // (a host address: nothing for the device)
    const int n_tiles = a.Npad / 32;
    if (a.MT == 2) gemm_head_persist_kernel<2><<<G, 512, 2 * 8 * 16 * 64 * 4 + 8 * 32 * 8, st>>>(a.Wp, a.xp, a.K / 16, a.K / 16, n_tiles, a, f);
    else gemm_head_persist_kernel<1><<<G, 512, 8 * 16 * 64 * 4 + 8 * 32 * 8, st>>>(a.Wp, a.xp, a.K / 16, a.K / 16, n_tiles, a, f);
    return true;
}

// ------------------------------------------------------------------------------------------------
// mlp_fused_kernel: the MLP half of a decode layer (gpt_bigcode/modeling_gpt_bigcode.py:645-660: c_fc -> GELU-tanh -> c_proj) as ONE
// launch of F/32 co-resident blocks (256 for StarVector-1B, one 8-wave block per CU), instead of gemm_skinny_kernel<8, true> (folded
// c_fc) and gemm_skinny_kernel<8, false> (down projection, split-K slabs) with a kernel boundary between them.  Round 4; on for an
// engine that owns its GPU (sv_config.exclusive_device), SV_EXP bit 128 / 512 = forced on / off.  Measured in process on one MI355X,
// BASELINE config 2: 1079 vs 1111 us per decode step, tokens bit-identical (every version's A/B and wall-clock
// trace: profiles/mlp_fused_r04_ab.log).
//
//   Why it pays (MI355X_MICROARCH.md price list: boundary, prefetch-credit): both GEMMs are pure weight streams (2 x 33.5 MB) and the
//   WEIGHTS of the second one depend on nothing.  Two launches pay a boundary (1.7-1.9 us) plus the second kernel's cold start; here a
//   wave requests the first half of its share of the down projection's weights as soon as its c_fc loop has issued its last MFMA, so
//   HBM keeps streaming under the c_fc reduction, epilogue and publish.
//
//   Phase 1  block L = (xcd = L & 7, i = L >> 3): c_fc tile nt1 = split * (T1 / S) + (xcd / S) * (T1 / 8) + i with split = xcd % S --
//            the 32 GELU output columns of a tile are 2 KiB contiguous in fragment order; they go LDS -> 16-byte sc1 (write-through)
//            stores.  No flag, no counter, nothing to drain.
//   Phase 2  the same block owns (tile nt2 = (xcd / S) * (T1 / 8) + i, K slice `split`) of the down projection = exactly the
//            (tile, slice) the XCD-aware assignment of the slab kernel gives block L.  Wave w needs the c_fc columns of its 16 k-steps
//            = the tiles of 8 producer blocks; it reads them with sc1 loads (L1 bypass) and recognises "not written yet" by the data
//            itself: the launch in front fills the buffer with the bf16 pair 0xFFFF'FFFF (two NaNs -- never a finite GELU output), a
//            k-step that still shows the pattern is re-requested after an s_sleep, BOUNDED (a give-up code in *err, never a hang).
//   What the traces taught (all of it in profiles/mlp_fused_r04_ab.log):
//            * one arrival counter per K slice (64 arrivals + 64 pollers per word): 25.5 us per launch against 17.8 for two launches;
//              one flag word per producer: 17.7 us (the publish store drained behind 24 MB of weight prefetch); in-band pattern: 14.9;
//              + the first half of the weights requested before the reduction: 13.8 us -- this version.
//            * a CU's miss queue takes ~64 KiB: a wave that requests more BLOCKS AT ISSUE until the queue drains, and a wave's loads
//              return IN ORDER -- a poll queued behind 16 KiB of its own weight requests completes after them.  Requesting all 16 KiB
//              at once, moving the reduction to two "critical-path" waves, and the attention output projection as a phase 0 in front
//              (a residual-stream ping-pong, 4 launches per layer) were all built, bit-identical, and slower or equal: removed.
//              Round 5 BUILT the loader wave (a ninth wave per block streams the block's W2 share into LDS with global_load_lds, the
//              consumers request nothing but polls after the publish): bit-identical, +2.0 ... +2.5 us per layer for every start
//              trigger / depth tried -- polls issued at the publish find the pattern and every failed poll is a ~2 us round trip beside
//              the CU's own fill stream; this form's polls ride behind the second half of the weights and arrive when the data is there.
//              What is left over the 12.2 us that 67 MB + the phase-2 tail take is ~1 us.  profiles/mlp_fused_r05_loader_ab.log; removed.
//   Results  per-wave k ranges, MFMA order, cross-wave reduction order, fold statistics and epilogues are those of the two kernels it
//            replaces: bit-identical slabs (tests/test_gpu_e2e.py::test_fused_mlp_launch_equals_the_two_launches_bit_for_bit).
//   Safety   needs all F/32 blocks resident at once (one per CU): enabled only when #CUs >= F/32 AND the engine owns the device -- two
//            processes decoding on one GPU could each hold part of the CUs and wait for blocks that cannot be scheduled (then both
//            give up after the bounded spin and the call fails: never a hang, never wrong tokens).  The pattern is written by the
//            kernel in front (gemm_cols_resid_kernel, ColsArgs::poison), never by this launch; a NaN pair with the pattern (corrupted
//            inputs only: arithmetic NaNs are 0x7FC0) ends in the give-up code, i.e. an error from sv_generate.
// ------------------------------------------------------------------------------------------------
struct MlpFusedKernarg { const bf16_t* W1; const bf16_t* x1; const bf16_t* W2; int KS1; int KS2; int S; MlpFusedArgs p; };
// amdgpu_waves_per_eu(2, 2): one 8-wave block per CU is the design point (2 waves per SIMD, 256 VGPRs each); without it the
// scheduler trades registers for a third wave that can never exist and serialises phase 2's 16 activation loads 3 at a time
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void mlp_fused_kernel(const bf16_t* W1_, const bf16_t* x1_, const bf16_t* W2_, int KS1_, int KS2_, int S_,
                                                        MlpFusedArgs p_unused) {
    constexpr int WAVES = 8, CH = 4, NB = 2, RPW = 2, KPW = 16;          // k-steps per wave in BOTH phases (host-checked)
    extern __shared__ __attribute__((aligned(16))) char sk_smem[];
    float (*red)[16][64] = reinterpret_cast<float (*)[16][64]>(sk_smem);          // [WAVES][16][64]
    float2* fst_s = reinterpret_cast<float2*>(sk_smem + (size_t)WAVES * 16 * 64 * 4);      // [WAVES][32] partial row statistics
    bf16_t* tile_s = reinterpret_cast<bf16_t*>(sk_smem + (size_t)WAVES * 16 * 64 * 4 + (size_t)WAVES * 32 * 8);   // 2 KiB: the c_fc tile
    int* flag_s = reinterpret_cast<int*>(tile_s + 1024);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 31, half = lane >> 5;
    const int L = blockIdx.x, xcd = L & 7, ii = L >> 3;
    const int T1 = gridDim.x, tpg = T1 >> 3;
    const int split = xcd % S_, grp = xcd / S_;
    const int nt2 = grp * tpg + ii;
    const int nt1 = split * (T1 / S_) + nt2;
    const long long t_start = wall_clock64();             // 100 MHz; only stored when the trace buffer is on (tools/mlp_trace.py)

    // ---- phase 1: folded c_fc, tile nt1 over the whole K1 (gemm_skinny_kernel<8, true>, long-range path) ----
    const int ks0 = wave * KPW;
    const u32x4* wptr = reinterpret_cast<const u32x4*>(W1_) + ((size_t)nt1 * KS1_ + ks0) * 64 + lane;
    const u32x4* xptr = reinterpret_cast<const u32x4*>(x1_) + (size_t)ks0 * 64 + lane;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    SkChunk<CH> ck[NB];
    float fs1 = 0.f, fs2 = 0.f;
    auto fold_acc = [&](const u32x4& xv) {
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float a = __uint_as_float(xv[w] << 16), b = __uint_as_float(xv[w] & 0xffff0000u);
            fs1 += a + b;
            fs2 = fmaf(a, a, fmaf(b, b, fs2));
        }
    };
#pragma unroll
    for (int b = 0; b < NB; ++b) sk_load_full<CH>(ck[b], wptr, xptr, b * CH);
    const MlpFusedArgs p = sv_late_args<MlpFusedArgs>(offsetof(MlpFusedKernarg, p));
    float c2v[RPW], c1v[RPW];
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const int r = wave * RPW + i;
        const int n = nt1 * 32 + 8 * (r >> 2) + 4 * half + (r & 3);
        c1v[i] = n < p.N1 ? p.fold_c1[n] : 0.f;
        c2v[i] = n < p.N1 ? p.fold_c2[n] : 0.f;
    }
    sk_settle<RPW>(c2v, c1v);
#pragma unroll
    for (int ks = 0; ks < KPW; ks += NB * CH) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag4(ck[b].w[u]), as_frag4(ck[b].x[u]), acc, 0, 0, 0);
                fold_acc(ck[b].x[u]);
            }
            if (ks + (b + NB) * CH < KPW) {
                __builtin_amdgcn_sched_barrier(0);
                sk_load_full<CH>(ck[b], wptr, xptr, ks + (b + NB) * CH);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    const long long t_loop1 = wall_clock64();
    const int ks2 = split * (WAVES * KPW) + wave * KPW;
    const u32x4* w2ptr = reinterpret_cast<const u32x4*>(W2_) + ((size_t)nt2 * KS2_ + ks2) * 64 + lane;
    u32x4 w2[KPW];
    // The down projection's weights of this wave (tile nt2, k-steps split * 128 + wave * 16 .. + 16: 16 KiB) depend on nothing: the first
    // half is requested NOW, so that the HBM stream does not pause while the block reduces and publishes (third version: requested after
    // the publish -- HBM idle for 2.5 us); 8 KiB per wave in flight is what the steady state of the stand-alone kernels keeps (all
    // 16 KiB at once put 24 MB of reads in front of every tile store of the chip: second version, 6 us from loop end to publish).
    // (round 5 re-measured the alternative "reduce and publish with an empty queue, then the whole 16 KiB share": publish 0.9 us earlier, but the
    //  MFMAs then wait 2.7 us for the weights -- 1100 vs 1066 us per step; removed.)
#pragma unroll
    for (int u = 0; u < 8; ++u) w2[u] = __builtin_nontemporal_load(w2ptr + (size_t)u * 64);

    // K reduction across the waves (wave order) + LayerNorm fold epilogue: gemm_skinny_kernel<8, true>'s, value for value
    float v[RPW];
    fs1 += __shfl_xor(fs1, 32, 64);
    fs2 += __shfl_xor(fs2, 32, 64);
    if (half == 0) fst_s[wave * 32 + m] = make_float2(fs1, fs2);
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const int r = wave * RPW + i;
        float t = red[0][r][lane];
#pragma unroll
        for (int w = 1; w < WAVES; ++w) t += red[w][r][lane];
        v[i] = t;
    }
    {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int q = 0; q < WAVES; ++q) { const float2 t = fst_s[q * 32 + m]; s1 += t.x; s2 += t.y; }
        const float invD = 1.0f / (float)p.fold_D;
        const float mean = s1 * invD;
        float var = s2 * invD - mean * mean;
        var = var > 0.f ? var : 0.f;
        const float rstd = rsqrtf(var + p.fold_eps);
        const int r = wave * RPW;
        const int nl = 8 * (r >> 2) + 4 * half + (r & 3);                 // local column of v[0] (v[1]: + 1)
        float o[RPW];
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            float x = 0.f;
            if (nt1 * 32 + nl + i < p.N1) {
                x = bfround(rstd * (v[i] - mean * c1v[i]) + c2v[i]);
                if (p.act != ACT_NONE) x = sv_act(x, p.act);
            }
            o[i] = x;
        }
        // the tile in fragment order (the image xp_index addresses: [k-step nl >> 4][64 lanes][8])
        *reinterpret_cast<uint32_t*>(tile_s + (((nl >> 4) * 64 + ((nl >> 3) & 1) * 32 + m) * 8 + (nl & 7))) = pack2bf(o[0], o[1]);
    }
    __syncthreads();
    // ---- hand-off, third version: NO flag and NO counter.  The launch in front of this one (gemm_cols_resid_kernel) fills the whole
    // activation buffer with the bf16 pair 0xFFFF'FFFF (two NaNs: no finite GELU output has that pattern).  A producer only writes its
    // 2 KiB tile (write-through 16-byte stores, nothing to wait for); a consumer WAVE polls the 16 KiB it needs itself -- the tiles of
    // the 8 producers behind its 16 k-steps, not of all 64 producers of the slice -- and re-requests only the k-steps that still
    // carry the pattern.  History (profiles/mlp_fused_r04_ab.log): one ticket word per K slice: 25.5 us per launch (64 arrivals + 64
    // pollers per word); one flag word per producer polled by wave 0: 17.7 us = the two launches, with 6 us of "publish" (the store
    // drained behind 24 MB of weight prefetch) and 4 us of "wait" on the critical path.
    const __amdgpu_buffer_rsrc_t rs_act = __builtin_amdgcn_make_buffer_rsrc(p.out_xp, 0, (unsigned)((size_t)p.out_KS * 1024), 0x00020000);
    if (wave < 2) {
        const u32x4 q = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(tile_s) + tid * 16);
        __builtin_amdgcn_raw_buffer_store_b128(q, rs_act, nt1 * 2048 + tid * 16, 0, 16);          // sc1: write-through
    }
    const long long t_pub = wall_clock64();
    // second half of the weights (the first half has landed during the reduction), then -- once half of THAT is in -- the activations:
    // their producers publish at about the same time as this block, and a request that finds the pattern costs a whole extra round trip
#pragma unroll
    for (int u = 8; u < 16; ++u) w2[u] = __builtin_nontemporal_load(w2ptr + (size_t)u * 64);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");

    // ---- phase 2: down projection (tile nt2, K slice `split`) -> fp32 slab, gemm_skinny_kernel<8, false>'s order ----
    // (round 5 tried rowln_cattn_kernel's lesson here -- hold the first poll until a fixed time after the block's start: 1007.1 us per step at 10.0 us
    //  against 1009.8 without, worse from 11 us on: the polls already sit where the data turns up; profiles/rowln_cattn_r05_ab.log)
    u32x4 x2[KPW];
#pragma unroll
    for (int u = 0; u < KPW; ++u) x2[u] = __builtin_amdgcn_raw_buffer_load_b128(rs_act, (ks2 + u) * 1024 + lane * 16, 0, 16);   // sc1: L1 bypass
    unsigned pending = 0xffffu;                              // k-steps whose activations are not (known to be) complete: wave-uniform
    int gave_up = 1;
    // Bounded by WALL CLOCK, not by a poll count (65536 polls of ~2 us each per wave, per layer, per step turned a
    // co-residency failure into minutes before the host saw it): a wave gives up after spin_ticks (5 ms) -- and at once when another
    // wave or an earlier launch has already raised the flag (every later layer of a void step then falls through without waiting).
    for (int it = 0;; ++it) {
        unsigned still = 0u;
#pragma unroll
        for (int u = 0; u < KPW; ++u) {
            if (pending & (1u << u)) {
                const bool bad = x2[u][0] == 0xffffffffu || x2[u][1] == 0xffffffffu || x2[u][2] == 0xffffffffu || x2[u][3] == 0xffffffffu;
                if (__any(bad)) still |= 1u << u;
            }
        }
        pending = still;
        if (!pending) { gave_up = 0; break; }
        if ((it & 7) == 7 && (wall_clock64() - t_pub > (long long)p.spin_ticks ||
                              __hip_atomic_load(p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) break;
        __builtin_amdgcn_s_sleep(8);
#pragma unroll
        for (int u = 0; u < KPW; ++u)
            if (pending & (1u << u)) x2[u] = __builtin_amdgcn_raw_buffer_load_b128(rs_act, (ks2 + u) * 1024 + lane * 16, 0, 16);
    }
    if (gave_up && lane == 0) atomicCAS(p.err, 0, 3);      // the step's result is void; the first code raised survives
    const long long t_go = wall_clock64();
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int u = 0; u < KPW; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag4(w2[u]), as_frag4(x2[u]), acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const int r = wave * RPW + i;
        float t = red[0][r][lane];
#pragma unroll
        for (int w = 1; w < WAVES; ++w) t += red[w][r][lane];
        v[i] = t;
    }
    {
        const int r = wave * RPW;
        const int n0 = nt2 * 32 + 8 * (r >> 2) + 4 * half + (r & 3);
        *reinterpret_cast<float2*>(p.ws + ((size_t)split * p.rows_ws + m) * p.ldws + n0) = make_float2(v[0], v[1]);
    }
    if (p.trace && tid == 0) {                              // block L: start | c_fc loop done | tile published | slice complete | end  (wave 0's clock)
        long long* q = p.trace + (size_t)L * 8;
        q[0] = t_start; q[1] = t_loop1; q[2] = t_pub; q[3] = t_go; q[4] = wall_clock64(); { unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); q[5] = (long long)(xcc & 0xf); }
    }
}
static size_t mlp_fused_smem() { return (size_t)8 * 16 * 64 * 4 + (size_t)8 * 32 * 8 + 2048 + 64; }

// 0 = launched; -1 = the shapes are outside the kernel's scope (the caller runs the two launches)
int launch_mlp_fused(const MlpFusedArgs& a, hipStream_t st) {
    const int KS1 = a.K1 / 16, KS2 = a.K2 / 16, T1 = a.N1pad / 32, T2 = a.N2pad / 32;
    if (a.splitk < 1 || 8 % a.splitk || T1 % 8 || KS1 != 8 * 16 || KS2 != a.splitk * 8 * 16) return -1;      // 16 k-steps per wave in both phases
    if (T2 * a.splitk != T1 || a.K2 != a.N1pad || a.N1 != a.N1pad || a.N2 != a.N2pad) return -1;
    if (!a.err || !a.fold_c1 || !a.fold_c2) return -1;
    mlp_fused_kernel<<<T1, 512, mlp_fused_smem(), st>>>(a.W1, a.x1, a.W2, KS1, KS2, a.splitk, a);
    return 0;
}

static size_t skinny_smem(int waves) { return (size_t)waves * 16 * 64 * 4 + (size_t)waves * 32 * 8 + 16; }

static int init_mt2_attrs();
int init_gemm_kernels() {
    // 16-wave blocks reduce through 64 KiB of LDS: above the default dynamic-LDS limit
    int r = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_skinny_kernel<16, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    if (!r) r = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_skinny_kernel<16, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    if (!r) r = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_skinny_kernel<16, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    if (!r) r = init_mt2_attrs();
    if (!r) r = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_head_persist_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 8 * 16 * 64 * 4 + 8 * 32 * 8);
    if (!r) r = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256_kernel<bf16_t>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 2 * G2_BUF);
    if (!r) r = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256_kernel<float>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 2 * G2_BUF);
    return r;
}

template <int W>
static void launch_sk(const SkinnyArgs& a, dim3 grid, hipStream_t st) {
    if (a.fold_c1) gemm_skinny_kernel<W, true><<<grid, W * 64, skinny_smem(W), st>>>(a.Wp, a.xp, a.K / 16, (a.K / 16) / a.splitk, a.xcd_remap, a);
    else if (W > 1 && a.out_mode == SK_OUT_F32 && (a.amax || a.poison))
        gemm_skinny_kernel<W, false, (W > 1)><<<grid, W * 64, skinny_smem(W), st>>>(a.Wp, a.xp, a.K / 16, (a.K / 16) / a.splitk, a.xcd_remap, a);
    else gemm_skinny_kernel<W, false><<<grid, W * 64, skinny_smem(W), st>>>(a.Wp, a.xp, a.K / 16, (a.K / 16) / a.splitk, a.xcd_remap, a);
}

// ------------------------------------------------------------------------------------------------
// skinny GEMM with fp8 (e4m3) weights: the decode step's weight stream at half the bytes.  Same structure as the slab
// pipeline's bf16 kernel (K split over the waves of a block and over `splitk` blocks, all waves share the epilogue),
// restricted to what that pipeline uses: fp32 slabs, packed activations with bias + activation, fp32 logits.
// A lane's 16 B load holds two k-steps of 8 fp8 each; they are widened to bf16 in registers (exact: e4m3 has 3 mantissa
// bits) and fed to the same bf16 MFMA, so activations and accumulation are untouched; the per-column scale multiplies the
// fp32 accumulator before the cross-wave reduction.  8 k-steps per register chunk keep as many bytes in flight per wave
// as the bf16 kernel has (one CU streams ~25 GB/s whatever the element size: it is the bytes in flight that count).
// ------------------------------------------------------------------------------------------------
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void gemm_skinny_fp8_kernel(const uint8_t* Wq_, const bf16_t* xp_, int KS_, int ks_per_split_, int flags_, SkinnyArgs p_unused) {
    constexpr int CH = 8;
    extern __shared__ __attribute__((aligned(16))) char sk_smem[];
    float (*red)[16][64] = reinterpret_cast<float (*)[16][64]>(sk_smem);          // [WAVES][16][64]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int nt = blockIdx.x, split = blockIdx.y;
    const int mt = blockIdx.z;
    if (flags_ & 1) {                                          // XCD-aware (tile, K slice) assignment: see gemm_skinny_kernel
        const int S = gridDim.y, L = blockIdx.x + gridDim.x * blockIdx.y;
        const int xcd = L & 7, tpg = (gridDim.x * S) >> 3;
        split = xcd % S;
        nt = (xcd / S) * tpg + (L >> 3);
    }
    const int KS = KS_;                                    // k-steps of the whole K and of this block's split (host-computed:
    const int ks_per_split = ks_per_split_;                //  no integer division in front of the first load)
    const int ks_per_wave = ks_per_split / WAVES;          // even (launcher)
    const int ks0 = split * ks_per_split + wave * ks_per_wave;
    const int m = lane & 31, half = lane >> 5;

    const u32x4* wq = reinterpret_cast<const u32x4*>(Wq_) + ((size_t)nt * (KS >> 1) + (ks0 >> 1)) * 64 + lane;
    const u32x4* xptr = reinterpret_cast<const u32x4*>(xp_) + ((size_t)mt * KS + ks0) * 64 + lane;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    constexpr int RPW = 16 / WAVES;                       // WAVES in {2, 4, 8}
    struct Chunk { u32x4 w[CH / 2]; u32x4 x[CH]; };
    Chunk ca, cb;
    auto load = [&](Chunk& c, int ks, auto full) {         // FULL: no guards (exact wait counts in the steady state, see sk_load_full)
        constexpr bool FULL = decltype(full)::value;
#pragma unroll
        for (int u2 = 0; u2 < CH / 2; ++u2)
            if (FULL || ks + 2 * u2 < ks_per_wave) c.w[u2] = __builtin_nontemporal_load(wq + (size_t)((ks >> 1) + u2) * 64);
#pragma unroll
        for (int u = 0; u < CH; ++u)
            if (FULL || ks + u < ks_per_wave) c.x[u] = xptr[(size_t)(ks + u) * 64];
    };
    auto compute = [&](Chunk& c, int ks, auto full) {
        constexpr bool FULL = decltype(full)::value;
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            if (FULL || ks + u < ks_per_wave) {
                const u32x4 wf = fp8x8_to_bf16x8(c.w[u >> 1][(u & 1) * 2], c.w[u >> 1][(u & 1) * 2 + 1]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag4(wf), as_frag4(c.x[u]), acc, 0, 0, 0);
            }
        }
    };
    SkinnyArgs p;
    float4 sc4[4];
    float bias_d[RPW];
    auto late = [&]() {     // scales and bias need the argument block (a scalar load): requested after the stream is in flight
        p = sv_late_args<SkinnyArgs>(offsetof(SkinnyKernarg, p));
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) sc4[rg] = *reinterpret_cast<const float4*>(p.wscale + nt * 32 + rg * 8 + half * 4);
        sk_bias<RPW>(p, bias_d, wave * RPW, nt, half);
        sk_settle<RPW>(bias_d);
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) asm volatile("" : "+v"(sc4[rg].x), "+v"(sc4[rg].y), "+v"(sc4[rg].z), "+v"(sc4[rg].w));
    };
    auto guarded = [&](int ks) {
        for (; ks < ks_per_wave; ks += 2 * CH) {
            compute(ca, ks, std::false_type{});
            if (ks + 2 * CH < ks_per_wave) load(ca, ks + 2 * CH, std::false_type{});
            if (ks + CH < ks_per_wave) compute(cb, ks + CH, std::false_type{});
            if (ks + 3 * CH < ks_per_wave) load(cb, ks + 3 * CH, std::false_type{});
        }
    };
    if (ks_per_wave >= 4 * CH) {                            // long ranges: branch-free steady state
        load(ca, 0, std::true_type{});
        load(cb, CH, std::true_type{});
        late();
        int ks = 0;
        for (; ks + 4 * CH <= ks_per_wave; ks += 2 * CH) {
            compute(ca, ks, std::true_type{});
            __builtin_amdgcn_sched_barrier(0);
            load(ca, ks + 2 * CH, std::true_type{});
            __builtin_amdgcn_sched_barrier(0);
            compute(cb, ks + CH, std::true_type{});
            __builtin_amdgcn_sched_barrier(0);
            load(cb, ks + 3 * CH, std::true_type{});
            __builtin_amdgcn_sched_barrier(0);
        }
        guarded(ks);
    } else {
        load(ca, 0, std::false_type{});
        if (CH < ks_per_wave) load(cb, CH, std::false_type{});
        late();
        guarded(0);
    }
    // per-column scale (accumulator row r <-> column 8 (r >> 2) + 4 half + (r & 3)), then the K reduction across waves
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
        acc[rg * 4 + 0] *= sc4[rg].x; acc[rg * 4 + 1] *= sc4[rg].y; acc[rg * 4 + 2] *= sc4[rg].z; acc[rg * 4 + 3] *= sc4[rg].w;
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
        for (int w = 1; w < WAVES; ++w) t += red[w][r][lane];
        v[i] = t;
    }
    sk_store<RPW>(p, v, bias_d, wave * RPW, nt, mt, split, m, half);
}

// Waves per block of the skinny kernels = how K is cut inside a block, i.e. the order in which a row's partial sums are
// added.  A function of the GEMM (N, K, split-K) ONLY -- never of the number of rows -- so that a row's bits do not depend on
// the batch it is computed in (one tile, two tiles per block, grid.z tiles all cut K the same way).
static int skinny_waves(int Npad, int KS, int splitk) {
    const int per_split = KS / splitk;
    // narrow outputs (few column tiles) get 16 waves per block so that no cross-block split-K is needed
    const bool narrow = (Npad / 32) * splitk < 160;
    if (per_split % 16 == 0 && narrow) return 16;
    if (per_split % 8 == 0) return 8;
    if (per_split % 4 == 0) return 4;
    if (per_split % 2 == 0) return 2;
    return 1;
}
static int skinny_waves_fp8(int KS, int splitk) {       // a lane's 16 bytes hold two k-steps: even k-steps per wave
    const int per_split = KS / splitk;
    if (per_split % 16 == 0) return 8;
    if (per_split % 8 == 0) return 4;
    if (per_split % 4 == 0) return 2;
    return 0;
}

// returns false when the shape / mode has no fp8 variant (the caller reports it)
static bool launch_gemm_skinny_fp8(const SkinnyArgs& a, hipStream_t st) {
    if (!(a.out_mode == SK_OUT_PARTIAL || ((a.out_mode == SK_OUT_PACKED_ACT || a.out_mode == SK_OUT_F32) && a.splitk == 1)))
        return false;
    const dim3 grid(a.Npad / 32, a.splitk, a.MT);
    const int per_split = (a.K / 16) / a.splitk;
    (void)per_split;
    const int waves = skinny_waves_fp8(a.K / 16, a.splitk);
    if (waves == 8) gemm_skinny_fp8_kernel<8><<<grid, 512, 8 * 16 * 64 * 4, st>>>(a.Wq, a.xp, a.K / 16, (a.K / 16) / a.splitk, a.xcd_remap, a);
    else if (waves == 4) gemm_skinny_fp8_kernel<4><<<grid, 256, 4 * 16 * 64 * 4, st>>>(a.Wq, a.xp, a.K / 16, (a.K / 16) / a.splitk, a.xcd_remap, a);
    else if (waves == 2) gemm_skinny_fp8_kernel<2><<<grid, 128, 2 * 16 * 64 * 4, st>>>(a.Wq, a.xp, a.K / 16, (a.K / 16) / a.splitk, a.xcd_remap, a);
    else return false;
    return true;
}

// ------------------------------------------------------------------------------------------------
// skinny GEMM for 33..64 rows (batch 64 of BASELINE config 5): TWO 32-row tiles per block.  The one-tile kernels put the
// row tile on grid.z, so at batch 64 every weight byte crossed HBM -> CU twice (8B text2svg, batch 64: 5.6 ms of GEMM per
// step against 3.4 ms at batch 16).  Here a wave feeds each weight fragment to two MFMAs (B operands = the fragments of row
// tile 0 and 1): the weight stream is read once, the per-row arithmetic -- same MFMA, same ascending k order, same wave /
// slab reduction order -- is the one-tile kernels', so a row's result does not depend on which kernel produced it.
// Scope = what the slab pipeline launches: fp32 slabs, packed activations with bias + activation, fp32 logits; bf16 or
// fp8 (e4m3, widened in registers, per-column scale on the accumulator) weights.  NBUF register chunks of CH k-steps ring.
// ------------------------------------------------------------------------------------------------
// NT = column tiles per block (1, 2 or 3).  Every block of a launch re-reads the SAME activation fragments out of L2 -- at 64 rows
// that is 2 (bf16 weights) or 4 (fp8) bytes per weight byte, and the L2 -> CU side, not HBM, bounds the launch
// (tools/diag/mem_mix.hip, profiles/mem_mix_r03.log: the same traffic without MFMA streams weights at 6.3 / 5.0 / 3.7 TB/s with
// 1 / 2 / 4 shared-operand bytes per weight byte).  NT = 2: a wave feeds the activation fragments it has loaded to the
// weight fragments of TWO adjacent column tiles -> half the re-reads per weight byte; per-column arithmetic unchanged (same k
// ranges per wave, same reduction order), so the results stay bit-identical to NT = 1 and to the one-tile kernels.
template <int WAVES, bool FP8, int NT>
__global__ __launch_bounds__(WAVES * 64) void gemm_skinny_mt2_kernel(const void* W_, const bf16_t* xp_, int KS_, int ks_per_split_, int n_tiles_, SkinnyArgs p_unused) {
    constexpr int CH = NT >= 3 ? 2 : 4;                    // k-steps per register chunk (three column tiles: 2, to stay under 256 VGPRs)
    constexpr int NBUF = (FP8 && NT == 1) ? 3 : 2;
    constexpr int WCH = FP8 ? CH / 2 : CH;                 // 16-byte weight loads per chunk, lane and column tile
    extern __shared__ __attribute__((aligned(16))) char sk_smem[];
    float (*red)[WAVES][16][64] = reinterpret_cast<float (*)[WAVES][16][64]>(sk_smem);          // [2][WAVES][16][64]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nt0 = blockIdx.x * NT, split = blockIdx.y, mt0 = blockIdx.z * 2;
    const int KS = KS_;                                    // k-steps of the whole K and of this block's split (host-computed:
    const int ks_per_split = ks_per_split_;                //  no integer division in front of the first load)
    const int ks_per_wave = ks_per_split / WAVES;          // fp8: even (launcher)
    const int ks0 = split * ks_per_split + wave * ks_per_wave;
    const int m = lane & 31, half = lane >> 5;

    const u32x4* wptr[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        int nt = nt0 + j;
        nt = nt < n_tiles_ ? nt : n_tiles_ - 1;             // odd tile count: the last block streams its only tile twice, stores it once
        wptr[j] = FP8 ? reinterpret_cast<const u32x4*>(W_) + ((size_t)nt * (KS >> 1) + (ks0 >> 1)) * 64 + lane
                      : reinterpret_cast<const u32x4*>(W_) + ((size_t)nt * KS + ks0) * 64 + lane;
    }
    const u32x4* xptr0 = reinterpret_cast<const u32x4*>(xp_) + ((size_t)mt0 * KS + ks0) * 64 + lane;
    const u32x4* xptr1 = xptr0 + (size_t)KS * 64;

    f32x16 acc[NT][2];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[j][0][r] = 0.f; acc[j][1][r] = 0.f; }

    constexpr int RPW = 16 / WAVES;                       // WAVES in {4, 8}
    struct Chunk { u32x4 w[NT][WCH]; u32x4 x0[CH]; u32x4 x1[CH]; };
    Chunk c[NBUF];
    // FULL = the chunk lies inside the wave's range: no guards, i.e. no branches around the loads -> exact s_waitcnt counts in the
    // steady-state loop (see sk_load_full)
    auto load = [&](Chunk& k, int ks, auto full) {         // ks multiple of CH; the last chunk of a wave may be ragged
        constexpr bool FULL = decltype(full)::value;
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int u = 0; u < WCH; ++u)
                if (FULL || ks + (FP8 ? 2 * u : u) < ks_per_wave)       // wave-uniform
                    k.w[j][u] = __builtin_nontemporal_load(wptr[j] + (size_t)(FP8 ? (ks >> 1) + u : ks + u) * 64);
#pragma unroll
        for (int u = 0; u < CH; ++u)
            if (FULL || ks + u < ks_per_wave) {
                k.x0[u] = xptr0[(size_t)(ks + u) * 64];
                k.x1[u] = xptr1[(size_t)(ks + u) * 64];
            }
    };
    auto compute = [&](Chunk& k, int ks, auto full) {
        constexpr bool FULL = decltype(full)::value;
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            if (!FULL && ks + u >= ks_per_wave) continue;   // wave-uniform
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                u32x4 wf;
                if constexpr (FP8) {
                    wf = fp8x8_to_bf16x8(k.w[j][u >> 1][(u & 1) * 2], k.w[j][u >> 1][(u & 1) * 2 + 1]);
                } else {
                    wf = k.w[j][u];
                }
                acc[j][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag4(wf), as_frag4(k.x0[u]), acc[j][0], 0, 0, 0);
                acc[j][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag4(wf), as_frag4(k.x1[u]), acc[j][1], 0, 0, 0);
            }
        }
    };
    SkinnyArgs p;
    float bias_d[NT][RPW];
    float4 sc4[NT][4];
    auto late = [&]() {                                     // the stream is in flight: now the rest of the arguments, bias, scales
        p = sv_late_args<SkinnyArgs>(offsetof(SkinnyKernarg, p));
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int nt = nt0 + j < n_tiles_ ? nt0 + j : n_tiles_ - 1;
            sk_bias<RPW>(p, bias_d[j], wave * RPW, nt, half);
            sk_settle<RPW>(bias_d[j]);
            if constexpr (FP8) {
                // per-column scale (accumulator row r <-> column 8 (r >> 2) + 4 half + (r & 3))
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    sc4[j][rg] = *reinterpret_cast<const float4*>(p.wscale + nt * 32 + rg * 8 + half * 4);
                    asm volatile("" : "+v"(sc4[j][rg].x), "+v"(sc4[j][rg].y), "+v"(sc4[j][rg].z), "+v"(sc4[j][rg].w));
                }
            }
        }
    };
    auto guarded = [&](int ks) {
        for (; ks < ks_per_wave; ks += NBUF * CH) {
#pragma unroll
            for (int b = 0; b < NBUF; ++b) {
                if (ks + b * CH < ks_per_wave) compute(c[b], ks + b * CH, std::false_type{});
                if (ks + (b + NBUF) * CH < ks_per_wave) load(c[b], ks + (b + NBUF) * CH, std::false_type{});
            }
        }
    };
    if (ks_per_wave >= 2 * NBUF * CH) {
        // long ranges: branch-free steady state (chunk b's MFMAs wait for chunk b only; the other chunks' loads stay in flight)
#pragma unroll
        for (int b = 0; b < NBUF; ++b) load(c[b], b * CH, std::true_type{});
        late();
        int ks = 0;
        for (; ks + 2 * NBUF * CH <= ks_per_wave; ks += NBUF * CH) {
#pragma unroll
            for (int b = 0; b < NBUF; ++b) {
                compute(c[b], ks + b * CH, std::true_type{});
                __builtin_amdgcn_sched_barrier(0);
                load(c[b], ks + (b + NBUF) * CH, std::true_type{});
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        guarded(ks);
    } else {
#pragma unroll
        for (int b = 0; b < NBUF; ++b)
            if (b * CH < ks_per_wave) load(c[b], b * CH, std::false_type{});
        late();
        guarded(0);
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        if constexpr (FP8) {
            // per-column scale before the reduction, as the one-tile kernel
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const float4 sc = sc4[j][rg];
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) {
                    acc[j][mi][rg * 4 + 0] *= sc.x; acc[j][mi][rg * 4 + 1] *= sc.y; acc[j][mi][rg * 4 + 2] *= sc.z; acc[j][mi][rg * 4 + 3] *= sc.w;
                }
            }
        }
        if (j > 0) __syncthreads();                        // the previous tile's reduction has been read
#pragma unroll
        for (int r = 0; r < 16; ++r) { red[0][wave][r][lane] = acc[j][0][r]; red[1][wave][r][lane] = acc[j][1][r]; }
        __syncthreads();
        if (nt0 + j < n_tiles_) {                           // block-uniform
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                float v[RPW];
#pragma unroll
                for (int i = 0; i < RPW; ++i) {
                    const int r = wave * RPW + i;
                    float t = red[mi][0][r][lane];
#pragma unroll
                    for (int w = 1; w < WAVES; ++w) t += red[mi][w][r][lane];
                    v[i] = t;
                }
                sk_store<RPW>(p, v, bias_d[j], wave * RPW, nt0 + j, mt0 + mi, split, m, half);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// gemm_skinny_mt2x_kernel (round 6): the two-row-tile kernel with the ACTIVATIONS landing in LDS and twice the waves per CU.
//   gemm_skinny_mt2_kernel holds both operands of its two register chunks in VGPRs (NT = 3: ~250 registers -> one 8-wave block
//   per CU, 8 x 14 KiB = 112 KiB in flight per CU) and runs at 34 GB/s per CU where the traffic-only probe (tools/diag/mem_mix.hip)
//   does 70: the launch is bound by the bytes a CU has in flight, not by HBM or by the L2 -> CU side.  Here
//     * the activation fragments (2 KiB per k-step: two row tiles) go global -> LDS by LDS-DMA into a WAVE-PRIVATE ring of two chunks
//       (8 KiB per wave; the ring IS the cross-wave reduction buffer, used after the loop): in flight they cost no register;
//     * the weights go straight to registers as before, but through inline-asm loads: beside an LDS-DMA hipcc waits vmcnt(0) for
//       every ordinary register load (cdna guide, section 5 "three .s-level traps" (b)), which would drain the ring at every chunk;
//       the wave counts its own queue: per chunk NX LDS-DMA pieces, then NW weight loads, `s_waitcnt vmcnt(NX + NW)` before a chunk's
//       first use = "everything but the refill issued last"; the weight registers are re-defined by an empty asm behind the wait, so
//       no use can be scheduled above it (guide section 5.7, form (ii));
//     * <= 128 VGPRs (two column tiles per block): two blocks = 16 waves per CU, ~12 KiB in flight each.
//   Same per-wave k ranges, same MFMA order, same LDS reduction in wave order, same epilogues as gemm_skinny_mt2_kernel and the
//   one-tile kernels: bit-identical results (tests/test_gpu_ops.py), so a row's tokens still do not depend on the batch around it.
// ------------------------------------------------------------------------------------------------
//   CH = k-steps per chunk: 2 -> 8 KiB of ring per wave, 64 KiB per block, <= 128 VGPRs: two blocks per CU (launches of >= 512 blocks);
//        4 -> 16 KiB per wave, 128 KiB per block, one block per CU, up to three column tiles per block (the shapes whose tile count
//        cannot fill 512 block slots: the depth per wave has to make up for the second block)
template <bool FP8, int NT, int CH>
__global__ __launch_bounds__(512, CH == 2 ? 4 : 2) void gemm_skinny_mt2x_kernel(const void* W_, const bf16_t* xp_, int KS_, int ks_per_split_, int n_tiles_, SkinnyArgs p_unused) {
    constexpr int WAVES = 8, RPW = 2;
    constexpr int RING = 2 * CH * 2048;                    // bytes of ring per wave
    constexpr int WCH = FP8 ? CH / 2 : CH;                 // 16-byte weight loads per chunk, lane and column tile
    constexpr int NW = NT * WCH, NX = 2 * CH;              // weight loads / LDS-DMA pieces a wave issues per chunk
    extern __shared__ __attribute__((aligned(16))) char sk_smem[];
    float (*red)[WAVES][16][64] = reinterpret_cast<float (*)[WAVES][16][64]>(sk_smem);          // [2][WAVES][16][64], after the loop
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* const xring = sk_smem + wave * RING;             // [2 slots][CH k-steps][2 row tiles][1 KiB]
    const int nt0 = blockIdx.x * NT, split = blockIdx.y, mt0 = blockIdx.z * 2;
    const int KS = KS_;
    const int ks_per_split = ks_per_split_;
    const int ks_per_wave = ks_per_split / WAVES;          // a multiple of CH, >= 2 * CH (launcher)
    const int ks0 = split * ks_per_split + wave * ks_per_wave;
    const int m = lane & 31, half = lane >> 5;

    const u32x4* wptr[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        int nt = nt0 + j;
        nt = nt < n_tiles_ ? nt : n_tiles_ - 1;
        wptr[j] = FP8 ? reinterpret_cast<const u32x4*>(W_) + ((size_t)nt * (KS >> 1) + (ks0 >> 1)) * 64 + lane
                      : reinterpret_cast<const u32x4*>(W_) + ((size_t)nt * KS + ks0) * 64 + lane;
    }
    const u32x4* xg0 = reinterpret_cast<const u32x4*>(xp_) + ((size_t)mt0 * KS + ks0) * 64 + lane;
    const u32x4* xg1 = xg0 + (size_t)KS * 64;

    f32x16 acc[NT][2];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[j][0][r] = 0.f; acc[j][1][r] = 0.f; }

    u32x4 wa[NT][WCH], wb[NT][WCH];                        // the two register slots of the weight stream (statically named: no rotation copies)
    auto issue = [&](int slot, int ks, u32x4 (&w)[NT][WCH]) {        // chunk = k-steps [ks, ks + CH) of this wave's range
        char* dst = xring + slot * (CH * 2048);
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            lds_dma16(xg0 + (size_t)(ks + u) * 64, dst + u * 2048);
            lds_dma16(xg1 + (size_t)(ks + u) * 64, dst + u * 2048 + 1024);
        }
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int u = 0; u < WCH; ++u) {
                const u32x4* g = wptr[j] + (size_t)(FP8 ? (ks >> 1) + u : ks + u) * 64;
                asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(w[j][u]) : "v"(g) : "memory");
            }
    };
    // everything but the LAST `keep` loads of this wave's queue has landed; the slot's weight registers are (re)defined here
    auto landed = [&](auto keep_tag, u32x4 (&w)[NT][WCH]) {
        constexpr int KEEP = decltype(keep_tag)::value;
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(KEEP) : "memory");
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int u = 0; u < WCH; ++u) asm volatile("" : "+v"(w[j][u]));
    };
    auto compute = [&](int slot, u32x4 (&w)[NT][WCH]) {
        const char* xs = xring + slot * (CH * 2048) + lane * 16;
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const u32x4 x0 = *reinterpret_cast<const u32x4*>(xs + u * 2048);
            const u32x4 x1 = *reinterpret_cast<const u32x4*>(xs + u * 2048 + 1024);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                u32x4 wf;
                if constexpr (FP8) wf = fp8x8_to_bf16x8(w[j][u >> 1][(u & 1) * 2], w[j][u >> 1][(u & 1) * 2 + 1]);
                else wf = w[j][u];
                acc[j][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag4(wf), as_frag4(x0), acc[j][0], 0, 0, 0);
                acc[j][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_frag4(wf), as_frag4(x1), acc[j][1], 0, 0, 0);
            }
        }
        // the slot's LDS reads are complete before its refill is issued (program order + the MFMAs above consumed them); make it explicit
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    // epilogue constants of the block's NT x 32 columns (fp8 scale, bias) live in LDS behind the ring, not in registers across the loop
    float* cs = reinterpret_cast<float*>(sk_smem + WAVES * RING);  // [NT][32] per-column scale
    float* cb = cs + NT * 32;                               // [NT][32] bias (PACKED_ACT), else 0
    SkinnyArgs p;
    auto late = [&]() {                                     // the stream is in flight: now the rest of the arguments
        p = sv_late_args<SkinnyArgs>(offsetof(SkinnyKernarg, p));
        if (tid < NT * 32) {
            const int j = tid >> 5;
            const int nt = nt0 + j < n_tiles_ ? nt0 + j : n_tiles_ - 1;
            const int n = nt * 32 + (tid & 31);
            float sc = 1.f, bi = 0.f;
            if constexpr (FP8) sc = p.wscale[n];
            if (p.out_mode == SK_OUT_PACKED_ACT && n < p.N && p.bias) bi = bf2f(p.bias[n]);
            cs[tid] = sc;
            cb[tid] = bi;
        }
    };
    const int nch = ks_per_wave / CH;                       // >= 2
    issue(0, 0, wa);
    issue(1, CH, wb);
    late();
    using Keep1 = std::integral_constant<int, NX + NW>;     // the refill issued last stays in flight
    using Keep0 = std::integral_constant<int, 0>;
    int c = 0;
    for (; c + 3 < nch; c += 2) {
        landed(Keep1{}, wa); compute(0, wa); issue(0, (c + 2) * CH, wa);
        landed(Keep1{}, wb); compute(1, wb); issue(1, (c + 3) * CH, wb);
    }
    {   // two or three chunks left: c, c + 1 (in flight), c + 2 (not yet issued)
        const bool three = c + 2 < nch;                     // wave-uniform
        landed(Keep1{}, wa); compute(0, wa);
        if (three) issue(0, (c + 2) * CH, wa);
        if (three) landed(Keep1{}, wb); else landed(Keep0{}, wb);
        compute(1, wb);
        if (three) { landed(Keep0{}, wa); compute(0, wa); }
    }
    __syncthreads();                                        // every wave is done with its ring: the buffer becomes the reduction buffer
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        if constexpr (FP8) {
            // per-column scale before the reduction, as the one-tile kernel (accumulator row r <-> column 8 (r >> 2) + 4 half + (r & 3))
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const float4 sc = *reinterpret_cast<const float4*>(cs + j * 32 + rg * 8 + half * 4);
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) {
                    acc[j][mi][rg * 4 + 0] *= sc.x; acc[j][mi][rg * 4 + 1] *= sc.y; acc[j][mi][rg * 4 + 2] *= sc.z; acc[j][mi][rg * 4 + 3] *= sc.w;
                }
            }
        }
        float bias_d[RPW];
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const int r = wave * RPW + i;
            bias_d[i] = cb[j * 32 + 8 * (r >> 2) + 4 * half + (r & 3)];
        }
        if (j > 0) __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r) { red[0][wave][r][lane] = acc[j][0][r]; red[1][wave][r][lane] = acc[j][1][r]; }
        __syncthreads();
        if (nt0 + j < n_tiles_) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                float v[RPW];
#pragma unroll
                for (int i = 0; i < RPW; ++i) {
                    const int r = wave * RPW + i;
                    float t = red[mi][0][r][lane];
#pragma unroll
                    for (int w = 1; w < WAVES; ++w) t += red[mi][w][r][lane];
                    v[i] = t;
                }
                sk_store<RPW>(p, v, bias_d, wave * RPW, nt0 + j, mt0 + mi, split, m, half);
            }
        }
    }
}

#define MT2X_SMEM(ch) (8 * 2 * (ch) * 2048 + 2 * 3 * 32 * 4)      // the ring / reduction buffer + [NT <= 3][32] scales + biases
std::atomic<int> g_mt2x{1};             // 33..64 rows: 0 = gemm_skinny_mt2_kernel, 1 = the LDS-ring form (chunk depth by block count), 2 / 3 = its CH = 2 / 4 form always
void set_mt2x(int on) { g_mt2x = on; }

std::atomic<int> g_op_col_tiles{0};     // op-level entry points only (SkinnyArgs.col_tiles == 0); engines always pass their plan
static int init_mt2_attrs() {
    int r = 0;
    auto set = [&](const void* f, int bytes) { if (!r) r = (int)hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes); };
    set(reinterpret_cast<const void*>(&gemm_skinny_mt2_kernel<8, true, 1>), 2 * 8 * 16 * 64 * 4);
    set(reinterpret_cast<const void*>(&gemm_skinny_mt2_kernel<8, false, 1>), 2 * 8 * 16 * 64 * 4);
    set(reinterpret_cast<const void*>(&gemm_skinny_mt2_kernel<8, true, 2>), 2 * 8 * 16 * 64 * 4);
    set(reinterpret_cast<const void*>(&gemm_skinny_mt2_kernel<8, false, 2>), 2 * 8 * 16 * 64 * 4);
    set(reinterpret_cast<const void*>(&gemm_skinny_mt2_kernel<8, true, 3>), 2 * 8 * 16 * 64 * 4);
    set(reinterpret_cast<const void*>(&gemm_skinny_mt2_kernel<8, false, 3>), 2 * 8 * 16 * 64 * 4);
    set(reinterpret_cast<const void*>(&gemm_skinny_mt2x_kernel<true, 1, 2>), MT2X_SMEM(2));
    set(reinterpret_cast<const void*>(&gemm_skinny_mt2x_kernel<false, 1, 2>), MT2X_SMEM(2));
    set(reinterpret_cast<const void*>(&gemm_skinny_mt2x_kernel<true, 2, 2>), MT2X_SMEM(2));
    set(reinterpret_cast<const void*>(&gemm_skinny_mt2x_kernel<false, 2, 2>), MT2X_SMEM(2));
    set(reinterpret_cast<const void*>(&gemm_skinny_mt2x_kernel<true, 1, 4>), MT2X_SMEM(4));
    set(reinterpret_cast<const void*>(&gemm_skinny_mt2x_kernel<false, 1, 4>), MT2X_SMEM(4));
    set(reinterpret_cast<const void*>(&gemm_skinny_mt2x_kernel<true, 2, 4>), MT2X_SMEM(4));
    set(reinterpret_cast<const void*>(&gemm_skinny_mt2x_kernel<false, 2, 4>), MT2X_SMEM(4));
    set(reinterpret_cast<const void*>(&gemm_skinny_mt2x_kernel<true, 3, 4>), MT2X_SMEM(4));
    set(reinterpret_cast<const void*>(&gemm_skinny_mt2x_kernel<false, 3, 4>), MT2X_SMEM(4));
    return r;
}
// two-row-tile launch: false when the shape / mode is outside the kernel's scope (the caller falls back to one tile per block)
static bool launch_gemm_skinny_mt2(const SkinnyArgs& a, hipStream_t st) {
    if (a.MT < 2 || (a.MT & 1)) return false;
    if (!(a.out_mode == SK_OUT_PARTIAL || ((a.out_mode == SK_OUT_PACKED_ACT || a.out_mode == SK_OUT_F32) && a.splitk == 1)))
        return false;
    if (a.Wq && !a.wscale) return false;
    if ((a.K / 16) % a.splitk) return false;
    const int n_tiles = a.Npad / 32, KS = a.K / 16, per = KS / a.splitk;
    // the SAME number of waves (= the same per-wave k ranges and reduction order) as the one-tile kernel of this GEMM
    const int waves = a.Wq ? skinny_waves_fp8(KS, a.splitk) : skinny_waves(a.Npad, KS, a.splitk);
    const void* W = a.Wq ? (const void*)a.Wq : (const void*)a.Wp;
    const int mode = g_mt2x.load(std::memory_order_relaxed);
    if (waves == 8 && mode && per % 8 == 0 && (per / 8) % 2 == 0 && per / 8 >= 4) {
        // the LDS-ring form (gemm_skinny_mt2x_kernel).  Column tiles per block = the plan's (the engine's pick_decode_plan; 0: this launcher's
        // default).  Chunk depth: launches that fill 512 block slots take the two-blocks-per-CU form (CH = 2, <= 2 column tiles); the others
        // one block per CU with 16 KiB of ring per wave (CH = 4) when the wave's k range is a multiple of 4 k-steps and >= 8.
        int ct = a.col_tiles ? a.col_tiles : g_op_col_tiles.load(std::memory_order_relaxed);
        if (ct == 0) ct = n_tiles * a.splitk >= 512 ? 2 : 1;
        const int pw = per / 8;
        const long blocks2 = (long)((n_tiles + 1) / 2) * a.splitk * (a.MT / 2), blocks1 = (long)n_tiles * a.splitk * (a.MT / 2);
        const bool deep_ok = pw % 4 == 0 && pw >= 8;
        int ch = 2;
        if (mode == 2) ch = 2; else if (mode == 3) ch = deep_ok ? 4 : 2;
        else ch = (deep_ok && (ct >= 3 || (ct == 2 ? blocks2 : blocks1) < 512)) ? 4 : 2;
        if (ch == 2 && ct > 2) ct = 2;
        const dim3 grid((n_tiles + ct - 1) / ct, a.splitk, a.MT / 2);
#define SV_MT2X(F, N_, C_) gemm_skinny_mt2x_kernel<F, N_, C_><<<grid, 512, MT2X_SMEM(C_), st>>>(W, a.xp, KS, per, n_tiles, a)
        const bool f8 = a.Wq != nullptr;
        if (ch == 2) {
            if (ct == 2) { if (f8) SV_MT2X(true, 2, 2); else SV_MT2X(false, 2, 2); }
            else { if (f8) SV_MT2X(true, 1, 2); else SV_MT2X(false, 1, 2); }
        } else {
            if (ct == 3) { if (f8) SV_MT2X(true, 3, 4); else SV_MT2X(false, 3, 4); }
            else if (ct == 2) { if (f8) SV_MT2X(true, 2, 4); else SV_MT2X(false, 2, 4); }
            else { if (f8) SV_MT2X(true, 1, 4); else SV_MT2X(false, 1, 4); }
        }
#undef SV_MT2X
        return true;
    }
    if (waves == 8) {
        // the engine picks (column tiles, split-K) together (engine_core.hip, pick_decode_plan); on its own (col_tiles = 0: the op-level
        // entry points) the launcher takes two column tiles when half the blocks still cover the chip
        const int ct = a.col_tiles ? a.col_tiles : g_op_col_tiles.load(std::memory_order_relaxed);
        if (ct == 3) {
            const dim3 grid((n_tiles + 2) / 3, a.splitk, a.MT / 2);
            if (a.Wq) gemm_skinny_mt2_kernel<8, true, 3><<<grid, 512, 2 * 8 * 16 * 64 * 4, st>>>(W, a.xp, KS, per, n_tiles, a);
            else gemm_skinny_mt2_kernel<8, false, 3><<<grid, 512, 2 * 8 * 16 * 64 * 4, st>>>(W, a.xp, KS, per, n_tiles, a);
            return true;
        }
        if (ct == 2 || (ct == 0 && n_tiles * a.splitk >= 512)) {
            const dim3 grid((n_tiles + 1) / 2, a.splitk, a.MT / 2);
            if (a.Wq) gemm_skinny_mt2_kernel<8, true, 2><<<grid, 512, 2 * 8 * 16 * 64 * 4, st>>>(W, a.xp, KS, per, n_tiles, a);
            else gemm_skinny_mt2_kernel<8, false, 2><<<grid, 512, 2 * 8 * 16 * 64 * 4, st>>>(W, a.xp, KS, per, n_tiles, a);
            return true;
        }
        const dim3 grid(n_tiles, a.splitk, a.MT / 2);
        if (a.Wq) gemm_skinny_mt2_kernel<8, true, 1><<<grid, 512, 2 * 8 * 16 * 64 * 4, st>>>(W, a.xp, KS, per, n_tiles, a);
        else gemm_skinny_mt2_kernel<8, false, 1><<<grid, 512, 2 * 8 * 16 * 64 * 4, st>>>(W, a.xp, KS, per, n_tiles, a);
        return true;
    }
    if (waves == 4) {
        const dim3 grid(n_tiles, a.splitk, a.MT / 2);
        if (a.Wq) gemm_skinny_mt2_kernel<4, true, 1><<<grid, 256, 2 * 4 * 16 * 64 * 4, st>>>(W, a.xp, KS, per, n_tiles, a);
        else gemm_skinny_mt2_kernel<4, false, 1><<<grid, 256, 2 * 4 * 16 * 64 * 4, st>>>(W, a.xp, KS, per, n_tiles, a);
        return true;
    }
    return false;       // 16-wave (narrow outputs) and 1/2-wave (tiny K) shapes keep one row tile per block
}

void skinny_plan(int Npad, int K, int splitk, int fp8, int MT, int* waves, int* two_row_tiles) {
    const int KS = K / 16;
    const int w = fp8 ? skinny_waves_fp8(KS, splitk) : skinny_waves(Npad, KS, splitk);
    *waves = w;
    *two_row_tiles = (MT >= 2 && !(MT & 1) && (w == 8 || w == 4) && KS % splitk == 0) ? 1 : 0;
}

// false: outside the scope of the tail split (see gemm_skinny_tailsplit_kernel)
static bool launch_skinny_tailsplit(const SkinnyArgs& a, hipStream_t st) {
    const char* ev = getenv("SV_TAILSPLIT");                 // read per call (A/B in one process; a captured graph keeps its choice)
    const int on = ev ? atoi(ev) : g_tailsplit.load(std::memory_order_relaxed);
    if (!on || !a.tail_ws || !a.tail_cnt) return false;
    if (a.out_mode != SK_OUT_PACKED_ACT || a.MT != 1 || a.Wq || a.fold_c1 || a.splitk != 1) return false;
    const int KS = a.K / 16, n_tiles = a.Npad / 32;
    if (KS % 32 || KS / 32 != 9 || skinny_waves(a.Npad, KS, 1) != 8) return false;      // the one instantiation: 9 k-steps per wave and quarter (K = 4608)
    static int cus = 0;
    if (!cus) {
        int dev = 0; hipDeviceProp_t pr;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) return false;
        cus = pr.multiProcessorCount;
    }
    const int slots = 2 * cus, T = n_tiles - slots;           // two 8-wave blocks of the ordinary kernel per CU
    if (T <= 0 || T > SV_TAIL_TILES || 4 * T > slots) return false;
    gemm_skinny_kernel<8, false><<<dim3(slots, 1, 1), 512, skinny_smem(8), st>>>(a.Wp, a.xp, KS, KS, 0, a);
    gemm_skinny_tailsplit_kernel<9><<<4 * T, 512, 8 * 16 * 64 * 4, st>>>(a.Wp, a.xp, KS, slots, 0, a);
    return true;
}

void launch_gemm_skinny(const SkinnyArgs& a, hipStream_t st) {
    if (launch_head_persist(a, st)) return;                      // the lm_head of a <= 32-row step: one round of blocks over several column tiles each
    if (launch_gemm_skinny_mt2(a, st)) return;                  // 33..64 rows: two row tiles per block, weights streamed once
    if (a.Wq && launch_gemm_skinny_fp8(a, st)) return;       // fp8 weights: its own kernel (falls through if unsupported)
    if (launch_skinny_tailsplit(a, st)) return;                 // whole-K tiles beyond the first round of blocks: split four ways along K (StarVector-8B's c_fc)
    dim3 grid(a.Npad / 32, a.splitk, a.MT);
    switch (skinny_waves(a.Npad, a.K / 16, a.splitk)) {
        case 16: launch_sk<16>(a, grid, st); break;
        case 8: launch_sk<8>(a, grid, st); break;
        case 4: launch_sk<4>(a, grid, st); break;
        case 2: launch_sk<2>(a, grid, st); break;
        default: launch_sk<1>(a, grid, st); break;
    }
}

// f32 -> bf16 through the hardware convert every epilogue uses (test surface)
__global__ void cvt_bf16_hw_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, size_t n) {
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (i + 1 < n) {
        *reinterpret_cast<uint32_t*>(y + i) = cvt_pk_bf16(x[i], x[i + 1]);
    } else if (i < n) {
        y[i] = (bf16_t)(cvt_pk_bf16(x[i], 0.f) & 0xffffu);
    }
}
void launch_cvt_bf16_hw(const float* x, bf16_t* y, size_t n, hipStream_t st) {
    cvt_bf16_hw_kernel<<<(unsigned)((n / 2 + 256) / 256), 256, 0, st>>>(x, y, n);
}

}  // namespace sv
