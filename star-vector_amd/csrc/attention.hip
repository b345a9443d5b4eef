// Attention kernels for gfx950.
//
//  * attn_prefill_kernel : flash-style forward (no S x S matrix) for the ViT's bidirectional MHSA
//    (clip_model.py:130-155, nn.MultiheadAttention 16 x 64) and the decoder's causal MQA prefill
//    (gpt_bigcode/modeling_gpt_bigcode.py:151-285).  v_mfma_f32_32x32x16_bf16 with the "swapped"
//    products  S^T = K.Q^T  and  O^T = V^T.P^T : every lane owns ONE query row, so the online-softmax
//    statistics and the rescale factor are lane-local (one cross-half shuffle per tile), and P feeds the
//    second MFMA straight from registers (the key order inside a 16-key k-step is a free permutation,
//    applied when V is transposed into LDS).
//  * attn_decode_kernel  : single-token MQA/GQA attention over the PAGED KV cache (HBM-bound).  The
//    16 query heads that share one KV head form the N=16 side of v_mfma_f32_16x16x32_bf16, so every
//    K/V byte is read once for all heads.  K and V^T live in the cache in MFMA fragment order: each
//    wave-load is one contiguous 1 KiB.  The kernel also finishes the c_attn split-K reduction
//    (+bias, bf16 round) for its sequence and appends the new token's K/V to the cache.
#include "kernels.h"

namespace sv {

__device__ __forceinline__ int swap23(int x) { return (x & ~0xC) | ((x & 4) << 1) | ((x & 8) >> 1); }

// ------------------------------------------------------------------------------------------------
// prefill
// ------------------------------------------------------------------------------------------------
// (Round 5 measured a form with ONE block per (image, head) and three 32-row query tiles per wave for the ViT's S = 257 -- no third block for one
// row, every K / V tile fetched and transposed once instead of three times, 512 blocks = one round: the same bits, and no faster (encoder attention
// 1.04 vs 0.96 ms of TTFT on a box whose GEMMs ran 8 % slower).  The launch is bound by instruction issue per SIMD -- per 32 x 64 tile pair ~300 VALU
// instructions of mask / exp / rescale / pack next to 16 MFMAs -- not by the global -> LDS -> barrier chain the restructure removed; 45 of those
// tile pairs cover 257 x 257 scores where 32.3 would do (the 257th key and the 257th row each cost a whole tile).  profiles/SUMMARY_r05.md; removed.)
template <int D, int HPB>
__global__ __launch_bounds__(256, 2) void attn_prefill_kernel(AttnPrefillArgs p) {      // two blocks per CU: <= 256 registers

    constexpr int KSTR = D + 8;          // K tile row stride (elements): +16 B pad -> conflict-free b128
    constexpr int VSTR = 64 + 8;         // V^T tile row stride
    constexpr int NKS = D / 16;          // k-steps of the QK^T product
    constexpr int NDV = D / 32;          // 32-wide dv tiles of the output
    __shared__ __attribute__((aligned(16))) bf16_t Ks[64 * KSTR];
    __shared__ __attribute__((aligned(16))) bf16_t Vt[D * VSTR];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane >> 5;
    const int b = blockIdx.z;
    const int S = p.S;
    int head, q0, qblock_end;
    // (q_tile0 > 0: only the trailing query tiles, e.g. the last prompt row's.  Round 6 numbered the causal tiles longest-first -- the last query tile walks 5 key tiles,
    // the first 1 -- so that the short ones fill the end of the launch: 41.4 vs 40.8 us per layer, nothing; profiles/prefill_small_r06.log.)
    const int qtile = blockIdx.x + p.q_tile0;
    if (HPB == 1) {
        head = blockIdx.y;
        q0 = qtile * 128 + wave * 32;
        qblock_end = qtile * 128 + 128;
    } else {
        head = blockIdx.y * HPB + wave;
        q0 = qtile * 32;
        qblock_end = q0 + 32;
    }
    const int kvh = (HPB == 1 ? head : blockIdx.y * HPB) / p.kv_group;
    const int qabs = q0 + (lane & 31);
    const int qrow = qabs < S ? qabs : S - 1;

    // Q fragments (B operand): lane (q = l&31, c = l>>5) holds Q[q][16 s + 8 c .. +8]
    bf16x8 qf[NKS];
    {
        const bf16_t* qp = p.q + ((size_t)b * S + qrow) * p.q_row_stride + (size_t)head * p.q_head_stride + c * 8;
#pragma unroll
        for (int s = 0; s < NKS; ++s) qf[s] = as_frag(*reinterpret_cast<const uint4*>(qp + s * 16));
    }

    f32x16 accO[NDV];
#pragma unroll
    for (int t = 0; t < NDV; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) accO[t][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;                     // m_run: running row maximum in the log2 domain (score * scale * log2 e)
    const float c2 = p.scale * 1.4426950408889634f;

    int last = (qblock_end < S ? qblock_end : S) - 1;        // last query row of the block
    const int ntiles = p.causal ? (last / 64 + 1) : ((S + 63) / 64);
    // sliding window (StarCoder2, prompts longer than the window): key tiles entirely below the block's first query's window
    // are never loaded; inside the first tiles a query row may see nothing yet (statistics stay at -inf / 0, see m_use below)
    const int win = p.causal ? p.window : 0;
    const int qfirst = HPB == 1 ? qtile * 128 : q0;
    const int kt0 = (win > 0 && qfirst - win + 1 > 0) ? (qfirst - win + 1) / 64 : 0;
    const bf16_t* kbase = p.k + (size_t)b * S * p.kv_row_stride + (size_t)kvh * p.kv_head_stride;
    const bf16_t* vbase = p.v + (size_t)b * S * p.kv_row_stride + (size_t)kvh * p.kv_head_stride;

    // K / V tiles go global -> registers -> LDS, one tile AHEAD: the loads of tile kt + 1 are issued before the MFMAs of tile kt
    // and land while they run.  (Loading inside the tile loop, V chunk by chunk with the transposing LDS writes in between,
    // was a chain of 4-5 dependent global round trips per tile at 2 blocks per CU: 84 us for 8.8 GFLOP of causal MQA.)
    constexpr int KPT = (64 * (D / 8)) / 256;      // 16-byte K chunks per thread and tile
    constexpr int VPT = (D / 8) / 4;               // 16-byte V chunks per thread and tile (lane = key, wave strides the chunks)
    u32x4 kreg[KPT], vreg[VPT];                    // (native vectors: arrays of HIP's uint4 class ended up on the stack here)
    auto gload = [&](int kt) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const int idx = tid + i * 256;
            const int row = idx / (D / 8), ch = idx % (D / 8);
            int key = kt * 64 + row;
            key = key < S ? key : S - 1;
            kreg[i] = *reinterpret_cast<const u32x4*>(kbase + (size_t)key * p.kv_row_stride + ch * 8);
        }
        int key = kt * 64 + lane;
        key = key < S ? key : S - 1;
#pragma unroll
        for (int i = 0; i < VPT; ++i)
            vreg[i] = *reinterpret_cast<const u32x4*>(vbase + (size_t)key * p.kv_row_stride + (wave + 4 * i) * 8);
    };
    auto lstore = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < KPT; ++i) {                      // K tile: 64 keys x D, row-major
            const int idx = tid + i * 256;
            const int row = idx / (D / 8), ch = idx % (D / 8);
            *reinterpret_cast<u32x4*>(Ks + row * KSTR + ch * 8) = kreg[i];
        }
        const int col = swap23(lane);                        // V tile transposed: lane <-> key, Vt[dv][swap23(key)]
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            bf16_t* dst = Vt + (wave + 4 * i) * 8 * VSTR + col;
            const u32x4 v = vreg[i];
            dst[0 * VSTR] = (bf16_t)(v[0] & 0xffffu); dst[1 * VSTR] = (bf16_t)(v[0] >> 16);
            dst[2 * VSTR] = (bf16_t)(v[1] & 0xffffu); dst[3 * VSTR] = (bf16_t)(v[1] >> 16);
            dst[4 * VSTR] = (bf16_t)(v[2] & 0xffffu); dst[5 * VSTR] = (bf16_t)(v[2] >> 16);
            dst[6 * VSTR] = (bf16_t)(v[3] & 0xffffu); dst[7 * VSTR] = (bf16_t)(v[3] >> 16);
        }
    };
    gload(kt0);
    for (int kt = kt0; kt < ntiles; ++kt) {
        if (kt > kt0) __syncthreads();                       // every wave is done with the previous tile
        lstore();
        __syncthreads();
        if (kt + 1 < ntiles) gload(kt + 1);
        // a wave whose 32 query rows all lie behind the sequence (S = 257: three of the four waves of every third block) only helps with the K / V tiles
        if (q0 >= S) continue;

        // S^T = K . Q^T : two 32-key sub-tiles
        f32x16 accS[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int r = 0; r < 16; ++r) accS[j][r] = 0.f;
#pragma unroll
            for (int s = 0; s < NKS; ++s) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ks + (32 * j + (lane & 31)) * KSTR + 16 * s + 8 * c);
                accS[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[s], accS[j], 0, 0, 0);
            }
        }
        // mask + online softmax; lane holds keys kt*64 + 32 j + (r&3) + 8 (r>>2) + 4 c of its query row
        float mt = -INFINITY;
        // Interior tiles need no mask: every key of the tile exists, lies at or below every query row of this wave (causal) and inside every
        // row's window.  The test is wave-uniform; the masked path costs 4-5 VALU instructions per score and the launch is issue-bound
        // (round 5: ~300 VALU instructions next to 16 MFMAs per tile pair).  Same values either way.
        const int k_lo = kt * 64, k_hi = kt * 64 + 63;
        const bool interior = k_hi < S && (!p.causal || k_hi <= q0) && (win <= 0 || k_lo > q0 + 31 - win);
        // the softmax runs in the log2 domain on the RAW products: p = 2^(s * c2 - m2) with c2 = scale * log2(e) -- one fma + one v_exp_f32 per
        // score instead of mul (scale), sub, mul (log2 e), v_exp_f32; the row maximum is taken before the scaling (c2 > 0)
        if (interior) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) mt = fmaxf(mt, accS[j][r]);
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = kt * 64 + 32 * j + (r & 3) + 8 * (r >> 2) + 4 * c;
                    const bool ok = key < S && (!p.causal || key <= qabs) && (win <= 0 || key > qabs - win);
                    const float sc = ok ? accS[j][r] : -INFINITY;
                    accS[j][r] = sc;
                    mt = fmaxf(mt, sc);
                }
        }
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64)) * c2;             // (-inf stays -inf)
        const float m_new = fmaxf(m_run, mt);
        const float m_use = m_new == -INFINITY ? 0.f : m_new;    // a row that has seen no key yet (window): 2^(-inf - 0) = 0, no NaN
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);
        float ls = 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(fmaf(accS[j][r], c2, -m_use));
                accS[j][r] = pv;
                ls += pv;
            }
        l_run = l_run * alpha + ls;
        m_run = m_new;
        // P fragments (B operand of O^T = V^T.P^T): element e of k-step (j, s2) = accS[j][8 s2 + e]
        bf16x8 pf[2][2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                uint4 u;
                u.x = pack2bf(accS[j][8 * s2 + 0], accS[j][8 * s2 + 1]);
                u.y = pack2bf(accS[j][8 * s2 + 2], accS[j][8 * s2 + 3]);
                u.z = pack2bf(accS[j][8 * s2 + 4], accS[j][8 * s2 + 5]);
                u.w = pack2bf(accS[j][8 * s2 + 6], accS[j][8 * s2 + 7]);
                pf[j][s2] = as_frag(u);
            }
        // the running maximum rarely moves after the first tiles: alpha == exp(0) == 1 for every row of the wave -> the rescale of the
        // 16 x NDV accumulators is the identity and is skipped (wave-uniform; same values)
        const bool rescale = !__all(alpha == 1.0f);
#pragma unroll
        for (int t = 0; t < NDV; ++t) {
            if (rescale) {
#pragma unroll
                for (int r = 0; r < 16; ++r) accO[t][r] *= alpha;
            }
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const bf16x8 vf = *reinterpret_cast<const bf16x8*>(
                        Vt + (32 * t + (lane & 31)) * VSTR + 32 * j + 16 * s2 + 8 * c);
                    accO[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[j][s2], accO[t], 0, 0, 0);
                }
        }
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (qabs < S) {
        bf16_t* op = p.o + ((size_t)b * S + qabs) * p.o_row_stride + (size_t)head * D;
#pragma unroll
        for (int t = 0; t < NDV; ++t)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                uint2 o;
                o.x = pack2bf(accO[t][rg * 4 + 0] * inv, accO[t][rg * 4 + 1] * inv);
                o.y = pack2bf(accO[t][rg * 4 + 2] * inv, accO[t][rg * 4 + 3] * inv);
                *reinterpret_cast<uint2*>(op + 32 * t + 8 * rg + 4 * c) = o;
            }
    }
}

static bool mqa4_tile(const AttnPrefillArgs& a) { return (a.kv_group % 4 == 0) && (a.H % 4 == 0); }
void launch_attn_prefill(const AttnPrefillArgs& a_in, hipStream_t st) {
    const bool mqa4 = mqa4_tile(a_in);
    // last_rows > 0: only the query tiles that hold the last `last_rows` rows of every sequence (same bits for those rows: a row's
    // online softmax walks the key tiles in the same order whatever other query tiles are launched)
    AttnPrefillArgs a = a_in;
    const int qt = mqa4_tile(a) ? 32 : 128;
    const int tiles_all = (a.S + qt - 1) / qt;
    a.q_tile0 = a.last_rows > 0 ? (a.S - (a.last_rows < a.S ? a.last_rows : a.S)) / qt : 0;
    const int tiles = tiles_all - a.q_tile0;
    if (mqa4) {
        dim3 grid(tiles, a.H / 4, a.B);
        if (a.head_dim == 128) attn_prefill_kernel<128, 4><<<grid, 256, 0, st>>>(a);
        else attn_prefill_kernel<64, 4><<<grid, 256, 0, st>>>(a);
    } else {
        dim3 grid(tiles, a.H, a.B);
        if (a.head_dim == 128) attn_prefill_kernel<128, 1><<<grid, 256, 0, st>>>(a);
        else attn_prefill_kernel<64, 1><<<grid, 256, 0, st>>>(a);
    }
}

// ------------------------------------------------------------------------------------------------
// paged KV cache, fragment order.  One page = 64 tokens of one layer (one KV head):
//   K part  [4 key-groups of 16][D/32 k-steps][64 lanes][8] : lane (key = l&15, c = l>>4) holds
//           K[16 kg + key][32 s + 8 c + e]                         (A operand of S^T = K.Q^T)
//   V part  [2 key-groups of 32][D/16 dv-tiles][64 lanes][8] : lane (dv = l&15, c = l>>4) holds
//           V[32 g + 16 (e>>2) + 4 c + (e&3)][16 t + dv]            (A operand of O^T = V^T.P^T)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ size_t kv_k_offset(int D, int t64, int d) {
    const int kg = t64 >> 4, key = t64 & 15, s = d >> 5, c = (d >> 3) & 3, e = d & 7;
    return ((size_t)((kg * (D >> 5) + s) * 64 + c * 16 + key)) * 16 + e * 2;
}
__device__ __forceinline__ size_t kv_v_offset(int D, int t64, int dv) {
    const int g = t64 >> 5, k32 = t64 & 31, u = k32 >> 4, c = (k32 >> 2) & 3, r = k32 & 3;
    const int e = 4 * u + r, t = dv >> 4, dl = dv & 15;
    return (size_t)64 * D * 2 + ((size_t)((g * (D >> 4) + t) * 64 + c * 16 + dl)) * 16 + e * 2;
}

// Prompt rows -> pages.  One block per (32-token group, sequence): the K rows are 16-byte pieces of the fragment image as they are; the V rows go
// through LDS and leave as whole 16-byte fragment pieces (8 tokens of one dv column each) -- the first form stored V element by element, eight 2-byte
// stores per thread: 8.7 us per layer for 8.5 MB of traffic.  Tokens of the group behind the prompt get V = 0 (a V column is only ever read for keys the
// attention admits; zero is what the decode attention leaves behind a sequence's position as well).
__global__ __launch_bounds__(256) void kv_write_prefill_kernel(const bf16_t* __restrict__ qkv, int row_stride, int k_off, int v_off,
                                                               char* __restrict__ pool, const int32_t* __restrict__ table,
                                                               int max_pages, int B, int S0, int D) {
    extern __shared__ __attribute__((aligned(16))) char kvw_smem[];
    bf16_t* Vs = reinterpret_cast<bf16_t*>(kvw_smem);                 // [32][D + 8]
    const int VST = D + 8;
    const int b = blockIdx.y, tok0 = blockIdx.x * 32, NC = D >> 3;
    const int page_bytes = kv_page_bytes(D);
    char* page = pool + (size_t)table[b * max_pages + (tok0 >> 6)] * page_bytes;
    for (int i = threadIdx.x; i < 32 * NC; i += 256) {
        const int r = i / NC, ch = i - r * NC, tok = tok0 + r;
        uint4 kq = make_uint4(0u, 0u, 0u, 0u), vq = kq;
        if (tok < S0) {
            const bf16_t* src = qkv + ((size_t)b * S0 + tok) * row_stride;
            kq = *reinterpret_cast<const uint4*>(src + k_off + ch * 8);
            vq = *reinterpret_cast<const uint4*>(src + v_off + ch * 8);
            *reinterpret_cast<uint4*>(page + kv_k_offset(D, tok & 63, ch * 8)) = kq;
        }
        *reinterpret_cast<uint4*>(Vs + r * VST + ch * 8) = vq;
    }
    __syncthreads();
    const int g = (tok0 >> 5) & 1;                                    // the page's 32-key group
    for (int i = threadIdx.x; i < (D >> 4) * 64; i += 256) {
        const int t = i >> 6, l = i & 63, dl = l & 15, c = l >> 4;
        uint32_t w[4];
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) {
            const int e = 2 * e2;
            const int ka = 16 * (e >> 2) + 4 * c + (e & 3);           // e and e + 1: consecutive tokens
            w[e2] = (uint32_t)Vs[ka * VST + 16 * t + dl] | ((uint32_t)Vs[(ka + 1) * VST + 16 * t + dl] << 16);
        }
        *reinterpret_cast<uint4*>(page + (size_t)64 * D * 2 + ((size_t)((g * (D >> 4) + t) * 64 + l)) * 16) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}
// rotary embedding over the prompt rows (Starcoder2: apply_rotary_pos_emb on q and k before the cache / attention)
__global__ void rope_prefill_kernel(bf16_t* __restrict__ qkv, int row_stride, int rows, int S0, int n_heads, int D,
                                    const float* __restrict__ cos_t, const float* __restrict__ sin_t) {
    const int half = D >> 1;
    const size_t total = (size_t)rows * n_heads * half;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int d = (int)(i % half);
        const size_t t = i / half;
        const int hd = (int)(t % n_heads);
        const size_t row = t / n_heads;
        const int pos = (int)(row % S0);
        bf16_t* pnt = qkv + row * row_stride + (size_t)hd * D;
        const float x1 = bf2f(pnt[d]), x2 = bf2f(pnt[d + half]);
        const float cs = cos_t[(size_t)pos * half + d], sn = sin_t[(size_t)pos * half + d];
        pnt[d] = f2bf(bfround(x1 * cs) + bfround(-x2 * sn));
        pnt[d + half] = f2bf(bfround(x2 * cs) + bfround(x1 * sn));
    }
}
void launch_rope_prefill(bf16_t* qkv, int row_stride, int rows, int S0, int n_heads, int head_dim, const float* cos_t,
                         const float* sin_t, hipStream_t st) {
    size_t total = (size_t)rows * n_heads * (head_dim / 2);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 16384) blocks = 16384;
    rope_prefill_kernel<<<blocks, 256, 0, st>>>(qkv, row_stride, rows, S0, n_heads, head_dim, cos_t, sin_t);
}

void launch_kv_write_prefill(const bf16_t* qkv, int row_stride, int k_off, int v_off, char* pool_layer,
                             const int32_t* block_table, int max_pages, int B, int S0, int head_dim,
                             hipStream_t st) {
    dim3 grid((S0 + 31) / 32, B);
    kv_write_prefill_kernel<<<grid, 256, (size_t)32 * (head_dim + 8) * 2, st>>>(qkv, row_stride, k_off, v_off, pool_layer, block_table,
                                                                               max_pages, B, S0, head_dim);
}

// ------------------------------------------------------------------------------------------------
// decode: one block of 8 waves per (sequence, context split).  A single CU streams only ~25 GB/s from
// HBM, so the KV pages of one sequence are spread over `act = clamp(ceil(groups/4), 1, AD_SPLIT)` blocks
// (<= 64 KiB of KV each); the grid is (B, AD_SPLIT) so the captured hipGraph is static and the inactive
// splits exit at once.  Active blocks own the 32-key
// groups g = split + act*(wave + 8*i), prefetch the next group's fragments while the MFMAs of the
// current one run, and (act > 1) leave a partial (m, l, O) in HBM; an arrival ticket elects the last
// block, which merges the partials in split order (bitwise deterministic).  Hand-off = write-through
// (sc1) partial stores + drained ticket + sc1 loads in the merger: no fences, placement independent, and NOBODY WAITS: a launch may
// have more blocks than the chip has CUs (64 rows: 512).  Round 4 measured two other hand-offs and removed both (profiles/
// attn_trace_r04.log): an in-band one (split 0's block polls the others' partial slots: needs every block resident -> NaNs at 64 rows)
// and an XCD-local one (all splits of a sequence on one XCD, partials through its L2: bit-identical, but no faster once the grid
// dispatches every sequence's split 0 first -- the store acknowledgement was never an HBM round trip).
// ------------------------------------------------------------------------------------------------
#define AD_WAVES 8
#define AD_SPLIT 16
#define AD_GROUPS_PER_BLOCK 4

template <int D>
struct KvFrags {
    u32x4 k[2][D / 32];
    u32x4 v[D / 16];
};

template <int D>
__device__ __forceinline__ void load_group(KvFrags<D>& f, const char* pool, int page_index, int page_bytes,
                                           int g, int lane) {
    constexpr int NKS = D / 32, NDV = D / 16;
    const int t0 = g << 5;
    const char* page = pool + (size_t)page_index * page_bytes;
    const int half = (t0 >> 5) & 1;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int s = 0; s < NKS; ++s)
            f.k[u][s] = *reinterpret_cast<const u32x4*>(page + ((size_t)(((half * 2 + u) * NKS + s) * 64 + lane)) * 16);
#pragma unroll
    for (int t = 0; t < NDV; ++t)
        f.v[t] = *reinterpret_cast<const u32x4*>(page + (size_t)64 * D * 2 + ((size_t)((half * NDV + t) * 64 + lane)) * 16);
}

__device__ __forceinline__ unsigned xcc_id() {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 15u;
}

// Leading scalar parameters: what the first loads (position, block table, KV fragments, c_attn slabs) need; they arrive
// preloaded in SGPRs with the dispatch (see gemm_skinny_kernel), the struct is read later.
struct AttnDecodeKernarg { const int32_t* positions; const int32_t* block_table; char* pool_layer; const float* ws; size_t kv_head_stride;
                           int max_pages, n_kv, max_splits, window, gpb; AttnDecodeArgs p; };     // the kernarg segment
template <int D>
__global__ __launch_bounds__(AD_WAVES * 64) void attn_decode_kernel(const int32_t* positions_, const int32_t* block_table_, char* pool_layer_,
                                                                     const float* ws_, size_t kv_head_stride_, int max_pages_, int n_kv_,
                                                                     int max_splits_, int window_, int gpb_, AttnDecodeArgs p_unused) {
    constexpr int NKS = D / 32;          // k-steps of S^T (over head_dim)
    constexpr int NDV = D / 16;          // dv tiles of O^T
    constexpr int PART = 32 + 16 * D;    // floats per partial: m[16], l[16], O[16][D]
    const long long t_start = wall_clock64();         // 100 MHz; stored only when the trace buffer is on (tools/attn_trace.py)
    const int bx = blockIdx.x;               // (sequence, KV head)
    const int b = bx / n_kv_;
    const int kvh = bx % n_kv_;
    const int split = blockIdx.y;
    // the first 64 entries of this sequence's block-table row (4096 tokens) are requested together with the position: the
    // page of a key group is then a cross-lane read instead of a second dependent global round trip (position -> table -> KV)
    const int32_t* table = block_table_ + (size_t)b * max_pages_;
    const int tl = threadIdx.x & 63;
    const int tpre = tl < max_pages_ ? table[tl] : 0;
    const int pos = positions_[b];
    const int L = pos + 1;
    const int ngroups = (L + 31) >> 5;
    // StarCoder2 sliding window: only keys win0 <= j <= pos are visible; whole groups (and pages) below are skipped
    const int win0 = (window_ > 0 && L > window_) ? L - window_ : 0;
    const int g0 = win0 >> 5;
    int act = (ngroups - g0 + gpb_ - 1) / gpb_;         // gpb_: 32-key groups a block takes before another context split joins
    act = act > max_splits_ ? max_splits_ : act;      // host cap: B * act <= #CUs (one block per CU) and <= AD_SPLIT
    act = act < 1 ? 1 : act;
    // AttnDecodeArgs::poison: every block of the launch stores 16 bytes per thread of the pattern -- the inactive splits before they
    // leave, the active ones once their KV stream is in flight (the arguments are read late in both cases: off the critical path)
    // (poison2 = the next layer's LayerNorm output buffer: only ever touched with write-through stores and L1-bypassing loads -- by this
    //  launch, the lm_head launch and rowln_cattn_kernel -- so that no XCD's L2 can hold a line of it that memory does not)
    auto poison2_store = [&](const AttnDecodeArgs& q, unsigned off) {
        const __amdgpu_buffer_rsrc_t rsp = __builtin_amdgcn_make_buffer_rsrc(q.poison2, 0, q.poison2_bytes, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b128(u32x4{0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}, rsp, (int)off, 0, 16);      // sc1
    };
    auto poison = [&](const AttnDecodeArgs& q) {
        if (q.poison) {
            const unsigned off = ((blockIdx.y * gridDim.x + blockIdx.x) * (AD_WAVES * 64) + threadIdx.x) * 16u;
            if (off < q.poison_bytes)
                *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(q.poison) + off) = u32x4{0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
            else if (q.poison2 && off - q.poison_bytes < q.poison2_bytes) poison2_store(q, off - q.poison_bytes);
        } else if (q.poison2) {
            const unsigned off = ((blockIdx.y * gridDim.x + blockIdx.x) * (AD_WAVES * 64) + threadIdx.x) * 16u;
            if (off < q.poison2_bytes) poison2_store(q, off);
        }
    };
    if (split >= act) {
        poison(sv_late_args<AttnDecodeArgs>(offsetof(AttnDecodeKernarg, p)));
        return;
    }

    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16_t* q_s = reinterpret_cast<bf16_t*>(smem);                       // [16][D]
    bf16_t* kv_new = q_s + 16 * D;                                       // [2][D]
    float* m_s = reinterpret_cast<float*>(kv_new + 2 * D);               // [AD_WAVES][16]
    float* l_s = m_s + AD_WAVES * 16;                                    // [AD_WAVES][16]
    float* O_s = l_s + AD_WAVES * 16;                                    // [AD_WAVES][16][D]
    int* flag_s = reinterpret_cast<int*>(O_s + AD_WAVES * 16 * D);       // [4]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* const pool = pool_layer_ + (size_t)kvh * kv_head_stride_;

    const int page_bytes = kv_page_bytes(D);
    auto page_of = [&](int grp) -> int {                   // grp is wave-uniform
        const int pi = grp >> 1;                           // 64-token pages, 32-key groups
        return pi < 64 ? __builtin_amdgcn_readlane(tpre, pi) : table[pi];
    };
    // the KV stream does not depend on q: request the first group before anything else
    const int stride = act * AD_WAVES;
    int g = g0 + split + act * wave;
    KvFrags<D> fa, fb;
    if (g < ngroups) load_group<D>(fa, pool, page_of(g), page_bytes, g, lane);
    const long long t_kvreq = wall_clock64();          // position + table landed, first KV group requested

    // the KV stream is in flight: now the rest of the arguments (common.h sv_late_args)
    const AttnDecodeArgs p = sv_late_args<AttnDecodeArgs>(offsetof(AttnDecodeKernarg, p));
    poison(p);
    const int H = p.H;                       // all query heads of the model
    const int G = H / n_kv_;                 // query heads sharing this KV head (<= 16): the MFMA N side
    const int HD = G * D;                    // output columns owned by this block

    // q / k_new / v_new of this sequence -> LDS (bf16): the fp32 split-K slabs of the c_attn GEMM, summed here in slab order + bias.
    // 4 columns per thread and every load issued before the first add: one memory round trip.
    for (int c4 = tid; c4 < (16 * D + 2 * D) / 4; c4 += AD_WAVES * 64) {
        const int n = c4 * 4;                    // LDS slot: q rows [0,16*D) (rows >= G are zero), k_new, v_new
        int col;                                 // column of the c_attn output: q heads | k heads | v heads
        if (n < 16 * D) col = (n / D) < G ? (kvh * G + n / D) * D + n % D : -1;
        else if (n < 17 * D) col = H * D + kvh * D + (n - 16 * D);
        else col = H * D + p.n_kv * D + kvh * D + (n - 17 * D);
        uint2 o = make_uint2(0u, 0u);
        if (col >= 0) {
            {
                float4 acc4[8];
                const uint2 bb = *reinterpret_cast<const uint2*>(p.bias + col);
#pragma unroll
                for (int sp = 0; sp < 8; ++sp)
                    if (sp < p.splitk)
                        acc4[sp] = *reinterpret_cast<const float4*>(ws_ + ((size_t)sp * p.rows_ws + b) * p.ldws + col);
                float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int sp = 0; sp < 8; ++sp)
                    if (sp < p.splitk) { a.x += acc4[sp].x; a.y += acc4[sp].y; a.z += acc4[sp].z; a.w += acc4[sp].w; }
                o.x = pack2bf(a.x + __uint_as_float(bb.x << 16), a.y + __uint_as_float(bb.x & 0xffff0000u));
                o.y = pack2bf(a.z + __uint_as_float(bb.y << 16), a.w + __uint_as_float(bb.y & 0xffff0000u));
            }
        }
        *reinterpret_cast<uint2*>(q_s + n) = o;
    }
    __syncthreads();
    if (p.rope_cos) {
        // rotary embedding of the G query rows and of k_new at position `pos` (rotate_half convention; the
        // products and the sum are rounded to bf16 one by one, as the reference's bf16 tensors are)
        const float* ct = p.rope_cos + (size_t)pos * (D / 2);
        const float* st_ = p.rope_sin + (size_t)pos * (D / 2);
        for (int i = tid; i < (G + 1) * (D / 2); i += AD_WAVES * 64) {
            const int r = i / (D / 2), d = i % (D / 2);
            bf16_t* row = r < G ? q_s + r * D : kv_new;
            const float x1 = bf2f(row[d]), x2 = bf2f(row[d + D / 2]);
            const float cs = ct[d], sn = st_[d];
            row[d] = f2bf(bfround(x1 * cs) + bfround(-x2 * sn));
            row[d + D / 2] = f2bf(bfround(x2 * cs) + bfround(x1 * sn));
        }
        __syncthreads();
    }

    // split 0 appends the new token's K / V to the cache (for FUTURE steps; in this launch every block
    // patches the new token into its fragments from LDS, so no block depends on another block's store)
    if (split == 0 && tid < 2 * D) {
        char* page = pool + (size_t)table[pos >> 6] * page_bytes;
        const int t64 = pos & 63;
        const int d = tid < D ? tid : tid - D;
        const size_t off = tid < D ? kv_k_offset(D, t64, d) : kv_v_offset(D, t64, d);
        *reinterpret_cast<bf16_t*>(page + off) = kv_new[tid];
    }

    const long long t_q = wall_clock64();              // q / k_new / v_new summed from the slabs and in LDS
    const int hd = lane & 15, c = lane >> 4;
    bf16x8 qf[NKS];
#pragma unroll
    for (int s = 0; s < NKS; ++s)
        qf[s] = *reinterpret_cast<const bf16x8*>(q_s + hd * D + 32 * s + 8 * c);

    f32x4 accO[NDV];
#pragma unroll
    for (int t = 0; t < NDV; ++t) accO[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;

    auto process = [&](KvFrags<D>& f, int g) {
        const int t0 = g << 5;
        if (pos >= t0 && pos < t0 + 32) {
            // this group holds the new token: take its K row / V column from LDS (wave-uniform branch)
            const int k32 = pos - t0;
            const int us = k32 >> 4, key_lo = k32 & 15, cs = (k32 >> 2) & 3, es = 4 * us + (k32 & 3);
            const bool klane = hd == key_lo;
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int s = 0; s < NKS; ++s) {
                    const u32x4 nk = *reinterpret_cast<const u32x4*>(kv_new + 32 * s + 8 * c);
                    if (klane && u == us) f.k[u][s] = nk;
                }
            const bool vlane = c == cs;
            const int ws = es >> 1, sh = (es & 1) * 16;
#pragma unroll
            for (int t = 0; t < NDV; ++t) {
                const uint32_t nv = (uint32_t)kv_new[D + 16 * t + hd] << sh;
                const uint32_t keep = ~(0xffffu << sh);
#pragma unroll
                for (int w = 0; w < 4; ++w)
                    if (vlane && w == ws) f.v[t][w] = (f.v[t][w] & keep) | nv;
            }
            // The keys of this group BEHIND the new token are stale rows of the page (whatever its last owner left there).  Their scores are
            // masked below by a select -- NaN-safe -- but their V rows still go through the P.V MFMA with P = 0, and 0 x NaN = NaN: a page that
            // ever held a non-finite row (a request with NaN inputs, a step voided by a give-up) would fail its NEXT owner, call after call,
            // until every stale row had been overwritten (round 5, tests/test_gpu_safety.py).  So the stale V columns are cleared here -- in
            // the one group per sequence and step that has any, a few dozen v_and per lane.  Element e of a lane's V fragment is key
            // 16 (e >> 2) + 4 c + (e & 3) of the group.
            uint32_t vm[4];
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const int klo = 16 * (w >> 1) + 4 * c + 2 * (w & 1);             // e = 2 w (low half of the dword), e + 1 (high half)
                vm[w] = (klo <= k32 ? 0x0000ffffu : 0u) | (klo + 1 <= k32 ? 0xffff0000u : 0u);
            }
#pragma unroll
            for (int t = 0; t < NDV; ++t)
#pragma unroll
                for (int w = 0; w < 4; ++w) f.v[t][w] &= vm[w];
        }
        f32x4 accS[2];
        float sc[2][4];
        float mt = -INFINITY;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            accS[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < NKS; ++s)
                accS[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag4(f.k[u][s]), qf[s], accS[u], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = t0 + 16 * u + 4 * c + r;
                sc[u][r] = (key <= pos && key >= win0) ? accS[u][r] * p.scale : -INFINITY;
                mt = fmaxf(mt, sc[u][r]);
            }
        }
        mt = fmaxf(mt, __shfl_xor(mt, 16, 64));
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
        const float m_new = fmaxf(m_run, mt);
        const float alpha = __expf(m_run - m_new);
        float pr[8];
        float ls = 0.f;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pr[4 * u + r] = __expf(sc[u][r] - m_new);
                ls += pr[4 * u + r];
            }
        l_run = l_run * alpha + ls;
        m_run = m_new;
        const bf16x8 pf = as_frag(pack8(pr));
#pragma unroll
        for (int t = 0; t < NDV; ++t) {
            accO[t] = accO[t] * alpha;
            accO[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_frag4(f.v[t]), pf, accO[t], 0, 0, 0);
        }
    };

    // two register buffers: the loads of group i+1 are in flight while group i is processed
    while (g < ngroups) {
        const int g1 = g + stride;
        if (g1 < ngroups) load_group<D>(fb, pool, page_of(g1), page_bytes, g1, lane);
        process(fa, g);
        if (g1 >= ngroups) break;
        const int g2 = g1 + stride;
        if (g2 < ngroups) load_group<D>(fa, pool, page_of(g2), page_bytes, g2, lane);
        process(fb, g1);
        g = g2;
    }

    const long long t_loop = wall_clock64();           // this wave's key groups processed
    // merge the waves of this block (LDS), in wave order.  Only the waves that had a key group take part (round 6): at the run's mean context a
    // block owns 4 groups, waves 4-7 carry (m, l, O) = (-inf, 0, 0) -- terms that add exactly zero, so the result is bit-identical -- and used to
    // cost half of the merge's LDS traffic (64 KiB written, every output thread reading 8 x 20 bytes instead of 4 x).  Two fully unrolled
    // instantiations (4 / 8 waves) picked by a block-uniform branch: a run-time loop bound measured SLOWER than the old form (LDS reads behind
    // branches: +1.2 us at context 1283).  p.merge_all (SV_EXP bit 2097152) = always 8.
    int nwa = (ngroups - g0 - split + act - 1) / act;
    nwa = p.merge_all ? AD_WAVES : (nwa < 1 ? 1 : (nwa > AD_WAVES ? AD_WAVES : nwa));
    float l_tot = l_run + __shfl_xor(l_run, 16, 64);
    l_tot += __shfl_xor(l_tot, 32, 64);
    // each thread owns 4 consecutive outputs (same head): 16-byte stores for the partial / 8-byte for the result
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        p.part, 0, (unsigned)((size_t)p.B * p.n_kv * AD_SPLIT * PART * sizeof(float)), 0x00020000);
    const int my_off = (int)(((size_t)bx * AD_SPLIT + split) * PART * 4);            // bytes
    const int col0 = kvh * G * D;            // this block's first output column ((kvh*G + h)*D + dv = col0 + idx)
    auto block_merge = [&](auto nw_tag) {
        constexpr int NWV = decltype(nw_tag)::value;
        if (wave < NWV) {
            if (c == 0) {
                m_s[wave * 16 + hd] = m_run;
                l_s[wave * 16 + hd] = l_tot;
            }
#pragma unroll
            for (int t = 0; t < NDV; ++t)
                *reinterpret_cast<float4*>(O_s + ((size_t)(wave * 16 + hd)) * D + 16 * t + 4 * c) =
                    make_float4(accO[t][0], accO[t][1], accO[t][2], accO[t][3]);
        }
        __syncthreads();
        for (int idx = tid * 4; idx < 16 * D; idx += AD_WAVES * 64 * 4) {
            const int h = idx / D, dv = idx % D;
            float M = -INFINITY;
#pragma unroll
            for (int w = 0; w < NWV; ++w) M = fmaxf(M, m_s[w * 16 + h]);
            float num[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < NWV; ++w) {
                const float mw = m_s[w * 16 + h];
                const float f = (mw == -INFINITY) ? 0.f : __expf(mw - M);
                const float4 o = *reinterpret_cast<const float4*>(O_s + ((size_t)(w * 16 + h)) * D + dv);
                // (explicit fused multiply-adds: what the compiler's contraction made of `num += f * o` in the one-instantiation form of rounds 1-5 --
                //  left to its mood, the two-instantiation form came out with other roundings and four more near-tie flips in the free-running test)
                num[0] = fmaf(f, o.x, num[0]); num[1] = fmaf(f, o.y, num[1]); num[2] = fmaf(f, o.z, num[2]); num[3] = fmaf(f, o.w, num[3]);
            }
            if (act == 1) {
                if (idx < HD) {
                    float den = 0.f;
#pragma unroll
                    for (int w = 0; w < NWV; ++w) {
                        const float mw = m_s[w * 16 + h];
                        den = fmaf((mw == -INFINITY) ? 0.f : __expf(mw - M), l_s[w * 16 + h], den);
                    }
                    const float inv = 1.0f / den;
                    uint2 o;
                    o.x = pack2bf(num[0] * inv, num[1] * inv);
                    o.y = pack2bf(num[2] * inv, num[3] * inv);
                    *reinterpret_cast<uint2*>(p.out_xp + xp_index(b >> 5, p.out_KS, b & 31, col0 + idx)) = o;
                }
            } else {
                u32x4 v;
                v[0] = __float_as_uint(num[0]); v[1] = __float_as_uint(num[1]);
                v[2] = __float_as_uint(num[2]); v[3] = __float_as_uint(num[3]);
                __builtin_amdgcn_raw_buffer_store_b128(v, rs, my_off + (32 + idx) * 4, 0, 16);   // write-through
            }
        }
        if (act > 1 && tid < 8) {
            // m[16] | l[16] of this block's partial: 8 x 16 bytes
            const bool is_l = tid >= 4;
            u32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int h = (tid & 3) * 4 + j;
                float M = -INFINITY;
#pragma unroll
                for (int w = 0; w < NWV; ++w) M = fmaxf(M, m_s[w * 16 + h]);
                float den = 0.f;
#pragma unroll
                for (int w = 0; w < NWV; ++w) {
                    const float mw = m_s[w * 16 + h];
                    den = fmaf((mw == -INFINITY) ? 0.f : __expf(mw - M), l_s[w * 16 + h], den);
                }
                v[j] = __float_as_uint(is_l ? den : M);
            }
            __builtin_amdgcn_raw_buffer_store_b128(v, rs, my_off + tid * 16, 0, 16);
        }
    };
    if (nwa <= 4) block_merge(std::integral_constant<int, 4>{});                 // contexts <= 1024 at 4 groups per block
    else if (nwa <= 6) block_merge(std::integral_constant<int, 6>{});            // <= 1536
    else block_merge(std::integral_constant<int, AD_WAVES>{});
    auto stamp = [&](long long t_part, long long t_tick, long long t_end) {
        if (p.trace && tid == 0) {
            long long* q = p.trace + ((size_t)bx * gridDim.y + split) * 16;
            q[0] = t_start; q[1] = t_kvreq; q[2] = t_q; q[3] = t_loop; q[4] = t_part; q[5] = t_tick; q[6] = t_end; q[7] = act; q[8] = ngroups;
        }
    };
    if (act == 1) { stamp(0, 0, wall_clock64()); return; }

    // hand-off without fences: write-through (sc1) partial, every storing wave drains, one relaxed
    // agent-scope ticket; the last arriver reads the partials with sc1 loads (L1 bypass)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const long long t_part = wall_clock64();           // partial stored and drained
    if (tid == 0) {
        const unsigned t = __hip_atomic_fetch_add(p.counters + bx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        flag_s[0] = (t == (unsigned)(act - 1)) ? 1 : 0;
    }
    __syncthreads();
    const long long t_tick = wall_clock64();           // ticket drawn
    if (!flag_s[0]) { stamp(t_part, t_tick, 0); return; }
    const int seq_off = (int)((size_t)bx * AD_SPLIT * PART * 4);
    // merge: every load (statistics of all splits for this thread's head + its O columns) is issued before
    // the first use -> one memory round trip
    for (int idx = tid * 4; idx < HD; idx += AD_WAVES * 64 * 4) {
        const int h = idx / D;
        u32x4 ov[AD_SPLIT];
        float ms[AD_SPLIT], ls[AD_SPLIT];
#pragma unroll
        for (int s2 = 0; s2 < AD_SPLIT; ++s2) {
            ov[s2] = u32x4{0u, 0u, 0u, 0u};
            ms[s2] = -INFINITY; ls[s2] = 0.f;
            if (s2 < act) {
                ms[s2] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, seq_off + (s2 * PART + h) * 4, 0, 16));
                ls[s2] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, seq_off + (s2 * PART + 16 + h) * 4, 0, 16));
                ov[s2] = __builtin_amdgcn_raw_buffer_load_b128(rs, seq_off + (s2 * PART + 32 + idx) * 4, 0, 16);
            }
        }
        float M = -INFINITY;
#pragma unroll
        for (int s2 = 0; s2 < AD_SPLIT; ++s2) M = fmaxf(M, ms[s2]);
        float num[4] = {0.f, 0.f, 0.f, 0.f}, den = 0.f;
#pragma unroll
        for (int s2 = 0; s2 < AD_SPLIT; ++s2) {
            const float f = (s2 < act && ms[s2] != -INFINITY) ? __expf(ms[s2] - M) : 0.f;
            num[0] += f * __uint_as_float(ov[s2][0]); num[1] += f * __uint_as_float(ov[s2][1]);
            num[2] += f * __uint_as_float(ov[s2][2]); num[3] += f * __uint_as_float(ov[s2][3]);
            den += f * ls[s2];
        }
        const float inv = 1.0f / den;
        uint2 o;
        o.x = pack2bf(num[0] * inv, num[1] * inv);
        o.y = pack2bf(num[2] * inv, num[3] * inv);
        *reinterpret_cast<uint2*>(p.out_xp + xp_index(b >> 5, p.out_KS, b & 31, col0 + idx)) = o;
    }
    if (tid == 0) __hip_atomic_store(p.counters + bx, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm
    stamp(t_part, t_tick, wall_clock64());
}

size_t attn_decode_part_floats(int head_dim) { return (size_t)AD_SPLIT * (32 + 16 * (size_t)head_dim); }

static size_t attn_decode_smem(int D) {
    return (size_t)(16 * D + 2 * D) * 2 + (size_t)AD_WAVES * 16 * 4 * 2 + (size_t)AD_WAVES * 16 * D * 4 + 16;
}

// dynamic LDS above the 64 KiB default needs an explicit opt-in.  Done once at engine creation (never
// inside a stream capture)
int init_attention_kernels() {
    hipError_t r = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_decode_kernel<128>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)attn_decode_smem(128));
    if (r != hipSuccess) return (int)r;
    r = hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_decode_kernel<64>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)attn_decode_smem(64));
    return (int)r;
}

void launch_attn_decode(const AttnDecodeArgs& a, hipStream_t st) {
    const size_t smem = attn_decode_smem(a.head_dim);
    // grid.y = the engine's split cap, not AD_SPLIT: the splits above it never work, and dispatching 8-wave blocks that exit
    // at once is not free (B = 32: 256 of 512 blocks)
    dim3 grid(a.B * a.n_kv, a.max_splits < 1 ? 1 : (a.max_splits > AD_SPLIT ? AD_SPLIT : a.max_splits));
    const int gpb = a.groups_per_block > 0 ? a.groups_per_block : AD_GROUPS_PER_BLOCK;
    if (a.head_dim == 128)
        attn_decode_kernel<128><<<grid, AD_WAVES * 64, smem, st>>>(a.positions, a.block_table, a.pool_layer, a.ws, a.kv_head_stride,
                                                                    a.max_pages, a.n_kv, a.max_splits, a.window, gpb, a);
    else
        attn_decode_kernel<64><<<grid, AD_WAVES * 64, smem, st>>>(a.positions, a.block_table, a.pool_layer, a.ws, a.kv_head_stride,
                                                                   a.max_pages, a.n_kv, a.max_splits, a.window, gpb, a);
}

// which XCD each block of a 1-D launch runs on (XCC_ID), with the footprint of the decode attention when asked: `lds_bytes` of dynamic
// LDS (one block per CU) and `spin_us` of residence, so that a grid above the CU count is dispatched in rounds like the real launch
__global__ __launch_bounds__(AD_WAVES * 64) void xcc_probe_kernel(int32_t* out, int spin_ticks) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (spin_ticks > 0) {
        smem[threadIdx.x] = 1;
        const long long t0 = wall_clock64();
        while (wall_clock64() - t0 < spin_ticks) __builtin_amdgcn_s_sleep(8);
    }
    if (threadIdx.x == 0) out[blockIdx.x] = (int32_t)xcc_id();
}
// occupy_kernel: a foreign tenant for the safety test of the launches whose blocks wait for each other (tests/test_gpu_safety.py): `blocks`
// 4-wave blocks that each pin `lds_bytes` of a CU's LDS for `ticks` x 10 ns and do nothing else -- with 144 KiB no block of the fused MLP /
// fused row-update launch fits beside one, i.e. those CUs are taken the way another process's kernels would take them.
__global__ __launch_bounds__(256) void occupy_kernel(int ticks) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    smem[threadIdx.x] = 1;
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
int launch_occupy(int blocks, int lds_bytes, int ticks, hipStream_t st) {
    const hipError_t r = hipFuncSetAttribute(reinterpret_cast<const void*>(&occupy_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (r != hipSuccess) return (int)r;
    occupy_kernel<<<blocks, 256, (size_t)lds_bytes, st>>>(ticks);
    return (int)hipGetLastError();
}
int launch_xcc_probe(int32_t* out, int blocks, int heavy, hipStream_t st) {
    if (heavy) {
        const hipError_t r = hipFuncSetAttribute(reinterpret_cast<const void*>(&xcc_probe_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 (int)attn_decode_smem(128));
        if (r != hipSuccess) return (int)r;
    }
    xcc_probe_kernel<<<blocks, AD_WAVES * 64, heavy ? attn_decode_smem(128) : 64, st>>>(out, heavy ? 1000 : 0);     // 10 us at 100 MHz
    return 0;
}

}  // namespace sv
