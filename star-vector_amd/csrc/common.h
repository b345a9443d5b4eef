// Shared device helpers for the StarVector gfx950 kernels (CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned short bf16_t;  // raw bfloat16 bits
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;   // one MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(16))) float f32x16;   // 32x32 MFMA accumulator
typedef __attribute__((ext_vector_type(4))) float f32x4;     // 16x16 MFMA accumulator
typedef __attribute__((ext_vector_type(2))) float f32x2;     // fp8 pair conversions

#define SV_WAVE 64

__device__ __forceinline__ float bf2f(bf16_t x) { return __uint_as_float(((uint32_t)x) << 16); }

// float -> bfloat16, round-to-nearest-even like torch's cast: gfx950's v_cvt_pk_bf16_f32 (two values per instruction; checked
// bit for bit against torch -- halfway cases, denormals -- by tests/test_gpu_ops.py::test_bf16_rounding_is_rne).  The software
// rounding this replaces (NaN test, add 0x7fff + lsb, shift: ~7 VALU operations per value) was a third of the big-M GEMMs'
// epilogue (profiles/gemm_r02_fixed_cost_sweep.log).  Written as a vector conversion the COMPILER selects the instruction
// for -- never as inline asm: gfx950 needs a wait state between a transcendental (v_exp / v_rcp) and a VALU instruction
// reading its result, the compiler inserts it for its own instructions and does not look inside an asm statement (the asm
// version read stale exp() results in the attention softmax: garbage probabilities, NaN logits).
typedef __attribute__((ext_vector_type(2))) __bf16 sv_bf16x2;
typedef __attribute__((ext_vector_type(2))) float sv_f32x2;
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
    const sv_f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, sv_bf16x2));
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack2bf(f, f) & 0xffffu); }
__device__ __forceinline__ float bfround(float f) { return __uint_as_float(pack2bf(f, f) << 16); }

// 8 bf16 <-> 8 floats
__device__ __forceinline__ void unpack8(const uint4& u, float* f) {
    f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xffff0000u);
    f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xffff0000u);
    f[4] = __uint_as_float(u.z << 16); f[5] = __uint_as_float(u.z & 0xffff0000u);
    f[6] = __uint_as_float(u.w << 16); f[7] = __uint_as_float(u.w & 0xffff0000u);
}
__device__ __forceinline__ uint4 pack8(const float* f) {
    uint4 u;
    u.x = pack2bf(f[0], f[1]); u.y = pack2bf(f[2], f[3]);
    u.z = pack2bf(f[4], f[5]); u.w = pack2bf(f[6], f[7]);
    return u;
}
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;   // native 16-byte vector (nontemporal loads)
__device__ __forceinline__ bf16x8 as_frag(const uint4& u) { return __builtin_bit_cast(bf16x8, u); }
__device__ __forceinline__ bf16x8 as_frag4(const u32x4& u) { return __builtin_bit_cast(bf16x8, u); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ------------------------------------------------------------------------------------------------
// XCD-aligned weight prefetch.  The decode step alternates weight-streaming GEMMs with latency-bound kernels (paged attention,
// row update) during which HBM is nearly idle; the small GEMMs that follow (c_attn, attention c_proj: ~1.1 MiB per XCD) are a
// single HBM round trip per wave, i.e. mostly first-touch latency.  Waves with nothing to do in the latency-bound kernel read
// the head of every consumer wave's weight stream with ordinary loads, so the lines sit in the L2 of the XCD THE CONSUMER BLOCK
// WILL RUN ON: workgroups are dealt round-robin to the 8 XCDs in launch order, by the SAME rule in every launch of a replayed
// graph (tools/diag/xcd_map.hip on MI355X: linear block id b runs on HW_REG_XCC_ID (b + 7) % 8, every kernel, every replay) -- a
// speed assumption only, a wrong guess costs the hint, never a result.  So a prefetching block with linear id p and consumer
// block (nt, split) of a skinny GEMM with NT % 8 == 0 share an XCD iff p % 8 == nt % 8: the prefetcher takes the tiles
// congruent to ITS OWN block id (the hardware register would be off by that rotation: round 3's first attempt).
// Round 2's linear slicing parked the lines in other XCDs' L2s / the Infinity Cache and bought nothing (profiles/prefetch_r02_ab.log).
// A piece = the first `piece_units` KiB of one consumer wave's stream; a unit = 1 KiB = one wave-wide 16-byte load.
// ------------------------------------------------------------------------------------------------
struct PrefetchDesc {
    const char* base;          // packed weight image (nullptr = off)
    unsigned tile_bytes;       // bytes of one 32-column tile (K/16 KiB for bf16)
    int n_tiles;               // column tiles of the consumer (its grid.x)
    int pieces;                // consumer wave streams per tile (splitk * waves per block)
    unsigned piece_stride;     // bytes between the starts of consecutive streams
    int piece_units;           // KiB prefetched at the head of each stream
};
// Issue up to NU loads: units slot, slot + nslots, ... of this XCD's share (units past the end re-read the first line).
// The caller keeps `dst` alive until sv_prefetch_sink (the loads are ordinary loads: the compiler tracks them).
template <int NU>
__device__ __forceinline__ void sv_prefetch_issue(const PrefetchDesc& d, int xcd, int slot, int nslots, int lane, u32x4* dst) {
    const int n_own = (d.n_tiles - xcd + 7) >> 3;                  // tiles xcd, xcd + 8, ...
    const int total = n_own * d.pieces * d.piece_units;
#pragma unroll
    for (int k = 0; k < NU; ++k) {
        const int u = slot + k * nslots;
        size_t off = 0;
        if (u < total) {                                            // wave-uniform
            const int q = u / d.piece_units, kb = u - q * d.piece_units;
            const int ti = q / d.pieces, pc = q - ti * d.pieces;
            off = (size_t)(xcd + 8 * ti) * d.tile_bytes + (size_t)pc * d.piece_stride + (size_t)kb * 1024;
        }
        dst[k] = *reinterpret_cast<const u32x4*>(d.base + off + (size_t)lane * 16);
    }
}
// Wait for the prefetch loads where the wave has time to (data-dependent, practically never taken store keeps them alive).
template <int NU>
__device__ __forceinline__ void sv_prefetch_sink(const u32x4* dst, unsigned* scratch_word) {
    unsigned x = 0;
#pragma unroll
    for (int k = 0; k < NU; ++k) x ^= dst[k][0] ^ dst[k][1] ^ dst[k][2] ^ dst[k][3];
    if (x == 0x9e3779b9u && scratch_word == reinterpret_cast<unsigned*>(1)) *scratch_word = x;      // never true
}

// activations used by the path's fused epilogues
enum { ACT_NONE = 0, ACT_QUICKGELU = 1, ACT_SWISH = 2, ACT_GELU_TANH = 3 };

__device__ __forceinline__ float sv_act(float x, int act) {
    switch (act) {
        // x * rcp(1 + e): v_rcp_f32 (1 ulp) instead of the ~10-instruction IEEE division; the result is rounded to bf16 by every
        // caller, and every kernel of the path uses this one function, so rows stay bit-identical across kernels
        case ACT_QUICKGELU: return x * __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * x));         // clip_model.py:126-128
        case ACT_SWISH:     return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x));                   // adapter.py:5-10
        case ACT_GELU_TANH: {                                                 // gelu_pytorch_tanh
            // 0.5 x (1 + tanh(u)) = x / (1 + exp(-2u)),  u = sqrt(2/pi) (x + 0.044715 x^3)
            const float k0 = 0.7978845608028654f, k1 = 0.044715f;
            const float u = k0 * (x + k1 * x * x * x);
            return x * __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * u));
        }
        default: return x;
    }
}

// Packed ("fragment order") layouts --------------------------------------------------------------
// Weight  W[N][K]  ->  Wp[N/32][K/16][64 lanes][8]:  lane l holds W[nt*32 + (l&31)][ks*16 + (l>>5)*8 + e]
//   = the A operand of v_mfma_f32_32x32x16_bf16; one 1 KiB wave-load per (n-tile, k-step).
// Skinny activations x[32][K] -> xp[K/16][64][8]:    lane l holds x[l&31][ks*16 + (l>>5)*8 + e]
//   = the B operand of the same instruction.
__device__ __forceinline__ size_t xp_index(int mt, int KS, int m, int k) {
    // element offset of x[mt*32 + m][k] in a packed activation buffer with KS = K/16 k-steps
    return ((((size_t)mt * KS + (k >> 4)) * 64) + (((k >> 3) & 1) * 32) + m) * 8 + (k & 7);
}
