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
// Late argument block.  A decode kernel's leading scalar / pointer parameters are preloaded into SGPRs with the dispatch
// (-amdgpu-kernarg-preload-count): enough to issue its first global loads at once.  The rest of its arguments travel as a
// by-value struct placed right after them in the kernarg segment; read through an ordinary `p.field` the compiler hoists the
// scalar loads to the top of the kernel and waits for them BEFORE the first global load (a cold K$ miss at every launch, 172
// launches per decode step).  sv_late_args reads the struct through the kernarg pointer laundered by an empty asm, so the
// scalar loads are issued where this is called -- after the kernel's first loads are in flight.
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T sv_late_args(unsigned byte_offset) {
    typedef const char __attribute__((address_space(4))) * kaptr_t;
    typedef const uint32_t __attribute__((address_space(4))) * kaw_t;
    kaptr_t ka = (kaptr_t)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ka) : : "memory");
    kaw_t w = (kaw_t)(ka + byte_offset);
    constexpr int N = (int)((sizeof(T) + 3) / 4);
    union U { T t; uint32_t w[N]; __device__ U() {} } u;          // T: a plain argument struct
#pragma unroll
    for (int i = 0; i < N; ++i) u.w[i] = w[i];                    // constant address space, uniform address: scalar loads
    return u.t;
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
