// Stand-alone check of float -> bf16 conversion variants on gfx950 against the host's round-to-nearest-even.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
__device__ __forceinline__ uint32_t pk_builtin(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ uint32_t pk_asm(float lo, float hi) {
    uint32_t r;
    asm volatile("s_nop 4\n\tv_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
// mode 0: plain values; mode 1: (x - m) * r * g + b like the LayerNorm prologue; mode 2: exp() like the softmax
template <int MODE, bool ASM>
__global__ void k(const float* x, const float* g, const float* b, float m, float r, uint32_t* y, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    float a0 = x[2 * i], a1 = x[2 * i + 1];
    if (MODE == 1) { a0 = (a0 - m) * r * g[2 * i] + b[2 * i]; a1 = (a1 - m) * r * g[2 * i + 1] + b[2 * i + 1]; }
    if (MODE == 2) { a0 = __expf(a0); a1 = __expf(a1); }
    y[i] = ASM ? pk_asm(a0, a1) : pk_builtin(a0, a1);
}
static uint16_t rne(float f) {
    uint32_t u; memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
int main() {
    const int n = 1 << 20;
    float *hx = (float*)malloc(n * 4), *hg = (float*)malloc(n * 4), *hb = (float*)malloc(n * 4);
    srand(1);
    for (int i = 0; i < n; ++i) {
        hx[i] = ((rand() / (float)RAND_MAX) - 0.5f) * 6.f; hg[i] = 1.f + 0.1f * ((rand() / (float)RAND_MAX) - 0.5f);
        hb[i] = 0.1f * ((rand() / (float)RAND_MAX) - 0.5f);
    }
    float *dx, *dg, *db; uint32_t* dy;
    hipMalloc(&dx, n * 4); hipMalloc(&dg, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dy, n * 2);
    hipMemcpy(dx, hx, n * 4, hipMemcpyHostToDevice); hipMemcpy(dg, hg, n * 4, hipMemcpyHostToDevice); hipMemcpy(db, hb, n * 4, hipMemcpyHostToDevice);
    uint32_t* hy = (uint32_t*)malloc(n * 2);
    const float m = 0.3f, r = 0.66f;
    for (int mode = 0; mode < 3; ++mode)
        for (int as = 0; as < 2; ++as) {
            dim3 grid(n / 2 / 256), blk(256);
            if (mode == 0 && !as) k<0, false><<<grid, blk>>>(dx, dg, db, m, r, dy, n);
            if (mode == 0 && as) k<0, true><<<grid, blk>>>(dx, dg, db, m, r, dy, n);
            if (mode == 1 && !as) k<1, false><<<grid, blk>>>(dx, dg, db, m, r, dy, n);
            if (mode == 1 && as) k<1, true><<<grid, blk>>>(dx, dg, db, m, r, dy, n);
            if (mode == 2 && !as) k<2, false><<<grid, blk>>>(dx, dg, db, m, r, dy, n);
            if (mode == 2 && as) k<2, true><<<grid, blk>>>(dx, dg, db, m, r, dy, n);
            hipMemcpy(hy, dy, n * 2, hipMemcpyDeviceToHost);
            long bad = 0, off1 = 0;
            for (int i = 0; i < n; ++i) {
                float a = hx[i];
                if (mode == 1) a = fmaf((a - m) * r, hg[i], hb[i]);
                if (mode == 2) continue;                      // exp differs from the host's: only modes 0 / 1 are compared
                const uint16_t got = (uint16_t)(hy[i / 2] >> ((i & 1) * 16)), ref = rne(a);
                if (got != ref) { ++bad; if (abs((int)got - (int)ref) <= 1) ++off1; }
            }
            printf("mode %d %s: %ld of %d differ from host RNE (%ld by one ulp)\n", mode, as ? "asm+nop" : "builtin", bad, n, off1);
        }
    return 0;
}
