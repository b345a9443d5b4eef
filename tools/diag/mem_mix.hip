// What bounds the decode GEMMs' memory side?  A block of the skinny kernels streams its own weights from HBM (every byte once)
// and re-reads the SAME small activation matrix out of L2 as every other block (R bytes of activations per weight byte: 1 for
// bf16 weights and <= 32 rows, 2 at 64 rows, 4 for fp8 weights at 64 rows).  This probe reproduces only that traffic -- 16-byte
// loads per lane, 1 KiB per wave instruction, 8 waves per block, no MFMA -- and sweeps R, the blocks per CU and the way the shared
// operand reaches the waves (every wave from L2, or once per block into LDS and ds_read from there).
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/mem_mix tools/diag/mem_mix.hip && /tmp/mem_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// W: [blocks][waves][steps][64 lanes] u32x4 (contiguous per wave, like a packed weight tile's K range)
// A: [a_steps][64 lanes] u32x4, the same for all blocks; wave w of a block starts at step w * steps * R (mod a_steps)
template <int R, int LDSMODE>
__global__ __launch_bounds__(512) void probe(const u32x4* __restrict__ W, const u32x4* __restrict__ A, int steps, int a_steps,
                                             unsigned* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32x4* w = W + ((size_t)(blockIdx.x * 8 + wave) * steps) * 64 + lane;
    u32x4 acc = {0u, 0u, 0u, 0u};
    if constexpr (LDSMODE == 0) {
        int ai = (wave * steps * R) % a_steps;
        for (int s = 0; s < steps; s += 4) {
            u32x4 wv[4], av[4 * (R > 0 ? R : 1)];
#pragma unroll
            for (int u = 0; u < 4; ++u) wv[u] = __builtin_nontemporal_load(w + (size_t)(s + u) * 64);
#pragma unroll
            for (int u = 0; u < 4 * R; ++u) {
                av[u] = A[(size_t)ai * 64 + lane];
                ai = ai + 1 == a_steps ? 0 : ai + 1;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) acc ^= wv[u];
#pragma unroll
            for (int u = 0; u < 4 * R; ++u) acc ^= av[u];
        }
    } else {
        // the shared operand goes global -> LDS once per block (the 8 waves split each 8-step slice), every wave then reads all
        // of what it needs from LDS: R ds_read_b128 per weight load.  Double-buffered slices of 8 KiB, one barrier per slice.
        u32x4* lds = reinterpret_cast<u32x4*>(smem);                    // [2][8 steps][64]
        const int slices = steps * (R > 0 ? R : 1) / 8;                 // LDS slices a wave consumes over its range
        int ai = 0;
        auto fill = [&](int buf, int sl) {
            const int st = (sl * 8 + wave) % a_steps;
            lds[(buf * 8 + wave) * 64 + lane] = A[(size_t)st * 64 + lane];
        };
        fill(0, 0);
        __syncthreads();
        int s = 0;
        for (int sl = 0; sl < slices; ++sl) {
            if (sl + 1 < slices) fill((sl + 1) & 1, sl + 1);
            // per slice: 8 LDS reads and 8 / R weight loads
            u32x4 wv[8 / (R > 0 ? R : 1)];
#pragma unroll
            for (int u = 0; u < 8 / R; ++u) wv[u] = __builtin_nontemporal_load(w + (size_t)(s + u) * 64);
            s += 8 / R;
#pragma unroll
            for (int u = 0; u < 8; ++u) acc ^= lds[((sl & 1) * 8 + u) * 64 + lane];
#pragma unroll
            for (int u = 0; u < 8 / R; ++u) acc ^= wv[u];
            __syncthreads();
        }
        (void)ai;
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) out[0] = 1;
}

template <int R, int LDSMODE>
static void run(const u32x4* W, const u32x4* A, unsigned* out, int blocks, int steps, int a_steps, int per_cu, hipStream_t st) {
    // blocks per CU: LDS is 160 KiB -> 1 block with 96 KiB, 2 with 64, 4 with 36
    const int smem = per_cu == 1 ? 96 * 1024 : per_cu == 2 ? 64 * 1024 : 36 * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<R, LDSMODE>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<R, LDSMODE><<<blocks, 512, smem, st>>>(W, A, steps, a_steps, out);
    hipEventRecord(e0, st);
    const int reps = 5;
    for (int i = 0; i < reps; ++i) probe<R, LDSMODE><<<blocks, 512, smem, st>>>(W, A, steps, a_steps, out);
    hipEventRecord(e1, st);
    hipStreamSynchronize(st);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps;
    const double wb = (double)blocks * 8 * steps * 1024.0, ab = wb * R;
    printf("R=%d %-9s blocks/CU %d  blocks %5d  %8.1f us   weights %6.2f TB/s   shared operand %6.2f TB/s   total into CUs %6.2f TB/s\n", R,
           LDSMODE ? "via LDS" : "from L2", per_cu, blocks, us, wb / us * 1e-6, ab / us * 1e-6, (wb + ab) / us * 1e-6);
}

int main() {
    const int steps = 128;                    // 128 KiB of weights per wave, 1 MiB per block
    const int blocks = 2304;                  // 2.4 GB per launch: past the 256 MiB Infinity Cache
    const int a_steps = 576;                  // 576 KiB shared operand (64 rows x 4608 bf16)
    u32x4 *W, *A;
    unsigned* out;
    hipMalloc(&W, (size_t)blocks * 8 * steps * 1024);
    hipMalloc(&A, (size_t)a_steps * 1024);
    hipMalloc(&out, 64);
    hipMemset(W, 1, (size_t)blocks * 8 * steps * 1024);
    hipMemset(A, 2, (size_t)a_steps * 1024);
    hipStream_t st;
    hipStreamCreate(&st);
    for (int per_cu = 1; per_cu <= 4; per_cu *= 2) {
        run<0, 0>(W, A, out, blocks, steps, a_steps, per_cu, st);
        run<1, 0>(W, A, out, blocks, steps, a_steps, per_cu, st);
        run<2, 0>(W, A, out, blocks, steps, a_steps, per_cu, st);
        run<4, 0>(W, A, out, blocks, steps, a_steps, per_cu, st);
        run<1, 1>(W, A, out, blocks, steps, a_steps, per_cu, st);
        run<2, 1>(W, A, out, blocks, steps, a_steps, per_cu, st);
        run<4, 1>(W, A, out, blocks, steps, a_steps, per_cu, st);
    }
    return 0;
}
