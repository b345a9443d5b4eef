// Standalone A/B lab for the main loop of the 256 x 256 x 64 big-M GEMM (star-vector_amd/csrc/gemm.hip: gemm256_kernel).
// Not part of the library: every variant below computes the SAME bits (same MFMA, same operand roles, ascending k), so a variant is
// checked bit for bit against variant 0 (= the shipped loop) and variant 0 against a plain fp32 reference; what differs is the schedule:
//   SCHED 0  the shipped loop: 4 phases per K-tile, one half-tile staged per phase ONE K-tile ahead, vmcnt(2) after every phase
//   SCHED 1  4 phases per K-tile, every half-tile staged into its slot two phases after the slot's last ds_read: 6 phases between
//            LDS-DMA issue and first read, vmcnt(8) (four half-tiles in flight) after every phase
//   SCHED 2  2 phases per K-tile (16 MFMAs between a barrier pair), half-tiles restaged ONE phase after the slot's last read (the reads
//            are waited for before the phase's first barrier), vmcnt(8) before the first barrier
//   EARLY e  the barrier that ends an MFMA section is issued before the section's last e MFMAs (its release latency overlaps them)
//   ABL   1  no LDS-DMA in the loop (LDS zero-filled), 2: neither LDS-DMA nor ds_read (barriers + MFMAs only)
// Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -o gemm_lab tools/diag/gemm_lab.hip        Run: ./gemm_lab [rounds]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include <string>
#include <type_traits>
#include "../../star-vector_amd/csrc/common.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
__device__ __forceinline__ void lds_dma16(const void* g, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)lds_wave_base, 16, 0, 0);
}

__global__ void pack_w_kernel(const bf16_t* __restrict__ W, bf16_t* __restrict__ Wp, int N, int K) {
    const int KS = K >> 4;
    const size_t total = (size_t)(N >> 5) * KS * 64;
    for (size_t c = (size_t)blockIdx.x * blockDim.x + threadIdx.x; c < total; c += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(c & 63);
        const size_t t = c >> 6;
        const int ks = (int)(t % KS), nt = (int)(t / KS);
        const int n = nt * 32 + (lane & 31), k0 = ks * 16 + (lane >> 5) * 8;
        *reinterpret_cast<uint4*>(Wp + c * 8) = *reinterpret_cast<const uint4*>(W + (size_t)n * K + k0);
    }
}
__global__ void fill_kernel(bf16_t* p, size_t n, unsigned seed, int zero) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u ^ seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        const float v = zero ? 0.f : ((float)(h >> 8) * (1.0f / 8388608.0f) - 1.0f);          // uniform [-1, 1)
        p[i] = f2bf(v);
    }
}
// fp32 reference at sampled positions: idx -> (m, n) by a hash; err[0] = max |C - ref|, err[1] = max |ref|
__global__ void ref_check_kernel(const bf16_t* A, const bf16_t* W, const bf16_t* C, int M, int N, int K, int samples, float* err) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= samples) return;
    unsigned h = (unsigned)s * 2654435761u + 12345u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    const int m = (int)(h % (unsigned)M);
    h *= 3266489917u; h ^= h >> 16;
    const int n = (int)(h % (unsigned)N);
    float acc = 0.f;
    for (int k = 0; k < K; ++k) acc += bf2f(A[(size_t)m * K + k]) * bf2f(W[(size_t)n * K + k]);
    const float d = fabsf(bf2f(C[(size_t)m * N + n]) - acc);
    atomicMax((int*)&err[0], __float_as_int(d));
    atomicMax((int*)&err[1], __float_as_int(fabsf(acc)));
}
__global__ void diff_kernel(const uint4* a, const uint4* b, size_t n16, unsigned* ndiff) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 x = a[i], y = b[i];
        if (x.x != y.x || x.y != y.y || x.z != y.z || x.w != y.w) atomicAdd(ndiff, 1u);
    }
}

#define G2_T 256
#define G2_HALF 16384
#define G2_BUF 65536
#define BAR() asm volatile("s_barrier" ::: "memory")
#define SB() __builtin_amdgcn_sched_barrier(0)

template <int SCHED, int EARLY, int ABL>
__global__ __launch_bounds__(512) void lab_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ Wp, bf16_t* __restrict__ C, int M, int N, int K,
                                                  int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int T = tiles_m * tiles_n;
    const int id = blockIdx.x;
    const int q = T >> 3, rem = T & 7, xcd = id & 7, loc = id >> 3;
    const int wg = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + loc;
    const int band = wg / (4 * tiles_n), r_in = wg - band * 4 * tiles_n;
    const int band_rows = min(4, tiles_m - band * 4);
    const int tm = band * 4 + r_in % band_rows, tn = r_in / band_rows;
    const int m0 = tm * G2_T, n0 = tn * G2_T;
    const int KS = K >> 4;
    const int KT = K >> 6;
    const int NT_total = (N + 31) >> 5;

    f32x16 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const bf16_t* xsrc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) {
            const int g = wave * 2 + pc;
            const int r = g * 8 + (lane >> 3);
            int grow = m0 + (r >> 6) * 128 + i * 64 + (r & 63);
            grow = grow < M ? grow : M - 1;
            const int c = (lane & 7) ^ ((r >> 1) & 7);
            xsrc[i][pc] = A + (size_t)grow * K + c * 8;
        }
    const bf16_t* wsrc[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) {
            const int f = wave * 2 + pc;
            int nt = (n0 >> 5) + (f >> 2) * 2 + j;
            nt = nt < NT_total ? nt : NT_total - 1;
            wsrc[j][pc] = Wp + (((size_t)nt * KS + (f & 3)) * 64 + lane) * 8;
        }
    // h: 0 X0, 1 W0, 2 W1, 3 X1 of K-tile kt into the 64 KiB buffer `buf`
    auto stage = [&](int h, int kt, char* buf) {
        if constexpr (ABL == 1 || ABL == 2) return;
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
    int xoff[2][4];
#pragma unroll
    for (int mt2 = 0; mt2 < 2; ++mt2)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int r = wr * 64 + mt2 * 32 + (lane & 31);
            xoff[mt2][ks] = r * 128 + (((2 * ks + (lane >> 5)) ^ ((r >> 1) & 7)) << 4);
        }
    const int woff = wc * 4096 + lane * 16;

    bf16x8 x0[2][4], x1[2][4], w0[4], w1[4];
    auto read_x = [&](bf16x8 (&x)[2][4], const char* half) {
        if constexpr (ABL == 2 || ABL == 4) return;
#pragma unroll
        for (int mt2 = 0; mt2 < 2; ++mt2)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) x[mt2][ks] = *reinterpret_cast<const bf16x8*>(half + xoff[mt2][ks]);
    };
    auto read_w = [&](bf16x8 (&w)[4], const char* half) {
        if constexpr (ABL == 2 || ABL == 4) return;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) w[ks] = *reinterpret_cast<const bf16x8*>(half + woff + ks * 1024);
    };
    if constexpr (ABL == 1 || ABL == 2) {                                           // ablations run on a zero-filled LDS image
        for (int i = tid; i < 2 * G2_BUF / 16; i += 512) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0u, 0u, 0u, 0u);
        __syncthreads();
    }
    if constexpr (ABL == 2 || ABL == 4) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            w0[ks] = *reinterpret_cast<const bf16x8*>(smem + woff + ks * 1024);
            w1[ks] = w0[ks];
#pragma unroll
            for (int mt2 = 0; mt2 < 2; ++mt2) { x0[mt2][ks] = *reinterpret_cast<const bf16x8*>(smem + xoff[mt2][ks]); x1[mt2][ks] = x0[mt2][ks]; }
        }
        asm volatile("" : "+v"(w0[0]), "+v"(w1[0]), "+v"(x0[0][0]), "+v"(x1[0][0]));
    }

    // one MFMA section: NQ quadrants of 8 MFMAs; the closing `tail` (wait + barrier) is issued before the last EARLY MFMAs
    auto mfma8 = [&](bf16x8 (&w)[4], bf16x8 (&x)[2][4], int j, int i, int first, int total, auto&& tail) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int mt2 = 0; mt2 < 2; ++mt2) {
                const int idx = first + ks * 2 + mt2;
                if (EARLY > 0 && idx == total - EARLY) { SB(); __builtin_amdgcn_s_setprio(0); tail(); __builtin_amdgcn_s_setprio(1); SB(); }
                if constexpr (ABL != 4) acc[j][2 * i + mt2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[ks], x[mt2][ks], acc[j][2 * i + mt2], 0, 0, 0);
            }
    };
    auto quad = [&](bf16x8 (&w)[4], bf16x8 (&x)[2][4], int j, int i, auto&& tail) {
        __builtin_amdgcn_s_setprio(1);
        mfma8(w, x, j, i, 0, 8, tail);
        __builtin_amdgcn_s_setprio(0);
        if (EARLY == 0) tail();
    };

    if constexpr (SCHED == 0) {
        stage(0, 0, smem); stage(1, 0, smem); stage(2, 0, smem); stage(3, 0, smem);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        BAR();
        if (wr == 1) BAR();
        read_x(x0, smem);
        for (int t = 0; t < KT; ++t) {
            char* buf = smem + (t & 1) * G2_BUF;
            char* nbuf = smem + ((t + 1) & 1) * G2_BUF;
            const bool more = t + 1 < KT;
            auto tail = [&]() {
                if (more) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                BAR();
            };
            read_w(w0, buf + 2 * G2_HALF);
            if (more) stage(0, t + 1, nbuf);
            BAR(); quad(w0, x0, 0, 0, tail);
            read_w(w1, buf + 3 * G2_HALF);
            if (more) stage(1, t + 1, nbuf);
            BAR(); quad(w1, x0, 1, 0, tail);
            read_x(x1, buf + G2_HALF);
            if (more) stage(2, t + 1, nbuf);
            BAR(); quad(w1, x1, 1, 1, tail);
            if (more) { read_x(x0, nbuf); stage(3, t + 1, nbuf); }
            BAR(); quad(w0, x1, 0, 1, tail);
        }
        if (wr == 0) BAR();
    } else if constexpr (SCHED == 1) {
        // steady state before K-tile t: all of t staged, plus X0 W0 W1 of t+1; X1(t+1) goes in phase 1 of t
        stage(0, 0, smem); stage(1, 0, smem); stage(2, 0, smem); stage(3, 0, smem);
        stage(0, 1, smem + G2_BUF); stage(1, 1, smem + G2_BUF); stage(2, 1, smem + G2_BUF);
        if constexpr (ABL == 0) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        BAR();
        if (wr == 1) BAR();
        read_x(x0, smem);
        auto ktile = [&](int t, auto mode_tag) {
            constexpr int MODE = decltype(mode_tag)::value;              // 0 steady, 1 t == KT-2, 2 t == KT-1
            char* buf = smem + (t & 1) * G2_BUF;
            char* nbuf = smem + ((t + 1) & 1) * G2_BUF;
            auto tail = [&]() {
                if constexpr (ABL == 3) {}
                else if constexpr (MODE == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                BAR();
            };
            read_w(w0, buf + 2 * G2_HALF);
            if constexpr (MODE <= 1) stage(3, t + 1, nbuf);              // X1(t+1): slot last read in phase 3 of t-1
            BAR(); quad(w0, x0, 0, 0, tail);
            read_w(w1, buf + 3 * G2_HALF);
            if constexpr (MODE == 0) stage(0, t + 2, buf);               // X0(t+2): slot last read in phase 4 of t-1
            BAR(); quad(w1, x0, 1, 0, tail);
            read_x(x1, buf + G2_HALF);
            if constexpr (MODE == 0) stage(1, t + 2, buf);               // W0(t+2): slot last read in phase 1 of t
            BAR(); quad(w1, x1, 1, 1, tail);
            if constexpr (MODE <= 1) read_x(x0, nbuf);
            if constexpr (MODE == 0) stage(2, t + 2, buf);               // W1(t+2): slot last read in phase 2 of t
            BAR(); quad(w0, x1, 0, 1, tail);
        };
        int t = 0;
        for (; t < KT - 2; ++t) ktile(t, std::integral_constant<int, 0>{});
        ktile(t, std::integral_constant<int, 1>{}); ++t;
        ktile(t, std::integral_constant<int, 2>{});
        if (wr == 0) BAR();
    } else if constexpr (SCHED == 3 || SCHED == 4) {
        // SCHED 3: 2 phases per K-tile, two half-tiles staged in EACH phase.  P1 = quadrants (0,0) (0,1): reads W0 W1 (8); P2 = (1,1) (1,0): reads
        // X1(t) and X0(t+1) (16).  Stages: P1(t): X1(t+1), X0(t+2);  P2(t): W0(t+2), W1(t+2) -- every slot one phase after its last read (the reads
        // are waited for before the phase's first barrier), three phases before its first read.  SCHED 4: the same with the LDS-DMA issued first.
        stage(0, 0, smem); stage(1, 0, smem); stage(2, 0, smem); stage(3, 0, smem);
        stage(0, 1, smem + G2_BUF); stage(1, 1, smem + G2_BUF); stage(2, 1, smem + G2_BUF);
        if constexpr (ABL == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        BAR();
        read_x(x0, smem);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        BAR();
        if (wr == 1) BAR();
        auto ktile = [&](int t, auto mode_tag) {
            constexpr int MODE = decltype(mode_tag)::value;              // 0 steady, 1 t == KT-2, 2 t == KT-1
            char* buf = smem + (t & 1) * G2_BUF;
            char* nbuf = smem + ((t + 1) & 1) * G2_BUF;
            auto lwait = [&]() {
                if constexpr (ABL == 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                else if constexpr (MODE == 0) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            };
            auto tail = [&]() { BAR(); };
            // P1
            if constexpr (SCHED == 4) { if constexpr (MODE <= 1) stage(3, t + 1, nbuf); if constexpr (MODE == 0) stage(0, t + 2, buf); }
            read_w(w0, buf + 2 * G2_HALF); read_w(w1, buf + 3 * G2_HALF);
            if constexpr (SCHED == 3) { if constexpr (MODE <= 1) stage(3, t + 1, nbuf); if constexpr (MODE == 0) stage(0, t + 2, buf); }
            lwait(); SB(); BAR();
            __builtin_amdgcn_s_setprio(1);
            mfma8(w0, x0, 0, 0, 0, 16, tail); mfma8(w1, x0, 1, 0, 8, 16, tail);
            __builtin_amdgcn_s_setprio(0);
            if (EARLY == 0) tail();
            // P2
            if constexpr (SCHED == 4) { if constexpr (MODE == 0) { stage(1, t + 2, buf); stage(2, t + 2, buf); } }
            read_x(x1, buf + G2_HALF);
            if constexpr (MODE <= 1) read_x(x0, nbuf);
            if constexpr (SCHED == 3) { if constexpr (MODE == 0) { stage(1, t + 2, buf); stage(2, t + 2, buf); } }
            lwait(); SB(); BAR();
            __builtin_amdgcn_s_setprio(1);
            mfma8(w1, x1, 1, 1, 0, 16, tail); mfma8(w0, x1, 0, 1, 8, 16, tail);
            __builtin_amdgcn_s_setprio(0);
            if (EARLY == 0) tail();
        };
        int t = 0;
        for (; t < KT - 2; ++t) ktile(t, std::integral_constant<int, 0>{});
        ktile(t, std::integral_constant<int, 1>{}); ++t;
        ktile(t, std::integral_constant<int, 2>{});
        if (wr == 0) BAR();
    } else {
        // SCHED 2: phase P1 = quadrants (0,0) (0,1), phase P2 = (1,1) (1,0).  Before K-tile t: all of t staged and X0 W0 W1 X1 of t+1.
        stage(0, 0, smem); stage(1, 0, smem); stage(2, 0, smem); stage(3, 0, smem);
        stage(0, 1, smem + G2_BUF); stage(1, 1, smem + G2_BUF); stage(2, 1, smem + G2_BUF); stage(3, 1, smem + G2_BUF);
        if constexpr (ABL == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        BAR();
        read_x(x0, smem);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        BAR();
        if (wr == 1) BAR();
        auto ktile = [&](int t, auto mode_tag) {
            constexpr int MODE = decltype(mode_tag)::value;
            char* buf = smem + (t & 1) * G2_BUF;
            char* nbuf = smem + ((t + 1) & 1) * G2_BUF;
            auto lwait = [&]() {
                if constexpr (MODE == 0) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            };
            auto tail = [&]() { BAR(); };
            auto none = [&]() {};
            // P1
            read_w(w0, buf + 2 * G2_HALF); read_w(w1, buf + 3 * G2_HALF); read_x(x1, buf + G2_HALF);
            if constexpr (MODE == 0) stage(0, t + 2, buf);               // X0(t+2): slot last read in P2 of t-1 (waited for before its barrier)
            lwait(); SB(); BAR();
            __builtin_amdgcn_s_setprio(1);
            mfma8(w0, x0, 0, 0, 0, 16, tail); mfma8(w1, x0, 1, 0, 8, 16, tail);
            __builtin_amdgcn_s_setprio(0);
            if (EARLY == 0) tail();
            // P2
            if constexpr (MODE <= 1) read_x(x0, nbuf);
            if constexpr (MODE == 0) { stage(1, t + 2, buf); stage(2, t + 2, buf); stage(3, t + 2, buf); }      // slots read in P1 of t
            lwait(); SB(); BAR();
            __builtin_amdgcn_s_setprio(1);
            mfma8(w1, x1, 1, 1, 0, 16, tail); mfma8(w0, x1, 0, 1, 8, 16, tail);
            __builtin_amdgcn_s_setprio(0);
            if (EARLY == 0) tail();
            (void)none;
        };
        int t = 0;
        for (; t < KT - 2; ++t) ktile(t, std::integral_constant<int, 0>{});
        ktile(t, std::integral_constant<int, 1>{}); ++t;
        ktile(t, std::integral_constant<int, 2>{});
        if (wr == 0) BAR();
    }

    // plain register epilogue: bf16 stores, one output row per lane (not what is being measured)
    const int half = lane >> 5;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int m = m0 + wr * 128 + (mt >> 1) * 64 + (mt & 1) * 32 + (lane & 31);
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int n = n0 + wc * 64 + j * 32 + rg * 8 + half * 4;
                if (n >= N) continue;
                uint2 o;
                o.x = pack2bf(acc[j][mt][rg * 4 + 0], acc[j][mt][rg * 4 + 1]);
                o.y = pack2bf(acc[j][mt][rg * 4 + 2], acc[j][mt][rg * 4 + 3]);
                *reinterpret_cast<uint2*>(C + (size_t)m * N + n) = o;
            }
    }
}

typedef void (*kern_t)(const bf16_t*, const bf16_t*, bf16_t*, int, int, int, int, int);
struct Variant { const char* name; kern_t k; int abl; };
#define V(S, E, AB) Variant{"sched" #S "_early" #E "_abl" #AB, lab_kernel<S, E, AB>, AB}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 5;
    const char* only = argc > 2 ? argv[2] : nullptr;
    std::vector<Variant> vs = {V(0, 0, 0), V(1, 0, 0), V(3, 0, 0), V(1, 0, 4), V(3, 0, 4)};
    for (auto& v : vs) CK(hipFuncSetAttribute((const void*)v.k, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * G2_BUF));
    struct Shape { const char* name; int M, N, K; };
    std::vector<Shape> shapes = {{"check", 1024, 768, 512}, {"8192^3", 8192, 8192, 8192}, {"c_proj", 8192, 2048, 2048}, {"c_fc", 8192, 8192, 2048},
                                 {"down", 8192, 2048, 8192}};
    size_t maxA = 0, maxW = 0, maxC = 0;
    for (auto& s : shapes) { maxA = std::max(maxA, (size_t)s.M * s.K); maxW = std::max(maxW, (size_t)s.N * s.K); maxC = std::max(maxC, (size_t)s.M * s.N); }
    bf16_t *A, *W, *Wp, *C0, *C1;
    CK(hipMalloc(&A, maxA * 2)); CK(hipMalloc(&W, maxW * 2)); CK(hipMalloc(&Wp, maxW * 2)); CK(hipMalloc(&C0, maxC * 2)); CK(hipMalloc(&C1, maxC * 2));
    float* err; unsigned* ndiff;
    CK(hipMalloc(&err, 8)); CK(hipMalloc(&ndiff, 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int zero = 0; zero < 2; ++zero) {
        for (auto& s : shapes) {
            if (zero && strcmp(s.name, "8192^3") && strcmp(s.name, "c_proj")) continue;
            const int M = s.M, N = s.N, K = s.K;
            fill_kernel<<<2048, 256>>>(A, (size_t)M * K, 0x1234u, zero);
            fill_kernel<<<2048, 256>>>(W, (size_t)N * K, 0x9876u, zero);
            pack_w_kernel<<<2048, 256>>>(W, Wp, N, K);
            CK(hipDeviceSynchronize());
            const int tiles_m = (M + 255) / 256, tiles_n = (N + 255) / 256;
            const double flops = 2.0 * M * N * K;
            auto launch = [&](const Variant& v, bf16_t* C) { v.k<<<tiles_m * tiles_n, 512, 2 * G2_BUF>>>(A, Wp, C, M, N, K, tiles_m, tiles_n); };
            // correctness: variant 0 against the fp32 reference (sampled), every full variant bit for bit against variant 0, three runs each
            if (!zero) {
                CK(hipMemset(C0, 0, (size_t)M * N * 2));
                launch(vs[0], C0);
                CK(hipMemset(err, 0, 8));
                const int samples = 1 << 16;
                ref_check_kernel<<<samples / 256, 256>>>(A, W, C0, M, N, K, samples, err);
                float h[2];
                CK(hipMemcpy(h, err, 8, hipMemcpyDeviceToHost));
                printf("[check %s %dx%dx%d] variant0 vs fp32 reference at %d samples: max|err| %.4f, max|ref| %.2f (%s)\n", s.name, M, N, K, samples, h[0], h[1],
                       h[0] <= 0.01f * h[1] + 0.02f ? "ok" : "MISMATCH");
                for (size_t vi = 1; vi < vs.size(); ++vi) {
                    if (vs[vi].abl) continue;   // (ABL 3 skips the waits: its output is not checked)
                    unsigned worst = 0;
                    for (int rep = 0; rep < 3; ++rep) {
                        CK(hipMemset(C1, 0xff, (size_t)M * N * 2));
                        launch(vs[vi], C1);
                        CK(hipMemset(ndiff, 0, 4));
                        diff_kernel<<<1024, 256>>>((const uint4*)C0, (const uint4*)C1, (size_t)M * N / 8, ndiff);
                        unsigned nd;
                        CK(hipMemcpy(&nd, ndiff, 4, hipMemcpyDeviceToHost));
                        worst = std::max(worst, nd);
                    }
                    if (worst) printf("[check %s] %s: %u 16-byte groups differ from variant0  <-- WRONG\n", s.name, vs[vi].name, worst);
                }
                fflush(stdout);
            }
            if (!strcmp(s.name, "check")) continue;
            // timing: interleaved rounds, median / min
            const int iters = std::max(3, (int)(2.0e13 / flops));                 // ~15-20 ms of launches per sample at ~1 PF
            std::vector<std::vector<double>> us(vs.size());
            for (auto& v : vs) { if (only && !strstr(v.name, only)) continue; launch(v, C1); }
            CK(hipDeviceSynchronize());
            for (int r = 0; r < rounds; ++r)
                for (size_t vi = 0; vi < vs.size(); ++vi) {
                    if (only && !strstr(vs[vi].name, only)) continue;
                    CK(hipEventRecord(e0));
                    for (int it = 0; it < iters; ++it) launch(vs[vi], C1);
                    CK(hipEventRecord(e1));
                    CK(hipEventSynchronize(e1));
                    float ms;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    us[vi].push_back(ms * 1000.0 / iters);
                }
            for (size_t vi = 0; vi < vs.size(); ++vi) {
                if (us[vi].empty()) continue;
                std::sort(us[vi].begin(), us[vi].end());
                const double med = us[vi][us[vi].size() / 2], mn = us[vi][0];
                printf("%-8s %-6s %-22s median %9.1f us %7.1f TF   min %9.1f us %7.1f TF\n", s.name, zero ? "zero" : "random", vs[vi].name, med, flops / med / 1e6, mn,
                       flops / mn / 1e6);
            }
            fflush(stdout);
        }
    }
    return 0;
}
