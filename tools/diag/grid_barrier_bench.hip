// How much does a grid-wide barrier cost inside ONE persistent launch on MI355X (256 blocks x 512 threads, one per CU)?
// Compared with the ~3.3 us dispatch-to-dispatch spacing of dependent launches in the replayed decode graph.
//   variant 0: atomic counter + spin (relaxed agent-scope atomics), no memory fences          (pure rendezvous)
//   variant 1: + release fence before the arrival and acquire fence after the wait           (agent scope: L2 write-back /
//              invalidate across the 8 XCDs -- what a producer/consumer hand-off through ordinary loads and stores needs)
//   variant 2: variant 1 + every block writes 2 KiB before the barrier and reads another block's 2 KiB after it
// Spins are bounded: a stuck barrier sets an error flag instead of hanging the GPU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__device__ __forceinline__ bool grid_bar(unsigned* counter, unsigned target, int* err, int variant) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        if (variant >= 1) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 20)) { *err = 1; ok = false; break; }
        }
        if (variant >= 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    return ok;
}

__global__ __launch_bounds__(512) void bar_kernel(unsigned* counter, int* err, float* buf, int nbar, int variant, float* sink) {
    const int nb = gridDim.x;
    float acc = 0.f;
    for (int i = 0; i < nbar; ++i) {
        if (variant == 2) buf[(size_t)blockIdx.x * 512 + threadIdx.x] = (float)(i + blockIdx.x);
        if (!grid_bar(counter, (unsigned)(i + 1) * nb, err, variant)) return;
        if (variant == 2) acc += buf[(size_t)((blockIdx.x + 97) % nb) * 512 + threadIdx.x];
    }
    if (acc == 12345.678f) *sink = acc;
}
__global__ void empty_kernel(float* p) { if (p && threadIdx.x == 9999) *p = 1.f; }

int main() {
    unsigned* counter; int* err; float *buf, *sink;
    hipMalloc(&counter, 4); hipMalloc(&err, 4); hipMalloc(&buf, 256 * 512 * 4); hipMalloc(&sink, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int nbar = 2000;
    for (int blocks : {256, 128}) {
        for (int variant = 0; variant < 3; ++variant) {
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                hipMemset(counter, 0, 4); hipMemset(err, 0, 4);
                hipEventRecord(e0);
                bar_kernel<<<blocks, 512>>>(counter, err, buf, nbar, variant, sink);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            int herr = 0; hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
            printf("%d blocks, variant %d: %.3f us per grid barrier%s\n", blocks, variant, best * 1e3f / nbar, herr ? "  (BARRIER TIMED OUT)" : "");
        }
    }
    // the alternative: dependent launches of a (nearly) empty kernel, back to back on one stream, and as a replayed hipGraph
    for (int g = 0; g < 2; ++g) {
        hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        const int n = 2000;
        hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr;
        if (g) {
            hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
            for (int i = 0; i < n; ++i) empty_kernel<<<256, 512, 0, st>>>(sink);
            hipStreamEndCapture(st, &graph);
            hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
            hipGraphLaunch(exec, st); hipStreamSynchronize(st);
        }
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0, st);
            if (g) hipGraphLaunch(exec, st);
            else for (int i = 0; i < n; ++i) empty_kernel<<<256, 512, 0, st>>>(sink);
            hipEventRecord(e1, st); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("dependent launches of an empty 256 x 512 kernel, %s: %.3f us per launch\n", g ? "replayed hipGraph" : "stream", best * 1e3f / n);
    }
    return 0;
}
