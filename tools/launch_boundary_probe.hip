// What one dependent kernel boundary costs on this part, three ways: launches on a stream (the host far ahead), the same chain
// captured into a hipGraph and replayed, and the chain with a small and a chip-filling grid.  Each kernel reads a word the previous
// one wrote (a real dependency through memory) and spins for about `work` clocks.
// build: hipcc --offload-arch=gfx950 -O3 tools/launch_boundary_probe.hip -o tools/_bin/launch_boundary_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_step(int* __restrict__ cell, int work) {
    const int v = __hip_atomic_load(cell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    long long t0 = clock64();
    while (clock64() - t0 < work) { }
    if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(cell, v + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 3000;
    int* cell; CHK(hipMalloc(&cell, 256));
    hipStream_t st; CHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int grids[3] = {1, 192, 512}, works[2] = {0, 10000};      // (10,000 clocks of 100 MHz = the s_memtime clock? printed as measured)
    printf("{\"launches\": %d, \"rows\": [", N);
    bool first = true;
    for (int gi = 0; gi < 3; ++gi) for (int wi = 0; wi < 2; ++wi) {
        const int grid = grids[gi], work = works[wi];
        CHK(hipMemsetAsync(cell, 0, 4, st));
        for (int i = 0; i < 200; ++i) k_step<<<grid, 512, 0, st>>>(cell, work);
        CHK(hipStreamSynchronize(st));
        // (a) stream launches
        double best_a = 1e30, best_g = 1e30;
        for (int rep = 0; rep < 3; ++rep) {
            CHK(hipEventRecord(e0, st));
            for (int i = 0; i < N; ++i) k_step<<<grid, 512, 0, st>>>(cell, work);
            CHK(hipEventRecord(e1, st));
            CHK(hipStreamSynchronize(st));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
            if (ms * 1000.0 / N < best_a) best_a = ms * 1000.0 / N;
        }
        // (b) the same chain as a graph
        hipGraph_t g; hipGraphExec_t ge;
        CHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < N; ++i) k_step<<<grid, 512, 0, st>>>(cell, work);
        CHK(hipStreamEndCapture(st, &g));
        CHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CHK(hipGraphLaunch(ge, st)); CHK(hipStreamSynchronize(st));
        for (int rep = 0; rep < 3; ++rep) {
            CHK(hipEventRecord(e0, st));
            CHK(hipGraphLaunch(ge, st));
            CHK(hipEventRecord(e1, st));
            CHK(hipStreamSynchronize(st));
            float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
            if (ms * 1000.0 / N < best_g) best_g = ms * 1000.0 / N;
        }
        CHK(hipGraphExecDestroy(ge)); CHK(hipGraphDestroy(g));
        printf("%s{\"grid\": %d, \"spin_clocks\": %d, \"stream_us_per_kernel\": %.3f, \"graph_us_per_kernel\": %.3f}", first ? "" : ", ", grid, work, best_a, best_g);
        first = false;
    }
    printf("]}\n");
    return 0;
}
