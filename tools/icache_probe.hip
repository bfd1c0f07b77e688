// How much straight-line code does a CU's instruction cache hold ACROSS launches, and what does a miss cost?  A chain of dependent
// kernels (each reads the word its predecessor wrote and writes it back).  The body is N dependent `v_fma_f32 v, v, v, v` (8 bytes
// each, inline asm: exactly N * 8 bytes of code), executed once per wavefront.  While the code stays resident between launches a
// kernel costs launch + N * (FMA latency); once it does not, every 64-byte line is a miss.  Also pairs / triples of kernels
// alternating, as a training step alternates its kernels.
// build: hipcc --offload-arch=gfx950 -O3 tools/icache_probe.hip -o tools/_bin/icache_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

#define F1(x) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x));
#define F4(x) F1(x) F1(x) F1(x) F1(x)
#define F16(x) F4(x) F4(x) F4(x) F4(x)
#define F64(x) F16(x) F16(x) F16(x) F16(x)
#define F256(x) F64(x) F64(x) F64(x) F64(x)       // 2 KB
#define F1K(x) F256(x) F256(x) F256(x) F256(x)    // 8 KB

template <int KB8, int SALT>       // KB8 blocks of 8 KB of code
__global__ __launch_bounds__(512) void k_code(int* __restrict__ cell, float* __restrict__ sink) {
    const int v = __hip_atomic_load(cell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    float x = 1.0f + (float)SALT * 0.0f;
    if constexpr (KB8 >= 1) { F1K(x) }
    if constexpr (KB8 >= 2) { F1K(x) }
    if constexpr (KB8 >= 3) { F1K(x) }
    if constexpr (KB8 >= 4) { F1K(x) }
    if constexpr (KB8 >= 5) { F1K(x) }
    if constexpr (KB8 >= 6) { F1K(x) }
    if constexpr (KB8 >= 7) { F1K(x) }
    if constexpr (KB8 >= 8) { F1K(x) }
    if constexpr (KB8 >= 9) { F1K(x) F1K(x) F1K(x) F1K(x) }      // 12 blocks = 96 KB
    if (x == 123.456f) sink[threadIdx.x] = x;
    if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(cell, v + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <typename F>
static double chain(F launch, int N, hipStream_t st) {
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int i = 0; i < 60; ++i) launch(i);
    CHK(hipStreamSynchronize(st));
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
        CHK(hipEventRecord(e0, st));
        for (int i = 0; i < N; ++i) launch(i);
        CHK(hipEventRecord(e1, st));
        CHK(hipStreamSynchronize(st));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        if (ms * 1000.0 / N < best) best = ms * 1000.0 / N;
    }
    return best;
}
#define ONE(KB8) printf(", \"%d_KB\": %.2f", KB8 == 9 ? 96 : KB8 * 8, chain([&](int) { k_code<KB8, 0><<<grid, 512, 0, st>>>(cell, sink); }, N, st));

int main() {
    const int N = 900;
    int* cell; CHK(hipMalloc(&cell, 256)); CHK(hipMemset(cell, 0, 256));
    float* sink; CHK(hipMalloc(&sink, 4096));
    hipStream_t st; CHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (int grid : {1, 192}) {
        printf("{\"grid\": %d, \"us_per_kernel\": {\"0\": 0", grid);
        ONE(1) ONE(2) ONE(3) ONE(4) ONE(5) ONE(6) ONE(7) ONE(8) ONE(9)
        printf("}, \"alternating\": {\"0\": 0");
        printf(", \"8+32_KB_pair_avg\": %.2f", chain([&](int i) { if (i & 1) k_code<1, 1><<<grid, 512, 0, st>>>(cell, sink); else k_code<4, 1><<<grid, 512, 0, st>>>(cell, sink); }, N, st));
        printf(", \"16+16+8_KB_triple_avg\": %.2f", chain([&](int i) { if (i % 3 == 0) k_code<2, 2><<<grid, 512, 0, st>>>(cell, sink); else if (i % 3 == 1) k_code<2, 3><<<grid, 512, 0, st>>>(cell, sink);
                                                                 else k_code<1, 2><<<grid, 512, 0, st>>>(cell, sink); }, N, st));
        printf(", \"24+24+24_KB_triple_avg\": %.2f", chain([&](int i) { if (i % 3 == 0) k_code<3, 4><<<grid, 512, 0, st>>>(cell, sink); else if (i % 3 == 1) k_code<3, 5><<<grid, 512, 0, st>>>(cell, sink);
                                                                 else k_code<3, 6><<<grid, 512, 0, st>>>(cell, sink); }, N, st));
        printf("}}\n");
    }
    return 0;
}
