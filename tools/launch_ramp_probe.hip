// How long does a chip-filling launch take to get ALL its workgroups started -- as a function of what a workgroup allocates?
// A chain of dependent kernels (each reads the word its predecessor wrote, spins ~4 us, writes it back); variants: threads per
// workgroup (256 / 512), static LDS per workgroup (0 / 36 KB / 76 KB), a register budget capped by waves-per-SIMD (2).
// Every workgroup stamps its start (wall clock) and the first workgroup's start: printed are the chain's us per kernel and the
// median start spread (last start - first start) of a launch.
// build: hipcc --offload-arch=gfx950 -O3 tools/launch_ramp_probe.hip -o tools/_bin/launch_ramp_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

template <int LDSF, int THREADS>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_step(int* __restrict__ cell, int work, long long* __restrict__ starts, int launch) {
    __shared__ float buf[LDSF > 0 ? LDSF : 1];
    const long long t0w = wall_clock64();
    if (threadIdx.x == 0 && starts) starts[(long long)launch * gridDim.x + blockIdx.x] = t0w;
    const int v = __hip_atomic_load(cell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (LDSF > 0) buf[threadIdx.x % LDSF] = (float)v;
    long long t0 = clock64();
    while (clock64() - t0 < work) { }
    if (LDSF > 0 && buf[(threadIdx.x + 1) % LDSF] == 12345.678f) cell[1] = 1;
    if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(cell, v + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int LDSF, int THREADS>
static void run(const char* name, int grid, int N, int* cell, long long* starts, hipStream_t st) {
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int i = 0; i < 100; ++i) k_step<LDSF, THREADS><<<grid, THREADS, 0, st>>>(cell, 8000, nullptr, 0);
    CHK(hipStreamSynchronize(st));
    CHK(hipEventRecord(e0, st));
    for (int i = 0; i < N; ++i) k_step<LDSF, THREADS><<<grid, THREADS, 0, st>>>(cell, 8000, starts, i);
    CHK(hipEventRecord(e1, st));
    CHK(hipStreamSynchronize(st));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> h((size_t)N * grid);
    CHK(hipMemcpy(h.data(), starts, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
    std::vector<double> spread;
    for (int i = 10; i < N; ++i) {
        long long lo = h[(size_t)i * grid], hi = lo;
        for (int b = 0; b < grid; ++b) { lo = std::min(lo, h[(size_t)i * grid + b]); hi = std::max(hi, h[(size_t)i * grid + b]); }
        spread.push_back((hi - lo) / 100.0);
    }
    std::sort(spread.begin(), spread.end());
    printf("{\"variant\": \"%s\", \"grid\": %d, \"threads\": %d, \"lds_bytes\": %d, \"us_per_kernel\": %.3f, \"start_spread_us_median\": %.2f}\n", name, grid, THREADS,
           LDSF * 4, ms * 1000.0 / N, spread[spread.size() / 2]);
}

int main() {
    const int N = 1500;
    int* cell; CHK(hipMalloc(&cell, 256)); CHK(hipMemset(cell, 0, 256));
    long long* starts; CHK(hipMalloc(&starts, (size_t)N * 512 * sizeof(long long)));
    hipStream_t st; CHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    // the same on a CU-masked stream (the library's training partition: mask bits 0 .. 191 of 256)
    hipStream_t sm;
    { uint32_t mask[8] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u}; CHK(hipExtStreamCreateWithCUMask(&sm, 8, mask)); }
    run<9216, 512>("36 KB LDS, CU-masked stream (192 of 256)", 192, N, cell, starts, sm);
    run<0, 512>("no LDS, CU-masked stream (192 of 256)", 192, N, cell, starts, sm);
    run<9216, 256>("36 KB LDS, CU-masked stream (192 of 256)", 192, N, cell, starts, sm);
    for (int grid : {192, 288}) {
        run<0, 512>("no LDS", grid, N, cell, starts, st);
        run<0, 256>("no LDS", grid, N, cell, starts, st);
        run<9216, 512>("36 KB LDS", grid, N, cell, starts, st);
        run<9216, 256>("36 KB LDS", grid, N, cell, starts, st);
        run<12288, 512>("48 KB LDS", grid, N, cell, starts, st);
        run<15872, 512>("62 KB LDS", grid, N, cell, starts, st);
    }
    return 0;
}
