// Does a busy neighbour raise the clock the latency-bound training kernels run at?  (measurement tool)
// A spinner kernel on a CU-masked stream of its own: `waves` wavefronts per CU of dependent FMAs until stopped.
// build: hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o tools/clock_keeper.so tools/clock_keeper.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
__global__ void k_keeper(volatile int* stop, float* sink, long long max_ticks, int mfma) {
    const long long t0 = wall_clock64();
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    for (;;) {
        if (mfma) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < 64; ++i) a = a * b + 0.5f;
        }
        if (*stop || wall_clock64() - t0 > max_ticks) break;
    }
    if (a == 12345.678f || acc[0] == 1.25f) sink[0] = a + acc[0];
}
static hipStream_t g_st = nullptr;
static int* g_stop = nullptr;
static float* g_sink = nullptr;
extern "C" int keeper_start(int cu_lo, int cu_hi, int blocks, int threads, double max_s, int mfma) {
    if (!g_st) {
        uint32_t mask[8] = {0};
        for (int b = cu_lo; b < cu_hi; ++b) mask[b / 32] |= 1u << (b % 32);
        if (hipExtStreamCreateWithCUMask(&g_st, 8, mask) != hipSuccess) return 1;
        hipHostMalloc(reinterpret_cast<void**>(&g_stop), sizeof(int), hipHostMallocMapped);
        hipMalloc(&g_sink, 64);
    }
    *g_stop = 0;
    int* dstop = nullptr;
    hipHostGetDevicePointer(reinterpret_cast<void**>(&dstop), g_stop, 0);
    k_keeper<<<blocks, threads, 0, g_st>>>(dstop, g_sink, (long long)(max_s * 1e8), mfma);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
extern "C" int keeper_stop() {
    if (!g_st) return 0;
    *g_stop = 1;
    return hipStreamSynchronize(g_st) == hipSuccess ? 0 : 3;
}
