// Where do the workgroups of a CU-masked stream run?  (measurement tool, not part of the library)
// build: hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o tools/placement_probe.so tools/placement_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
__global__ void k_where(uint32_t* out, int spin) {
    if (threadIdx.x == 0) {
        const uint32_t xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xf;     // HW_REG_XCC_ID
        const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);             // HW_REG_HW_ID
        out[blockIdx.x * 2] = xcc;
        out[blockIdx.x * 2 + 1] = hw;
    }
    const uint64_t t0 = wall_clock64();
    while (wall_clock64() - t0 < (uint64_t)spin) {}
}
extern "C" int probe(const uint32_t* mask, int n_words, int n_blocks, int threads, int spin, uint32_t* host_out) {
    hipStream_t st;
    hipError_t e = n_words > 0 ? hipExtStreamCreateWithCUMask(&st, (uint32_t)n_words, mask) : hipStreamCreate(&st);
    if (e != hipSuccess) { fprintf(stderr, "stream: %s\n", hipGetErrorString(e)); return 1; }
    uint32_t* d;
    hipMalloc(&d, (size_t)n_blocks * 8);
    k_where<<<n_blocks, threads, 0, st>>>(d, spin);
    e = hipStreamSynchronize(st);
    if (e != hipSuccess) { fprintf(stderr, "run: %s\n", hipGetErrorString(e)); return 2; }
    hipMemcpy(host_out, d, (size_t)n_blocks * 8, hipMemcpyDeviceToHost);
    hipFree(d);
    hipStreamDestroy(st);
    return 0;
}
