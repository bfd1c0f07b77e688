// LDS atomic throughput on one CU: clocks per wavefront instruction for ds_add (no return), ds_add_rtn (returning) and plain
// ds_write / ds_read on random and on conflict-free addresses, with 1..16 wavefronts per workgroup.
//   hipcc --offload-arch=gfx950 -O3 tools/lds_atomic_rate_probe.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE, int RANDOM>
__global__ __launch_bounds__(1024) void k(uint32_t* out, long long* clocks, int iters, int bins) {
    __shared__ uint32_t cnt[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) cnt[i] = 0;
    __syncthreads();
    uint32_t x = threadIdx.x * 2654435761u + 12345u;
    uint32_t acc = 0;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            x = x * 1664525u + 1013904223u;
            const uint32_t b = RANDOM ? (x >> 10) & (uint32_t)(bins - 1) : ((threadIdx.x + 64u * u) & (uint32_t)(bins - 1));
            if (MODE == 0) atomicAdd(&cnt[b], 1u);                         // result unused: ds_add_u32
            else if (MODE == 1) acc += atomicAdd(&cnt[b], 1u);             // ds_add_rtn_u32
            else if (MODE == 2) cnt[b] = x;                                // ds_write_b32
            else acc += cnt[b];                                            // ds_read_b32
        }
    }
    const long long t1 = clock64();
    __syncthreads();
    if (threadIdx.x == 0) clocks[blockIdx.x] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + cnt[threadIdx.x & 4095];
}

int main() {
    uint32_t* out; long long* clocks;
    CHK(hipMalloc(&out, 1024 * 1024 * 4)); CHK(hipMalloc(&clocks, 1024 * 8));
    const char* names[4] = {"ds_add (no return)", "ds_add_rtn", "ds_write_b32", "ds_read_b32"};
    const int iters = 2000;
    for (int mode = 0; mode < 4; ++mode)
        for (int rnd = 0; rnd < 2; ++rnd)
            for (int waves = 1; waves <= 16; waves *= 4) {
                for (int rep = 0; rep < 2; ++rep) {
                    dim3 g(1), b(64 * waves);
#define L(M, R) hipLaunchKernelGGL((k<M, R>), g, b, 0, 0, out, clocks, iters, 1024)
                    if (mode == 0) { if (rnd) L(0, 1); else L(0, 0); }
                    if (mode == 1) { if (rnd) L(1, 1); else L(1, 0); }
                    if (mode == 2) { if (rnd) L(2, 1); else L(2, 0); }
                    if (mode == 3) { if (rnd) L(3, 1); else L(3, 0); }
                    CHK(hipDeviceSynchronize());
                }
                long long c; CHK(hipMemcpy(&c, clocks, 8, hipMemcpyDeviceToHost));
                const double per_instr = (double)c / (iters * 8.0);          // s_memtime / clock64 ticks per wave instruction (one wave's view)
                printf("%-20s %-14s %2d waves: %7.1f ticks per instruction per wave, %6.2f ticks per instruction overall\n", names[mode],
                       rnd ? "random bins" : "conflict-free", waves, per_instr, per_instr / waves);
            }
    return 0;
}
