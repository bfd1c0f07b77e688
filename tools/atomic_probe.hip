// Throughput of random device-scope atomicOr over a table (measurement tool): n ops over `words` 32-bit words.
// build: hipcc --offload-arch=gfx950 -O2 -o tools/atomic_probe tools/atomic_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
__global__ void k_idx(uint32_t* idx, long long n, uint32_t words) {
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint64_t z = (uint64_t)i * 0x9e3779b97f4a7c15ull; z ^= z >> 31; z *= 0xbf58476d1ce4e5b9ull; z ^= z >> 29;
    idx[i] = (uint32_t)(z % words);
}
__global__ void k_or(uint32_t* state, const uint32_t* idx, long long n, uint32_t* dup) {
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t bit = 1u << (2 * (i & 15));
    const uint32_t old = atomicOr(&state[idx[i]], bit);
    if (old & bit) atomicOr(&state[idx[i]], bit << 1);
}
__global__ void k_read(const uint32_t* state, const uint32_t* idx, long long n, unsigned char* out) {
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    out[i] = (state[idx[i]] >> (2 * (i & 15) + 1)) & 1;
}
int main(int argc, char** argv) {
    const long long n = argc > 1 ? atoll(argv[1]) : 12600000;
    const uint32_t words = argc > 2 ? (uint32_t)atoll(argv[2]) : 10000000u;
    uint32_t *state, *idx; unsigned char* out;
    hipMalloc(&state, (size_t)words * 4); hipMalloc(&idx, (size_t)n * 4); hipMalloc(&out, n);
    k_idx<<<(unsigned)((n + 255) / 256), 256>>>(idx, n, words);
    hipEvent_t e0, e1, e2, e3; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2); hipEventCreate(&e3);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipMemsetAsync(state, 0, (size_t)words * 4);
        hipEventRecord(e1);
        k_or<<<(unsigned)((n + 255) / 256), 256>>>(state, idx, n, nullptr);
        hipEventRecord(e2);
        k_read<<<(unsigned)((n + 255) / 256), 256>>>(state, idx, n, out);
        hipEventRecord(e3);
        hipDeviceSynchronize();
        float a, b, c; hipEventElapsedTime(&a, e0, e1); hipEventElapsedTime(&b, e1, e2); hipEventElapsedTime(&c, e2, e3);
        printf("n=%lld words=%u: memset %.1f us, atomicOr pass %.1f us (%.2f G/s), read pass %.1f us\n", n, words, a * 1e3, b * 1e3, n / b / 1e6, c * 1e3);
    }
    return 0;
}
