// Ceiling probe for the bare a3 gradient pass: 128-byte random row reads + in-place row writes with no arithmetic to
// speak of.  hipcc --offload-arch=gfx950 -O3 tools/micro_gather.hip -o gpurun_out/micro_gather && ./gpurun_out/micro_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <random>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ __launch_bounds__(256) void k(const int64_t* __restrict__ tri, int B, float* wu, float* wi, int mode) {
    const int gid = blockIdx.x * 256 + threadIdx.x;
    const int t = gid / 8, sub = gid % 8;
    if (t >= B) return;
    const int64_t iu = tri[(int64_t)t * 3], ii = tri[(int64_t)t * 3 + 1], in = tri[(int64_t)t * 3 + 2];
    f32x4* pu = reinterpret_cast<f32x4*>(wu + iu * 32 + sub * 4);
    f32x4* pi = reinterpret_cast<f32x4*>(wi + ii * 32 + sub * 4);
    f32x4* pn = reinterpret_cast<f32x4*>(wi + in * 32 + sub * 4);
    f32x4 u = *pu, i = *pi, n = *pn;
    if (mode == 0) { if (u[0] + i[0] + n[0] == 12345.678f) *pu = u; return; }          // reads only
    u += 1e-9f; i += 1e-9f; n += 1e-9f;
    if (NT) { __builtin_nontemporal_store(u, pu); __builtin_nontemporal_store(i, pi); __builtin_nontemporal_store(n, pn); }
    else { *pu = u; *pi = i; *pn = n; }
}
int main() {
    const int64_t U = 10000000, I = 1000000; const int B = 262144, NB = 16;
    float *wu, *wi; int64_t* tri;
    hipMalloc(&wu, U * 128); hipMalloc(&wi, I * 128); hipMalloc(&tri, (size_t)NB * B * 24);
    hipMemset(wu, 0, U * 128); hipMemset(wi, 0, I * 128);
    std::vector<int64_t> h((size_t)NB * B * 3); std::mt19937_64 g(1);
    for (size_t e = 0; e < (size_t)NB * B; ++e) { h[3 * e] = g() % U; h[3 * e + 1] = g() % I; h[3 * e + 2] = g() % I; }
    hipMemcpy(tri, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            for (int b = 0; b < NB; ++b) {
                if (mode == 2) k<true><<<B * 8 / 256, 256>>>(tri + (size_t)b * B * 3, B, wu, wi, 1);
                else k<false><<<B * 8 / 256, 256>>>(tri + (size_t)b * B * 3, B, wu, wi, mode);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) {
                const double bytes = (double)NB * B * (24 + 384 + (mode ? 384 : 0));
                printf("mode %d (%s): %.1f us per 262144-triple launch, %.2f TB/s\n", mode,
                       mode == 0 ? "3 row reads" : mode == 1 ? "3 row reads + 3 row writes" : "same, nontemporal stores",
                       1000.0 * ms / NB, bytes / (ms * 1e-3) / 1e12);
            }
        }
    }
    return 0;
}
