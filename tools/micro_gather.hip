// Ceiling probe for the bare a3 gradient pass: random row reads + in-place row writes with no arithmetic to speak of, at the
// three row shapes of the bench line's a3 legs (d = 32 fp32: 128-byte rows on 10M x 1M; d = 64 fp32: 256-byte rows on
// 10M x 1M; d = 128 fp16: 256-byte rows on 50M x 5M).  One JSON object on stdout (committed under profiles/ per round;
// bench.py's a3 object quotes `ceiling_frac` = its kernels' algorithmic rate over the rate measured here).
//   hipcc --offload-arch=gfx950 -O3 tools/micro_gather.hip -o gpurun_out/micro_gather && ./gpurun_out/micro_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <random>
typedef float f32x4 __attribute__((ext_vector_type(4)));
// a row is LPR lanes x 16 bytes
template <int LPR, bool NT>
__global__ __launch_bounds__(256) void k(const int64_t* __restrict__ tri, int B, char* wu, char* wi, int mode) {
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t t = gid / LPR; const int sub = (int)(gid % LPR);
    if (t >= B) return;
    const int64_t iu = tri[t * 3], ii = tri[t * 3 + 1], in = tri[t * 3 + 2];
    f32x4* pu = reinterpret_cast<f32x4*>(wu + iu * (LPR * 16) + sub * 16);
    f32x4* pi = reinterpret_cast<f32x4*>(wi + ii * (LPR * 16) + sub * 16);
    f32x4* pn = reinterpret_cast<f32x4*>(wi + in * (LPR * 16) + sub * 16);
    f32x4 u = *pu, i = *pi, n = *pn;
    if (mode == 0) {                                                                  // reads only (every byte of the rows is used)
        const f32x4 s = u + i + n;
        if (s[0] + s[1] + s[2] + s[3] == 12345.678f) *pu = u;
        return;
    }
    u += 1e-9f; i += 1e-9f; n += 1e-9f;
    if (NT) { __builtin_nontemporal_store(u, pu); __builtin_nontemporal_store(i, pi); __builtin_nontemporal_store(n, pn); }
    else { *pu = u; *pi = i; *pn = n; }
}
template <int LPR>
static void shape(const char* name, int64_t U, int64_t I, bool last) {
    const int B = 262144, NB = 16;
    const size_t rb = (size_t)LPR * 16;
    char *wu, *wi; int64_t* tri;
    if (hipMalloc(&wu, U * rb) != hipSuccess || hipMalloc(&wi, I * rb) != hipSuccess || hipMalloc(&tri, (size_t)NB * B * 24) != hipSuccess) {
        printf("  \"%s\": null%s\n", name, last ? "" : ","); return;
    }
    hipMemset(wu, 0, U * rb); hipMemset(wi, 0, I * rb);
    std::vector<int64_t> h((size_t)NB * B * 3); std::mt19937_64 g(1);
    for (size_t e = 0; e < (size_t)NB * B; ++e) { h[3 * e] = g() % U; h[3 * e + 1] = g() % I; h[3 * e + 2] = g() % I; }
    hipMemcpy(tri, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("  \"%s\": {\"users\": %lld, \"items\": %lld, \"row_bytes\": %d, \"triples_per_launch\": %d, \"launches\": %d", name, (long long)U, (long long)I,
           (int)rb, B, NB);
    const char* mname[3] = {"reads_only", "reads_writes", "reads_writes_nontemporal"};
    for (int mode = 0; mode < 3; ++mode) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0);
            const unsigned grid = (unsigned)(((int64_t)B * LPR + 255) / 256);
            for (int b = 0; b < NB; ++b) {
                if (mode == 2) k<LPR, true><<<grid, 256>>>(tri + (size_t)b * B * 3, B, wu, wi, 1);
                else k<LPR, false><<<grid, 256>>>(tri + (size_t)b * B * 3, B, wu, wi, mode);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep && ms < best) best = ms;
        }
        const double bytes = (double)NB * B * (24 + 3.0 * rb + (mode ? 3.0 * rb : 0));
        printf(", \"%s\": {\"us_per_launch\": %.2f, \"bytes_per_triple\": %d, \"TBps\": %.3f}", mname[mode], 1000.0 * best / NB,
               (int)(24 + 3 * rb + (mode ? 3 * rb : 0)), bytes / (best * 1e-3) / 1e12);
    }
    printf("}%s\n", last ? "" : ",");
    hipFree(wu); hipFree(wi); hipFree(tri);
}
int main() {
    printf("{\"what\": \"tools/micro_gather.hip: three random row reads (+ three in-place row writes) per triple, nothing else in the kernel; best of 3 timed repeats of 16 launches\",\n");
    shape<8>("d32_fp32", 10000000, 1000000, false);
    shape<16>("d64_fp32", 10000000, 1000000, false);
    shape<16>("d128_fp16", 50000000, 5000000, true);
    printf("}\n");
    return 0;
}
