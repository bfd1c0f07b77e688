// L1-hit load rate per CU on gfx950: how many clocks a wavefront-wide global_load_dwordx4 / dwordx2 / dword costs the CU's
// vector-memory path when every line is L1-resident, with all lanes active and with 8 / 16 of 64 lanes active.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/_bin/l1_rate_probe tools/l1_rate_probe.hip ; run: tools/_bin/l1_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// MODE 0: dwordx4 all lanes, 8 rows of 128 B per wave-load (8 lanes per row); 1: dwordx4, lanes & 7 == 0 only (8 active);
// 2: dwordx2 all lanes; 3: dword all lanes; 4: dwordx4 all lanes, ALL lanes of a 8-lane group the same 16 B (broadcast);
// 5: dwordx4, 16 lanes active (lanes < 16)
template <int MODE>
__global__ __launch_bounds__(1024) void k_probe(const float* __restrict__ buf, int iters, float* __restrict__ out, long long* __restrict__ clk) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // every wavefront cycles through its own 4 KB (32 lines): 16 waves * 4 KB = 64 KB > L1?  keep it 1 KB per wave = 16 KB per CU
    const char* base = reinterpret_cast<const char*>(buf) + (size_t)blockIdx.x * 65536 + wv * 1024;
    float acc = 0.f;
    bool on = true;
    int off;
    if (MODE == 0) off = lane * 16;
    else if (MODE == 1) { off = lane * 16; on = (lane & 7) == 0; }
    else if (MODE == 2) off = lane * 8;
    else if (MODE == 3) off = lane * 4;
    else if (MODE == 4) off = (lane >> 3) * 128;
    else { off = lane * 16; on = lane < 16; }
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int o = (off + j * 128) & 1023;
            if (on) {
                if (MODE == 2) { f32x2 v; asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(v) : "v"(base + o) : "memory"); asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); acc += 0.f; (void)v; }
            }
        }
        if (MODE != 2) {
            f32x4 v[8]; float w[8];
            if (on) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int o = (off + j * 128) & 1023;
                    if (MODE == 3) asm volatile("global_load_dword %0, %1, off" : "=v"(w[j]) : "v"(base + o) : "memory");
                    else asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[j]) : "v"(base + o) : "memory");
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int j = 0; j < 8; ++j) acc += MODE == 3 ? w[j] : v[j][0];
            }
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    const long long t1 = clock64();
    if (acc == 12345.f) out[0] = acc;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}
template <int MODE> void run(const char* name, const float* buf, float* out, long long* clk, int waves) {
    const int iters = 2000, blocks = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k_probe<MODE><<<blocks, waves * 64>>>(buf, 10, out, clk);
    hipEventRecord(e0);
    k_probe<MODE><<<blocks, waves * 64>>>(buf, iters, out, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // wave-loads per CU: waves * iters * 8; clocks at 2.4 GHz
    const double clks = ms * 1e-3 * 2.4e9, loads = (double)waves * iters * 8;
    printf("{\"mode\": \"%s\", \"waves_per_cu\": %d, \"us\": %.1f, \"clk_per_wave_load\": %.2f}\n", name, waves, ms * 1000, clks / loads);
}
int main() {
    float *buf, *out; long long* clk;
    hipMalloc(&buf, 256 * 65536 + 4096); hipMemset(buf, 0, 256 * 65536 + 4096); hipMalloc(&out, 64); hipMalloc(&clk, 256 * 8);
    for (int waves : {4, 8, 16}) {
        run<0>("dwordx4 all lanes (8 lines per load)", buf, out, clk, waves);
        run<4>("dwordx4 all lanes, 8 lanes share 16 B", buf, out, clk, waves);
        run<1>("dwordx4, 8 of 64 lanes active", buf, out, clk, waves);
        run<5>("dwordx4, lanes 0-15 active", buf, out, clk, waves);
        run<2>("dwordx2 all lanes", buf, out, clk, waves);
        run<3>("dword all lanes", buf, out, clk, waves);
    }
    return 0;
}
