// VALU issue cost on gfx950: clocks a SIMD spends per wavefront-wide v_fma_f32, v_pk_fma_f32 (two fp32 FMAs per lane),
// DPP-modified v_add_f32 and v_cndmask_b32, at 1 / 2 / 4 wavefronts per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/_bin/valu_rate_probe tools/valu_rate_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, int iters, float s) {
    float a0 = threadIdx.x, a1 = 1.f, a2 = 2.f, a3 = 3.f, a4 = 4.f, a5 = 5.f, a6 = 6.f, a7 = 7.f;
    f32x2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, ss = {s, s * 0.5f};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (MODE == 0) {
                asm volatile("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s));
            } else if (MODE == 1) {
                asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(ss));
            } else if (MODE == 2) {
                asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n s_nop 1\n v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n s_nop 1\n"
                             "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n s_nop 1\n v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n s_nop 1"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
            } else {
                asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(s) : "vcc");
            }
        }
    }
    if (a0 + a1 + a2 + a3 + p0[0] + p1[1] + p2[0] + p3[1] == 123.f) out[0] = 1.f;
}
template <int MODE> void run(const char* name, float* out, int waves_per_simd) {
    const int iters = 4000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE><<<256, waves_per_simd * 256>>>(out, 10, 1.0001f);
    (void)hipEventRecord(e0);
    k<MODE><<<256, waves_per_simd * 256>>>(out, iters, 1.0001f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double clks = ms * 1e-3 * 2.4e9, instr_per_simd = (double)waves_per_simd * iters * 64;
    printf("{\"op\": \"%s\", \"waves_per_simd\": %d, \"clk_per_instr_per_simd\": %.2f}\n", name, waves_per_simd, clks / instr_per_simd);
}
int main() {
    float* out; (void)hipMalloc(&out, 64);
    for (int w : {1, 2, 4}) {
        run<0>("v_fma_f32", out, w);
        run<1>("v_pk_fma_f32", out, w);
        run<2>("v_add_f32_dpp (+ s_nop 1)", out, w);
        run<3>("v_cndmask_b32", out, w);
    }
    return 0;
}
