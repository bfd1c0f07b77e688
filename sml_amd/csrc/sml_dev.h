// Shared device-side definitions for libsml_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define SML_HID 512   // fc1 width                    (reference model/conv_transfer.py:33)
#define SML_C1 10     // conv1 output channels        (model/conv_transfer.py:26-27)
#define SML_C2 5      // conv2 output channels        (model/conv_transfer.py:29-30)
#define SML_R 32      // rows per workgroup tile = M of v_mfma_f32_32x32x2_f32

// ---- flat layout of one net inside theta (floats; every tensor 16-byte aligned) ----
#define SML_OFF_C1W 0      // [10][3]
#define SML_OFF_C1B 32     // [10]
#define SML_OFF_C2W 44     // [5][10]
#define SML_OFF_C2B 96     // [5]
#define SML_OFF_F1W 104    // [512][5d]
__host__ __device__ constexpr int sml_off_f1b(int d) { return SML_OFF_F1W + SML_HID * SML_C2 * d; }
__host__ __device__ constexpr int sml_off_f2w(int d) { return sml_off_f1b(d) + SML_HID; }   // [d][512]
__host__ __device__ constexpr int sml_off_f2b(int d) { return sml_off_f2w(d) + SML_HID * d; }
__host__ __device__ constexpr int sml_net_size(int d) { return sml_off_f2b(d) + d; }

// ---- MFMA operand images ("packed" weights), per net --------------------------------
// Each image stores, for output tile T (32 columns) and k-step S (8 reduction indices),
// 64 lanes x 4 floats: lane l, element e = W[col = T*32 + (l&31)][red = S*8 + 4*(l>>5) + e]
// so one wave-wide 16-byte load feeds four v_mfma_f32_32x32x2_f32 (k pairs {e, 4+e}).
//   P1  : fc1 forward    Z1 = A1 * W1^T   cols n (512), red k (5d)      W = fc1.weight[n][k]
//   P1B : fc1 backward   dA1 = dZ1 * W1   cols k (5d),  red n (512)     W = fc1.weight[n][k]
//   P2  : fc2 forward    Out = a2 * W2^T  cols j (d),   red n (512)     W = fc2.weight[j][n]
//   P2B : fc2 backward   dA2 = dOut * W2  cols n (512), red j (d)       W = fc2.weight[j][n]
__host__ __device__ constexpr int sml_pk_p1(int d) { return 0; }
__host__ __device__ constexpr int sml_pk_p1b(int d) { return SML_HID * SML_C2 * d; }
__host__ __device__ constexpr int sml_pk_p2(int d) { return 2 * SML_HID * SML_C2 * d; }
__host__ __device__ constexpr int sml_pk_p2b(int d) { return 2 * SML_HID * SML_C2 * d + SML_HID * d; }
__host__ __device__ constexpr int sml_pk_size(int d) { return 2 * SML_HID * SML_C2 * d + 2 * SML_HID * d; }

__device__ __forceinline__ int pk_index(int ksteps, int tile, int kstep, int lane, int e) {
    return ((tile * ksteps + kstep) * 64 + lane) * 4 + e;
}
// element (col, red) -> position in an image with `ksteps` k-steps per tile
__device__ __forceinline__ int pk_pos(int ksteps, int col, int red) {
    return pk_index(ksteps, col >> 5, red >> 3, (col & 31) + 32 * ((red >> 2) & 1), red & 3);
}

// sigmoid on the transcendental unit: v_exp_f32 (2^x) and v_rcp_f32, each ~1 ulp -- the result is
// within ~3e-7 relative of the correctly rounded value, far inside the 1e-4 parity budget, at
// ~5 VALU issues instead of ~35 for expf() plus an IEEE divide.
__device__ __forceinline__ float sml_sigmoid(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x));
}
// Gelu of the reference: x * sigmoid(1.702 x)  (model/conv_transfer.py:9-10)
__device__ __forceinline__ float sml_gelu(float x) { return x * sml_sigmoid(1.702f * x); }
__device__ __forceinline__ float sml_gelu_grad(float x) {
    const float s = sml_sigmoid(1.702f * x);
    return s + 1.702f * x * s * (1.0f - s);
}

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
// accumulator register q of lane l holds D[row][col = l&31]:
__device__ __forceinline__ int mfma32_row(int q, int lane) { return (q & 3) + 8 * (q >> 2) + 4 * (lane >> 5); }

// ---- Adam (torch.optim.Adam single-tensor path; reference model/transfer.py:392-393) ----
#define SML_BETA1 0.9f
#define SML_BETA2 0.999f
#define SML_EPS 1e-8f
// schedule entry for step k (1-based): x = lr / (1 - beta1^k), y = sqrt(1 - beta2^k)
struct SmlSched { float step_size; float bc2_sqrt; };

// one Adam step with gradient g
__device__ __forceinline__ void adam_apply(float& p, float& m, float& v, float g, SmlSched s) {
    m = m + (1.0f - SML_BETA1) * (g - m);                 // exp_avg.lerp_(grad, 1-beta1)
    v = v * SML_BETA2 + (1.0f - SML_BETA2) * g * g;       // mul_(beta2).addcmul_(g, g, 1-beta2)
    const float denom = sqrtf(v) / s.bc2_sqrt + SML_EPS;
    p = p + (-s.step_size * m) / denom;                   // addcdiv_(m, denom, value=-step_size)
}
// replay the zero-gradient steps (from+1 .. to) a dense Adam would have applied
__device__ __forceinline__ void adam_replay(float& p, float& m, float& v, int from, int to,
                                            const SmlSched* __restrict__ sched) {
    for (int k = from + 1; k <= to; ++k) adam_apply(p, m, v, 0.0f, sched[k]);
}
