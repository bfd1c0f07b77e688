// Shared device-side definitions for libsml_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/sml_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define SML_HID 512   // fc1 width                    (reference model/conv_transfer.py:33)
#define SML_C1 10     // conv1 output channels        (model/conv_transfer.py:26-27)
#define SML_C2 5      // conv2 output channels        (model/conv_transfer.py:29-30)
#define SML_R 32      // padding unit of the per-batch scratch runs (covers 16- and 32-row workgroup tiles)
#define SML_TM 16     // rows of one MFMA row-tile = M of v_mfma_f32_16x16x4_f32

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
// Each image stores, for output tile T (16 columns) and k-step S (16 reduction indices),
// 64 lanes x 4 floats: lane l, element e = W[col = T*16 + (l&15)][red = S*16 + 4*(l>>4) + e]
// so one wave-wide 16-byte load feeds four v_mfma_f32_16x16x4_f32 (MFMA e covers the four
// reduction indices {S*16 + 4g + e, g = 0..3}; the A operand uses the same split).
//   P1  : fc1 forward    Z1 = A1 * W1^T   cols n (512), red k (5d)      W = fc1.weight[n][k]
//   P1B : fc1 backward   dA1 = dZ1 * W1   cols k (5d),  red n (512)     W = fc1.weight[n][k]
//   P2  : fc2 forward    Out = a2 * W2^T  cols j (d),   red n (512)     W = fc2.weight[j][n]
//   P2B : fc2 backward   dA2 = dOut * W2  cols n (512), red j (d)       W = fc2.weight[j][n]
__host__ __device__ constexpr int sml_pk_p1(int d) { return 0; }
__host__ __device__ constexpr int sml_pk_p1b(int d) { return SML_HID * SML_C2 * d; }
__host__ __device__ constexpr int sml_pk_p2(int d) { return 2 * SML_HID * SML_C2 * d; }
__host__ __device__ constexpr int sml_pk_p2b(int d) { return 2 * SML_HID * SML_C2 * d + SML_HID * d; }
__host__ __device__ constexpr int sml_pk_size(int d) { return 2 * SML_HID * SML_C2 * d + 2 * SML_HID * d; }

// element (col, red) -> position in an image with `ksteps` k-steps (of 16) per tile
__host__ __device__ constexpr int pk_pos(int ksteps, int col, int red) {
    return ((((col >> 4) * ksteps + (red >> 4)) * 64) + ((col & 15) + 16 * ((red >> 2) & 3))) * 4 + (red & 3);
}

// conv1/conv2 parameters (95 floats of the 104-float head of a net) <-> compact index 0..94
__host__ __device__ constexpr bool conv_slot_used_host(int off) {
    return (off < 30) || (off >= SML_OFF_C1B && off < SML_OFF_C1B + 10) ||
           (off >= SML_OFF_C2W && off < SML_OFF_C2W + 50) || (off >= SML_OFF_C2B && off < SML_OFF_C2B + 5);
}
__host__ __device__ constexpr int conv_compact(int off) {
    return off < 30 ? off : off < SML_OFF_C2W ? off - 2 : off < SML_OFF_C2B ? off - 4 : off - 6;
}
#define SML_CG 96     // compact conv-gradient vector (95 used)

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
// v_mfma_f32_16x16x4_f32: lane l supplies A[i = l&15][k = l>>4], B[k = l>>4][j = l&15];
// accumulator register q of lane l holds D[row = 4*(l>>4) + q][col = l&15]
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// loss term and d loss / d s_pos, d loss / d s_neg for one pair
// BCE: model/conv_transfer.py:124-126 (means over the batch -> inv_b); BPR: :128-134 (sum)
__device__ __forceinline__ void pair_terms(int kind, float sp, float sn, float inv_b, float& lt, float& dsp,
                                           float& dsn) {
    if (kind == SML_LOSS_BCE) {
        const float gp = sml_sigmoid(sp), gn = sml_sigmoid(sn);
        const float ap = gp + 1e-15f, an = (1.0f - gn) + 1e-15f;
        lt = -(logf(ap) + logf(an)) * inv_b;
        dsp = -inv_b * gp * (1.0f - gp) / ap;
        dsn = inv_b * gn * (1.0f - gn) / an;
    } else {
        const float x = sp - sn;
        // -logsigmoid(x) = max(-x,0) + log1p(exp(-|x|))
        lt = fmaxf(-x, 0.0f) + log1pf(expf(-fabsf(x)));
        const float g = -sml_sigmoid(-x);
        dsp = g;
        dsn = -g;
    }
}


// ---- Adam (torch.optim.Adam single-tensor path; reference model/transfer.py:392-393) ----
#define SML_BETA1 0.9f
#define SML_BETA2 0.999f
#define SML_EPS 1e-8f
// schedule entry for step k (1-based): x = lr / (1 - beta1^k), y = sqrt(1 - beta2^k)
struct SmlSched { float step_size; float bc2_sqrt; };

// one Adam step with gradient g
// Every operation is PINNED (explicit roundings): the same step is inlined into several kernels (row updates, the
// weight-gradient tiles, the theta kernel, the forward's deferred conv step), and with the compiler free to contract
// a * b + c differently from one context to the next those copies part ways by an ulp -- found when the conv parameters'
// step moved from the merged launch into the next forward (round 4): replicas and A/B runs must stay bit-identical whichever
// kernel takes the step.  The sequence pinned is the one the compiler had chosen for the theta kernel all along (first
// moment: multiply, then add; second moment: v*beta2 fused with g * ((1-beta2)*g)) -- the form the G3 / G4 / G12 parity
// margins were measured with; a TR trajectory is chaotic enough that another rounding order moves G12's worst per-batch
// loss error from 3e-5 to 2e-4 (measured with ATen's own order, three roundings in the second moment).
__device__ __forceinline__ void adam_apply(float& p, float& m, float& v, float g, SmlSched s) {
    m = __fadd_rn(m, __fmul_rn(1.0f - SML_BETA1, __fsub_rn(g, m)));                        // exp_avg.lerp_(grad, 1-beta1)
    v = __fmaf_rn(g, __fmul_rn(1.0f - SML_BETA2, g), __fmul_rn(v, SML_BETA2));             // mul_(beta2).addcmul_(g, g, 1-beta2)
    const float denom = __fadd_rn(__fdiv_rn(sqrtf(v), s.bc2_sqrt), SML_EPS);
    p = __fsub_rn(p, __fdiv_rn(__fmul_rn(s.step_size, m), denom));                        // addcdiv_(m, denom, value=-step_size)
}
// gradient + weight_decay * p, pinned for the same reason
__device__ __forceinline__ float adam_wd(float g, float wd, float p) { return __fmaf_rn(wd, p, g); }
// one Adam step with ZERO gradient (a row that was not in the batch), algebraically
//   p -= step_size * m / (sqrt(v)/bc2 + eps)  =  step_size*bc2*m / (sqrt(v) + eps*bc2)
// on the transcendental unit (v_sqrt_f32, v_rcp_f32: ~1 ulp each).  A replayed window is at most
// one epoch long, so the accumulated deviation from the IEEE form stays ~1e-5 of an update that is
// itself ~lr -- invisible next to the 1e-4 parity budget -- at ~1/6 of the instructions.
__device__ __forceinline__ void adam_zero_step(float& p, float& m, float& v, SmlSched s) {
    m = m - (1.0f - SML_BETA1) * m;
    v = v * SML_BETA2;
    const float den = __builtin_amdgcn_sqrtf(v) + SML_EPS * s.bc2_sqrt;
    p = p - (s.step_size * s.bc2_sqrt) * m * __builtin_amdgcn_rcpf(den);
}
// replay the zero-gradient steps (from+1 .. to) a dense Adam would have applied
__device__ __forceinline__ void adam_replay(float& p, float& m, float& v, int from, int to,
                                            const SmlSched* __restrict__ sched) {
    if (from < 0 || (m == 0.0f && v == 0.0f)) return;
#ifdef SML_DBG_NOREPLAY      // measurement hook (wrong results): what the replay loops cost the kernels that carry them
    return;
#endif
    for (int k = from + 1; k <= to; ++k) adam_zero_step(p, m, v, sched[k]);
}
// the same with the most recent SML_SW schedule entries staged in LDS (win[i] = sched[wbase + i]):
// a dependent global load per replayed step would dominate the loop otherwise
#define SML_SW 256
__device__ __forceinline__ void sched_window_load(SmlSched* win, const SmlSched* __restrict__ sched, int upto,
                                                  int tid) {
    const int wbase = upto - SML_SW + 1;
    if (tid < SML_SW) { const int k = wbase + tid; if (k >= 0) win[tid] = sched[k]; }
}
__device__ __forceinline__ void adam_replay_w(float& p, float& m, float& v, int from, int to,
                                              const SmlSched* __restrict__ sched, const SmlSched* win, int upto) {
    // from < 0: the row was never touched.  m = v = 0: every zero-gradient step leaves p, m, v exactly
    // unchanged (m - 0.1*0 = 0, 0.999*0 = 0, p - c*0 = p), so there is nothing to replay.
    if (from < 0 || (m == 0.0f && v == 0.0f)) return;
#ifdef SML_DBG_NOREPLAY
    return;
#endif
    const int wbase = upto - SML_SW + 1;
    for (int k = from + 1; k <= to; ++k) adam_zero_step(p, m, v, k >= wbase ? win[k - wbase] : sched[k]);
}

// four elements of one row at once: ONE walk over the steps (one schedule read per step, four independent chains in flight)
// instead of four walks -- the same operations per element, so the same bits as four adam_replay_w calls.  (An element
// with m = v = 0 needs no special case: a zero-gradient step leaves it exactly unchanged.)
__device__ __forceinline__ void adam_replay_w4(float (&p)[4], float (&m)[4], float (&v)[4], int from, int to,
                                               const SmlSched* __restrict__ sched, const SmlSched* win, int upto) {
    if (from < 0) return;
#ifdef SML_DBG_NOREPLAY
    return;
#endif
    const int wbase = upto - SML_SW + 1;
    for (int k = from + 1; k <= to; ++k) {
        const SmlSched s = k >= wbase ? win[k - wbase] : sched[k];
#pragma unroll
        for (int e = 0; e < 4; ++e) adam_zero_step(p[e], m[e], v[e], s);
    }
}

// ---- closed-form replay (round 6) ----------------------------------------------------------------------------------------
// n pending zero-gradient steps of a row element (last stepped at f, brought to `to` = f + n) are, exactly,
//     m_k = beta1^j m_f,  sqrt(v_k) = s sigma^j  (s = sqrt(v_f), sigma = sqrt(beta2), j = k - f),
//     p_to = p_f - m_f * SUM_j w_j / (s + e_j),   w_j = c_k rho^j,  e_j = eps bc2_k sigma^-j,  c_k = step_size_k bc2_k,  rho = beta1 / sigma
// -- the loop above walks that sum term by term (74 terms x ~64 issue cycles at the end of an MF epoch: 2.5 us of the MF
// forward's 16 and most of k_adam_flush, `tools/bench_steps.py --lib ..noreplay..`).  The e_j of one replay differ from
// their w-weighted mean ebar by a few per cent at most once bc2_k has flattened (k >= SML_RP_K0: under 3 % over the ~30 terms
// that carry the weight), so with t = s + ebar and the weighted central moments mu2, mu3 of the e_j
//     SUM_j w_j / (s + e_j) = (M0 / t) * (1 + mu2 / t^2 - mu3 / t^3 + O((delta / t)^4)),        (delta / t)^4 < 1e-6 for EVERY s >= 0
// (the first-order term vanishes by the choice of ebar).  The raw moments M_q(f, to) = SUM_k c_k E_k^q rho_q^(k-f), rho_q =
// beta1 sigma^-(1+q), come from ONE-dimensional host tables in double precision -- H_q[k0] = SUM_{k >= k0} c_k E_k^q rho_q^(k-k0+1),
// so M_q = H_q[f+1] - rho_q^n H_q[to+1] -- appended to the schedule table (ensure_sched): every workgroup turns them into the
// 256 entries {M0, ebar, mu2 / ebar^2, mu3 / ebar^3, beta1^n, beta2^n} of ITS launch's `to` while its gather is in flight, and
// an element's replay is one LDS read, one square root, one reciprocal and eight multiply-adds whatever n is.
// Steps below SML_RP_K0 (the first 1.4 periods of a run, every golden fixture), rows older than SML_RP_N steps and
// SML_REPLAY_CLOSED=0 keep the loop.  Against a float64 dense Adam the closed form is as close as the loop
// (tests/test_hip_parity.py::test_closed_form_replay_...).
#define SML_RP_N 256
#define SML_RP_K0 1024
struct __attribute__((aligned(16))) SmlReplayEnt { float M0, ebar, nu2, nu3, b1, b2, pad0, pad1; };
// layout behind the schedule's `len` entries: H[len + 1][4] doubles, R[SML_RP_N + 1][4] doubles (rho_q^n), B[SML_RP_N + 1][2] floats (beta1^n, beta2^n)
__device__ __forceinline__ void replay_table_build(SmlReplayEnt* tab, const SmlSched* __restrict__ sched, int len, int to, int tid) {
    if (tid >= SML_RP_N) return;
    const int n = tid + 1, f = to - n;
    SmlReplayEnt e;
    e.M0 = 0.f; e.ebar = 1.f; e.nu2 = 0.f; e.nu3 = 0.f; e.b1 = 1.f; e.b2 = 1.f; e.pad0 = 0.f; e.pad1 = 0.f;
    if (f >= 0) {
        const double* __restrict__ H = reinterpret_cast<const double*>(sched + len);
        const double* __restrict__ R = H + 4 * ((int64_t)len + 1);
        const float* __restrict__ B = reinterpret_cast<const float*>(R + 4 * (SML_RP_N + 1));
        double M[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) M[q] = H[4 * ((int64_t)f + 1) + q] - R[4 * n + q] * H[4 * ((int64_t)to + 1) + q];
        if (M[0] > 0.0 && M[1] > 0.0) {          // (lr = 0: every weight is zero -- the row does not move, its moments still decay)
            const double i0 = 1.0 / M[0];
            const double eb = M[1] * i0, r2 = M[2] * i0, r3 = M[3] * i0;
            const double ie = 1.0 / eb;
            const double mu2 = r2 - eb * eb, mu3 = r3 - 3.0 * eb * r2 + 2.0 * eb * eb * eb;
            e.M0 = (float)M[0]; e.ebar = (float)eb; e.nu2 = (float)(mu2 * ie * ie); e.nu3 = (float)(mu3 * ie * ie * ie);
        }
        e.b1 = B[2 * n]; e.b2 = B[2 * n + 1];
    }
    tab[tid] = e;
}
__device__ __forceinline__ void adam_replay_closed(float& p, float& m, float& v, const SmlReplayEnt& e) {
    const float s = __builtin_amdgcn_sqrtf(v);
    const float x = __builtin_amdgcn_rcpf(s + e.ebar);
    const float y = e.ebar * x;                                      // in (0, 1]
    const float corr = 1.0f + (y * y) * (e.nu2 - e.nu3 * y);
    p = p - ((m * e.M0) * x) * corr;
    m = m * e.b1;
    v = v * e.b2;
}
// replay through the table when it covers the row (tab: this launch's entries in LDS, null: the loop), else the loop
__device__ __forceinline__ void adam_replay_t(float& p, float& m, float& v, int from, int to, const SmlSched* __restrict__ sched,
                                              const SmlSched* win, int upto, const SmlReplayEnt* tab) {
    if (from < 0 || (m == 0.0f && v == 0.0f)) return;
    const int n = to - from;
    if (tab != nullptr && n >= 1 && n <= SML_RP_N) { adam_replay_closed(p, m, v, tab[n - 1]); return; }
    adam_replay_w(p, m, v, from, to, sched, win, upto);
}
__device__ __forceinline__ void adam_replay_t4(float (&p)[4], float (&m)[4], float (&v)[4], int from, int to,
                                               const SmlSched* __restrict__ sched, const SmlSched* win, int upto, const SmlReplayEnt* tab) {
    if (from < 0) return;
    const int n = to - from;
    if (tab != nullptr && n >= 1 && n <= SML_RP_N) {
        const SmlReplayEnt e = tab[n - 1];
#pragma unroll
        for (int q = 0; q < 4; ++q) adam_replay_closed(p[q], m[q], v[q], e);
        return;
    }
    adam_replay_w4(p, m, v, from, to, sched, win, upto);
}

// ---- one-shot exchange over peer mappings: device side ------------------------------------------------------------
// A pusher's data stores are system-scope write-through stores (on their way over xGMI while the kernel still
// computes; nothing left dirty in this XCD's L2 for the release to walk); then fence + barrier + one counter
// increment per destination.  A consumer polls its own counters (one lane per source rank), then acquires.
__device__ __forceinline__ void peer_store(float* p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void peer_store16(float* p, const f32x4& v) {
    // (s_nop: the hazard recogniser does not look inside inline asm -- a VALU instruction that overwrites the data
    // registers of a store wider than 8 bytes in the very next slots corrupts the last dwords: found with fp16 tables,
    // where two of these stores follow each other from recycled registers)
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}
// NO system-scope fences here: __threadfence_system() is a write-back + invalidate of this XCD's whole L2, executed by
// every wave of every pushing / polling workgroup (measured on one GPU: +25 us on the weight-gradient launch, +12 us
// on the Adam launch).  They are not needed: every exchanged byte moves with sc0 sc1 accesses -- write-through stores
// that are acknowledged (vmcnt = 0) once they have reached the system coherence point, loads that bypass the caches --
// and the counters are system-scope atomics.
template <typename P>
__device__ __forceinline__ void peer_signal(const P& p) {          // every thread of the workgroup calls this
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // this thread's pushes are acknowledged
    __syncthreads();
    if (threadIdx.x == 0)        // (uniform index into the kernel-argument array: scalar loads, no private copy)
        for (int q = 0; q < p.world; ++q) __hip_atomic_fetch_add(p.flag[q], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
template <typename P>
__device__ __forceinline__ void peer_wait(const P& p) {            // every thread of the workgroup calls this
    if ((int)threadIdx.x < p.world) {
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(p.flag0 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < p.expect) {
            if (wall_clock64() - t0 > p.timeout) { atomicAdd(p.err, 1); break; }
            __builtin_amdgcn_s_sleep(8);
        }
    }
    __syncthreads();             // (the slots are then read with sc0 sc1 loads, or by a later kernel)
}
__device__ __forceinline__ float peer_load(const float* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// eight 16-byte system-scope (cache-bypassing) loads in flight together, complete when the call returns: ONE asm
// statement holds the loads and their s_waitcnt, so no compiler-scheduled instruction can read a destination early
__device__ __forceinline__ void peer_load16x8(f32x4 (&v)[8], const float* const (&p)[8]) {
    asm volatile(
        "global_load_dwordx4 %0, %8, off sc0 sc1\n"
        "global_load_dwordx4 %1, %9, off sc0 sc1\n"
        "global_load_dwordx4 %2, %10, off sc0 sc1\n"
        "global_load_dwordx4 %3, %11, off sc0 sc1\n"
        "global_load_dwordx4 %4, %12, off sc0 sc1\n"
        "global_load_dwordx4 %5, %13, off sc0 sc1\n"
        "global_load_dwordx4 %6, %14, off sc0 sc1\n"
        "global_load_dwordx4 %7, %15, off sc0 sc1\n"
        "s_waitcnt vmcnt(0)"
        : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
        : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7])
        : "memory");
}
// eight 4-byte agent-scope (this device's L2s bypassed) loads in flight together, complete when the call returns
__device__ __forceinline__ void agent_load4x8(float (&v)[8], const float* const (&p)[8]) {
    asm volatile(
        "global_load_dword %0, %8, off sc1\n"
        "global_load_dword %1, %9, off sc1\n"
        "global_load_dword %2, %10, off sc1\n"
        "global_load_dword %3, %11, off sc1\n"
        "global_load_dword %4, %12, off sc1\n"
        "global_load_dword %5, %13, off sc1\n"
        "global_load_dword %6, %14, off sc1\n"
        "global_load_dword %7, %15, off sc1\n"
        "s_waitcnt vmcnt(0)"
        : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
        : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7])
        : "memory");
}
__device__ __forceinline__ void peer_load16x2(f32x4 (&v)[2], const float* p0, const float* p1) {
    asm volatile(
        "global_load_dwordx4 %0, %2, off sc0 sc1\n"
        "global_load_dwordx4 %1, %3, off sc0 sc1\n"
        "s_waitcnt vmcnt(0)"
        : "=&v"(v[0]), "=&v"(v[1])
        : "v"(p0), "v"(p1)
        : "memory");
}
__device__ __forceinline__ void peer_load16x4(f32x4 (&v)[4], const float* const (&p)[4]) {
    asm volatile(
        "global_load_dwordx4 %0, %4, off sc0 sc1\n"
        "global_load_dwordx4 %1, %5, off sc0 sc1\n"
        "global_load_dwordx4 %2, %6, off sc0 sc1\n"
        "global_load_dwordx4 %3, %7, off sc0 sc1\n"
        "s_waitcnt vmcnt(0)"
        : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3])
        : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3])
        : "memory");
}
