// Embedding-row kernels for gfx950: pair loss (BCE/BPR), the bare fused embed+loss
// gradient pass, deterministic segmented SGD / lazy-Adam row updates, Adam flush,
// MFbasemode.forward and the evaluation rank kernel.
//
// Row access pattern: a row of d elements is owned by LPR = d*sizeof(T)/16 adjacent
// lanes, 16 bytes per lane, so every gather is a set of full 128/256-byte line reads
// and a wavefront carries 64/LPR rows at once; dot products finish with xor-shuffles
// inside the lane group (no LDS).
#include <hip/hip_fp16.h>
#include "sml_dev.h"
#include "sml_kernels.h"
#include "../../include/sml_hip.h"

#ifndef SML_NT
#define SML_NT 7    // bit 0: user-row loads, bit 1: user-row stores, bit 2: item-row stores of the in-place pass are nontemporal
#endif
namespace {

template <typename T> struct RowVec;
template <> struct RowVec<float> {
    static constexpr int VEC = 4;
    __device__ static void load(const float* p, float (&x)[4]) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(p);
        x[0] = v[0]; x[1] = v[1]; x[2] = v[2]; x[3] = v[3];
    }
    __device__ static void store(float* p, const float (&x)[4]) {
        f32x4 v; v[0] = x[0]; v[1] = x[1]; v[2] = x[2]; v[3] = x[3];
        *reinterpret_cast<f32x4*>(p) = v;
    }
    // streaming forms: a row that is touched once per batch should not displace re-used lines
    __device__ static void load_nt(const float* p, float (&x)[4]) {
        const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
        x[0] = v[0]; x[1] = v[1]; x[2] = v[2]; x[3] = v[3];
    }
    __device__ static void store_nt(float* p, const float (&x)[4]) {
        f32x4 v; v[0] = x[0]; v[1] = x[1]; v[2] = x[2]; v[3] = x[3];
        __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
    }
    // the lane's 16 bytes as they arrived through a cache-bypassing load (peer_load16x2)
    __device__ static void decode(const f32x4& v, float (&x)[4]) { x[0] = v[0]; x[1] = v[1]; x[2] = v[2]; x[3] = v[3]; }
};
template <> struct RowVec<__half> {
    static constexpr int VEC = 8;
    __device__ static void load(const __half* p, float (&x)[8]) {
        const uint4 raw = *reinterpret_cast<const uint4*>(p);
        const __half2* h = reinterpret_cast<const __half2*>(&raw);
#pragma unroll
        for (int i = 0; i < 4; ++i) { const float2 f = __half22float2(h[i]); x[2 * i] = f.x; x[2 * i + 1] = f.y; }
    }
    __device__ static void store(__half* p, const float (&x)[8]) {
        uint4 raw;
        __half2* h = reinterpret_cast<__half2*>(&raw);
#pragma unroll
        for (int i = 0; i < 4; ++i) h[i] = __floats2half2_rn(x[2 * i], x[2 * i + 1]);
        *reinterpret_cast<uint4*>(p) = raw;
    }
    __device__ static void load_nt(const __half* p, float (&x)[8]) { load(p, x); }
    __device__ static void store_nt(__half* p, const float (&x)[8]) { store(p, x); }
    __device__ static void decode(const f32x4& v, float (&x)[8]) {
        const __half2* h = reinterpret_cast<const __half2*>(&v);
#pragma unroll
        for (int i = 0; i < 4; ++i) { const float2 f = __half22float2(h[i]); x[2 * i] = f.x; x[2 * i + 1] = f.y; }
    }
};

template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int off = LPR / 2; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// The evaluation's lane-group sums on DPP (data-parallel primitives: the cross-lane move rides in the VALU
// instruction) instead of ds_bpermute shuffles.  rocprofv3 counters put the rank kernels at two thirds VALU-bound
// (293 M vector instructions per 75k x 1001 evaluation, 31 per candidate and lane group): so the candidates are
// reduced EIGHT AT A TIME in a transposing butterfly -- every lane brings its partial dot product of eight
// candidates, each exchange step halves the candidates a lane still carries (stride LPR/2, LPR/4, LPR/8), and the
// lane ends up with the complete score of ONE candidate: 25 cross-lane instructions per eight candidates instead of
// 40, one compare per wavefront step instead of eight.  The pairing order is the plain xor butterfly's (stride LPR/2
// down to 1) for every candidate, so these sums equal group_sum<LPR>() bit for bit: the positive's score, the
// remainder candidates and the eight-at-a-time path all round alike (scores are compared exactly).
template <int CTRL, int BANK_MASK>
__device__ __forceinline__ float dpp_f(float old, float v) {
    // (full-mask permutations: bound_ctrl set, so `old` is never read and the move folds into the consuming VALU op)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, 0xf, BANK_MASK,
                                                                 BANK_MASK == 0xf));
}
// the value lane ^ S holds (S = 1, 2, 4, 8: inside a 16-lane DPP row)
template <int S>
__device__ __forceinline__ float lane_xor(float v) {
    if constexpr (S == 1) return dpp_f<0xB1, 0xf>(0.0f, v);               // quad_perm [1,0,3,2]
    else if constexpr (S == 2) return dpp_f<0x4E, 0xf>(0.0f, v);          // quad_perm [2,3,0,1]
    else if constexpr (S == 8) return dpp_f<0x128, 0xf>(0.0f, v);         // row_ror:8
    else {
        static_assert(S == 4, "stride");
        const float t = dpp_f<0x104, 0x5>(0.0f, v);                       // row_shl:4 -> lanes 0-3, 8-11 of a row take lane + 4
        return dpp_f<0x114, 0xa>(t, v);                                   // row_shr:4 -> lanes 4-7, 12-15 take lane - 4
    }
}
template <int LPR>
__device__ __forceinline__ float eval_group_sum(float v) {              // == group_sum<LPR>(v), on DPP
    if constexpr (LPR >= 32) v += __shfl_xor(v, 16, 64);
    if constexpr (LPR >= 16) v += lane_xor<8>(v);
    if constexpr (LPR >= 8) v += lane_xor<4>(v);
    v += lane_xor<2>(v);
    v += lane_xor<1>(v);
    return v;
}
// one exchange step of the transposing butterfly: N candidates in, N/2 out; lanes with bit S of `sub` keep the upper half
template <int S, int N>
__device__ __forceinline__ void butterfly_split(float (&a)[8], int sub) {
    const bool hi = (sub & S) != 0;
#pragma unroll
    for (int i = 0; i < N / 2; ++i) {
        if constexpr (S == 4) {
            // partner's share of the candidate THIS lane keeps, in two bank-masked moves; then one add
            float t = dpp_f<0x104, 0x5>(0.0f, a[i]);                      // low lanes (keep i): partner's a[i]
            t = dpp_f<0x114, 0xa>(t, a[i + N / 2]);                       // high lanes (keep i + N/2): partner's a[i + N/2]
            a[i] = (hi ? a[i + N / 2] : a[i]) + t;
        } else {
            const float x = a[i] + lane_xor<S>(a[i]), y = a[i + N / 2] + lane_xor<S>(a[i + N / 2]);
            a[i] = hi ? y : x;
        }
    }
}
// scores of eight candidates from the lanes' partial dot products a[0..7]; returns the complete score of candidate
// eval8_index<LPR>(sub) (lanes whose `sub` differ only below LPR/8 hold the same candidate)
template <int LPR>
__device__ __forceinline__ float eval8_reduce(float (&a)[8], int sub) {
    static_assert(LPR == 8 || LPR == 16, "eight-at-a-time path: d = 32 or 64");
    butterfly_split<LPR / 2, 8>(a, sub);
    butterfly_split<LPR / 4, 4>(a, sub);
    butterfly_split<LPR / 8, 2>(a, sub);
    float v = a[0];
    if constexpr (LPR == 16) v += lane_xor<1>(v);
    return v;
}
template <int LPR>
__device__ __forceinline__ int eval8_index(int sub) {
    return ((sub & (LPR / 2)) ? 4 : 0) + ((sub & (LPR / 4)) ? 2 : 0) + ((sub & (LPR / 8)) ? 1 : 0);
}

// 4-element partial dot product with a pinned operation order (one multiply, three fused multiply-adds):
// the evaluation compares scores exactly, so every kernel variant must round them identically
__device__ __forceinline__ float dot4(const float (&a)[4], const float (&b)[4]) {
    return __fmaf_rn(a[3], b[3], __fmaf_rn(a[2], b[2], __fmaf_rn(a[1], b[1], __fmul_rn(a[0], b[0]))));
}

// deterministic 256-thread block sum; result valid in thread 0
__device__ __forceinline__ float block_sum256(float v, float* sh4) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((threadIdx.x & 63) == 0) sh4[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh4[0] + sh4[1] + sh4[2] + sh4[3];
}

// one wavefront per batch: lanes stride over the batch's partials, fixed-order tree (deterministic)
// --need_adaptive (reference model/transfer.py:490-499): loss += sum over the batch's UNIQUE users of
// beta * count_u / ||w_u|| (detached) * ||w_u||^2 -- per occurrence that is beta * ||w_u|| of loss and
// 2 * beta * w_u / ||w_u|| of gradient.  Runs after the backward: x_hat of user slot t is row 1 of xin[t] (the row as the
// forward saw it, pending Adam steps replayed), its gradient row is dx[t].  One workgroup, fixed order: deterministic.
template <int D>
__global__ __launch_bounds__(1024) void k_adaptive_users(const float* __restrict__ xin, float* __restrict__ dx, int B, float beta,
                                                         float* __restrict__ loss_slot) {
    constexpr int LPR = D / 4;
    __shared__ float part[16];
    const int grp = threadIdx.x / LPR, sub = threadIdx.x % LPR;
    float acc = 0.0f;
    for (int t = grp; t < B; t += 1024 / LPR) {
        float x[4], g[4];
        RowVec<float>::load(xin + ((int64_t)t * 3 + 1) * D + sub * 4, x);
        const float nrm = sqrtf(group_sum<LPR>(x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3]));
        RowVec<float>::load(dx + (int64_t)t * D + sub * 4, g);
#pragma unroll
        for (int e = 0; e < 4; ++e) g[e] += 2.0f * beta * x[e] / nrm;
        RowVec<float>::store(dx + (int64_t)t * D + sub * 4, g);
        if (sub == 0) acc += beta * nrm;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) { float s2 = 0.0f; for (int w = 0; w < 16; ++w) s2 += part[w]; *loss_slot += s2; }
}

// (round 5: 256 threads per batch, eight loads in flight per thread.  One wavefront walking a bare batch's 8,192 partials with one
// dependent load per trip took 49 us per epoch -- 3 us per 262,144-triple batch of the a3 step's END-TO-END time.  Fixed order:
// thread t adds elements t, t + 256, ... in ascending order, lanes by the xor butterfly, the four wavefronts in index order.)
__global__ __launch_bounds__(256) void k_loss_finalize(const float* __restrict__ part, int n_batches, int stride,
                                                      float* __restrict__ out) {
    __shared__ float sh4[4];
    const int b = blockIdx.x;
    if (b >= n_batches) return;
    const float* p = part + (int64_t)b * stride;
    float s = 0.0f;
    int i = threadIdx.x;
    for (; i + 7 * 256 < stride; i += 8 * 256) {
        float x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = p[i + u * 256];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += x[u];
    }
    for (; i < stride; i += 256) s += p[i];
    const float tot = block_sum256(s, sh4);
    if (threadIdx.x == 0) out[b] = tot;
}

// ------------------------------------------------------------------------------------
// bare fused embed + loss gradient pass: gather u,i,j rows, two dot
// products, loss, per-occurrence gradient rows.  model/baseline.py:188-201 (BCE),
// model/MF.py:139-144 without biases (BPR).
// ------------------------------------------------------------------------------------
// LAZY (the baselines' dense-Adam form, model/baseline.py:343-361): the three rows first take the
// zero-gradient Adam steps a dense optimiser would have applied since they were last touched (replayed
// in registers, nothing written here), every occurrence emits its gradient row, and k_run_update<Adam>
// finishes the step.
template <int D, typename T, bool LAZY, bool SH = false>
__global__ __launch_bounds__(256) void k_bare_grad(const int64_t* __restrict__ p_tri, const uint8_t* __restrict__ p_uniq, int p_B, SmlBareArgs a) {
    // (round 6: the first round trip's operands -- the triples, the "occurs once" marks, the batch length -- are leading scalar
    // parameters: preloaded into SGPRs with the wavefront (sml_amd/build.py), not fetched with the argument segment)
    constexpr int VEC = RowVec<T>::VEC;
    constexpr int LPR = D / VEC;
    __shared__ float sh4[4];
    __shared__ SmlSched swin[LAZY ? SML_SW : 1];
    if constexpr (LAZY) {
        sched_window_load(swin, a.sched, a.cur_step - 1, threadIdx.x);
        __syncthreads();
    }
    const int gid = blockIdx.x * 256 + threadIdx.x;
    const int t = gid / LPR, sub = gid % LPR;
    float contrib = 0.0f;
    if (t < p_B) {
        const int64_t iu = p_tri[(int64_t)t * 3], ii = p_tri[(int64_t)t * 3 + 1], in = p_tri[(int64_t)t * 3 + 2];
        // (the "row occurs once" marks ride in the FIRST round trip, with the triple: fetched where they are used they were a
        // third dependent trip ahead of the stores)
        uint8_t mk_u = 0, mk_i = 0, mk_n = 0;
        if (!LAZY && p_uniq != nullptr) { mk_u = p_uniq[t]; mk_i = p_uniq[p_B + t]; mk_n = p_uniq[2 * p_B + t]; }
        __builtin_amdgcn_sched_barrier(0);
        float u[VEC], it[VEC], ng[VEC];
        if (SML_NT & 1) RowVec<T>::load_nt(reinterpret_cast<const T*>(a.w_user) + iu * D + sub * VEC, u);
        else RowVec<T>::load(reinterpret_cast<const T*>(a.w_user) + iu * D + sub * VEC, u);
        // (item-sharded form: head rows from the local replica, tail rows from their owner's shard over the peer mapping)
        int qi = -1, qn = -1;                 // owners of the two item rows (-1: head / not sharded)
        auto item_row = [&](int64_t row, int& q) -> const T* {
            if constexpr (SH) {
                if (row >= a.head_rows) {
                    const int64_t r = row - a.head_rows;
                    q = (int)(r / a.shard_rows);
                    return reinterpret_cast<const T*>(a.shard_tab[q]) + (r - (int64_t)q * a.shard_rows) * D;
                }
            }
            return reinterpret_cast<const T*>(a.w_item) + row * D;
        };
        if constexpr (SH) {
            // Another rank REWRITES the rows this reads, every batch, from another device.  Its update kernel has ended
            // (and released: its L2 is written back) before its "done" signal is pushed, so the new row is in its memory;
            // what could still be stale is a line in THIS device's L2 from the previous batch.  System-scope
            // (sc0 sc1) loads never take such a line: both item rows are fetched past the caches, head rows included
            // (they are local; one code path).
            f32x4 raw[2];
            peer_load16x2(raw, reinterpret_cast<const float*>(item_row(ii, qi) + sub * VEC), reinterpret_cast<const float*>(item_row(in, qn) + sub * VEC));
            RowVec<T>::decode(raw[0], it);
            RowVec<T>::decode(raw[1], ng);
        } else {
            RowVec<T>::load(item_row(ii, qi) + sub * VEC, it);
            RowVec<T>::load(item_row(in, qn) + sub * VEC, ng);
        }
        if constexpr (LAZY) {
            static_assert(!LAZY || VEC == 4, "lazy Adam runs on fp32 tables");
            float m[3][4], v[3][4];
            RowVec<float>::load(a.m_user + iu * D + sub * 4, m[0]); RowVec<float>::load(a.v_user + iu * D + sub * 4, v[0]);
            RowVec<float>::load(a.m_item + ii * D + sub * 4, m[1]); RowVec<float>::load(a.v_item + ii * D + sub * 4, v[1]);
            RowVec<float>::load(a.m_item + in * D + sub * 4, m[2]); RowVec<float>::load(a.v_item + in * D + sub * 4, v[2]);
            const int fu = a.last_user[iu], fi = a.last_item[ii], fn = a.last_item[in];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                adam_replay_w(u[k], m[0][k], v[0][k], fu, a.cur_step - 1, a.sched, swin, a.cur_step - 1);
                adam_replay_w(it[k], m[1][k], v[1][k], fi, a.cur_step - 1, a.sched, swin, a.cur_step - 1);
                adam_replay_w(ng[k], m[2][k], v[2][k], fn, a.cur_step - 1, a.sched, swin, a.cur_step - 1);
            }
            if (a.xrep != nullptr) {          // the row update continues from these copies: no second replay
                const int64_t o[3] = {(int64_t)t * D + sub * 4, (int64_t)(p_B + t) * D + sub * 4, (int64_t)(2 * p_B + t) * D + sub * 4};
                RowVec<float>::store(a.xrep + o[0], reinterpret_cast<const float(&)[4]>(u[0]));
                RowVec<float>::store(a.xrep + o[1], reinterpret_cast<const float(&)[4]>(it[0]));
                RowVec<float>::store(a.xrep + o[2], reinterpret_cast<const float(&)[4]>(ng[0]));
#pragma unroll
                for (int r3 = 0; r3 < 3; ++r3) { RowVec<float>::store(a.mrep + o[r3], m[r3]); RowVec<float>::store(a.vrep + o[r3], v[r3]); }
            }
        }
        // The element-wise stretches on PACKED fp32 (v_pk_mul_f32 / v_pk_fma_f32: two elements of a row per lane and
        // instruction; a plain wavefront-wide VALU instruction costs its SIMD 4.1 clocks, a packed one 4.9 --
        // tools/valu_rate_probe.hip): element pairs (k, k + 1) ride together, every element sees the same operations as before.
        f32x2 sp2 = {0.f, 0.f}, sn2 = {0.f, 0.f}, squ2 = {0.f, 0.f}, sqi2 = {0.f, 0.f};
#pragma unroll
        for (int k = 0; k < VEC; k += 2) {
            const f32x2 uu = {u[k], u[k + 1]}, ii2 = {it[k], it[k + 1]}, nn = {ng[k], ng[k + 1]};
            sp2 += uu * ii2; sn2 += uu * nn;
            squ2 += uu * uu; sqi2 += ii2 * ii2 + nn * nn;
        }
        float sp = sp2[0] + sp2[1], sn = sn2[0] + sn2[1];
        const float sq_u = squ2[0] + squ2[1], sq_i = sqi2[0] + sqi2[1];
        sp = group_sum<LPR>(sp); sn = group_sum<LPR>(sn);
        float lt, dsp, dsn;
        // The pair terms on the transcendental unit (v_log_f32 / v_exp_f32 / v_rcp_f32, ~1 ulp each) instead of logf / log1pf /
        // IEEE divides: every one of a triple's D/4 lanes runs this scalar stretch, and the pass is close enough to the memory
        // system's pace that its instruction count shows (round 5, d = 32 uniform: 40.7 -> 39.0 us per 262,144 triples).  The loss
        // TERM moves by ~1e-7 absolute, the coefficients by an ulp: far inside the oracle tolerances of every bare-step test.
        // (BCE: a mean over the GLOBAL batch -> inv_b; the BPR sum ignores it.)
        {
            const float inv_b = a.scale / (float)p_B;
            if (a.kind == SML_LOSS_BCE) {
                const float gp = sml_sigmoid(sp), gn = sml_sigmoid(sn);
                const float ap = gp + 1e-15f, an = (1.0f - gn) + 1e-15f;
                lt = -(__logf(ap) + __logf(an)) * inv_b;
                dsp = -inv_b * gp * (1.0f - gp) * __builtin_amdgcn_rcpf(ap);
                dsn = inv_b * gn * (1.0f - gn) * __builtin_amdgcn_rcpf(an);
            } else {
                const float x = sp - sn;
                lt = fmaxf(-x, 0.0f) + __logf(1.0f + __expf(-fabsf(x)));
                dsp = -sml_sigmoid(-x);
                dsn = -dsp;
            }
        }
        // gradient rows.  An occurrence whose row appears ONCE in this batch is read by nobody else in
        // the batch, so its synchronous-SGD update is applied in place right here (exact); the others
        // hand their gradient row to the segmented update.
        float gx[VEC], gy[VEC], gz[VEC];
        {
            const f32x2 dsp2 = {dsp, dsp}, dsn2 = {dsn, dsn}, lu2 = {a.lam_user, a.lam_user}, li2 = {a.lam_item, a.lam_item};
#pragma unroll
            for (int e = 0; e < VEC; e += 2) {
                const f32x2 uu = {u[e], u[e + 1]}, ii2 = {it[e], it[e + 1]}, nn = {ng[e], ng[e + 1]};
                const f32x2 x = dsp2 * ii2 + dsn2 * nn + lu2 * uu, y = dsp2 * uu + li2 * ii2, z = dsn2 * uu + li2 * nn;
                gx[e] = x[0]; gx[e + 1] = x[1]; gy[e] = y[0]; gy[e + 1] = y[1]; gz[e] = z[0]; gz[e + 1] = z[1];
            }
        }
        const bool one_u = mk_u != 0, one_i = mk_i != 0, one_n = mk_n != 0;
        auto emit = [&](bool in_place, T* wrow, const float (&row)[VEC], const float (&g)[VEC], float* dxrow, bool nt) {
            if (in_place) {
                float nw[VEC];
                const f32x2 nlr = {-a.lr, -a.lr};
#pragma unroll
                for (int e = 0; e < VEC; e += 2) {
                    const f32x2 r2 = {row[e], row[e + 1]}, g2 = {g[e], g[e + 1]};
                    const f32x2 w2 = nlr * g2 + r2;
                    nw[e] = w2[0]; nw[e + 1] = w2[1];
                }
                if (nt) RowVec<T>::store_nt(wrow, nw); else RowVec<T>::store(wrow, nw);
            } else {
#pragma unroll
                for (int h = 0; h < VEC / 4; ++h)
                    RowVec<float>::store(dxrow + h * 4, reinterpret_cast<const float(&)[4]>(g[h * 4]));
            }
        };
        emit(one_u, reinterpret_cast<T*>(a.w_user) + iu * D + sub * VEC, u, gx, a.dx + (int64_t)t * D + sub * VEC, (SML_NT & 2) != 0);
        if constexpr (SH) {
            // a tail occurrence's gradient row goes straight into its owner's inbox, slot t (positive) / B + t (negative)
            // of this rank's row slot; a head occurrence's row stays in the local dx (dense reduction later)
            auto push = [&](int q, int slot, const float (&g)[VEC], float* dxrow) {
                float* dst = q >= 0 ? a.inbox_tab[q] + a.push_off + (int64_t)slot * D + sub * VEC : dxrow;
#pragma unroll
                for (int h = 0; h < VEC / 4; ++h) {
                    f32x4 v; v[0] = g[h * 4]; v[1] = g[h * 4 + 1]; v[2] = g[h * 4 + 2]; v[3] = g[h * 4 + 3];
                    if (q >= 0) peer_store16(dst + h * 4, v); else *reinterpret_cast<f32x4*>(dst + h * 4) = v;
                }
            };
            push(qi, t, gy, a.dx + (int64_t)(p_B + t) * D + sub * VEC);
            push(qn, p_B + t, gz, a.dx + (int64_t)(2 * p_B + t) * D + sub * VEC);
        } else {
            emit(one_i, reinterpret_cast<T*>(a.w_item) + ii * D + sub * VEC, it, gy, a.dx + (int64_t)(p_B + t) * D + sub * VEC, (SML_NT & 4) != 0);
            emit(one_n, reinterpret_cast<T*>(a.w_item) + in * D + sub * VEC, ng, gz, a.dx + (int64_t)(2 * p_B + t) * D + sub * VEC, (SML_NT & 4) != 0);
        }
        contrib = (sub == 0 ? lt : 0.0f) + 0.5f * (a.lam_user * sq_u + a.lam_item * sq_i);
    }
    const float tot = block_sum256(contrib, sh4);
    if (threadIdx.x == 0) a.loss_part[blockIdx.x] = tot;
    if constexpr (SH) peer_signal(a.peer);          // this workgroup's pushes are acknowledged: +1 on every owner's counter
}


// w_head[row] -= lr * (sum over ranks, in rank order, of the ranks' dense head partials): every replica applies the same bits
template <int D, typename T>
__global__ __launch_bounds__(256) void k_head_apply(T* __restrict__ w, long long head_rows, float lr, SmlPeerPoll p) {
    constexpr int VEC = RowVec<T>::VEC;
    constexpr int LPR = D / VEC;
    peer_wait(p);
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long row = gid / LPR;
    const int sub = (int)(gid % LPR);
    if (row >= head_rows) return;
    float g[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) g[k] = 0.0f;
    for (int q = 0; q < p.world; ++q) {
        const float* src = p.slot0 + q * p.slot_stride + row * D + sub * VEC;
#pragma unroll
        for (int k = 0; k < VEC; ++k) { const float x = peer_load(src + k); g[k] = q == 0 ? x : g[k] + x; }
    }
    float x[VEC];
    RowVec<T>::load(w + row * D + sub * VEC, x);
#pragma unroll
    for (int k = 0; k < VEC; ++k) x[k] -= lr * g[k];
    RowVec<T>::store(w + row * D + sub * VEC, x);
}
__global__ __launch_bounds__(64) void k_peer_signal(SmlPeerPush p) { peer_signal(p); }


// per epoch, over one sorted list: a record for every position (len = 0 unless the position
// starts a run of equal keys); optionally the selection flag of duplicated runs (len >= 2) and
// the "row occurs once in its batch" mark of every occurrence (indexed by batch and slot).
template <typename K>
__global__ void k_mark_runs(const K* __restrict__ keys, const uint32_t* __restrict__ vals, int64_t n, int row_bits,
                            SmlRun* __restrict__ rec, uint8_t* __restrict__ flag_dup, uint8_t* __restrict__ uniq,
                            int64_t uniq_stride, int64_t uniq_item_base) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    constexpr int PROBE = 8;
    const K k = keys[q];
    K nb[PROBE + 1];
    nb[0] = q > 0 ? keys[q - 1] : (K)~k;
#pragma unroll
    for (int j = 1; j <= PROBE; ++j) nb[j] = q + j < n ? keys[q + j] : (K)~k;
    const bool head = nb[0] != k;
    int len = 0;
    if (head) {
        len = 1;
#pragma unroll
        for (int j = 1; j <= PROBE; ++j) len += (len == j && nb[j] == k) ? 1 : 0;
        if (len > PROBE) {   // upper bound of `k` in (q+PROBE, n) by bisection
            int64_t lo = q + PROBE + 1, hi = n;
            while (lo < hi) {
                const int64_t mid = (lo + hi) >> 1;
                if (keys[mid] == k) lo = mid + 1; else hi = mid;
            }
            len = (int)(lo - q);
        }
    }
    SmlRun r;
    r.row = row_bits >= 32 ? (uint32_t)k : (uint32_t)(k & (((K)1 << (row_bits & 31)) - 1));
    r.pos = (uint32_t)q; r.len = (uint32_t)len; r.pad = 0;
#pragma unroll
    for (int j = 0; j < SML_RUN_INL; ++j) r.slot[j] = j < len ? vals[q + j] : 0u;
    rec[q] = r;
    if (flag_dup != nullptr) flag_dup[q] = (head && len >= 2) ? 1 : 0;
    if (uniq != nullptr) {
        const int64_t b = row_bits >= 32 ? (int64_t)((uint64_t)k >> 32) : (int64_t)(k >> (row_bits & 31));
        const bool one = head && len == 1;      // (a non-head position belongs to a longer run)
        uniq[b * uniq_stride + uniq_item_base + vals[q]] = one ? 1 : 0;
    }
}




// this batch's slice of the run lists
struct RunLists { const SmlRun* run_u; int n_u; const SmlRun* run_i; int n_i; };
__device__ __forceinline__ RunLists run_lists(const SmlRunArgs& a, const SmlRun* run_u, const SmlRun* run_i, const int* off_u, const int* off_i,
                                              int n_u, int n_i, int batch_index) {
    RunLists L{run_u, n_u, run_i, n_i};
    if (off_u != nullptr) {
        const int u0 = off_u[batch_index], i0 = off_i[batch_index];
        L.run_u += u0; L.n_u = off_u[batch_index + 1] - u0;
        L.run_i += i0; L.n_i = off_i[batch_index + 1] - i0;
        if (a.cnt_u != nullptr) { L.n_u = a.cnt_u[batch_index * SML_PREP_CNT_STRIDE]; L.n_i = a.cnt_i[batch_index * SML_PREP_CNT_STRIDE]; }
    }
    return L;
}

// ------------------------------------------------------------------------------------
// hot rows: chunk partial sums.  One workgroup per (hot run, chunk of SML_HOT_CHUNK occurrences),
// grid-stride; the flattened chunk index -> (run, chunk) map is a prefix sum over the hot list,
// recomputed per workgroup in LDS (the list has at most a few thousand entries).  Fixed order.
// ------------------------------------------------------------------------------------
template <int D>
__device__ __forceinline__ void hot_partial_body(const SmlRunArgs& a, int block, int n_blocks, int* pre, int* tsum,
                                                 float (*rows)[D]) {
    constexpr int VEC = 4;
    constexpr int LPR = D / VEC;
    constexpr int GB = 256 / LPR;            // lane groups per workgroup
    const int tid = threadIdx.x;
    const int nh = min(*a.hot_count, a.hot_cap);
    if (nh == 0) return;
    // exclusive prefix of chunk counts: each thread owns a contiguous slice, then a serial pass over 256 sums
    const int per = (nh + 255) / 256;
    int local = 0;
    for (int e = tid * per; e < min(nh, (tid + 1) * per); ++e)
        local += ((int)a.hot_list[3 * e + 1] + SML_HOT_CHUNK - 1) / SML_HOT_CHUNK;
    {   // block-wide inclusive scan of `local`: shuffles inside each wavefront, then the four wave totals
        const int ln = tid & 63, wvi = tid >> 6;
        int inc = local;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(inc, off, 64); if (ln >= off) inc += t; }
        if (ln == 63) tsum[wvi] = inc;           // tsum[0..3]: wave totals (scratch use of the first entries)
        __syncthreads();
        int before = 0;
#pragma unroll
        for (int w2 = 0; w2 < 4; ++w2) before += (w2 < wvi) ? tsum[w2] : 0;
        __syncthreads();
        tsum[tid + 1] = before + inc;
        if (tid == 0) tsum[0] = 0;
        __syncthreads();
    }
    {
        int run = tsum[tid];
        for (int e = tid * per; e < min(nh, (tid + 1) * per); ++e) {
            pre[e] = run;
            run += ((int)a.hot_list[3 * e + 1] + SML_HOT_CHUNK - 1) / SML_HOT_CHUNK;
        }
        if (tid == 255) pre[nh] = tsum[256];
    }
    __syncthreads();
    const int grp = tid / LPR, sub = tid % LPR;
    for (int w = block; w < pre[nh]; w += n_blocks) {
        int lo = 0, hi2 = nh - 1;                 // largest h with pre[h] <= w
        while (lo < hi2) { const int mid = (lo + hi2 + 1) >> 1; if (pre[mid] <= w) lo = mid; else hi2 = mid - 1; }
        const int h = lo, c = w - pre[h];
        if (c == 0 && tid == 0) a.hot_first[h] = w;
        const uint32_t packed = a.hot_list[3 * h];
        const int is_item = packed >> 31, pos0 = (int)(packed & 0x7fffffffu), len = (int)a.hot_list[3 * h + 1];
        const uint32_t* vals = (is_item ? a.val_i : a.val_u) + pos0;
        const float* dx = is_item ? a.dx_i : a.dx;
        const int q_begin = c * SML_HOT_CHUNK, q_end = min(len, q_begin + SML_HOT_CHUNK);
        float acc[VEC] = {0.f, 0.f, 0.f, 0.f};
        for (int q0 = q_begin + grp; q0 < q_end; q0 += 16 * GB) {
            float x[16][VEC];
            uint32_t sl[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) sl[j] = q0 + j * GB < q_end ? vals[q0 + j * GB] : 0u;
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (q0 + j * GB < q_end) RowVec<float>::load(dx + (int64_t)sl[j] * D + sub * VEC, x[j]);
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (q0 + j * GB < q_end) {
#pragma unroll
                    for (int k = 0; k < VEC; ++k) acc[k] += x[j][k];
                }
        }
        __syncthreads();                          // the previous trip's readers of rows[] are done
        RowVec<float>::store(&rows[grp][sub * VEC], acc);
        __syncthreads();
        if (tid < D) {
            float s2 = 0.0f;
#pragma unroll 8
            for (int g2 = 0; g2 < GB; ++g2) s2 += rows[g2][tid];
            a.hot_part[(int64_t)w * D + tid] = s2;
        }
    }
}

// ------------------------------------------------------------------------------------
// segmented row update over run records: one lane group per record (grid-stride).  The group
// sums the run's per-occurrence gradient rows in slot order (deterministic) and writes the row
// once; the row (and its Adam state) is fetched before the sum, so the two latencies overlap.
// Runs longer than LONG are summed by the whole wavefront; runs longer than SML_HOT (SGD, large
// batches) go to the workgroup-level reducers.  OPT 0: SGD.  OPT 1: Adam with the skipped
// zero-gradient steps replayed.
// ------------------------------------------------------------------------------------
template <int D, typename T, int OPT, bool HOTB>
__global__ __launch_bounds__(256) void k_run_update(const SmlRun* __restrict__ p_run_u, const SmlRun* __restrict__ p_run_i, const int* __restrict__ p_off_u,
                                                    const int* __restrict__ p_off_i, int p_n_u, int p_n_i, int p_batch_index, int p_known, int p_hot_blocks,
                                                    SmlRunArgs a) {
    // (round 6: what the first round trip -- the run records -- needs as 13 preloaded dwords; see transfer_net.hip k_transfer_fwd)
    constexpr int VEC = RowVec<T>::VEC;
    constexpr int LPR = D / VEC;
    constexpr int G = 64 / LPR;             // lane groups (records) per wavefront
    // (an SGD epoch without hot runs is light-tailed: shallower unrolls, 71 instead of 112 VGPRs, 7 waves per SIMD)
    constexpr bool LIGHT = OPT == 0 && !HOTB;
    // runs longer than this are summed by the whole wavefront.  (ONE value for both SGD variants: which of them runs
    // depends on whether the epoch's longest run is known yet, and the two must round identically)
    constexpr int LONG = OPT == 0 ? 4 : 8;
    constexpr int LD = (VEC == 4 && !LIGHT) ? 8 : 4;    // ... with LD rows per lane group in flight
    __shared__ SmlSched swin[OPT == 1 ? SML_SW : 1];
    __shared__ SmlReplayEnt rtab[OPT == 1 ? SML_RP_N : 1];      // closed-form replay entries of this launch (rows this kernel replays itself)
    const SmlReplayEnt* tab = nullptr;
    if (OPT == 1) {                       // the Adam schedule of the last SML_SW steps, staged once per block
        sched_window_load(swin, a.sched, a.cur_step, threadIdx.x);
        if (a.sched_len > 0) { replay_table_build(rtab, a.sched, a.sched_len, a.cur_step - 1, threadIdx.x); tab = rtab; }
        __syncthreads();
    }
    // several GPUs (one-shot exchange): the ranks' gradient rows have landed in this rank's inbox slots when its counters
    // reach the step's value -- polled HERE (one lane per source rank, every workgroup) instead of by a launch of its own
    if constexpr (OPT == 1) { if (a.wait.world > 0) peer_wait(a.wait); }
    int run_blocks = gridDim.x;
    if constexpr (OPT == 0 && HOTB) {
        // the first hot_blocks workgroups reduce the hot rows' chunks (independent of the runs below; theirs is
        // the longest chain of the launch, so they are dispatched first)
        __shared__ int pre[SML_HOT_MAXCAP + 1];
        __shared__ int tsum[257];
        __shared__ __attribute__((aligned(16))) float hrows[256 / (D / 4)][D];
        run_blocks -= p_hot_blocks;
        if ((int)blockIdx.x < p_hot_blocks) { hot_partial_body<D>(a, (int)blockIdx.x, p_hot_blocks, pre, tsum, hrows); return; }
    }
    const int lane = threadIdx.x & 63;
    const int grp = lane / LPR, sub = lane % LPR;
    const RunLists L = run_lists(a, p_run_u, p_run_i, p_off_u, p_off_i, p_n_u, p_n_i, p_batch_index);
    const int total = L.n_u + L.n_i;
    const int wave_id = (((int)blockIdx.x - ((OPT == 0 && HOTB) ? p_hot_blocks : 0)) * 256 + threadIdx.x) >> 6;
    const int n_waves = (run_blocks * 256) >> 6;
    // Compacted run lists (bare step): record k = (trip * G + grp) * n_waves + wave -- neighbouring records
    // (hot rows are neighbours in a sorted list when popular rows have neighbouring ids) go to different
    // wavefronts, so their long sums run side by side instead of one after the other in one wave.
    // Per-position records (MF stage): a run of length L is followed by L-1 empty records, which spaces
    // the heads out already; consecutive records per wave keep the record loads coalesced.
    const bool strided = p_off_u != nullptr || p_known != 0;
    for (int base = 0; base < total; base += n_waves * G) {                 // wave-uniform trip count
        const int k = strided ? base + grp * n_waves + wave_id : base + wave_id * G + grp;
        const bool valid = k < total;
        const int is_item = (valid && k >= L.n_u) ? 1 : 0;
        SmlRun run;
        {   // two 16-byte loads; an out-of-range group gets an empty record
            const uint4* src = reinterpret_cast<const uint4*>(is_item ? L.run_i + (k - L.n_u) : L.run_u + k);
            uint4 r0 = make_uint4(0u, 0u, 0u, 0u), r1 = r0;
            if (valid) { r0 = src[0]; r1 = src[1]; }
            run.row = r0.x; run.pos = r0.y; run.len = r0.z; run.pad = r0.w;
            run.slot[0] = r1.x; run.slot[1] = r1.y; run.slot[2] = r1.z; run.slot[3] = r1.w;
        }
        int len = (int)run.len;
        bool head = len > 0;
        // a hot row is in the batch's hot list (built with the index lists): the workgroup-level reducers own it
        if (OPT == 0 && a.hot_list != nullptr && len > SML_HOT) head = false;
        // the row and its optimiser state do not depend on the gradient sum: fetch them first
        T* w = reinterpret_cast<T*>(is_item ? a.w_item : a.w_user);
        const int64_t row = run.row;
        float p[VEC];
        float m[4] = {0.f, 0.f, 0.f, 0.f}, v[4] = {0.f, 0.f, 0.f, 0.f};
        int from = 0;
        if (head) {
            bool from_scratch = false;
            if constexpr (OPT == 1) from_scratch = a.rep_x != nullptr && (is_item ? a.rep_i : a.rep_u);
            if constexpr (OPT == 1) {
                if (from_scratch) {
                    // the forward's replayed copy of this row (any occurrence holds the same values: the first one)
                    const int64_t s0 = run.slot[0];
                    RowVec<float>::load(a.rep_x + s0 * a.rep_x_stride + a.rep_x_off + sub * 4, reinterpret_cast<float(&)[4]>(p[0]));
                    RowVec<float>::load(a.rep_m + s0 * D + sub * 4, m);
                    RowVec<float>::load(a.rep_v + s0 * D + sub * 4, v);
                    from = a.cur_step - 1;
                } else {
                    RowVec<T>::load(w + row * D + sub * VEC, p);
                    RowVec<float>::load((is_item ? a.m_item : a.m_user) + row * D + sub * 4, m);
                    RowVec<float>::load((is_item ? a.v_item : a.v_user) + row * D + sub * 4, v);
                    from = (is_item ? a.last_item : a.last_user)[row];
                }
            } else {
                RowVec<T>::load(w + row * D + sub * VEC, p);
            }
        }
        float g[VEC];
#pragma unroll
        for (int q = 0; q < VEC; ++q) g[q] = 0.0f;
        // ---- short runs: the record's own lane group sums them, up to LONG rows in flight, in slot order
        if (head && len <= LONG) {
            const uint32_t* vals = (is_item ? a.val_i : a.val_u) + run.pos;
            const float* dx = is_item ? a.dx_i : a.dx;
            uint32_t sl[LONG];
#pragma unroll
            for (int j = 0; j < LONG; ++j) sl[j] = j < SML_RUN_INL ? run.slot[j] : (j < len ? vals[j] : 0u);
            float x[LONG][VEC];
#pragma unroll
            for (int j = 0; j < LONG; ++j)
                if (j < len) {
                    const float* src = dx + (int64_t)sl[j] * D + sub * VEC;
#pragma unroll
                    for (int h = 0; h < VEC / 4; ++h) RowVec<float>::load(src + h * 4, reinterpret_cast<float(&)[4]>(x[j][h * 4]));
                }
#pragma unroll
            for (int j = 0; j < LONG; ++j)
                if (j < len) {
#pragma unroll
                    for (int q = 0; q < VEC; ++q) g[q] += x[j][q];
                }
        }
        // ---- long runs: the wavefront's G lane groups each sum a strided share, then the shares are
        // added across groups in a fixed xor order (deterministic)
        unsigned long long todo = __ballot(head && len > LONG && sub == 0);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const int l_pos = __shfl((int)run.pos, leader, 64), l_len = __shfl(len, leader, 64), l_item = __shfl(is_item, leader, 64);
            const uint32_t* vals = (l_item ? a.val_i : a.val_u) + l_pos;
            const float* dx = l_item ? a.dx_i : a.dx;
            float acc[VEC];
#pragma unroll
            for (int q = 0; q < VEC; ++q) acc[q] = 0.0f;
            for (int q0 = grp; q0 < l_len; q0 += LD * G) {
                float x[LD][VEC];
                uint32_t sl[LD];
#pragma unroll
                for (int j = 0; j < LD; ++j) sl[j] = q0 + j * G < l_len ? vals[q0 + j * G] : 0u;
#pragma unroll
                for (int j = 0; j < LD; ++j)
                    if (q0 + j * G < l_len) {
                        const float* src = dx + (int64_t)sl[j] * D + sub * VEC;
#pragma unroll
                        for (int h = 0; h < VEC / 4; ++h) RowVec<float>::load(src + h * 4, reinterpret_cast<float(&)[4]>(x[j][h * 4]));
                    }
#pragma unroll
                for (int j = 0; j < LD; ++j)
                    if (q0 + j * G < l_len) {
#pragma unroll
                        for (int q = 0; q < VEC; ++q) acc[q] += x[j][q];
                    }
            }
#pragma unroll
            for (int off = LPR; off < 64; off <<= 1)
#pragma unroll
                for (int q = 0; q < VEC; ++q) acc[q] += __shfl_xor(acc[q], off, 64);
            if (lane / LPR == leader / LPR) {
#pragma unroll
                for (int q = 0; q < VEC; ++q) g[q] = acc[q];
            }
        }
        if (!head) continue;
        if constexpr (OPT == 0) {
#pragma unroll
            for (int q = 0; q < VEC; ++q) p[q] -= a.lr * g[q];
            RowVec<T>::store(w + row * D + sub * VEC, p);
        } else {
            const SmlSched sc = swin[SML_SW - 1];      // = sched[cur_step]
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                adam_replay_t(p[q], m[q], v[q], from, a.cur_step - 1, a.sched, swin, a.cur_step, tab);
                adam_apply(p[q], m[q], v[q], g[q], sc);
            }
            RowVec<T>::store(w + row * D + sub * VEC, p);
            RowVec<float>::store((is_item ? a.m_item : a.m_user) + row * D + sub * 4, m);
            RowVec<float>::store((is_item ? a.v_item : a.v_user) + row * D + sub * 4, v);
            if (sub == 0) (is_item ? a.last_item : a.last_user)[row] = a.cur_step;
        }
    }
}

// hot rows: sum each run's chunk partials and take the SGD step.  One wavefront per hot row: its lane
// groups sum strided shares of the partials, then the shares meet in a fixed xor order (deterministic).
template <int D, typename T>
__global__ __launch_bounds__(256) void k_hot_apply(SmlRunArgs a) {
    constexpr int VEC = RowVec<T>::VEC;
    constexpr int LPR = D / VEC;
    constexpr int G = 64 / LPR;
    const int lane = threadIdx.x & 63, grp = lane / LPR, sub = lane % LPR;
    const int h = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const int nh = min(*a.hot_count, a.hot_cap);
    if (h >= nh) return;
    const uint32_t packed = a.hot_list[3 * h];
    const int is_item = packed >> 31, len = (int)a.hot_list[3 * h + 1];
    const int first = a.hot_first[h], nchunks = (len + SML_HOT_CHUNK - 1) / SML_HOT_CHUNK;
    const int64_t row = a.hot_list[3 * h + 2];
    T* w = reinterpret_cast<T*>(is_item ? a.w_item : a.w_user);
    float p[VEC];
    if (grp == 0) RowVec<T>::load(w + row * D + sub * VEC, p);
    float g[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) g[k] = 0.0f;
    for (int c0 = grp; c0 < nchunks; c0 += 8 * G) {
        float x[8][VEC];
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (c0 + j * G < nchunks) {
                const float* src = a.hot_part + (int64_t)(first + c0 + j * G) * D + sub * VEC;
#pragma unroll
                for (int hh = 0; hh < VEC / 4; ++hh) RowVec<float>::load(src + hh * 4, reinterpret_cast<float(&)[4]>(x[j][hh * 4]));
            }
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (c0 + j * G < nchunks) {
#pragma unroll
                for (int k = 0; k < VEC; ++k) g[k] += x[j][k];
            }
    }
#pragma unroll
    for (int off = LPR; off < 64; off <<= 1)
#pragma unroll
        for (int k = 0; k < VEC; ++k) g[k] += __shfl_xor(g[k], off, 64);
    if (grp != 0) return;
#pragma unroll
    for (int k = 0; k < VEC; ++k) p[k] -= a.lr * g[k];
    RowVec<T>::store(w + row * D + sub * VEC, p);
}

// bring every row up to `cur_step` (all pending steps have zero gradient).  sched_len > 0: the closed-form tables behind the
// schedule cover this launch (sml_dev.h) -- the workgroup builds its 256 entries first, rows within reach take one evaluation
// per element instead of one loop trip per pending step.  Grid-stride over row groups: the table is built once per workgroup.
template <int D>
__global__ __launch_bounds__(256) void k_adam_flush(float* __restrict__ w, float* __restrict__ mt, float* __restrict__ vt,
                                                    int32_t* __restrict__ last, int64_t rows,
                                                    const SmlSched* __restrict__ sched, int cur_step, int sched_len) {
    constexpr int LPR = D / 4;
    __shared__ SmlSched swin[SML_SW];
    __shared__ SmlReplayEnt rtab[SML_RP_N];
    sched_window_load(swin, sched, cur_step, threadIdx.x);
    const bool closed = sched_len > 0;
    if (closed) replay_table_build(rtab, sched, sched_len, cur_step, threadIdx.x);
    __syncthreads();
    const SmlReplayEnt* tab = closed ? rtab : nullptr;
    const int64_t total = rows * LPR;
    for (int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x; gid < total; gid += (int64_t)gridDim.x * 256) {
        const int64_t row = gid / LPR;
        const int sub = (int)(gid % LPR);
        const int from = last[row];
        if (from < 0 || from >= cur_step) continue;          // never touched (state is zero), or already current
        float p[4], m[4], v[4];
        RowVec<float>::load(w + row * D + sub * 4, p);
        RowVec<float>::load(mt + row * D + sub * 4, m);
        RowVec<float>::load(vt + row * D + sub * 4, v);
        adam_replay_t4(p, m, v, from, cur_step, sched, swin, cur_step, tab);
        RowVec<float>::store(w + row * D + sub * 4, p);
        RowVec<float>::store(mt + row * D + sub * 4, m);
        RowVec<float>::store(vt + row * D + sub * 4, v);
        // (the row's lanes share this wavefront and sit in the same trip: all have read `from`.  256 % LPR == 0 and the stride is
        // a multiple of 256, so a row never straddles two trips or two workgroups)
        if (sub == 0) last[row] = cur_step;
    }
}

// ------------------------------------------------------------------------------------
// MFbasemode.forward (model/MF.py:34-43)
// ------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void k_mf_forward(const float* __restrict__ wu, const float* __restrict__ wi,
                                                    const int64_t* __restrict__ user, const int64_t* __restrict__ item,
                                                    int64_t n, int norm, float* __restrict__ uemb,
                                                    float* __restrict__ iemb, float* __restrict__ score) {
    constexpr int LPR = D / 4;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t t = gid / LPR;
    const int sub = (int)(gid % LPR);
    if (t >= n) return;
    float u[4], it[4];
    RowVec<float>::load(wu + user[t] * D + sub * 4, u);
    RowVec<float>::load(wi + item[t] * D + sub * 4, it);
    float s = 0.f, uu = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) { s += u[k] * it[k]; uu += u[k] * u[k]; }
    s = group_sum<LPR>(s);
    if (norm) s = s / sqrtf(group_sum<LPR>(uu));
    RowVec<float>::store(uemb + t * D + sub * 4, u);
    RowVec<float>::store(iemb + t * D + sub * 4, it);
    if (sub == 0) score[t] = s;
}

// ------------------------------------------------------------------------------------
// evaluation: rank of the positive among 1+neg candidates (MFbasemode.test, model/MF.py:45-60)
// one wavefront per test row; 64/LPR candidate rows in flight per step, unrolled x4
// ------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void k_eval_ranks(const float* __restrict__ wu, const float* __restrict__ wi,
                                                    const int64_t* __restrict__ rows, int64_t n, int n_cols,
                                                    int32_t* __restrict__ rank) {
    constexpr int LPR = D / 4;
    constexpr int G = 64 / LPR;
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    const int grp = lane / LPR, sub = lane % LPR;
    const int64_t* R = rows + r * n_cols;
    float u[4], x[4];
    RowVec<float>::load(wu + R[0] * D + sub * 4, u);
    RowVec<float>::load(wi + R[1] * D + sub * 4, x);
    const float s0 = eval_group_sum<LPR>(dot4(u, x));
    int cnt = 0;
    int c = 2;
    if constexpr (LPR <= 16) {
        // eight candidates per lane group and trip: 8 * G consecutive candidates per wavefront
        for (; c + 8 * G <= n_cols; c += 8 * G) {
            const int cb = c + grp * 8;
            float y[8][4], a[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) RowVec<float>::load(wi + R[cb + e] * D + sub * 4, y[e]);
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] = dot4(u, y[e]);
            const float sc = eval8_reduce<LPR>(a, sub);
            cnt += ((sub & (LPR / 8 - 1)) == 0 && sc > s0) ? 1 : 0;
        }
    }
    for (c += grp; c < n_cols; c += G) {            // the remainder, one candidate per lane group
        RowVec<float>::load(wi + R[c] * D + sub * 4, x);
        const float sc = eval_group_sum<LPR>(dot4(u, x));
        cnt += (sub == 0 && sc > s0) ? 1 : 0;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
    if (lane == 0) rank[r] = cnt;
}

// ------------------------------------------------------------------------------------
// L2-blocked evaluation.  The item table (15.7 MB at Yelp scale) does not fit one XCD's 4 MB L2, so
// the plain kernel's gathers mostly miss to Infinity Cache.  Once per test set, every row's
// candidates are grouped into SML_EVB item ranges (k_eval_bucketize); the rank kernel then gives
// range x to the workgroups that run on XCD x (workgroup b is dispatched to XCD b % 8 -- observed
// placement, used for cache affinity only: any other placement is merely slower), so each XCD
// gathers from a 1/8 slice of the table that stays in its own L2.  Partial counts meet in an integer
// atomicAdd (exact, order-independent).
// ------------------------------------------------------------------------------------
#define SML_EVB 8
// rows_out: int32 [n, n_cols] (user, positive, then the candidates grouped by item range): half the
// index stream of the int64 input, and that stream is the evaluation's only HBM traffic of note
__global__ __launch_bounds__(256) void k_eval_bucketize(const int64_t* __restrict__ rows, int64_t n, int n_cols,
                                                        int64_t n_item, int32_t* __restrict__ rows_out,
                                                        int32_t* __restrict__ bucket_off) {
    __shared__ int hist[4][SML_EVB], cursor[4][SML_EVB];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t r = (int64_t)blockIdx.x * 4 + wv;
    if (lane < SML_EVB) hist[wv][lane] = 0;
    __syncthreads();
    const int64_t per = (n_item + SML_EVB - 1) / SML_EVB;
    const bool ok = r < n;
    const int64_t* R = rows + (ok ? r : 0) * n_cols;
    if (ok) for (int c = 2 + lane; c < n_cols; c += 64) atomicAdd(&hist[wv][(int)min((int64_t)SML_EVB - 1, R[c] / per)], 1);
    __syncthreads();
    if (ok && lane == 0) {
        int run = 0;
        int32_t* off = bucket_off + r * (SML_EVB + 1);
        for (int b = 0; b < SML_EVB; ++b) { off[b] = run; cursor[wv][b] = run; run += hist[wv][b]; }
        off[SML_EVB] = run;
    }
    __syncthreads();
    if (!ok) return;
    int32_t* O = rows_out + r * n_cols;
    if (lane < 2) O[lane] = (int32_t)R[lane];
    for (int c = 2 + lane; c < n_cols; c += 64) {
        const int64_t it = R[c];
        const int slot = atomicAdd(&cursor[wv][(int)min((int64_t)SML_EVB - 1, it / per)], 1);
        O[2 + slot] = (int32_t)it;          // order inside a bucket is irrelevant: the rank is a count
    }
}

// Persistent form: workgroup b serves item range b % 8 (= its XCD); wavefront w of it walks the test rows
// (b/8 + k * gridDim/8) * 4 + w, k = 0, 1, ...  The grid is capped (an evaluation that runs underneath training
// kernels on a side stream should not flood every CU's wave slots), so what a wavefront does per row is a chain of
// dependent memory round trips -- header (bucket bounds, user, positive) -> candidate ids -> item rows -> scores --
// and with one wavefront per SIMD nothing else hides it.  The loop is therefore software-pipelined on UNITS of 8*G
// candidates (one trip of the eight-at-a-time reduction):
//   headers    for 64 rows at a time: lane l fetches the header of row k0 + l, the walk reads them with v_readlane
//              (scalar control flow; the next 64 headers are in flight while these are consumed);
//   unit k+2   candidate ids: one coalesced nontemporal load;
//   unit k+1   ids -> byte offsets through an LDS row of the wavefront -> eight item-row gathers + the user row
//              (+ the positive's row on a row's first unit) into the idle register buffer;
//   unit k     scores, exact compare against the positive's score, count by ballot / popcount (scalar).
// Two register buffers alternate by unrolling the loop twice: no register is copied while its load is in flight.
template <int D>
__global__ __launch_bounds__(256) void k_eval_ranks_bucketed(const float* __restrict__ wu, const float* __restrict__ wi,
                                                             const int32_t* __restrict__ rows, const int32_t* __restrict__ bucket_off,
                                                             int64_t n, int n_cols, int32_t* __restrict__ rank) {
    constexpr int LPR = D / 4;
    constexpr int G = 64 / LPR;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int x = blockIdx.x % SML_EVB;                          // this workgroup's item range (= its XCD)
    const int grp = lane / LPR, sub = lane % LPR;
    const int64_t n_groups = (n + 3) / 4;
    const int64_t g0 = blockIdx.x / SML_EVB, gstep = gridDim.x / SML_EVB;
    if constexpr (LPR <= 16) {
        constexpr int UN = 8 * G;                                // candidates per unit
        __shared__ uint32_t xch[4][64];
        const int64_t K = g0 < n_groups ? (n_groups - g0 + gstep - 1) / gstep : 0;     // rows of this wavefront
        if (K == 0) return;
        const char* const wib = reinterpret_cast<const char*>(wi);
        const char* const wub = reinterpret_cast<const char*>(wu);

        struct Hdr { int c0, c1, user, pos; };
        auto load_headers = [&](int64_t k0) {                    // lane l: row k0 + l of this wavefront's walk
            const int64_t k = k0 + lane, r = (g0 + k * gstep) * 4 + wv;
            const bool ok = k < K && r < n;
            const int64_t rr = ok ? r : 0;
            Hdr h;
            h.c0 = 2 + bucket_off[rr * (SML_EVB + 1) + x];
            h.c1 = 2 + bucket_off[rr * (SML_EVB + 1) + x + 1];
            h.user = rows[rr * n_cols];
            h.pos = rows[rr * n_cols + 1];
            if (!ok) h.c1 = h.c0;
            return h;
        };
        struct Unit { int valid, cb, c1, user, pos, first, last; int64_t r; };
        // ---- the walk (all scalar): rows in order, each cut into units of UN candidates; empty buckets yield none
        Hdr hq = load_headers(0), hq_next = load_headers(64);
        int64_t kb = 0;
        int j = -1, p = 0, c0 = 0, c1 = 0, user = 0, pos = 0;
        int64_t r = 0;
        auto next_unit = [&]() {
            Unit u;
            u.valid = 0; u.cb = 0; u.c1 = 0; u.user = 0; u.pos = 0; u.first = 0; u.last = 0; u.r = 0;
            for (;;) {
                if (p < c1) {
                    u.valid = 1; u.cb = p; u.c1 = c1; u.user = user; u.pos = pos; u.first = p == c0; u.r = r;
                    p += UN;
                    u.last = p >= c1;
                    return u;
                }
                ++j;
                if (kb + j >= K) { --j; return u; }
                if (j == 64) { hq = hq_next; kb += 64; j = 0; hq_next = load_headers(kb + 64); }
                c0 = __builtin_amdgcn_readlane(hq.c0, j);
                c1 = __builtin_amdgcn_readlane(hq.c1, j);
                user = __builtin_amdgcn_readlane(hq.user, j);
                pos = __builtin_amdgcn_readlane(hq.pos, j);
                r = (g0 + (kb + j) * gstep) * 4 + wv;
                p = c0;
            }
        };
        auto issue_ids = [&](const Unit& u) -> uint32_t {        // (a slot past the bucket's end re-reads its last candidate)
            if (!u.valid) return 0u;
            return (uint32_t)__builtin_nontemporal_load(rows + u.r * n_cols + min(u.cb + lane, u.c1 - 1));
        };
        auto issue_rows = [&](const Unit& u, uint32_t id, float (&y)[8][4], float (&uu)[4], float (&xx)[4]) {
            if (!u.valid) return;
            xch[wv][lane] = id * (uint32_t)(D * 4);
            const uint4 o0 = *reinterpret_cast<const uint4*>(&xch[wv][grp * 8]);
            const uint4 o1 = *reinterpret_cast<const uint4*>(&xch[wv][grp * 8 + 4]);
            const uint32_t off[8] = {o0.x, o0.y, o0.z, o0.w, o1.x, o1.y, o1.z, o1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) RowVec<float>::load(reinterpret_cast<const float*>(wib + (size_t)(off[e] + (uint32_t)(sub * 16))), y[e]);
            RowVec<float>::load(reinterpret_cast<const float*>(wub + (int64_t)u.user * (D * 4)) + sub * 4, uu);
            if (u.first) RowVec<float>::load(reinterpret_cast<const float*>(wib + (int64_t)u.pos * (D * 4)) + sub * 4, xx);
        };
        float s0 = 0.f;
        int cnt = 0;
        auto score = [&](const Unit& u, float (&y)[8][4], float (&uu)[4], float (&xx)[4]) {
            if (!u.valid) return;
            if (u.first) s0 = eval_group_sum<LPR>(dot4(uu, xx));
            float a[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] = dot4(uu, y[e]);
            const float sc = eval8_reduce<LPR>(a, sub);
            const bool hit = (sub & (LPR / 8 - 1)) == 0 && u.cb + grp * 8 + eval8_index<LPR>(sub) < u.c1 && sc > s0;
            cnt += __popcll(__ballot(hit));
            if (u.last) {
                if (cnt && lane == 0) atomicAdd(&rank[u.r], cnt);
                cnt = 0;
            }
        };
        float yP[8][4], uP[4], xP[4], yQ[8][4], uQ[4], xQ[4];
        Unit U0 = next_unit();
        uint32_t idA = issue_ids(U0);
        issue_rows(U0, idA, yP, uP, xP);
        Unit U1 = next_unit();
        idA = issue_ids(U1);
        uint32_t idB;
        while (U0.valid) {
            // U0's rows are landing in P, U1's ids in idA
            Unit U2 = next_unit();
            idB = issue_ids(U2);
            issue_rows(U1, idA, yQ, uQ, xQ);
            score(U0, yP, uP, xP);
            if (!U1.valid) break;
            // U1's rows are landing in Q, U2's ids in idB
            Unit U3 = next_unit();
            idA = issue_ids(U3);
            issue_rows(U2, idB, yP, uP, xP);
            score(U1, yQ, uQ, xQ);
            U0 = U2;
            U1 = U3;
        }
    } else {
        for (int64_t gi = g0; gi < n_groups; gi += gstep) {
            const int64_t r = gi * 4 + wv;
            if (r >= n) continue;
            const int32_t* R = rows + r * n_cols;
            const int c0 = 2 + bucket_off[r * (SML_EVB + 1) + x], c1 = 2 + bucket_off[r * (SML_EVB + 1) + x + 1];
            if (c0 == c1) continue;
            float u[4], xr[4];
            const int32_t pos_item = R[1];
            RowVec<float>::load(wu + (int64_t)R[0] * D + sub * 4, u);
            RowVec<float>::load(wi + (int64_t)pos_item * D + sub * 4, xr);
            const float s0 = eval_group_sum<LPR>(dot4(u, xr));
            int cnt = 0;
            constexpr int UD = 4;              // candidate rows in flight per lane group
            for (int c = c0 + grp; c < c1; c += UD * G) {
                float y[UD][4];
                int32_t id[UD];
#pragma unroll
                for (int j = 0; j < UD; ++j) id[j] = (c + j * G < c1) ? __builtin_nontemporal_load(R + c + j * G) : pos_item;
#pragma unroll
                for (int j = 0; j < UD; ++j) RowVec<float>::load(wi + (int64_t)id[j] * D + sub * 4, y[j]);
#pragma unroll
                for (int j = 0; j < UD; ++j) {
                    const float sc = eval_group_sum<LPR>(dot4(u, y[j]));
                    cnt += (sub == 0 && c + j * G < c1 && sc > s0) ? 1 : 0;
                }
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
            if (lane == 0 && cnt) atomicAdd(&rank[r], cnt);
        }
    }
}


// ------------------------------------------------------------------------------------
// LDS-sliced evaluation (d = 32).  The gathering kernels above pay one 128-byte L1 line per candidate row (5.3
// clocks per line and CU, whatever the bytes or the instruction count: DESIGN.md section 5); here the item rows are
// read from LDS instead.  Once per test set (k_evs_count -> k_evs_scan -> k_evs_scatter) the candidates are re-ordered
// SLICE-MAJOR: slice s = item >> 10 (1,024 item rows = 128 KB of LDS), inside a slice by MINI-BLOCK
// of 64 test rows, inside a mini-block by test row; the candidates of one (test row, slice) UNIT are padded to an even
// count, so entries 2k and 2k + 1 of a segment always belong to one test row.  An entry is
//     (row in mini-block) << 25 | (item & 1023) * 128 | pad flag:
// e >> 18 is the byte offset of the user row inside the mini-block's dense user rows, e & 0x1ff80 the LDS byte offset of
// the item row.  Per evaluation:
//   k_evs_pre    user row and the positive's score of every test row, once, into dense arrays (ug, s0);
//   k_evs_ranks  workgroup (slice, part) stages its slice in LDS; each wavefront walks whole (slice, mini-block)
//                segments, 64 entries per trip: lane group g takes entries 8g..8g+7 (ds_bpermute from the lane that loaded
//                them), FOUR user-chunk loads per trip (one per entry pair: a wavefront-wide 16-byte load costs the CU's
//                vector-memory path 16 clocks whatever its lanes fetch -- tools/l1_rate_probe.hip -- and that path is what
//                bounds the pass), item chunks by ds_read_b128, the eight-at-a-time butterfly leaves lane l with the score
//                of entry l -- the entry it loaded itself; hits are counted in 64 wavefront-private LDS counters (one per
//                test row of the mini-block) and leave as one coalesced 128-byte store of partial[slice][row] (uint16: no
//                atomics, no zeroing).  Software-pipelined over trips: the user chunks of trip t + 1 are in flight while
//                trip t is scored (two register buffers, the loop unrolled twice) -- without that the sixteen wavefronts
//                of a CU fall into step, all loading, then all computing;
//   k_evs_sum    rank[r] = sum over slices of partial[s][r].
// Scores round exactly as in k_eval_ranks (dot4 per lane, the xor butterfly's pairing order), so ranks are identical.
// ------------------------------------------------------------------------------------
#define SML_EVS_SHIFT 10
#define SML_EVS_S (1 << SML_EVS_SHIFT)          // item rows per slice
#define SML_EVS_STRIDE 128                      // bytes between item rows in LDS (a multiple of 128: the lane's chunk offset is OR-ed in)
#define SML_EVS_MB 64                           // test rows per mini-block
#define SML_EVS_WAVES 16                        // wavefronts of a rank workgroup
#define SML_EVS_NS_MAX 1024                     // slices the preparation's LDS histograms hold
#define SML_EVS_PAD 1u

// per mini-block and wavefront (16 rows each): entries per slice (every row's count rounded up to even); mbcnt is
// slice-major so that its flat exclusive scan IS the segment table
__global__ __launch_bounds__(256) void k_evs_count(const int64_t* __restrict__ rows, int64_t n, int n_cols, int ns, int n_mb,
                                                   int32_t* __restrict__ whist, int32_t* __restrict__ mbcnt) {
    extern __shared__ int evs_hist[];            // [4][ns] per wavefront + [4][ns] per row
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int mb = blockIdx.x;
    int* H = evs_hist + wv * ns;
    int* Rw = evs_hist + (4 + wv) * ns;
    for (int i = lane; i < ns; i += 64) H[i] = 0;
    for (int k = 0; k < 16; ++k) {
        const int64_t r = (int64_t)mb * SML_EVS_MB + wv * 16 + k;
        if (r >= n) break;
        const int64_t* R = rows + r * n_cols;
        for (int i = lane; i < ns; i += 64) Rw[i] = 0;
        __builtin_amdgcn_wave_barrier();
        for (int c = 2 + lane; c < n_cols; c += 64) {
            const int s = (int)min((int64_t)ns - 1, max((int64_t)0, __builtin_nontemporal_load(R + c) >> SML_EVS_SHIFT));
            atomicAdd(&Rw[s], 1);
        }
        __builtin_amdgcn_wave_barrier();
        for (int i = lane; i < ns; i += 64) H[i] += (Rw[i] + 1) & ~1;
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    for (int i = threadIdx.x; i < ns; i += 256) {
        int tot = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { const int h = evs_hist[w * ns + i]; whist[((int64_t)mb * 4 + w) * ns + i] = h; tot += h; }
        mbcnt[(int64_t)i * n_mb + mb] = tot;
    }
}
// exclusive scan of cnt[0..m) into off[0..m] (one workgroup; m is a few hundred thousand at most)
__global__ __launch_bounds__(1024) void k_evs_scan(const int32_t* __restrict__ cnt, int64_t m, int32_t* __restrict__ off) {
    __shared__ int part[1024];
    const int64_t per = (m + 1023) / 1024;
    const int64_t b = min(m, (int64_t)threadIdx.x * per), e = min(m, b + per);
    int sum = 0;
    for (int64_t i = b; i < e; ++i) sum += cnt[i];
    part[threadIdx.x] = sum;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int v = threadIdx.x >= o ? part[threadIdx.x - o] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    int run = part[threadIdx.x] - sum;
    for (int64_t i = b; i < e; ++i) { off[i] = run; run += cnt[i]; }
    if (threadIdx.x == 1023) off[m] = part[1023];
}
__global__ __launch_bounds__(256) void k_evs_scatter(const int64_t* __restrict__ rows, int64_t n, int n_cols, int ns, int n_mb,
                                                     const int32_t* __restrict__ whist, const int32_t* __restrict__ seg_off,
                                                     uint32_t* __restrict__ entries) {
    extern __shared__ int evs_cur[];             // [4][ns]: next free position of (wavefront, slice) + [4][ns]: where the row began
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int mb = blockIdx.x;
    for (int i = threadIdx.x; i < ns; i += 256) {
        int run = seg_off[(int64_t)i * n_mb + mb];
#pragma unroll
        for (int w = 0; w < 4; ++w) { evs_cur[w * ns + i] = run; run += whist[((int64_t)mb * 4 + w) * ns + i]; }
    }
    __syncthreads();
    int* C = evs_cur + wv * ns;
    int* B = evs_cur + (4 + wv) * ns;
    for (int k = 0; k < 16; ++k) {               // a wavefront's rows in order: a segment ends up sorted by test row
        const int64_t r = (int64_t)mb * SML_EVS_MB + wv * 16 + k;
        if (r >= n) break;
        const int64_t* R = rows + r * n_cols;
        const uint32_t tag = (uint32_t)(wv * 16 + k) << 25;
        for (int i = lane; i < ns; i += 64) B[i] = C[i];
        __builtin_amdgcn_wave_barrier();
        for (int c = 2 + lane; c < n_cols; c += 64) {
            const int64_t it = __builtin_nontemporal_load(R + c);
            const int s = (int)min((int64_t)ns - 1, max((int64_t)0, it >> SML_EVS_SHIFT));
            const int pos = atomicAdd(&C[s], 1);
            entries[pos] = tag | (uint32_t)(it & (SML_EVS_S - 1)) * SML_EVS_STRIDE;
        }
        __builtin_amdgcn_wave_barrier();
        for (int i = lane; i < ns; i += 64) {    // an odd unit gets one padding entry (scored, never counted)
            const int c = C[i];
            if ((c - B[i]) & 1) { entries[c] = tag | SML_EVS_PAD; C[i] = c + 1; }
        }
        __builtin_amdgcn_wave_barrier();
    }
}
template <int D>
__global__ __launch_bounds__(256) void k_evs_pre(const float* __restrict__ wu, const float* __restrict__ wi, const int64_t* __restrict__ rows,
                                                 int64_t n, int n_cols, float* __restrict__ ug, float* __restrict__ s0) {
    constexpr int LPR = D / 4;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t r = gid / LPR;
    const int sub = (int)(gid % LPR);
    if (r >= n) return;                          // (a lane group is inside or outside as a whole)
    float u[4], x[4];
    RowVec<float>::load(wu + rows[r * n_cols] * D + sub * 4, u);
    RowVec<float>::load(wi + rows[r * n_cols + 1] * D + sub * 4, x);
    const float s = eval_group_sum<LPR>(dot4(u, x));
    RowVec<float>::store(ug + r * D + sub * 4, u);
    if (sub == 0) s0[r] = s;
}
template <int D>
__global__ __launch_bounds__(SML_EVS_WAVES * 64) void k_evs_ranks(const float* __restrict__ ug, const float* __restrict__ s0v,
                                                                   const float* __restrict__ wi, const uint32_t* __restrict__ entries,
                                                                   const int32_t* __restrict__ seg_off, int64_t n, int64_t n_item,
                                                                   int n_mb, int parts, uint16_t* __restrict__ partial) {
    static_assert(D == 32, "the sliced evaluation is cut for d = 32 (eight lanes per row)");
    constexpr int LPR = D / 4;
    extern __shared__ __align__(16) unsigned char evs_smem[];
    int* const cnt = reinterpret_cast<int*>(evs_smem + (size_t)SML_EVS_S * SML_EVS_STRIDE);        // [waves][64]
    const int s = blockIdx.x / parts, part = blockIdx.x % parts;
    const int64_t i0 = (int64_t)s << SML_EVS_SHIFT;
    const int live = (int)min((int64_t)SML_EVS_S, n_item - i0);
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(wi + i0 * D);
        for (int i = threadIdx.x; i < SML_EVS_S * LPR; i += SML_EVS_WAVES * 64) {
            const int row = i / LPR, ch = i % LPR;
            *reinterpret_cast<f32x4*>(evs_smem + row * SML_EVS_STRIDE + ch * 16) = row < live ? src[i] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    __syncthreads();
    // the rank loop addresses the slice by LDS byte offset (one v_and_or per item chunk): the kernel's only LDS is this
    // dynamic block, so it starts at LDS address 0 -- checked, not assumed
    typedef const __attribute__((address_space(3))) f32x4* lds_f32x4_ptr;
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)evs_smem != 0u) __builtin_trap();
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int grp = lane >> 3, sub = lane & 7;
    const uint32_t sub16 = (uint32_t)sub * 16u;
    int* const mycnt = cnt + wv * 64;
    const int64_t n_pad = (int64_t)n_mb * SML_EVS_MB;
    const int32_t* const so = seg_off + (int64_t)s * n_mb;
    uint16_t* const pout = partial + (int64_t)s * n_pad;
    for (int mb = part * SML_EVS_WAVES + wv; mb < n_mb; mb += SML_EVS_WAVES * parts) {
        const int seg0 = so[mb], seg1 = so[mb + 1];
        const int64_t r0 = (int64_t)mb * SML_EVS_MB;
        if (seg0 == seg1) { pout[r0 + lane] = 0; continue; }          // no candidate of these 64 rows in this slice
        const float s0r = r0 + lane < n ? s0v[r0 + lane] : 0.f;
        mycnt[lane] = 0;
        const char* const ub = reinterpret_cast<const char*>(ug + r0 * D);
        auto load_e = [&](int t) -> uint32_t { return t < seg1 ? __builtin_nontemporal_load(entries + min(t + lane, seg1 - 1)) : 0u; };   // (a 320 MB stream per pass: not to displace the training stream's tables from the memory-side cache)
        // stage 2 of a trip: the lane group's eight entries, the four user chunks (one per entry pair)
        auto fetch = [&](uint32_t e, uint32_t (&x)[8], f32x4 (&u)[4]) {
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = (uint32_t)__shfl((int)e, grp * 8 + j, 64);
#pragma unroll
            for (int k = 0; k < 4; ++k) u[k] = *reinterpret_cast<const f32x4*>(ub + ((x[2 * k] >> 18) | sub16));
        };
        // stage 3: item chunks from LDS, scores, counts
        auto score = [&](uint32_t e, int t, const uint32_t (&x)[8], const f32x4 (&u)[4]) {
            float a[8];
            f32x4 y[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) y[j] = *(lds_f32x4_ptr)(uintptr_t)((x[j] & 0x1ff80u) | sub16);      // (LDS byte address: the slice starts at 0)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const f32x4 uu = u[j >> 1];
                a[j] = __fmaf_rn(uu[3], y[j][3], __fmaf_rn(uu[2], y[j][2], __fmaf_rn(uu[1], y[j][1], __fmul_rn(uu[0], y[j][0]))));   // == dot4
            }
            const float sc = eval8_reduce<LPR>(a, sub);          // eval8_index<8>(sub) == sub: the score of entry t + lane
            const int myrow = (int)(e >> 25);
            const float s0c = __shfl(s0r, myrow, 64);
            if (t + lane < seg1 && !(e & SML_EVS_PAD) && sc > s0c) atomicAdd(&mycnt[myrow], 1);
        };
        uint32_t eA = load_e(seg0), eB = load_e(seg0 + 64);
        uint32_t xA[8], xB[8];
        f32x4 uA[4], uB[4];
        fetch(eA, xA, uA);
        for (int t = seg0; t < seg1; t += 128) {
            // trip t is landing in A, the entries of trip t + 64 are in eB
            const uint32_t eC = load_e(t + 128);
            if (t + 64 < seg1) fetch(eB, xB, uB);
            score(eA, t, xA, uA);
            if (t + 64 >= seg1) break;
            const uint32_t eD = load_e(t + 192);
            if (t + 128 < seg1) fetch(eC, xA, uA);
            score(eB, t + 64, xB, uB);
            eA = eC;
            eB = eD;
        }
        __builtin_nontemporal_store((uint16_t)mycnt[lane], pout + r0 + lane);
    }
}
__global__ __launch_bounds__(256) void k_evs_sum(const uint16_t* __restrict__ partial, int64_t n, int64_t n_pad, int ns, int32_t* __restrict__ rank) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    int acc = 0;
    for (int s = 0; s < ns; ++s) acc += __builtin_nontemporal_load(partial + (int64_t)s * n_pad + r);
    rank[r] = acc;
}

// ------------------------------------------------------------------------------------
// On-device batch supply (fast mode; the reference-exact numpy stream stays on the host): for every
// (user, item) pair of an epoch, a negative drawn uniformly from the period's item set, redrawn while it is
// one of the user's own items (data/dataset.py:63-71 as a distribution, not as a random-number stream).
// Counter-based generator: element e of epoch `seed` always gets the same draws, whatever the launch shape.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t splitmix64(uint64_t& s) {
    uint64_t z = (s += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
__global__ __launch_bounds__(256) void k_sample_negatives(const int64_t* __restrict__ users, int64_t n,
                                                          const int64_t* __restrict__ item_all, int64_t pop,
                                                          const int64_t* __restrict__ user_ptr, int64_t n_users,
                                                          const int64_t* __restrict__ user_items, uint64_t seed,
                                                          int64_t* __restrict__ negs, int* __restrict__ failed) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    const int64_t u = users[e];
    int64_t b = 0, t = 0;
    if (u >= 0 && u < n_users) { b = user_ptr[u]; t = user_ptr[u + 1]; }
    uint64_t st = seed ^ ((uint64_t)e * 0xd1342543de82ef95ull + 0x632be59bd9b4e019ull);
    int64_t c = -1, cand = item_all[0];
    for (int tries = 0; tries < 4096; ++tries) {
        const uint64_t r = splitmix64(st);
        cand = item_all[(int64_t)__umul64hi(r, (uint64_t)pop)];                    // uniform over [0, pop)
        int64_t lo = b, hi = t;
        while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (user_items[mid] < cand) lo = mid + 1; else hi = mid; }
        if (!(lo < t && user_items[lo] == cand)) { c = cand; break; }
    }
    // a user who owns (almost) every item of the period: counted (the driver raises when it reads the counter), and
    // the element still carries a VALID item index -- the last candidate -- so nothing downstream gathers row -1
    if (c < 0) { atomicAdd(failed, 1); c = cand; }
    negs[e] = c;
}

// One shuffled pass over a period's rows, assembled ON THE DEVICE (the device form of a DataLoader(shuffle=True) pass over
// trainDataset_withPreSample / offlineDataset_withsample, data/dataset2.py:172-201, data/dataset.py:41-71): element e of
// the epoch is row perm(e), where perm is a COUNTER-BASED permutation of [0, n) -- a keyed 4-round Feistel network over
// the 2*hb-bit square that covers n, cycle-walked back into range (a Feistel network is a bijection of its domain; walking
// the cycles of a bijection until the value falls below n is a bijection of [0, n)).  No sort, no state: every element
// is computed independently from (seed, e), so every rank of a job derives the same epoch from the shared seed.
// out3[e] = (ui[r][0], ui[r][1], mat ? mat[r * stride + col] : untouched).
__device__ __forceinline__ uint64_t feistel_perm(uint64_t x, uint64_t n, int hb, uint64_t seed) {
    const uint64_t mask = (1ull << hb) - 1;
    do {
        uint64_t l = x >> hb, r = x & mask;
#pragma unroll
        for (int rd = 0; rd < 4; ++rd) {
            uint64_t z = r + seed + (uint64_t)(rd + 1) * 0x9e3779b97f4a7c15ull;
            z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
            z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
            z ^= z >> 31;
            const uint64_t t = l ^ (z & mask);
            l = r; r = t;
        }
        x = (l << hb) | r;
    } while (x >= n);
    return x;
}
template <typename E>
__global__ __launch_bounds__(256) void k_device_epoch(const int64_t* __restrict__ ui, const E* __restrict__ mat, int64_t stride, int64_t col,
                                                      int64_t n, int hb, uint64_t seed, int64_t* __restrict__ out3) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    const int64_t r = (int64_t)feistel_perm((uint64_t)e, (uint64_t)n, hb, seed);
    out3[3 * e] = ui[2 * r];
    out3[3 * e + 1] = ui[2 * r + 1];
    if (mat != nullptr) out3[3 * e + 2] = (int64_t)mat[r * stride + col];
}
// hits and NDCG sum over ranks (model/MF.py:60-78): hit iff rank < topk, NDCG = 1/log2(rank+2).
// One 1024-thread block; fixed reduction tree (deterministic).
__global__ __launch_bounds__(1024) void k_eval_metrics(const int32_t* __restrict__ rank, int64_t n, int topk,
                                                       float* __restrict__ out) {
    __shared__ float sh[2][16];
    float hits = 0.f, nd = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 1024) {
        const int rk = rank[i];
        if (rk < topk) { hits += 1.0f; nd += 1.0f / log2f((float)rk + 2.0f); }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { hits += __shfl_xor(hits, off, 64); nd += __shfl_xor(nd, off, 64); }
    if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = hits; sh[1][threadIdx.x >> 6] = nd; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float h = 0.f, d = 0.f;
        for (int w = 0; w < 16; ++w) { h += sh[0][w]; d += sh[1][w]; }
        out[0] = h; out[1] = d;
    }
}

#ifdef SML_TEST_PREP_REFERENCE        // (test build only: helper kernels of the library-sort reference path, tests/csrc)
#define SML_PREP_REF_DEVICE_PART
#include "../../tests/csrc/prep_cub_kernels.inc"
#undef SML_PREP_REF_DEVICE_PART
#endif
}  // namespace

// ---------------------------------------------------------------------------- launchers
#define SML_DISPATCH_D(d, ...)              \
    switch (d) {                             \
        case 32: { constexpr int DD = 32; __VA_ARGS__; } break;   \
        case 64: { constexpr int DD = 64; __VA_ARGS__; } break;   \
        case 128: { constexpr int DD = 128; __VA_ARGS__; } break; \
        default: return hipErrorInvalidValue; \
    }

hipError_t sml_launch_adaptive_users(int d, const float* xin, float* dx, int B, float beta, float* loss_slot, hipStream_t st) {
    if (B <= 0) return hipSuccess;
    SML_DISPATCH_D(d, k_adaptive_users<DD><<<dim3(1), dim3(1024), 0, st>>>(xin, dx, B, beta, loss_slot));
    return hipGetLastError();
}
hipError_t sml_launch_loss_finalize(const float* part, int n_batches, int stride, const int*, float* out, hipStream_t st) {
    k_loss_finalize<<<dim3(n_batches), dim3(256), 0, st>>>(part, n_batches, stride, out);
    return hipGetLastError();
}
hipError_t sml_launch_bare_grad(int d, int dtype_bytes, const SmlBareArgs& a, int* n_blocks, hipStream_t st) {
    const int lpr = d * dtype_bytes / 16;
    const int nb = (int)(((int64_t)a.B * lpr + 255) / 256);
    if (n_blocks) *n_blocks = nb;
    if (a.sched != nullptr) {            // lazy dense-Adam form (fp32 tables)
        if (dtype_bytes != 4) return hipErrorInvalidValue;
        SML_DISPATCH_D(d, k_bare_grad<DD, float, true><<<dim3(nb), dim3(256), 0, st>>>(a.tri, a.uniq, a.B, a));
    } else if (a.shard_tab != nullptr) {  // item-sharded form over peer mappings
        if (dtype_bytes == 4) { SML_DISPATCH_D(d, k_bare_grad<DD, float, false, true><<<dim3(nb), dim3(256), 0, st>>>(a.tri, a.uniq, a.B, a)); }
        else if (dtype_bytes == 2) { SML_DISPATCH_D(d, k_bare_grad<DD, __half, false, true><<<dim3(nb), dim3(256), 0, st>>>(a.tri, a.uniq, a.B, a)); }
        else return hipErrorInvalidValue;
    } else if (dtype_bytes == 4) {
        SML_DISPATCH_D(d, k_bare_grad<DD, float, false><<<dim3(nb), dim3(256), 0, st>>>(a.tri, a.uniq, a.B, a));
    } else if (dtype_bytes == 2) {
        SML_DISPATCH_D(d, k_bare_grad<DD, __half, false><<<dim3(nb), dim3(256), 0, st>>>(a.tri, a.uniq, a.B, a));
    } else return hipErrorInvalidValue;
    return hipGetLastError();
}
hipError_t sml_launch_head_apply(int d, int dtype_bytes, void* w_head, long long head_rows, float lr, const SmlPeerPoll& p, hipStream_t st) {
    if (head_rows <= 0) return hipSuccess;
    const int lpr = d * dtype_bytes / 16;
    const unsigned nb = (unsigned)((head_rows * lpr + 255) / 256);
    if (dtype_bytes == 4) { SML_DISPATCH_D(d, k_head_apply<DD, float><<<dim3(nb), dim3(256), 0, st>>>((float*)w_head, head_rows, lr, p)); }
    else if (dtype_bytes == 2) { SML_DISPATCH_D(d, k_head_apply<DD, __half><<<dim3(nb), dim3(256), 0, st>>>((__half*)w_head, head_rows, lr, p)); }
    else return hipErrorInvalidValue;
    return hipGetLastError();
}
// eight pointers from the kernel arguments into a device table (per-lane indexed by the sharded gradient pass)
struct SmlPtr8 { void* p[8]; };
__global__ void k_set_ptr_tab(void** dst, SmlPtr8 v) { if (threadIdx.x < 8) dst[threadIdx.x] = v.p[threadIdx.x]; }
hipError_t sml_launch_set_ptr_tab(void** dst, void* const* src, int n, hipStream_t st) {
    SmlPtr8 v;
    for (int q = 0; q < 8; ++q) v.p[q] = q < n ? src[q] : nullptr;
    k_set_ptr_tab<<<dim3(1), dim3(64), 0, st>>>(dst, v);
    return hipGetLastError();
}
hipError_t sml_launch_peer_signal(const SmlPeerPush& p, hipStream_t st) {
    k_peer_signal<<<dim3(1), dim3(64), 0, st>>>(p);
    return hipGetLastError();
}
hipError_t sml_launch_mark_runs(int key_bytes, const void* keys, const uint32_t* vals, int64_t n, int row_bits, SmlRun* rec,
                                uint8_t* flag_dup, uint8_t* uniq, int64_t uniq_stride, int64_t uniq_item_base, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    const dim3 grid((unsigned)((n + 255) / 256));
    if (key_bytes == 4) k_mark_runs<uint32_t><<<grid, dim3(256), 0, st>>>((const uint32_t*)keys, vals, n, row_bits, rec, flag_dup, uniq, uniq_stride, uniq_item_base);
    else k_mark_runs<uint64_t><<<grid, dim3(256), 0, st>>>((const uint64_t*)keys, vals, n, row_bits, rec, flag_dup, uniq, uniq_stride, uniq_item_base);
    return hipGetLastError();
}
// grid for a run kernel: one lane group per record, capped (the kernels stride)
static int run_grid(int64_t records, int lpr, int cap_blocks) {
    int64_t nb = (records * lpr + 255) / 256;
    if (nb < 1) nb = 1;
    return (int)(nb > cap_blocks ? cap_blocks : nb);
}
hipError_t sml_launch_run_adam(int d, const SmlRunArgs& a, int64_t max_records, hipStream_t st) {
    const int nb = run_grid(max_records, d / 4, 4096);
    SML_DISPATCH_D(d, k_run_update<DD, float, 1, false><<<dim3(nb), dim3(256), 0, st>>>(a.run_u, a.run_i, a.off_u, a.off_i, a.n_u, a.n_i, a.batch_index, a.known, a.hot_blocks, a));
    return hipGetLastError();
}
hipError_t sml_launch_run_sgd(int d, int dtype_bytes, const SmlRunArgs& a, int64_t max_records, hipStream_t st) {
    const int nb = run_grid(max_records, d * dtype_bytes / 16, 4096) + a.hot_blocks;
    // the hot-chunk reducer rides in the launch only when the epoch has hot runs (tried as its own launch on a
    // second stream beside this one: the fork/join cost more than the overlap gave, 2.45 -> 2.17 G triples/s).
    // Without it the epoch is light-tailed: shallower unrolls, half the registers, no LDS to speak of.
    const bool hotb = a.hot_blocks > 0;
    if (dtype_bytes == 4) {
        if (hotb) { SML_DISPATCH_D(d, k_run_update<DD, float, 0, true><<<dim3(nb), dim3(256), 0, st>>>(a.run_u, a.run_i, a.off_u, a.off_i, a.n_u, a.n_i, a.batch_index, a.known, a.hot_blocks, a)); }
        else { SML_DISPATCH_D(d, k_run_update<DD, float, 0, false><<<dim3(nb), dim3(256), 0, st>>>(a.run_u, a.run_i, a.off_u, a.off_i, a.n_u, a.n_i, a.batch_index, a.known, a.hot_blocks, a)); }
    } else if (dtype_bytes == 2) {
        if (hotb) { SML_DISPATCH_D(d, k_run_update<DD, __half, 0, true><<<dim3(nb), dim3(256), 0, st>>>(a.run_u, a.run_i, a.off_u, a.off_i, a.n_u, a.n_i, a.batch_index, a.known, a.hot_blocks, a)); }
        else { SML_DISPATCH_D(d, k_run_update<DD, __half, 0, false><<<dim3(nb), dim3(256), 0, st>>>(a.run_u, a.run_i, a.off_u, a.off_i, a.n_u, a.n_i, a.batch_index, a.known, a.hot_blocks, a)); }
    } else return hipErrorInvalidValue;
    return hipGetLastError();
}
hipError_t sml_launch_hot_apply(int d, int dtype_bytes, const SmlRunArgs& a, hipStream_t st) {
    const int nb_apply = (a.hot_cap + 3) / 4;              // one wavefront per hot row
    if (dtype_bytes == 4) {
        SML_DISPATCH_D(d, k_hot_apply<DD, float><<<dim3(nb_apply), dim3(256), 0, st>>>(a));
    } else if (dtype_bytes == 2) {
        SML_DISPATCH_D(d, k_hot_apply<DD, __half><<<dim3(nb_apply), dim3(256), 0, st>>>(a));
    } else return hipErrorInvalidValue;
    return hipGetLastError();
}
hipError_t sml_launch_adam_flush(int d, float* w, float* m, float* v, int32_t* last, int64_t rows,
                                 const SmlSched* sched, int cur_step, int sched_len, hipStream_t st) {
    const int lpr = d / 4;
    int64_t nb = (rows * lpr + 255) / 256;
    if (sched_len > 0 && nb > 2048) nb = 2048;          // closed form: the table is built per workgroup -- eight resident rounds of workgroups, grid-stride
    SML_DISPATCH_D(d, k_adam_flush<DD><<<dim3((unsigned)nb), dim3(256), 0, st>>>(w, m, v, last, rows, sched, cur_step, sched_len));
    return hipGetLastError();
}
hipError_t sml_launch_mf_forward(int d, const float* wu, const float* wi, const int64_t* user, const int64_t* item,
                                 int64_t n, int norm, float* uemb, float* iemb, float* score, hipStream_t st) {
    const int lpr = d / 4;
    const int64_t nb = (n * lpr + 255) / 256;
    SML_DISPATCH_D(d, k_mf_forward<DD><<<dim3((unsigned)nb), dim3(256), 0, st>>>(wu, wi, user, item, n, norm, uemb, iemb, score));
    return hipGetLastError();
}
hipError_t sml_launch_eval_ranks(int d, const float* wu, const float* wi, const int64_t* rows, int64_t n, int n_cols,
                                 int32_t* rank, hipStream_t st) {
    const int64_t nb = (n + 3) / 4;
    SML_DISPATCH_D(d, k_eval_ranks<DD><<<dim3((unsigned)nb), dim3(256), 0, st>>>(wu, wi, rows, n, n_cols, rank));
    return hipGetLastError();
}
hipError_t sml_launch_eval_bucketize(const int64_t* rows, int64_t n, int n_cols, int64_t n_item, int32_t* rows_out,
                                     int32_t* bucket_off, hipStream_t st) {
    k_eval_bucketize<<<dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st>>>(rows, n, n_cols, n_item, rows_out, bucket_off);
    return hipGetLastError();
}
// a11 save_MF_weight / evaluation snapshots: up to four table copies in ONE launch (a hipMemcpyAsync per table costs
// a blit launch plus the runtime's bookkeeping around it, ~30 us of stream time each at these sizes)
struct SmlCopyJobs { f32x4* dst[4]; const f32x4* src[4]; long long n16[4]; int n_jobs; };
__global__ __launch_bounds__(256) void k_copy_tables(SmlCopyJobs j) {
    const long long stride = (long long)gridDim.x * 256;
    for (int q = 0; q < j.n_jobs; ++q) {
        f32x4* __restrict__ d = j.dst[q];
        const f32x4* __restrict__ s = j.src[q];
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < j.n16[q]; i += stride)
            __builtin_nontemporal_store(__builtin_nontemporal_load(s + i), d + i);
    }
}
hipError_t sml_launch_copy_tables(int n_jobs, void* const* dst, const void* const* src, const long long* bytes, hipStream_t st) {
    SmlCopyJobs j;
    long long most = 0;
    j.n_jobs = n_jobs;
    for (int q = 0; q < 4; ++q) {
        j.dst[q] = q < n_jobs ? reinterpret_cast<f32x4*>(dst[q]) : nullptr;
        j.src[q] = q < n_jobs ? reinterpret_cast<const f32x4*>(src[q]) : nullptr;
        j.n16[q] = q < n_jobs ? bytes[q] / 16 : 0;
        if (j.n16[q] > most) most = j.n16[q];
    }
    if (most == 0) return hipSuccess;
    long long nb = (most + 1023) / 1024;            // four 16-byte chunks per thread and table
    if (nb > 4096) nb = 4096;
    k_copy_tables<<<dim3((unsigned)nb), dim3(256), 0, st>>>(j);
    return hipGetLastError();
}

// Device-side ordering between two streams without a cross-queue barrier packet: the signalling stream runs
// k_flag_set after the work to be waited for (in-order queue: that work is complete and released when the kernel
// starts), the waiting stream runs k_flag_wait before its dependent kernels (their start-of-kernel acquire then sees
// the data).  One lane polls a system-scope word.  flag[0] is the sequence word, flag[1] counts waiters that gave up
// after `timeout` 100-MHz ticks (a hang guard): a time-out is an INCIDENT the host finds in flag[1] -- it does not
// touch the sequence word, so every later waiter is still ordered behind its own signal.
__global__ void k_flag_set(int* flag, int value) {
    __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void k_flag_wait(int* flag, int value, long long timeout) {
    if (threadIdx.x != 0) return;
    const long long t0 = wall_clock64();
    for (;;) {
        const int v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (v >= value) return;
        if (wall_clock64() - t0 > timeout) { __hip_atomic_fetch_add(flag + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); return; }
        __builtin_amdgcn_s_sleep(32);
    }
}
// ---- one-shot exchange over peer mappings: generic push / wait / sum ------------------------------------------
// push: n16 16-byte chunks of `src` into the same offsets of every destination slot, then one counter increment per
// workgroup and destination.  The grid is a function of the size alone (every rank pushes the same size per step, so
// every counter grows by the same amount per step on every rank).
__global__ __launch_bounds__(256) void k_peer_push(const f32x4* __restrict__ src, long long n16, SmlPeerPush p) {
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) {
        const f32x4 v = src[i];
        for (int q = 0; q < p.world; ++q) peer_store16(p.dst[q] + 4 * i, v);
    }
    peer_signal(p);
}
__global__ __launch_bounds__(64) void k_peer_wait(SmlPeerPoll p) { peer_wait(p); }
// dst[i] = slot 0 [i] + slot 1 [i] + ... in rank order (the start-up self-check of the theta path)
// theta_net > 0: the buffer is a flat theta gradient (two nets of theta_net floats) -- the conv block's alignment padding
// is never pushed and may hold anything an earlier use of the slot left there (the start-up self-check's pattern): it
// leaves as zero, as in the flat gradient of the other carriers (k_grad_sumsq reads the whole buffer)
__global__ __launch_bounds__(256) void k_peer_sum(float* __restrict__ dst, long long n, SmlPeerPoll p, int theta_net) {
    peer_wait(p);
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float g = peer_load(p.slot0 + i);
    for (int q = 1; q < p.world; ++q) g += peer_load(p.slot0 + q * p.slot_stride + i);
    if (theta_net > 0) { const int off = (int)(i % theta_net); if (off < SML_OFF_F1W && !conv_slot_used_host(off)) g = 0.0f; }
    dst[i] = g;
}
// dst <- src in 16-byte chunks through system-scope (cache-bypassing) loads: how a rank reads memory another DEVICE writes
// (the start-up check of the item shards' visibility; the same load the sharded gradient pass uses for item rows)
__global__ __launch_bounds__(256) void k_peer_read(const float* __restrict__ src, f32x4* __restrict__ dst, long long n16) {
    const long long stride = (long long)gridDim.x * 256 * 2;
    for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 2; i < n16; i += stride) {
        f32x4 v[2];
        const long long j = i + 1 < n16 ? i + 1 : i;
        peer_load16x2(v, src + 4 * i, src + 4 * j);
        dst[i] = v[0];
        if (j != i) dst[j] = v[1];
    }
}
hipError_t sml_launch_peer_read(const void* src, void* dst, long long n16, hipStream_t st) {
    long long nb = (n16 + 511) / 512;
    nb = nb < 1 ? 1 : nb > 1024 ? 1024 : nb;
    k_peer_read<<<dim3((unsigned)nb), dim3(256), 0, st>>>(reinterpret_cast<const float*>(src), reinterpret_cast<f32x4*>(dst), n16);
    return hipGetLastError();
}
int sml_peer_push_blocks(long long n_floats) {
    long long nb = (n_floats / 4 + 1023) / 1024;      // four 16-byte chunks per thread
    return (int)(nb < 1 ? 1 : nb > 256 ? 256 : nb);
}
hipError_t sml_launch_peer_push(const float* src, long long n_floats, const SmlPeerPush& p, hipStream_t st) {
    k_peer_push<<<dim3(sml_peer_push_blocks(n_floats)), dim3(256), 0, st>>>(reinterpret_cast<const f32x4*>(src), n_floats / 4, p);
    return hipGetLastError();
}
hipError_t sml_launch_peer_wait(const SmlPeerPoll& p, hipStream_t st) {
    k_peer_wait<<<dim3(1), dim3(64), 0, st>>>(p);
    return hipGetLastError();
}
hipError_t sml_launch_peer_sum(float* dst, long long n_floats, const SmlPeerPoll& p, hipStream_t st, int theta_net) {
    k_peer_sum<<<dim3((unsigned)((n_floats + 255) / 256)), dim3(256), 0, st>>>(dst, n_floats, p, theta_net);
    return hipGetLastError();
}

hipError_t sml_launch_flag_set(int* flag, int value, hipStream_t st) {
    k_flag_set<<<dim3(1), dim3(1), 0, st>>>(flag, value);
    return hipGetLastError();
}
hipError_t sml_launch_flag_wait(int* flag, int value, long long timeout_ticks, hipStream_t st) {
    k_flag_wait<<<dim3(1), dim3(64), 0, st>>>(flag, value, timeout_ticks);
    return hipGetLastError();
}
hipError_t sml_launch_eval_ranks_bucketed(int d, const float* wu, const float* wi, const int32_t* rows_b,
                                          const int32_t* bucket_off, int64_t n, int n_cols, int32_t* rank, int max_blocks,
                                          hipStream_t st) {
    hipError_t e = hipMemsetAsync(rank, 0, (size_t)n * sizeof(int32_t), st);
    if (e != hipSuccess) return e;
    // persistent: at most four workgroups per CU (the pipelined walk holds 4 waves per SIMD), fewer on request
    int64_t nb = ((n + 3) / 4) * SML_EVB;
    const int64_t cap = max_blocks > 0 ? max_blocks : (d <= 64 ? 1024 : 0x7fffff00);
    if (nb > cap) nb = (cap / SML_EVB > 0 ? cap / SML_EVB : 1) * SML_EVB;
    SML_DISPATCH_D(d, k_eval_ranks_bucketed<DD><<<dim3((unsigned)nb), dim3(256), 0, st>>>(wu, wi, rows_b, bucket_off, n, n_cols, rank));
    return hipGetLastError();
}
// ---- LDS-sliced evaluation (d = 32): geometry, preparation, rank pass -------------------------------------------
int sml_evs_slices(int d, int64_t n_item) {
    if (d != 32 || n_item <= 0) return 0;
    const int64_t ns = (n_item + SML_EVS_S - 1) >> SML_EVS_SHIFT;
    return ns <= SML_EVS_NS_MAX ? (int)ns : 0;
}
hipError_t sml_launch_evs_prepare(const int64_t* rows, int64_t n, int n_cols, int ns, int32_t* whist, int32_t* mbcnt, int32_t* seg_off,
                                  uint32_t* entries, hipStream_t st) {
    const int n_mb = (int)((n + SML_EVS_MB - 1) / SML_EVS_MB);
    const size_t lds = (size_t)8 * ns * sizeof(int);
    k_evs_count<<<dim3((unsigned)n_mb), dim3(256), lds, st>>>(rows, n, n_cols, ns, n_mb, whist, mbcnt);
    k_evs_scan<<<dim3(1), dim3(1024), 0, st>>>(mbcnt, (int64_t)ns * n_mb, seg_off);
    k_evs_scatter<<<dim3((unsigned)n_mb), dim3(256), lds, st>>>(rows, n, n_cols, ns, n_mb, whist, seg_off, entries);
    return hipGetLastError();
}
hipError_t sml_launch_evs_ranks(int d, const float* wu, const float* wi, const int64_t* rows, const uint32_t* entries, const int32_t* seg_off,
                                int64_t n, int n_cols, int64_t n_item, int ns, float* ug, float* s0, uint16_t* partial, int32_t* rank,
                                int max_blocks, hipStream_t st) {
    if (d != 32) return hipErrorInvalidValue;
    const size_t lds = (size_t)SML_EVS_S * SML_EVS_STRIDE + (size_t)SML_EVS_WAVES * 64 * sizeof(int);
    {   // (per device, so set on every call: a process may drive several devices)
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_evs_ranks<32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    const int n_mb = (int)((n + SML_EVS_MB - 1) / SML_EVS_MB);
    k_evs_pre<32><<<dim3((unsigned)((n * 8 + 255) / 256)), dim3(256), 0, st>>>(wu, wi, rows, n, n_cols, ug, s0);
    // one workgroup per CU (its slice fills the LDS): `parts` workgroups share a slice's mini-blocks
    const int want = max_blocks > 0 ? max_blocks : 256;
    int parts = want / ns;
    if (parts < 1) parts = 1;
    if (parts > (n_mb + SML_EVS_WAVES - 1) / SML_EVS_WAVES) parts = (n_mb + SML_EVS_WAVES - 1) / SML_EVS_WAVES;
    k_evs_ranks<32><<<dim3((unsigned)(ns * parts)), dim3(SML_EVS_WAVES * 64), lds, st>>>(ug, s0, wi, entries, seg_off, n, n_item, n_mb, parts, partial);
    k_evs_sum<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(partial, n, (int64_t)n_mb * SML_EVS_MB, ns, rank);
    return hipGetLastError();
}
hipError_t sml_launch_sample_negatives(const int64_t* users, int64_t n, const int64_t* item_all, int64_t pop, const int64_t* user_ptr,
                                       int64_t n_users, const int64_t* user_items, uint64_t seed, int64_t* negs, int* failed,
                                       hipStream_t st) {
    if (n <= 0) return hipSuccess;
    k_sample_negatives<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st>>>(users, n, item_all, pop, user_ptr, n_users, user_items,
                                                                              seed, negs, failed);
    return hipGetLastError();
}
hipError_t sml_launch_device_epoch(const int64_t* ui, const void* mat, int elem_bytes, int64_t stride, int64_t col, int64_t n, uint64_t seed,
                                   int64_t* out3, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    int bits = 1;
    while (bits < 63 && (1ull << bits) < (uint64_t)n) ++bits;
    const int hb = (bits + 1) / 2 > 0 ? (bits + 1) / 2 : 1;
    const dim3 grid((unsigned)((n + 255) / 256));
    if (elem_bytes == 4) k_device_epoch<int32_t><<<grid, dim3(256), 0, st>>>(ui, (const int32_t*)mat, stride, col, n, hb, seed, out3);
    else k_device_epoch<int64_t><<<grid, dim3(256), 0, st>>>(ui, (const int64_t*)mat, stride, col, n, hb, seed, out3);
    return hipGetLastError();
}

hipError_t sml_launch_eval_metrics(const int32_t* rank, int64_t n, int topk, float* out, hipStream_t st) {
    k_eval_metrics<<<dim3(1), dim3(1024), 0, st>>>(rank, n, topk, out);
    return hipGetLastError();
}
#ifdef SML_TEST_PREP_REFERENCE
#define SML_PREP_REF_HOST_PART
#include "../../tests/csrc/prep_cub_kernels.inc"
#undef SML_PREP_REF_HOST_PART
#endif
