// Transfer-net kernels for gfx950: ConvTransfer_com / one_transfer of the reference
// (model/conv_transfer.py:18-50, 87-135) forward, backward-to-input, backward-to-theta
// and the theta Adam step, with fc1/fc2 on v_mfma_f32_32x32x2_f32 (exact fp32).
//
// Tiling: one 256-thread workgroup (4 waves, one per SIMD) carries a tile of SML_R = 32
// rows through the WHOLE net, so the per-coordinate 3->10->5 prologue, both GEMMs and
// the Gelu epilogues never leave the CU: activations live in LDS, weights stream from
// L2 as pre-arranged MFMA operand images (sml_dev.h), one 16-byte load per lane per
// four MFMAs.
#include "sml_dev.h"
#include "sml_kernels.h"

namespace {

__host__ __device__ constexpr bool conv_slot_used_host(int off) {
    return (off < 30) || (off >= SML_OFF_C1B && off < SML_OFF_C1B + 10) ||
           (off >= SML_OFF_C2W && off < SML_OFF_C2W + 50) || (off >= SML_OFF_C2B && off < SML_OFF_C2B + 5);
}

// ------------------------------------------------------------------------------------
// prologue shared by forward and backward: per-coordinate conv1 -> Gelu -> conv2
// (model/conv_transfer.py:38-44).  cw = the net's 104 conv floats staged in LDS (every
// lane reads the same address: a broadcast, no bank conflict).
// ------------------------------------------------------------------------------------
struct Pro {
    float h1p[SML_C1];
    float h1[SML_C1];
    float h2p[SML_C2];
};

__device__ __forceinline__ void conv_prologue(const float* cw, float x0, float x1, float x2, Pro& o) {
#pragma unroll
    for (int c = 0; c < SML_C1; ++c) {
        float s = cw[SML_OFF_C1B + c];
        s += cw[SML_OFF_C1W + c * 3 + 0] * x0;
        s += cw[SML_OFF_C1W + c * 3 + 1] * x1;
        s += cw[SML_OFF_C1W + c * 3 + 2] * x2;
        o.h1p[c] = s;
        o.h1[c] = sml_gelu(s);
    }
#pragma unroll
    for (int q = 0; q < SML_C2; ++q) {
        float s = cw[SML_OFF_C2B + q];
#pragma unroll
        for (int c = 0; c < SML_C1; ++c) s += cw[SML_OFF_C2W + q * SML_C1 + c] * o.h1[c];
        o.h2p[q] = s;
    }
}

__device__ __forceinline__ int64_t seg_row_index(const SmlSeg& s, int r) {
    if (s.tri == nullptr) return r;
    if (!s.is_item) return s.tri[(int64_t)r * 3];
    return r < s.B ? s.tri[(int64_t)r * 3 + 1] : s.tri[(int64_t)(r - s.B) * 3 + 2];
}

// acc[t] += A(32 x 8*NK, from LDS rows) * B(image tiles), with the B operand images prefetched
// PFD k-steps ahead in a register ring (fully unrolled, so every ring index is static).
//   arow : this lane's LDS row pointer (+ 4*hi), advanced 8 floats per k-step
//   bimg : image base; tile t, k-step ks lives at bimg[(tile_of(t) * ksteps_total + ks0 + ks) * 64 + lane]
template <int NT, int NK, int PFD, typename TileOf>
__device__ __forceinline__ void mma_rows(f32x16 (&acc)[NT], const float* arow, const f32x4* __restrict__ bimg,
                                         int ksteps_total, int ks0, int lane, TileOf tile_of) {
    f32x4 ring[PFD][NT];
#pragma unroll
    for (int i = 0; i < PFD && i < NK; ++i)
#pragma unroll
        for (int t = 0; t < NT; ++t) ring[i][t] = bimg[(tile_of(t) * ksteps_total + ks0 + i) * 64 + lane];
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
        const f32x4 av = *reinterpret_cast<const f32x4*>(arow + (ks0 + ks) * 8);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[t] = mfma32(av[e], ring[ks % PFD][t][e], acc[t]);
        if (ks + PFD < NK) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
                ring[ks % PFD][t] = bimg[(tile_of(t) * ksteps_total + ks0 + ks + PFD) * 64 + lane];
        }
    }
}

// ------------------------------------------------------------------------------------
// forward: rows -> out, optionally saving z1 / (x_t, x_hat, x_com) / a1 for backward.
// The z1 / xin / a1 scratch is padded to whole tiles by the caller, so those stores are
// unconditional; `out` may be a table (updata) and is bounds-checked.
// ------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void k_transfer_fwd(SmlFwdArgs a) {
    constexpr int K1 = SML_C2 * D;       // fc1 reduction length
    constexpr int S1 = K1 + 4;           // LDS row stride of A1 (S1/4 odd: conflict-free b128 reads)
    constexpr int S2 = SML_HID + 4;
    constexpr int KS1 = K1 / 8;
    constexpr int EPT = SML_R * D / 256;
    constexpr int JT = D / 32;
    __shared__ __attribute__((aligned(16))) float smem[SML_R * S1 + SML_R * S2 + 104];
    float* A1s = smem;
    float* a2s = smem + SML_R * S1;
    float* cws = smem + SML_R * S1 + SML_R * S2;
    float* xts = a2s;                    // [32][D+1], dead before a2s is written
    float* nrm = a2s + SML_R * (D + 1);
    float* part = smem;                  // [4][32][D+1], aliases A1s after fc1

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int sidx = (int)blockIdx.x >= a.tiles0;
    const SmlSeg& sg = a.seg[sidx];
    const int row0 = ((int)blockIdx.x - (sidx ? a.tiles0 : 0)) * SML_R;
    const float* __restrict__ theta = sg.theta;
    if (tid < 104) cws[tid] = theta[tid];

    // ---- P1: gather x_t and x_hat; all index loads, then all row loads, are in flight together
    float xt[EPT], xh[EPT];
    {
        int64_t idx[EPT];
        bool ok[EPT];
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const int row = row0 + (q * 256 + tid) / D;
            ok[q] = row < sg.n_rows;
            idx[q] = ok[q] ? seg_row_index(sg, row) : 0;
        }
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const int w = (q * 256 + tid) % D;
            xt[q] = sg.xt_tab[idx[q] * D + w];
            xh[q] = sg.xh_tab[idx[q] * D + w];
        }
        if (sg.last_tab != nullptr) {      // replay the row's pending zero-gradient Adam steps
            float m[EPT], v[EPT];
            int from[EPT];
#pragma unroll
            for (int q = 0; q < EPT; ++q) {
                const int w = (q * 256 + tid) % D;
                m[q] = sg.m_tab[idx[q] * D + w];
                v[q] = sg.v_tab[idx[q] * D + w];
                from[q] = sg.last_tab[idx[q]];
            }
#pragma unroll
            for (int q = 0; q < EPT; ++q) adam_replay(xh[q], m[q], v[q], from[q], a.cur_step - 1, a.sched);
        }
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const int e = q * 256 + tid;
            if (!ok[q]) { xt[q] = 1.0f; xh[q] = 0.0f; }
            xts[(e / D) * (D + 1) + (e % D)] = xt[q];
        }
    }
    __syncthreads();
    if (tid < SML_R) {
        float s = 0.0f;
#pragma unroll 8
        for (int w = 0; w < D; ++w) { const float t = xts[tid * (D + 1) + w]; s += t * t; }
        nrm[tid] = sqrtf(s);
    }
    __syncthreads();
    // ---- P2: x_com, conv1, Gelu, conv2, Gelu -> A1 tile (channel-major flatten c*D + w)
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
        const int e = q * 256 + tid, r = e / D, w = e % D;
        const int row = row0 + r;
        const float xc = (xt[q] * xh[q]) / nrm[r];     // no epsilon, as model/conv_transfer.py:99
        Pro p;
        conv_prologue(cws, xt[q], xh[q], xc, p);
#pragma unroll
        for (int c = 0; c < SML_C2; ++c) {
            const float v = sml_gelu(p.h2p[c]);
            A1s[r * S1 + c * D + w] = v;
            if (sg.a1 != nullptr) sg.a1[(int64_t)row * K1 + c * D + w] = v;
        }
        if (sg.xin != nullptr) {
            float* x = sg.xin + (int64_t)row * 3 * D;
            x[w] = xt[q];
            x[D + w] = xh[q];
            x[2 * D + w] = xc;
        }
    }
    __syncthreads();

    // ---- fc1: Z1[32 x 512] = A1[32 x K1] * W1^T ; wave wv owns n-tiles 4wv..4wv+3
    {
        f32x16 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[t][q] = 0.0f;
        mma_rows<4, KS1, 2>(acc, A1s + l31 * S1 + 4 * hi, reinterpret_cast<const f32x4*>(sg.pk + sml_pk_p1(D)),
                            KS1, 0, lane, [wv](int t) { return wv * 4 + t; });
        // + bias, save z1, Gelu -> a2 tile.  (xts/nrm are dead: every wave passed the barrier above)
        float* z1 = sg.z1;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int n = (wv * 4 + t) * 32 + l31;
            const float bias = theta[sml_off_f1b(D) + n];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int r = mfma32_row(q, lane);
                const float z = acc[t][q] + bias;
                if (z1 != nullptr) z1[(int64_t)(row0 + r) * SML_HID + n] = z;
                a2s[r * S2 + n] = sml_gelu(z);
            }
        }
    }
    __syncthreads();

    // ---- fc2: Out[32 x D] = a2[32 x 512] * W2^T ; the four waves split K = 512
    {
        f32x16 acc[JT];
#pragma unroll
        for (int t = 0; t < JT; ++t)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[t][q] = 0.0f;
        mma_rows<JT, 16, (JT == 1 ? 8 : 4)>(acc, a2s + l31 * S2 + 4 * hi,
                                            reinterpret_cast<const f32x4*>(sg.pk + sml_pk_p2(D)), 64, wv * 16, lane,
                                            [](int t) { return t; });
#pragma unroll
        for (int t = 0; t < JT; ++t)
#pragma unroll
            for (int q = 0; q < 16; ++q)
                part[(wv * SML_R + mfma32_row(q, lane)) * (D + 1) + t * 32 + l31] = acc[t][q];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
        const int e = q * 256 + tid, r = e / D, j = e % D;
        float s = theta[sml_off_f2b(D) + j];
#pragma unroll
        for (int w4 = 0; w4 < 4; ++w4) s += part[(w4 * SML_R + r) * (D + 1) + j];
        if (row0 + r < sg.n_rows) sg.out[(int64_t)(row0 + r) * D + j] = s;
    }
}

// ------------------------------------------------------------------------------------
// backward: dOut -> (MF stage) dx_hat + l2*x_hat, or (TR stage) dZ1 rows + conv-grad partials.
// dx / dz1 scratch is padded to whole tiles (unconditional stores).
// ------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void k_transfer_bwd(SmlBwdArgs a) {
    constexpr int K1 = SML_C2 * D;
    constexpr int S2 = SML_HID + 4;
    constexpr int SD = D + 4;
    constexpr int KSD = D / 8;
    constexpr int EPT = SML_R * D / 256;
    constexpr int KSPLIT = (D == 32) ? 4 : (D == 64 ? 2 : 1);   // waves along the n reduction
    constexpr int PSTR = K1 + 1;
    constexpr int SZ_A = SML_R * S2 + SML_R * SD;
    constexpr int SZ_B = KSPLIT * SML_R * PSTR;
    constexpr int SZ = SZ_A > SZ_B ? SZ_A : SZ_B;
    __shared__ __attribute__((aligned(16))) float smem[SZ + 104];
    __shared__ float red[4][104];
    float* dZs = smem;                    // [32][516]
    float* dOs = smem + SML_R * S2;       // [32][D+4]
    float* part = smem;                   // [KSPLIT][32][5D+1], aliases dZs after the second GEMM
    float* cws = smem + SZ;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int sidx = (int)blockIdx.x >= a.tiles0;
    const SmlBwdSeg& sg = a.seg[sidx];
    const int row0 = ((int)blockIdx.x - (sidx ? a.tiles0 : 0)) * SML_R;
    if (tid < 104) cws[tid] = sg.theta[tid];

#pragma unroll
    for (int q = 0; q < EPT; ++q) {
        const int e = q * 256 + tid, r = e / D, j = e % D;
        dOs[r * SD + j] = (row0 + r < sg.n_rows) ? sg.dout[(int64_t)(row0 + r) * D + j] : 0.0f;
    }
    // the (x_t, x_hat, x_com) rows of the tail: issue the loads now, use them after both GEMMs
    float x0[EPT], x1[EPT], x2[EPT];
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
        const int e = q * 256 + tid, r = e / D, w = e % D;
        const float* x = sg.xin + (int64_t)(row0 + r) * 3 * D;
        x0[q] = x[w]; x1[q] = x[D + w]; x2[q] = x[2 * D + w];
    }
    __syncthreads();
    // ---- dA2[32 x 512] = dOut[32 x D] * W2 ; dZ1 = dA2 * Gelu'(z1)
    {
        f32x16 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[t][q] = 0.0f;
        // z1 of this wave's 4 column tiles: issue with the GEMM, consume in its epilogue
        mma_rows<4, KSD, (KSD < 4 ? KSD : 4)>(acc, dOs + l31 * SD + 4 * hi,
                                              reinterpret_cast<const f32x4*>(sg.pk + sml_pk_p2b(D)), KSD, 0, lane,
                                              [wv](int t) { return wv * 4 + t; });
        float* dz1 = sg.dz1;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int n = (wv * 4 + t) * 32 + l31;
            float z[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) z[q] = sg.z1[(int64_t)(row0 + mfma32_row(q, lane)) * SML_HID + n];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int r = mfma32_row(q, lane);
                const float dz = acc[t][q] * sml_gelu_grad(z[q]);
                dZs[r * S2 + n] = dz;
                if (dz1 != nullptr) dz1[(int64_t)(row0 + r) * SML_HID + n] = dz;
            }
        }
    }
    __syncthreads();
    // ---- dA1[32 x 5D] = dZ1[32 x 512] * W1 ; waves = KSPLIT (reduction) x TSPLIT (5 tiles each)
    {
        const int kq = wv % KSPLIT, tq = wv / KSPLIT;
        constexpr int KPER = 64 / KSPLIT;      // k-steps (of 8) per wave
        f32x16 acc[5];
#pragma unroll
        for (int t = 0; t < 5; ++t)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[t][q] = 0.0f;
        mma_rows<5, KPER, 2>(acc, dZs + l31 * S2 + 4 * hi, reinterpret_cast<const f32x4*>(sg.pk + sml_pk_p1b(D)), 64,
                             kq * KPER, lane, [tq](int t) { return tq * 5 + t; });
        __syncthreads();                        // every wave is done reading dZs
#pragma unroll
        for (int t = 0; t < 5; ++t)
#pragma unroll
            for (int q = 0; q < 16; ++q)
                part[(kq * SML_R + mfma32_row(q, lane)) * PSTR + (tq * 5 + t) * 32 + l31] = acc[t][q];
    }
    __syncthreads();
    // ---- per-coordinate tail: Gelu'(h2) -> conv2^T -> Gelu'(h1) -> conv1^T (row 1 = x_hat)
    float cg[104];
    const bool want_cg = a.convg_part != nullptr;
    if (want_cg) {
#pragma unroll
        for (int i = 0; i < 104; ++i) cg[i] = 0.0f;
    }
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
        const int e = q * 256 + tid, r = e / D, w = e % D;
        const int row = row0 + r;
        const bool ok = row < sg.n_rows;
        Pro p;
        conv_prologue(cws, x0[q], x1[q], x2[q], p);
        float dh2p[SML_C2];
#pragma unroll
        for (int c = 0; c < SML_C2; ++c) {
            float s = 0.0f;
#pragma unroll
            for (int k = 0; k < KSPLIT; ++k) s += part[(k * SML_R + r) * PSTR + c * D + w];
            dh2p[c] = s * sml_gelu_grad(p.h2p[c]);
        }
        float dxh = 0.0f;
        float dh1p[SML_C1];
#pragma unroll
        for (int c = 0; c < SML_C1; ++c) {
            float s = 0.0f;
#pragma unroll
            for (int o = 0; o < SML_C2; ++o) s += dh2p[o] * cws[SML_OFF_C2W + o * SML_C1 + c];
            dh1p[c] = s * sml_gelu_grad(p.h1p[c]);
            dxh += dh1p[c] * cws[SML_OFF_C1W + c * 3 + 1];
        }
        if (sg.dx != nullptr) sg.dx[(int64_t)row * D + w] = dxh + a.l2 * x1[q];
        if (want_cg && ok) {
#pragma unroll
            for (int c = 0; c < SML_C1; ++c) {
                cg[SML_OFF_C1W + c * 3 + 0] += dh1p[c] * x0[q];
                cg[SML_OFF_C1W + c * 3 + 1] += dh1p[c] * x1[q];
                cg[SML_OFF_C1W + c * 3 + 2] += dh1p[c] * x2[q];
                cg[SML_OFF_C1B + c] += dh1p[c];
            }
#pragma unroll
            for (int o = 0; o < SML_C2; ++o) {
#pragma unroll
                for (int c = 0; c < SML_C1; ++c) cg[SML_OFF_C2W + o * SML_C1 + c] += dh2p[o] * p.h1[c];
                cg[SML_OFF_C2B + o] += dh2p[o];
            }
        }
    }
    if (want_cg) {
        // deterministic tree: lanes (xor shuffles), then waves in index order
#pragma unroll
        for (int i = 0; i < 104; ++i) {
            if (!conv_slot_used_host(i)) continue;
            float v = cg[i];
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
            if (lane == 0) red[wv][i] = v;
        }
        __syncthreads();
        if (tid < 104)
            a.convg_part[(int64_t)blockIdx.x * 104 + tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
    }
}

// ------------------------------------------------------------------------------------
// weight gradients (TR stage): dW1 = dZ1^T A1, db1, dW2 = dOut^T Gelu(z1), db2
// one workgroup per 32x32 output tile; the four waves split the batch rows
// ------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void k_transfer_wgrad(SmlWgArgs a) {
    constexpr int K1 = SML_C2 * D;
    constexpr int KT = K1 / 32;
    constexpr int JT = D / 32;
    constexpr int T1 = 16 * KT, T2 = JT * 16, TN = T1 + T2;
    __shared__ float part[4][32][33];
    __shared__ float csum[4][32];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int net = (int)blockIdx.x / TN;
    const int tl = (int)blockIdx.x % TN;
    const SmlWgSeg& sg = a.seg[net];
    const bool is_w1 = tl < T1;
    int ti, tj;                       // tile along output rows / cols
    if (is_w1) { ti = tl / KT; tj = tl % KT; } else { ti = (tl - T1) / 16; tj = (tl - T1) % 16; }
    const float* __restrict__ Asrc = is_w1 ? sg.dz1 : sg.dout;   // A[i][r] = Asrc[r][ti*32 + i]
    const int lda = is_w1 ? SML_HID : D;
    const float* __restrict__ Bsrc = is_w1 ? sg.a1 : sg.z1;      // B[r][j] = Bsrc[r][tj*32 + j]
    const int ldb = is_w1 ? K1 : SML_HID;
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.0f;
    float colsum = 0.0f;
    // each wave takes a contiguous quarter of the batch rows, 32 rows (4 k-steps) per trip with all
    // 32 operand loads of the trip in flight before its 16 MFMAs
    const int rows_per_wave = ((sg.n_rows + 127) / 128) * 32;
    const int r_begin = wv * rows_per_wave;
    const int r_end = min(sg.n_rows, r_begin + rows_per_wave);
    for (int rb = r_begin; rb < r_end; rb += 32) {
        float av[4][4], bv[4][4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = rb + s4 * 8 + 4 * hi + e;
                const bool ok = r < r_end;
                av[s4][e] = ok ? Asrc[(int64_t)r * lda + ti * 32 + l31] : 0.0f;
                bv[s4][e] = ok ? Bsrc[(int64_t)r * ldb + tj * 32 + l31] : 0.0f;
            }
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = rb + s4 * 8 + 4 * hi + e;
                float b = bv[s4][e];
                if (!is_w1) b = (r < r_end) ? sml_gelu(b) : 0.0f;
                colsum += av[s4][e];
                acc = mfma32(av[s4][e], b, acc);
            }
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) part[wv][mfma32_row(q, lane)][l31] = acc[q];
    colsum += __shfl_xor(colsum, 32, 64);
    if (lane < 32) csum[wv][lane] = colsum;
    __syncthreads();
    float* __restrict__ g = sg.grad;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int e = q * 256 + tid, i = e >> 5, j = e & 31;
        const float s = part[0][i][j] + part[1][i][j] + part[2][i][j] + part[3][i][j];
        if (is_w1) g[SML_OFF_F1W + (int64_t)(ti * 32 + i) * K1 + tj * 32 + j] = s;
        else g[sml_off_f2w(D) + (int64_t)(ti * 32 + i) * SML_HID + tj * 32 + j] = s;
    }
    if (tj == 0 && tid < 32) {
        const float s = csum[0][tid] + csum[1][tid] + csum[2][tid] + csum[3][tid];
        if (is_w1) g[sml_off_f1b(D) + ti * 32 + tid] = s;
        else g[sml_off_f2b(D) + ti * 32 + tid] = s;
    }
}

// ------------------------------------------------------------------------------------
// theta Adam (torch.optim.Adam with weight_decay added to the gradient,
// model/transfer.py:393, 728) + refresh of the MFMA operand images
// ------------------------------------------------------------------------------------
template <int D>
__device__ __forceinline__ void pack_store(float* __restrict__ pk, int off, float p) {
    constexpr int K1 = SML_C2 * D;
    if (off >= SML_OFF_F1W && off < sml_off_f1b(D)) {
        const int n = (off - SML_OFF_F1W) / K1, k = (off - SML_OFF_F1W) % K1;
        pk[sml_pk_p1(D) + pk_pos(K1 / 8, n, k)] = p;
        pk[sml_pk_p1b(D) + pk_pos(SML_HID / 8, k, n)] = p;
    } else if (off >= sml_off_f2w(D) && off < sml_off_f2b(D)) {
        const int j = (off - sml_off_f2w(D)) / SML_HID, n = (off - sml_off_f2w(D)) % SML_HID;
        pk[sml_pk_p2(D) + pk_pos(SML_HID / 8, j, n)] = p;
        pk[sml_pk_p2b(D) + pk_pos(D / 8, n, j)] = p;
    }
}

template <int D>
__global__ __launch_bounds__(256) void k_theta_pack(const float* __restrict__ theta, float* __restrict__ pk) {
    constexpr int NS = sml_net_size(D);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 2 * NS) return;
    const int net = i / NS, off = i % NS;
    pack_store<D>(pk + (int64_t)net * sml_pk_size(D), off, theta[i]);
}

__device__ __forceinline__ bool conv_slot_used(int off) { return conv_slot_used_host(off); }

template <int D>
__global__ __launch_bounds__(256) void k_theta_adam(SmlThetaAdamArgs a) {
    constexpr int NS = sml_net_size(D);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 2 * NS) return;
    const int net = i / NS, off = i % NS;
    float g;
    if (off < SML_OFF_F1W) {
        if (!conv_slot_used(off)) return;
        if (a.convg_part != nullptr) {
            g = 0.0f;
            const int t0 = net ? a.tiles0 : 0, t1 = net ? a.tiles_total : a.tiles0;
            for (int t = t0; t < t1; ++t) g += a.convg_part[(int64_t)t * 104 + off];
            a.grad[i] = g;                           // keep the flat gradient complete (all-reduce input)
        } else {
            g = a.grad[i];
        }
    } else {
        g = a.grad[i];
    }
    if (a.grad_only) return;
    float p = a.theta[i], m = a.m[i], v = a.v[i];
    g = g + a.weight_decay * p;
    SmlSched s; s.step_size = a.step_size; s.bc2_sqrt = a.bc2_sqrt;
    adam_apply(p, m, v, g, s);
    a.theta[i] = p; a.m[i] = m; a.v[i] = v;
    pack_store<D>(a.pk + (int64_t)net * sml_pk_size(D), off, p);
}

// ------------------------------------------------------------------------------------
// lane-map self test: D = A(32 x 16) * W(32 cols x 16)^T through the same operand paths
// ------------------------------------------------------------------------------------
__global__ void k_selftest(const float* __restrict__ A, const float* __restrict__ W, float* __restrict__ pk,
                           float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float As[32 * 20];
    const int lane = threadIdx.x, l31 = lane & 31, hi = lane >> 5;
    for (int e = lane; e < 32 * 16; e += 64) {
        As[(e / 16) * 20 + (e % 16)] = A[e];
        pk[pk_pos(2, e / 16, e % 16)] = W[e];       // W[col][red]
    }
    __syncthreads();
    f32x16 acc;
    for (int q = 0; q < 16; ++q) acc[q] = 0.0f;
    const f32x4* P = reinterpret_cast<const f32x4*>(pk);
    for (int ks = 0; ks < 2; ++ks) {
        const f32x4 av = *reinterpret_cast<const f32x4*>(As + l31 * 20 + ks * 8 + 4 * hi);
        const f32x4 bv = P[(0 * 2 + ks) * 64 + lane];
        for (int e = 0; e < 4; ++e) acc = mfma32(av[e], bv[e], acc);
    }
    for (int q = 0; q < 16; ++q) out[mfma32_row(q, lane) * 32 + l31] = acc[q];
}

}  // namespace

// ---------------------------------------------------------------------------- launchers
#define SML_DISPATCH_D(d, ...)              \
    switch (d) {                             \
        case 32: { constexpr int DD = 32; __VA_ARGS__; } break;   \
        case 64: { constexpr int DD = 64; __VA_ARGS__; } break;   \
        case 128: { constexpr int DD = 128; __VA_ARGS__; } break; \
        default: return hipErrorInvalidValue; \
    }

hipError_t sml_launch_fwd(int d, const SmlFwdArgs& a, int tiles_total, hipStream_t st) {
    if (tiles_total <= 0) return hipSuccess;
    SML_DISPATCH_D(d, k_transfer_fwd<DD><<<dim3(tiles_total), dim3(256), 0, st>>>(a));
    return hipGetLastError();
}
hipError_t sml_launch_bwd(int d, const SmlBwdArgs& a, int tiles_total, hipStream_t st) {
    if (tiles_total <= 0) return hipSuccess;
    SML_DISPATCH_D(d, k_transfer_bwd<DD><<<dim3(tiles_total), dim3(256), 0, st>>>(a));
    return hipGetLastError();
}
hipError_t sml_launch_wgrad(int d, const SmlWgArgs& a, hipStream_t st) {
    const int tn = 16 * (SML_C2 * d / 32) + (d / 32) * 16;
    SML_DISPATCH_D(d, k_transfer_wgrad<DD><<<dim3(2 * tn), dim3(256), 0, st>>>(a));
    return hipGetLastError();
}
hipError_t sml_launch_theta_adam(int d, const SmlThetaAdamArgs& a, hipStream_t st) {
    const int n = 2 * sml_net_size(d);
    SML_DISPATCH_D(d, k_theta_adam<DD><<<dim3((n + 255) / 256), dim3(256), 0, st>>>(a));
    return hipGetLastError();
}
hipError_t sml_launch_theta_pack(int d, const float* theta, float* pk, hipStream_t st) {
    const int n = 2 * sml_net_size(d);
    SML_DISPATCH_D(d, k_theta_pack<DD><<<dim3((n + 255) / 256), dim3(256), 0, st>>>(theta, pk));
    return hipGetLastError();
}
hipError_t sml_launch_selftest(const float* A, const float* W, float* pk, float* out, hipStream_t st) {
    k_selftest<<<dim3(1), dim3(64), 0, st>>>(A, W, pk, out);
    return hipGetLastError();
}
