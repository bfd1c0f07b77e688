// Transfer-net kernels for gfx950: ConvTransfer_com / one_transfer of the reference
// (model/conv_transfer.py:18-50, 87-135) forward, backward-to-input, backward-to-theta
// and the theta Adam step, with fc1/fc2 on v_mfma_f32_16x16x4_f32 (exact fp32).
//
// Tiling: one 512-thread workgroup (8 waves, two per SIMD so one wave's operand loads and
// VALU epilogues overlap the other's MFMAs) carries MT row-tiles of 16 rows through the
// WHOLE net, so the per-coordinate 3->10->5 prologue, both GEMMs and the Gelu epilogues
// never leave the CU: activations live in LDS, weights stream from L2 as pre-arranged MFMA
// operand images (sml_dev.h), one 16-byte load per lane per four MFMAs, prefetched through
// a small register ring.  MT = 1 for a training batch (a TR batch is only 768 rows: more,
// shorter tiles keep more CUs busy and halve the serial chain); MT = 2 for table-sized calls
// (updata), where reusing each weight fragment for 32 rows halves the L2 traffic.
#include <cstdlib>
#include "sml_dev.h"
#include "sml_kernels.h"

// In-kernel timeline (tools/timeline_probe.py): built only with -DSML_TIMELINE (SML_EXTRA_FLAGS of sml_amd/build.py) --
// even a null check at every stamp costs the TR step more than a microsecond.  Workgroup 0 and the last workgroup
// of every instrumented launch take a record of 16 stamps (100 MHz wall clock): [0] = kernel id * 2 + (last block).
#ifdef SML_TIMELINE
__device__ long long* g_timeline = nullptr;     // [0] = record counter, records from [16]
#ifndef SML_TL_SECOND
#define SML_TL_SECOND (gridDim.x - 1)      // the second workgroup that takes a record (default: the last one)
#endif
#define TL_BEGIN(KID) long long* tl_rec = nullptr; const int tl_all = 70000 + (KID) * 1024; { if (g_timeline && threadIdx.x == 0) g_timeline[tl_all + 512 + blockIdx.x] = wall_clock64(); } { long long* tl = g_timeline; if (tl && threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == SML_TL_SECOND)) { \
    const unsigned long long slot = atomicAdd(reinterpret_cast<unsigned long long*>(tl), 1ull); tl_rec = tl + 16 + slot * 16; tl_rec[0] = (KID) * 2 + (blockIdx.x != 0); tl_rec[1] = wall_clock64(); for (int q = 2; q < 16; ++q) tl_rec[q] = 0; } }
#define TL(i) do { if (tl_rec) tl_rec[(i)] = wall_clock64(); } while (0)
// every workgroup's end joins a running maximum ([2]); workgroup 0 of the next launch copies it into its stamp 8:
// stamp 1 - stamp 8 = the true gap between the two launches (last workgroup out -> first workgroup in)
#define TL_DONE() do { long long* tl = g_timeline; if (tl && threadIdx.x == 0) { const long long now_ = wall_clock64(); \
    atomicMax(reinterpret_cast<unsigned long long*>(tl + 2), (unsigned long long)now_); tl[tl_all + blockIdx.x] = now_; } } while (0)   /* per-workgroup end: the LAST launch of each kernel survives */
#define TL_PREV() do { if (tl_rec) tl_rec[8] = (long long)__hip_atomic_load(reinterpret_cast<unsigned long long*>(g_timeline + 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while (0)
#else
#define TL_BEGIN(KID) const int tl_all = 0; (void)tl_all; long long* const tl_rec = nullptr; (void)tl_rec
#define TL(i) do { } while (0)
#define TL_DONE() do { } while (0)
#define TL_PREV() do { } while (0)
#endif
// How a kernel's outputs leave the CU.  The gap between two dependent launches grows with the bytes the first one
// leaves DIRTY in the L2s (measured with the timeline: about 1 us + 1 us per MB -- the end-of-kernel release walks and
// writes back every dirty line before the next dispatch may start).  Outputs that the next launch reads from OTHER
// XCDs anyway are therefore stored write-through (agent scope, `sc1`): they are on their way to memory while the
// kernel is still computing, and the release finds little left.  MODE 0 plain, 1 nontemporal (measured: no effect),
// 2 sc1.  (Per-dword sc1 stores cost more per byte than plain ones: the weight-gradient kernel's theta / m / v stores
// stay plain -- measured slower with sc1 -- until they are 16-byte vectors.)
#ifndef SML_WT_FWD
#define SML_WT_FWD 2
#endif
#ifndef SML_WT_BWD
#define SML_WT_BWD 2
#endif
#ifndef SML_WT_MFB
#define SML_WT_MFB 2
#endif
#ifndef SML_WT_FWD_LOCAL
#define SML_WT_FWD_LOCAL 0
#endif
__device__ __forceinline__ void st_out16_wt(float* p, const f32x4& v) {       // one 16-byte write-through store
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");    // (s_nop: see peer_store16)
}
template <int MODE>
__device__ __forceinline__ void st_out(float* p, float v) {
    if constexpr (MODE == 1) __builtin_nontemporal_store(v, p);
    else if constexpr (MODE == 2) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
namespace {

struct Pro {
    float h1p[SML_C1];
    float h1[SML_C1];
    float h2p[SML_C2];
};

// per-coordinate conv1 -> Gelu -> conv2 (model/conv_transfer.py:38-44).  cw = the net's 104
// conv floats staged in LDS (all lanes read one address: a broadcast).
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

// acc[mt][t] += A(MT x 16 rows x 16*NK, LDS) * B(image tiles).  B operand images are prefetched
// PFD k-steps ahead in a register ring.  The k loop runs in chunks of PFD steps: the chunk body is
// unrolled (static ring indices), the chunk loop is NOT, which bounds the live weight registers to the
// ring (a fully unrolled loop lets the compiler hoist every load of a 32-step reduction to the top).
//   arow : this lane's LDS pointer (row l&15, column 4*(l>>4)); M-tile mt is a_mt floats further
//   bimg : tile t, k-step ks at bimg[(tile_of(t) * ksteps_total + ks0 + kofs(t) + ks) * 64 + lane]
//   SPLITK: the NT "tiles" are k-ranges of ONE column tile (kofs(t) differs): independent accumulator
//   chains for a wave that owns a single column tile; the caller adds the NT accumulators at the end.
// first PFD k-steps of every tile's operand image into the ring.  Separate from the product so a kernel can
// issue these loads (they depend on nothing but theta) at its very top, under its gather / pair-loss stage
template <int NT, int PFD, typename TileOf, typename KOfs>
__device__ __forceinline__ void ring_preload(f32x4 (&ring)[PFD][NT], const f32x4* __restrict__ bimg, int ksteps_total, int ks0,
                                             int lane, TileOf tile_of, KOfs kofs) {
#pragma unroll
    for (int i = 0; i < PFD; ++i)
#pragma unroll
        for (int t = 0; t < NT; ++t) ring[i][t] = bimg[(tile_of(t) * ksteps_total + ks0 + kofs(t) + i) * 64 + lane];
    // the ring IS the prefetch distance: without this the scheduler sinks most of these loads in between the
    // products (seen in the ISA: one or two operand loads in flight per wave instead of PFD * NT), and a
    // workgroup's weight stream runs at a tenth of its CU's L2 port.  (VALU, SALU and LDS may still cross.)
    __builtin_amdgcn_sched_barrier(0x86);
}

template <int MT, int NT, int NK, int PFD, bool SPLITK, typename TileOf, typename KOfs>
__device__ __forceinline__ void mma16_ring(f32x4 (&acc)[MT][NT], f32x4 (&ring)[PFD][NT], const float* arow, int a_mt,
                                           const f32x4* __restrict__ bimg, int ksteps_total, int ks0, int lane,
                                           TileOf tile_of, KOfs kofs) {
    static_assert(PFD >= 1 && PFD <= NK, "ring depth");
    constexpr int NCH = NK / PFD, REM = NK % PFD;
    constexpr int NA = SPLITK ? NT : 1;
    // the A fragments (LDS) of step k+1 are fetched BEFORE the products of step k: an LDS round trip per k-step is
    // otherwise exposed in front of every burst of MFMAs (the ISA had ds_read_b128; s_waitcnt lgkmcnt(0); v_mfma ...)
    f32x4 av[NA][MT];
    auto load_a = [&](int ks, f32x4 (&dst)[NA][MT]) {
#pragma unroll
        for (int u = 0; u < NA; ++u)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                dst[u][mt] = *reinterpret_cast<const f32x4*>(arow + mt * a_mt + (ks0 + (SPLITK ? kofs(u) : 0) + ks) * 16);
    };
    load_a(0, av);
    auto step = [&](int ks, int slot, bool refill) {
        f32x4 an[NA][MT];
        const int kn = ks + 1 < NK ? ks + 1 : ks;          // (the last step re-reads its own fragment: no branch)
        load_a(kn, an);
        // e-major: consecutive products go to DIFFERENT accumulators (a product that reads the accumulator the previous one
        // writes cannot issue until that one has left the pipe)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt][t] = mfma16(av[SPLITK ? t : 0][mt][e], ring[slot][t][e], acc[mt][t]);
        if (refill) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
                ring[slot][t] = bimg[(tile_of(t) * ksteps_total + ks0 + kofs(t) + ks + PFD) * 64 + lane];
        }
        __builtin_amdgcn_sched_barrier(0x86);      // refills stay behind their step's products, ahead of the next step's
#pragma unroll
        for (int u = 0; u < NA; ++u)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) av[u][mt] = an[u][mt];
    };
#pragma unroll 1
    for (int ch = 0; ch < NCH; ++ch) {
#pragma unroll
        for (int j = 0; j < PFD; ++j) step(ch * PFD + j, j, ch * PFD + j + PFD < NK);
    }
#pragma unroll
    for (int j = 0; j < REM; ++j) step(NCH * PFD + j, j, false);
}

template <int MT, int NT, int NK, int PFD, bool SPLITK, typename TileOf, typename KOfs>
__device__ __forceinline__ void mma16_rows_k(f32x4 (&acc)[MT][NT], const float* arow, int a_mt,
                                             const f32x4* __restrict__ bimg, int ksteps_total, int ks0, int lane,
                                             TileOf tile_of, KOfs kofs) {
    f32x4 ring[PFD][NT];
    ring_preload<NT, PFD>(ring, bimg, ksteps_total, ks0, lane, tile_of, kofs);
    mma16_ring<MT, NT, NK, PFD, SPLITK>(acc, ring, arow, a_mt, bimg, ksteps_total, ks0, lane, tile_of, kofs);
}
template <int MT, int NT, int NK, int PFD, typename TileOf>
__device__ __forceinline__ void mma16_rows(f32x4 (&acc)[MT][NT], const float* arow, int a_mt,
                                           const f32x4* __restrict__ bimg, int ksteps_total, int ks0, int lane,
                                           TileOf tile_of) {
    mma16_rows_k<MT, NT, NK, PFD, false>(acc, arow, a_mt, bimg, ksteps_total, ks0, lane, tile_of, [](int) { return 0; });
}

template <int MT, int NT>
__device__ __forceinline__ void zero_acc(f32x4 (&acc)[MT][NT]) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[mt][t][q] = 0.0f;
}

constexpr int cmax(int a, int b) { return a > b ? a : b; }

// ------------------------------------------------------------------------------------
// forward: rows -> out, optionally saving z1 / (x_t, x_hat, x_com) / a1 for backward.
// The z1 / xin / a1 scratch is padded to whole tiles by the caller, so those stores are
// unconditional; `out` may be a table (updata) and is bounds-checked.
//
// NS > 1 (training batches): the 512 hidden units are split over NS workgroups per row tile, so a
// 768-row TR batch becomes 192 workgroups instead of 48 and the fc1 chain per workgroup is NS
// times shorter.  Workgroup (tile, h) repeats the cheap gather + conv prologue, computes
// z1[:, h*HL .. (h+1)*HL), and writes the fc2 PARTIAL sum over its hidden slice to plane h of
// `out` (planes out_pstride floats apart; the bias rides in plane 0).  The consumer (the pair loss
// at the head of k_transfer_bwd) adds the NS planes in index order: deterministic, no atomics.
// ------------------------------------------------------------------------------------
// HSEQ > 1 (table-sized calls): the workgroup walks its hidden units in HSEQ sequential passes -- fc1 of a pass, Gelu, then
// that pass's share of fc2 accumulated in registers -- so the a2 tile in LDS holds 512 / HSEQ columns: a 32-row workgroup
// then needs 68 KB of LDS and <= 128 VGPRs, TWO fit a CU, and one's gather / conv prologue / epilogues run under the
// other's MFMA phases (a single 48-row workgroup per CU left the matrix pipe idle 45 % of the time).
// Round 6 -- the operands of the kernel's FIRST round of loads are leading scalar parameters (16 SGPRs: built with
// -amdgpu-kernarg-preload-count=16 they arrive with the wavefront; everything else in the by-value struct comes behind one
// scalar-load round trip of the argument segment, 0.2-0.45 us that the operand rings, the index loads and the pending conv step
// no longer wait for): the kernel patches its copy of the struct with them and the body reads the struct as before.  The
// launcher (sml_launch_fwd) fills them from the struct and checks the layout the patch assumes: segment 1 = segment 0 advanced
// by one net (theta + net size, images + image size), the same triples and batch length, rows of segment 1 = the item columns.
// p_cg = cg_split | cg_total << 15.
// (14 dwords is what the hardware preloads: 16 user SGPRs less the argument-segment pointer.)  p_tiles = tiles0 | (segment 1 present: its ceil(2 B / 16) tiles follow) << 31,
// p_cg = cg_split | cg_total << 15; rows of segment 1: 2 B -- rounded up to whole tiles when segment 0's rows are (the distinct-row form).
#define SML_FWD_HOT_PARAMS const float* __restrict__ p_pk, const float* __restrict__ p_theta, const int64_t* __restrict__ p_tri,             \
                           const float* __restrict__ p_cs_in, const float* __restrict__ p_cg_part, int p_B, int p_n0, int p_tiles, int p_cg
#define SML_FWD_HOT_PATCH(a, D)                                                                                                               \
    (a).tiles0 = p_tiles & 0x7fffffff; (a).tiles_total = (p_tiles & 0x7fffffff) + (p_tiles < 0 ? (2 * p_B + SML_TM - 1) / SML_TM : 0);       \
    (a).cs_in = p_cs_in; (a).cg_part = p_cg_part; (a).cg_split = p_cg & 0x7fff; (a).cg_total = (p_cg >> 15) & 0xffff;                         \
    (a).seg[0].pk = p_pk; (a).seg[1].pk = p_pk + sml_pk_size(D); (a).seg[0].theta = p_theta; (a).seg[1].theta = p_theta + sml_net_size(D); \
    (a).seg[0].tri = p_tri; (a).seg[1].tri = p_tri; (a).seg[0].B = p_B; (a).seg[1].B = p_B; (a).seg[0].n_rows = p_n0;                        \
    (a).seg[1].n_rows = p_n0 == p_B ? 2 * p_B : ((2 * p_B + SML_TM - 1) / SML_TM) * SML_TM;                                                   \
    (a).seg[0].is_item = 0; (a).seg[1].is_item = 1
template <int D, int MT, int NS, int HSEQ = 1>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(HSEQ > 1 ? 4 : 2, HSEQ > 1 ? 4 : 2))) void k_transfer_fwd(SML_FWD_HOT_PARAMS, SmlFwdArgs a_in) {
    SmlFwdArgs a = a_in;
    SML_FWD_HOT_PATCH(a, D);
#include "transfer_fwd_body.inc"
}
// The same table-sized forward under a name of its own: launched by the EVALUATION stream's context (eval_submit_transferred:
// forwards that only an evaluation reads, on that stream's 64 CUs), so that kernel traces and the per-class timing tell it from
// the training stream's launches -- it runs at a third of their rate by design and is not on the critical path.
template <int D>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_side_transfer_fwd(SmlFwdArgs a) {
    constexpr int MT = 2, NS = 1, HSEQ = 2;
#include "transfer_fwd_body.inc"
}

// ------------------------------------------------------------------------------------
// Round 5: the TABLE-SIZED forward (updata, model/transfer.py:884-902: every row of both tables through its net) on the BF16
// matrix rate with fp32-grade results.  fc1 / fc2 are the one place of the period where the matrix pipe itself is the bound
// (k_transfer_fwd<32,2,1,2>: 61 % MFMA-busy on its partition), and gfx950 runs v_mfma_f32_16x16x32_bf16 at 16 times the rate of
// v_mfma_f32_16x16x4_f32 (no xf32 / tf32 forms on this part).  Every fp32 operand is split EXACTLY into three bf16 terms,
//     x = x1 + x2 + x3 (+ a remainder below 2^-26 |x|):   x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2)
// (the subtractions are exact in fp32), and a product a*b is the six bf16 products whose weight is at least 2^-16 of it,
//     a1 b1 + a1 b2 + a2 b1 + a2 b2 + a1 b3 + a3 b1,   accumulated in fp32 by the matrix core
// -- what is dropped (a2 b3, a3 b2, a3 b3) is below 2^-24 |a b|, i.e. below the rounding of ONE fp32 product.  Six bf16 products
// of K = 32 replace eight fp32 products of K = 4: 2.5 times fewer matrix-pipe cycles, for 1.5 times the operand bytes.  The
// weights' three planes are laid out per (tile, k-step, plane, lane) once per call (k_theta_pack_bx3), the activations are split
// when they are written to LDS.  d = 32, rows through identity indexing, no saves (the training forwards keep the fp32 products:
// they are latency-bound, and z1 feeds a backward that is pinned to them).  Against k_transfer_fwd<32,2,1,2> on the same rows:
// max relative difference ~1e-6 (tests).  SML_FWD_BX3=0: the fp32 kernel.
// ------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x4 mfma_bf(const uint4& a, const uint4& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ uint32_t bf16_rne(float x) {          // round to nearest even (a NaN stays a NaN: 0x7fc0 + carry-free)
    uint32_t u = __float_as_uint(x);
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}
__device__ __forceinline__ void split3(float x, unsigned short& h, unsigned short& m, unsigned short& l) {
    const uint32_t a = bf16_rne(x);
    const float r1 = x - __uint_as_float(a << 16);
    const uint32_t b = bf16_rne(r1);
    const float r2 = r1 - __uint_as_float(b << 16);
    h = (unsigned short)a; m = (unsigned short)b; l = (unsigned short)bf16_rne(r2);
}
// operand images of one net: P1x [32 column tiles][K1/32 k-steps][3 planes][64 lanes][8 bf16], P2x [D/16][16][3][64][8]
__host__ __device__ constexpr int sml_bx3_p1(int d) { return 0; }
__host__ __device__ constexpr int sml_bx3_p2(int d) { return 32 * (SML_C2 * d / 32) * 3 * 512; }                 // (ushorts)
__host__ __device__ constexpr int sml_bx3_size(int d) { return sml_bx3_p2(d) + (d / 16) * 16 * 3 * 512; }
template <int D>
__global__ __launch_bounds__(256) void k_theta_pack_bx3(const float* __restrict__ theta, unsigned short* __restrict__ pkx) {
    constexpr int K1 = SML_C2 * D, KS1 = K1 / 32, NF1 = SML_HID * K1, NF2 = D * SML_HID;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 2 * (NF1 + NF2)) return;
    const int net = i / (NF1 + NF2), e = i % (NF1 + NF2);
    const float* th = theta + (int64_t)net * sml_net_size(D);
    unsigned short* out = pkx + (int64_t)net * sml_bx3_size(D);
    float x; int base;
    if (e < NF1) {                                               // fc1.weight[n][k]: column n, reduction k
        const int n = e / K1, k = e % K1;
        x = th[SML_OFF_F1W + e];
        base = sml_bx3_p1(D) + (((n >> 4) * KS1 + (k >> 5)) * 3 * 64 + ((n & 15) + 16 * ((k >> 3) & 3))) * 8 + (k & 7);
    } else {                                                     // fc2.weight[j][n]: column j, reduction n
        const int f = e - NF1, j = f / SML_HID, n = f % SML_HID;
        x = th[sml_off_f2w(D) + f];
        base = sml_bx3_p2(D) + (((j >> 4) * 16 + (n >> 5)) * 3 * 64 + ((j & 15) + 16 * ((n >> 3) & 3))) * 8 + (n & 7);
    }
    unsigned short h, m, l;
    split3(x, h, m, l);
    out[base] = h; out[base + 512] = m; out[base + 1024] = l;    // (planes are 64 lanes * 8 = 512 entries apart)
}

template <int D, bool SIDE>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_transfer_fwd_bx3(SmlFwdArgs a, const unsigned short* __restrict__ pkx) {
    static_assert(D == 32, "bf16x3 table-sized forward: d = 32");
    constexpr int R = 32, MT = 2;                                // rows per workgroup
    constexpr int K1 = SML_C2 * D, KS1 = K1 / 32;                // fc1 reduction: 160 = 5 k-steps of 32
    constexpr int HSEQ = 4, HL = SML_HID / HSEQ;                 // hidden units per pass: 128 = one column tile per wave
    constexpr int S1B = K1 + 8, S2B = HL + 8;                    // LDS row strides (bf16 elements; rows stay 16-byte aligned)
    constexpr int EPT = R * D / 512;
    constexpr int JT = D / 16;                                   // fc2 column tiles (2)
    constexpr int KSPL = 8 / JT;                                 // fc2: waves along K (4: one k-step of the pass each) x JT column tiles
    static_assert(HL / 32 == KSPL && HL / 16 == 8, "one fc1 column tile and one fc2 k-step per wave and pass");
    __shared__ __attribute__((aligned(16))) unsigned short A1b[3][R][S1B];      // the A1 tile's three bf16 planes
    __shared__ __attribute__((aligned(16))) unsigned short A2b[3][R][S2B];      // Gelu(z1) of the current pass
    __shared__ float cws[104];
    float* part = reinterpret_cast<float*>(&A2b[0][0][0]);       // [KSPL][R][D + 1] fc2 partials, after the last pass
    static_assert(sizeof(A2b) >= KSPL * R * (D + 1) * sizeof(float), "part fits");
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, g4 = lane >> 4;
    const int tile = (int)blockIdx.x;
    const SmlSeg sg = a.seg[0];
    const int row0 = tile * R;
    const float* __restrict__ theta = sg.theta;
    if (tid < 104) cws[tid] = theta[tid];
    const unsigned short* __restrict__ p1x = pkx + sml_bx3_p1(D);
    const unsigned short* __restrict__ p2x = pkx + sml_bx3_p2(D);
    // fc1 operand planes (this wave's column tile of the pass; 3 planes x 16 bytes per lane and k-step): a two-deep ring over the
    // HSEQ * KS1 steps of the whole kernel -- step s + 1 is fetched while step s multiplies (static indices: the loops are unrolled)
    uint4 bw[2][3];
    auto load_b1 = [&](int s_, uint4 (&dst)[3]) {
        const int hp_ = s_ / KS1, ks_ = s_ % KS1;
        const int T = hp_ * (HL / 16) + wv;
#pragma unroll
        for (int p = 0; p < 3; ++p)
            dst[p] = *reinterpret_cast<const uint4*>(p1x + ((((int64_t)T * KS1 + ks_) * 3 + p) * 64 + lane) * 8);
    };
    // ---- gather x_t, x_hat (identity indexing: table-sized calls pass the tables themselves)
    float xt[EPT], xh[EPT], nr2[EPT];
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
        const int e = q * 512 + tid, r = e / D, w = e % D;
        const int row = row0 + r;
        const bool ok = row < sg.n_rows;
        xt[q] = ok ? sg.xt_tab[(int64_t)row * D + w] : 1.0f;
        xh[q] = ok ? sg.xh_tab[(int64_t)row * D + w] : 0.0f;
    }
    load_b1(0, bw[0]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
        float s2 = xt[q] * xt[q];
#pragma unroll
        for (int off = D / 2; off >= 1; off >>= 1) s2 += __shfl_xor(s2, off, 64);
        nr2[q] = s2;
    }
    __syncthreads();                                             // cws
    // ---- x_com, conv1, Gelu, conv2, Gelu -> A1 (channel-major flatten c*D + w), split into its three planes
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
        __builtin_amdgcn_sched_barrier(0);                       // (one element's prologue at a time: interleaved they spilled)
        const int e = q * 512 + tid, r = e / D, w = e % D;
        const float xc = a.k2 ? 0.0f : (xt[q] * xh[q]) / sqrtf(nr2[q]);        // no epsilon, as model/conv_transfer.py:99
        Pro p;
        conv_prologue(cws, xt[q], xh[q], xc, p);
#pragma unroll
        for (int c = 0; c < SML_C2; ++c) {
            unsigned short h, m, l;
            split3(sml_gelu(p.h2p[c]), h, m, l);
            A1b[0][r][c * D + w] = h; A1b[1][r][c * D + w] = m; A1b[2][r][c * D + w] = l;
        }
    }
    float bias2[EPT];
#pragma unroll
    for (int q = 0; q < EPT; ++q) bias2[q] = theta[sml_off_f2b(D) + (q * 512 + tid) % D];
    __syncthreads();
    const int kq = wv % KSPL, jq = wv / KSPL;                    // fc2: this wave's k-step of the pass and its column tile
    f32x4 acc2[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc2[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int hp = 0; hp < HSEQ; ++hp) {
        // fc2 operand planes of this pass (global k-step hp * KSPL + kq, column tile jq), and this pass's bias
        uint4 b2[3];
#pragma unroll
        for (int p = 0; p < 3; ++p)
            b2[p] = *reinterpret_cast<const uint4*>(p2x + ((((int64_t)jq * 16 + hp * KSPL + kq) * 3 + p) * 64 + lane) * 8);
        const float bias1 = theta[sml_off_f1b(D) + hp * HL + wv * 16 + l15];
        // ---- fc1: Z1[R x 16] of this wave's column tile; six bf16 products per (row tile, k-step)
        f32x4 acc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) {
            const int st_ = hp * KS1 + ks;                       // (compile-time: both loops are unrolled)
            if (st_ + 1 < HSEQ * KS1) load_b1(st_ + 1, bw[(st_ + 1) & 1]);
            uint4 av[MT][3];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    av[mt][p] = *reinterpret_cast<const uint4*>(&A1b[p][mt * 16 + l15][ks * 32 + 8 * g4]);
            // (plane pairs in ascending weight: the small terms first; consecutive products go to different accumulators)
#pragma unroll
            for (int pp = 0; pp < 6; ++pp) {
                constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[mt] = mfma_bf(av[mt][PA[pp]], bw[st_ & 1][PB[pp]], acc[mt]);
            }
            __builtin_amdgcn_sched_barrier(0);                   // (the unrolled steps stay steps: hoisted operand reads were spilling)
        }
        if (hp > 0) __syncthreads();                             // the previous pass's fc2 is done with the a2 tile
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = mt * 16 + 4 * g4 + q;
                unsigned short h, m, l;
                split3(sml_gelu(acc[mt][q] + bias1), h, m, l);
                A2b[0][r][wv * 16 + l15] = h; A2b[1][r][wv * 16 + l15] = m; A2b[2][r][wv * 16 + l15] = l;
            }
        __syncthreads();
        // ---- fc2: Out[R x 16 (tile jq)] += a2[R x 32 (k-step kq of the pass)] * W2^T
        {
            uint4 av[MT][3];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int p = 0; p < 3; ++p)
                    av[mt][p] = *reinterpret_cast<const uint4*>(&A2b[p][mt * 16 + l15][kq * 32 + 8 * g4]);
#pragma unroll
            for (int pp = 0; pp < 6; ++pp) {
                constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc2[mt] = mfma_bf(av[mt][PA[pp]], b2[PB[pp]], acc2[mt]);
            }
        }
    }
    __syncthreads();                                             // every wave is done reading the a2 tile: `part` may overwrite it
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            part[(kq * R + mt * 16 + 4 * g4 + q) * (D + 1) + jq * 16 + l15] = acc2[mt][q];
    __syncthreads();
    float sres[EPT];
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
        const int e = q * 512 + tid, r = e / D, j = e % D;
        float sacc = bias2[q];
#pragma unroll
        for (int k = 0; k < KSPL; ++k) sacc += part[(k * R + r) * (D + 1) + j];
        sres[q] = sacc;
    }
    if (a.unit_rows) {            // ConvTransfer.forward(type='user'): x / ||x|| (model/conv_transfer.py:62-64); a row = 32 neighbouring lanes
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            float n2 = sres[q] * sres[q];
#pragma unroll
            for (int off = D / 2; off >= 1; off >>= 1) n2 += __shfl_xor(n2, off, 64);
            sres[q] = sres[q] / sqrtf(n2);
        }
    }
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
        const int e = q * 512 + tid, r = e / D, j = e % D;
        if (row0 + r < sg.n_rows) st_out<SML_WT_FWD>(&sg.out[(int64_t)(row0 + r) * D + j], sres[q]);
    }
}

// ------------------------------------------------------------------------------------
// The MF stage's forward (one 16-row tile per workgroup, the whole net: k_transfer_fwd<32,1,1>'s job) with fc1 / fc2 on the
// bf16x3 products above.  A 16-row tile's fc1 is 160 fp32 matrix products per wave = 4.3 us of ONE CU's matrix pipe -- the
// phase that kernel's timeline is longest in -- against 120 bf16 products at a third of the cycles each; the operand planes are
// packed once per epoch (theta does not move during the MF stage).  Gather (triples or the distinct-row records), lazy-Adam
// replay, the saves for the backward (x_t / x_hat / x_com, z1 in fp32, the replayed moments) and `out` as in the fp32 kernel;
// the backward keeps the fp32 products (its GEMMs come next).  SML_MF_BX3=0: k_transfer_fwd<32,1,1>.
// ------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_mf_fwd_bx3(const unsigned short* __restrict__ pkx, const float* __restrict__ p_theta,
        const SmlSched* __restrict__ p_sched, const void* __restrict__ p_idx, const SmlTileHdr* __restrict__ p_hdr, int p_B, int p_cur_step, int p_sched_len,
        SmlFwdArgs a_in) {
    static_assert(D == 32, "bf16x3 MF forward: d = 32");
    // (round 6: the first round of loads' operands as 13 preloaded dwords -- see k_transfer_fwd.  p_idx: the distinct-row records
    // of segment 0 when p_hdr != null, else the triples; segment 1's records / headers follow segment 0's at the item slot /
    // after the user tiles; rows of a segment = whole tiles in the distinct-row form.  sml_launch_mf_fwd_bx3 checks this layout.)
    SmlFwdArgs a = a_in;
    {
        const bool dn = p_hdr != nullptr;
        const int t0 = (p_B + SML_TM - 1) / SML_TM;
        a.tiles0 = t0; a.cur_step = p_cur_step; a.sched = p_sched; a.sched_len = p_sched_len;
        a.seg[0].theta = p_theta; a.seg[1].theta = p_theta + sml_net_size(D);
        a.seg[0].tri = a.seg[1].tri = dn ? a_in.seg[0].tri : static_cast<const int64_t*>(p_idx);
        a.seg[0].B = a.seg[1].B = p_B; a.seg[0].is_item = 0; a.seg[1].is_item = 1;
        a.seg[0].n_rows = dn ? SML_TM * t0 : p_B; a.seg[1].n_rows = dn ? SML_TM * ((2 * p_B + SML_TM - 1) / SML_TM) : 2 * p_B;
        a.seg[0].drec = dn ? static_cast<const SmlRun*>(p_idx) : nullptr;
        a.seg[1].drec = dn ? static_cast<const SmlRun*>(p_idx) + (int64_t)SML_R * ((p_B + SML_R - 1) / SML_R) : nullptr;
        a.seg[0].hdr = p_hdr; a.seg[1].hdr = dn ? p_hdr + t0 : nullptr;
    }
    constexpr int R = SML_TM;
    constexpr int K1 = SML_C2 * D, KS1 = K1 / 32;
    constexpr int CT = SML_HID / 16 / 8;                         // fc1 column tiles per wave (4)
    constexpr int S1B = K1 + 8, S2B = SML_HID + 8;
    constexpr int JT = D / 16, KPW = SML_HID / 32 / 8;           // fc2: 8 waves along K, KPW k-steps of 32 each (2); JT column tiles
    __shared__ __attribute__((aligned(16))) unsigned short A1b[3][R][S1B];
    __shared__ __attribute__((aligned(16))) unsigned short A2b[3][R][S2B];
    __shared__ float cws[104];
    __shared__ SmlSched swin[SML_SW];
    __shared__ SmlReplayEnt rtab[SML_RP_N];                      // closed-form replay: this launch's entries (a.sched_len > 0)
    float* part = reinterpret_cast<float*>(&A2b[0][0][0]);       // [8][R][D + 1] fc2 partials, once the a2 tile is done with
    static_assert(sizeof(A2b) >= 8 * R * (D + 1) * sizeof(float), "part fits");
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, g4 = lane >> 4;
    const int tile = (int)blockIdx.x;
    const int sidx = tile >= a.tiles0;
    const SmlSeg sg = sidx ? a.seg[1] : a.seg[0];
    const int tl = tile - (sidx ? a.tiles0 : 0);
    const int row0 = tl * R;
    const float* __restrict__ theta = sg.theta;
    const unsigned short* __restrict__ p1x = pkx + (int64_t)sidx * sml_bx3_size(D) + sml_bx3_p1(D);
    const unsigned short* __restrict__ p2x = pkx + (int64_t)sidx * sml_bx3_size(D) + sml_bx3_p2(D);
    if (tid < 104) cws[tid] = theta[tid];
    const bool lazy = sg.last_tab != nullptr;
    if (lazy) sched_window_load(swin, a.sched, a.cur_step - 1, tid);
    // ---- gather: one element per thread (row r = tid / D of the tile, coordinate w)
    const int r = tid / D, w = tid % D;
    bool ok; int64_t idx;
    if (sg.drec != nullptr) {                                    // distinct-row form: header and records in one round trip
        const int live = (int)sg.hdr[tl].nrows;
        const uint32_t trow = sg.drec[row0 + r].row;
        ok = r < live; idx = ok ? (int64_t)trow : 0;
        if (live == 0) return;
    } else {
        ok = row0 + r < sg.n_rows;
        idx = ok ? seg_row_index(sg, row0 + r) : 0;
    }
    float xt = sg.xt_tab[idx * D + w], xh = sg.xh_tab[idx * D + w];
    float m = 0.0f, v = 0.0f; int from = -1;
    if (lazy) { m = sg.m_tab[idx * D + w]; v = sg.v_tab[idx * D + w]; from = sg.last_tab[idx]; }
    const bool closed = lazy && a.sched_len > 0;
    if (closed) replay_table_build(rtab, a.sched, a.sched_len, a.cur_step - 1, tid);      // (double-precision work under the gather's round trips)
    // fc1 operand planes: wave wv owns column tiles wv * CT + t; a two-deep ring over the KS1 k-steps, behind the gather's loads
    uint4 bw[2][CT][3];
    auto load_b1 = [&](int ks, uint4 (&dst)[CT][3]) {
#pragma unroll
        for (int t = 0; t < CT; ++t)
#pragma unroll
            for (int p = 0; p < 3; ++p)
                dst[t][p] = *reinterpret_cast<const uint4*>(p1x + ((((int64_t)(wv * CT + t) * KS1 + ks) * 3 + p) * 64 + lane) * 8);
    };
    load_b1(0, bw[0]);
    float bias1[CT];
#pragma unroll
    for (int t = 0; t < CT; ++t) bias1[t] = theta[sml_off_f1b(D) + (wv * CT + t) * 16 + l15];
    const float bias2 = theta[sml_off_f2b(D) + w];
    __builtin_amdgcn_sched_barrier(0);
    if (!ok) { xt = 1.0f; xh = 0.0f; }
    float nr2 = xt * xt;
#pragma unroll
    for (int off = D / 2; off >= 1; off >>= 1) nr2 += __shfl_xor(nr2, off, 64);
    __syncthreads();                                             // cws and the schedule window are in LDS
    if (lazy) {
        if (ok) adam_replay_t(xh, m, v, from, a.cur_step - 1, a.sched, swin, a.cur_step - 1, closed ? rtab : nullptr);
        if (sg.mrep != nullptr) {                                // the row update continues from these (same tile, same XCD: plain stores)
            sg.mrep[(int64_t)(row0 + r) * D + w] = ok ? m : 0.0f;
            sg.vrep[(int64_t)(row0 + r) * D + w] = ok ? v : 0.0f;
        }
    }
    {   // ---- x_com, conv1, Gelu, conv2, Gelu -> A1 (channel-major flatten), split into its three planes; the backward's inputs
        const float xc = a.k2 ? 0.0f : (xt * xh) / sqrtf(nr2);   // no epsilon, as model/conv_transfer.py:99
        Pro p;
        conv_prologue(cws, xt, xh, xc, p);
#pragma unroll
        for (int c = 0; c < SML_C2; ++c) {
            unsigned short h, mm, l;
            split3(sml_gelu(p.h2p[c]), h, mm, l);
            A1b[0][r][c * D + w] = h; A1b[1][r][c * D + w] = mm; A1b[2][r][c * D + w] = l;
        }
        if (sg.xin != nullptr) {
            float* x = sg.xin + (int64_t)(row0 + r) * 3 * D;
            x[w] = xt; x[D + w] = xh; x[2 * D + w] = xc;
        }
    }
    __syncthreads();
    // ---- fc1: Z1[16 x 512]; wave wv: CT column tiles, KS1 k-steps, six bf16 products per (tile, k-step)
    f32x4 acc[CT];
#pragma unroll
    for (int t = 0; t < CT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) {
        if (ks + 1 < KS1) load_b1(ks + 1, bw[(ks + 1) & 1]);
        uint4 av[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) av[p] = *reinterpret_cast<const uint4*>(&A1b[p][l15][ks * 32 + 8 * g4]);
#pragma unroll
        for (int pp = 0; pp < 6; ++pp) {
            constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int t = 0; t < CT; ++t) acc[t] = mfma_bf(av[PA[pp]], bw[ks & 1][t][PB[pp]], acc[t]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    // fc2 operand planes of this wave's KPW k-steps (issued now: they ride under the epilogue)
    uint4 b2[KPW][JT][3];
#pragma unroll
    for (int k2 = 0; k2 < KPW; ++k2)
#pragma unroll
        for (int jt = 0; jt < JT; ++jt)
#pragma unroll
            for (int p = 0; p < 3; ++p)
                b2[k2][jt][p] = *reinterpret_cast<const uint4*>(p2x + ((((int64_t)jt * 16 + wv * KPW + k2) * 3 + p) * 64 + lane) * 8);
    // + bias, save z1 (fp32: the backward's Gelu'), Gelu -> the a2 tile's three planes
#pragma unroll
    for (int t = 0; t < CT; ++t) {
        const int n = (wv * CT + t) * 16 + l15;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int rr = 4 * g4 + q;
            const float z = acc[t][q] + bias1[t];
            if (sg.z1 != nullptr) sg.z1[(int64_t)(row0 + rr) * SML_HID + n] = z;
            unsigned short h, mm, l;
            split3(sml_gelu(z), h, mm, l);
            A2b[0][rr][n] = h; A2b[1][rr][n] = mm; A2b[2][rr][n] = l;
        }
    }
    __syncthreads();
    // ---- fc2: Out[16 x D] = a2[16 x 512] * W2^T; wave wv: k-steps wv * KPW .. + KPW - 1, both column tiles
    f32x4 acc2[JT];
#pragma unroll
    for (int jt = 0; jt < JT; ++jt) acc2[jt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k2 = 0; k2 < KPW; ++k2) {
        uint4 av[3];
#pragma unroll
        for (int p = 0; p < 3; ++p) av[p] = *reinterpret_cast<const uint4*>(&A2b[p][l15][(wv * KPW + k2) * 32 + 8 * g4]);
#pragma unroll
        for (int pp = 0; pp < 6; ++pp) {
            constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
            for (int jt = 0; jt < JT; ++jt) acc2[jt] = mfma_bf(av[PA[pp]], b2[k2][jt][PB[pp]], acc2[jt]);
        }
    }
    __syncthreads();                                             // every wave is done with the a2 tile: `part` overwrites it
#pragma unroll
    for (int jt = 0; jt < JT; ++jt)
#pragma unroll
        for (int q = 0; q < 4; ++q) part[(wv * R + 4 * g4 + q) * (D + 1) + jt * 16 + l15] = acc2[jt][q];
    __syncthreads();
    float sres = bias2;
#pragma unroll
    for (int k = 0; k < 8; ++k) sres += part[(k * R + r) * (D + 1) + w];
    if (row0 + r < sg.n_rows) st_out<SML_WT_FWD>(&sg.out[(int64_t)(row0 + r) * D + w], sres);
}

// ------------------------------------------------------------------------------------
// MF stage: the row update inside the backward (SmlFusedUpdate; replaces the k_run_update<Adam> launch on one GPU).
// Called by ALL threads of the workgroup for their element (row, w) of the tile's x_hat gradient `g`; x1 = the forward's
// replayed x_hat element.  A row's D elements sit in D neighbouring lanes of ONE wavefront (D <= 64).
//   * the row occurs once in the batch: Adam step here, from the forward's replayed moments -- k_run_update's from_scratch
//     path element by element (same pinned arithmetic: the same bits);
//   * duplicated row: the gradient row leaves as agent-scope (write-through) stores; once they are acknowledged the row's
//     first lane bumps the run's arrival counter (L2-bypassing atomic); the occurrence that arrives LAST adds the run's
//     gradient rows with cache-bypassing loads IN k_run_update's ORDER -- up to 8 rows one after the other in slot order;
//     longer runs by the whole wavefront: 64 / (D/4) lane groups take strided shares, eight 16-byte loads in flight each,
//     the shares meet in the same xor order -- steps the row and clears the counter for the next launch.
// No fence: a __threadfence() here is a write-back + invalidate of the XCD's whole L2 (see k_tr_wgrad2's election).
// ------------------------------------------------------------------------------------
// (the loads that do not depend on the gradient are issued early by the caller where it can: FusedPre)
struct FusedPre { uint32_t info; int64_t trow; float m, v; uint4 r0, r1; };
template <int D>
__device__ __forceinline__ void fused_prefetch(const SmlBwdArgs& a, int sidx, int row, int w, bool ok, FusedPre& f) {
    const SmlFusedUpdate& fu = a.fu;
    const int slot = (sidx ? a.ioff : 0) + row;
    f.info = SML_SLOT_ONCE; f.trow = 0; f.m = 0.0f; f.v = 0.0f;
    if (ok) {
        f.info = fu.slot_info[slot];
        const int t = (sidx && row >= a.B) ? row - a.B : row;
        f.trow = fu.tri[(int64_t)t * 3 + (sidx ? (row >= a.B ? 2 : 1) : 0)];
        f.m = fu.mrep[(int64_t)slot * D + w]; f.v = fu.vrep[(int64_t)slot * D + w];
    }
}
// the record of a duplicated row's run (needs f.info)
__device__ __forceinline__ void fused_prefetch_record(const SmlBwdArgs& a, int sidx, bool ok, FusedPre& f) {
    f.r0 = make_uint4(0u, 0u, 0u, 0u); f.r1 = f.r0;
    if (ok && f.info != SML_SLOT_ONCE) {
        const uint4* rec = reinterpret_cast<const uint4*>(a.fu.rec[sidx] + f.info);
        f.r0 = rec[0]; f.r1 = rec[1];
    }
}
// distinct-row form: the table row and the forward's replayed moments of scratch row `row` (every live row steps in place)
template <int D>
__device__ __forceinline__ void dense_prefetch(const SmlBwdArgs& a, int sidx, int row, int w, bool ok, FusedPre& f) {
    const int slot = (sidx ? a.ioff : 0) + row;
    f.info = SML_SLOT_ONCE; f.trow = 0; f.m = 0.0f; f.v = 0.0f;
    if (ok) {
        f.trow = (int64_t)a.dn.drec[slot].row;
        f.m = a.fu.mrep[(int64_t)slot * D + w]; f.v = a.fu.vrep[(int64_t)slot * D + w];
    }
}
template <int D>
__device__ __forceinline__ void fused_row_update(const SmlBwdArgs& a, int sidx, int row, int w, bool ok, float x1, float g, const FusedPre& f) {
    static_assert(D <= 64, "a row inside one wavefront");
    constexpr int LPR = D / 4, G = 64 / LPR, LONG = 8, LD = 8;       // (k_run_update<D, float, 1, false>'s constants)
    const SmlFusedUpdate& fu = a.fu;
    const int lane = threadIdx.x & 63;
    const int lead = lane & ~(D - 1) & 63;                            // first lane of this row
    const bool first = (lane & (D - 1)) == 0;
    const int slot = (sidx ? a.ioff : 0) + row;
    const uint32_t info = f.info;
    const int64_t trow = f.trow;
    float m = f.m, v = f.v;
    const SmlSched sc = fu.sched[fu.cur_step];
    const bool dup = ok && info != SML_SLOT_ONCE;
    bool step_it = ok && !dup, long_last = false;
    float gsum = g;
    uint32_t rpos = 0, rlen = 0;
    if (dup) {
        st_out<2>(&fu.dx_all[(int64_t)slot * D + w], g);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // (the row's D stores are one instruction of this wavefront)
        const uint4 r0 = f.r0, r1 = f.r1;
        rpos = r0.y; rlen = r0.z;
        int* cnt = fu.arrive + (sidx ? a.B : 0) + info;
        int old = 0;
        if (first) old = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        old = __shfl(old, lead, 64);
        if (old == (int)rlen - 1) {                                   // every other occurrence's row is in memory
            if (first) __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (rlen <= (uint32_t)LONG) {
                // all of the run's rows in ONE round trip (unconditional loads, clamped addresses; the compiler waits
                // after every atomic load of its own)
                const uint32_t* vals = fu.val[sidx] + rpos;
                const uint32_t inl[4] = {r1.x, r1.y, r1.z, r1.w};
                uint32_t sl[LONG];
#pragma unroll
                for (int j = 0; j < LONG; ++j) sl[j] = j < SML_RUN_INL ? inl[j < SML_RUN_INL ? j : 0] : ((uint32_t)j < rlen ? vals[j] : inl[0]);
                const float* src[LONG];
#pragma unroll
                for (int j = 0; j < LONG; ++j) src[j] = fu.dx_all + (int64_t)((uint32_t)j < rlen ? sl[j] : inl[0]) * D + w;
                float x[LONG];
                agent_load4x8(x, src);
                gsum = 0.0f;
#pragma unroll
                for (int j = 0; j < LONG; ++j) if ((uint32_t)j < rlen) gsum += x[j];
                step_it = true;
            } else {
                long_last = true;
            }
        }
    }
    // long runs: the whole wavefront, one run after the other (all 64 lanes are here: the caller's loop is workgroup-uniform)
    unsigned long long todo = __ballot(long_last && first);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const uint32_t l_pos = (uint32_t)__shfl((int)rpos, leader, 64), l_len = (uint32_t)__shfl((int)rlen, leader, 64);
        const uint32_t* vals = fu.val[sidx] + l_pos;
        const int grp = lane / LPR, sub = lane % LPR;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (uint32_t q0 = (uint32_t)grp; q0 < l_len; q0 += LD * G) {
            const float* src[LD];
#pragma unroll
            for (int j = 0; j < LD; ++j) {
                const uint32_t sl = q0 + j * G < l_len ? vals[q0 + j * G] : 0u;      // (clamped address, skipped below)
                src[j] = fu.dx_all + (int64_t)sl * D + sub * 4;
            }
            f32x4 x[LD];
            peer_load16x8(x, src);
#pragma unroll
            for (int j = 0; j < LD; ++j) if (q0 + j * G < l_len) acc += x[j];
        }
#pragma unroll
        for (int off = LPR; off < 64; off <<= 1)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] += __shfl_xor(acc[q], off, 64);
        // element w of the sum sits in component w & 3 of the lanes with sub == w >> 2
        const float e0 = __shfl(acc[0], w >> 2, 64), e1 = __shfl(acc[1], w >> 2, 64), e2 = __shfl(acc[2], w >> 2, 64), e3 = __shfl(acc[3], w >> 2, 64);
        if (long_last && lead == leader) {
            gsum = (w & 3) == 0 ? e0 : (w & 3) == 1 ? e1 : (w & 3) == 2 ? e2 : e3;
            step_it = true;
        }
    }
    if (step_it) {
        float p = x1;
        adam_apply(p, m, v, gsum, sc);
        const int64_t o = trow * D + w;
        fu.w[sidx][o] = p; fu.m[sidx][o] = m; fu.v[sidx][o] = v;
        if (first) fu.last[sidx][trow] = fu.cur_step;
    }
}

// ------------------------------------------------------------------------------------
// backward, one workgroup per row tile (batches large enough to fill the chip without the
// coordinate split below): dOut -> (MF stage) dx_hat + l2*x_hat, or (TR stage) dZ1 rows +
// conv-grad partials.  dx / dz1 scratch is padded to whole tiles (unconditional stores).
// ------------------------------------------------------------------------------------
// (round 6: the head's first round of loads -- tile header and entries, the conv weights, the partner rows -- reads through 12
// preloaded dwords; see k_transfer_fwd.  sml_launch_bwd fills them from the struct.)
template <int D, int MT, bool TR>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_transfer_bwd_full(const SmlTileHdr* __restrict__ p_hdr, const uint2* __restrict__ p_ent,
        const float* __restrict__ p_theta, const float* __restrict__ p_out_all, int p_tiles0, int p_world, int p_tiles_live, int p_B, SmlBwdArgs a) {
    // (the struct is NOT copied and patched here as in k_transfer_fwd: SmlFusedUpdate's per-table arrays are indexed dynamically, a local
    // copy would live in scratch memory; the preloaded values replace the struct's fields by name below)
    constexpr int R = SML_TM * MT;
    constexpr int K1 = SML_C2 * D;
    constexpr int S2 = SML_HID + 4;
    constexpr int SD = D + 4;
    constexpr int KSD = D / 16;
    constexpr int EPT = R * D / 512;
    constexpr int KSPL = (D == 32) ? 4 : (D == 64 ? 2 : 1);      // dA1: waves along the n reduction ...
    constexpr int TSPL = 8 / KSPL;                                // ... x waves along the 5D outputs (5 tiles each)
    constexpr int KPER = 32 / KSPL;
    constexpr int PSTR = K1 + 1;
    constexpr int SZ = cmax(R * S2 + R * SD, KSPL * R * PSTR);
    static_assert((K1 / 16) == 5 * TSPL, "5 column tiles per wave");
    constexpr int CGS = 20;               // conv-grad operand row stride (16-byte aligned)
    __shared__ __attribute__((aligned(16))) float smem[SZ + 104];
    __shared__ __attribute__((aligned(16))) float cgst[TR ? 2 * 256 * CGS : 4];   // conv-gradient operands of 256 elements (A, B)
    __shared__ float red[TR ? 8 : 1][256];
    __shared__ float cf[4][SML_TM * MT];
    __shared__ float lred[8];
    // MF stage, distinct-row form (SmlDense): the per-occurrence dOut contributions of one chunk of the tile's entries, and the rows' entry ranges
    constexpr bool DENSEOK = !TR && MT == 1 && D <= 64;
    __shared__ __attribute__((aligned(16))) float Cs[DENSEOK ? SML_TILE_ENT * D : 4];
    __shared__ uint32_t rlen[16], rstart[17];
    __shared__ float Ps[DENSEOK ? 512 : 4];                       // the strided shares of a long row's sum
    float* dZs = smem;                    // [R][516]
    float* dOs = smem + R * S2;           // [R][D+4]
    float* part = smem;                   // [KSPL][R][5D+1], aliases dZs after the second GEMM
    float* cws = smem + SZ;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, g4 = lane >> 4;
    TL_BEGIN(5); TL_PREV();
    const int sidx = (int)blockIdx.x >= p_tiles0;
    const SmlBwdSeg& sg = a.seg[sidx];
    const float* __restrict__ sg_theta = p_theta + (sidx ? sml_net_size(D) : 0);       // = sg_theta, from the preloaded parameter
    const int row0 = ((int)blockIdx.x - (sidx ? p_tiles0 : 0)) * R;
    if constexpr (!TR) {
        if (p_world > 0 && (int)blockIdx.x >= p_tiles_live) { peer_signal(a.push); return; }     // (a batch shorter than the cap)
    }
    if (tid < 104) cws[tid] = sg_theta[tid];
    // both GEMMs' first operand k-steps are on their way before the pair loss starts (they depend on theta alone)
#ifndef SML_PREB
#define SML_PREB 0
#endif
    // (off: round 2 measured no gain from hoisting these under the per-occurrence pair loss; round 5 tried again under the
    // distinct-row head, which is two dependent round trips long -- issued behind the head's first loads: MF step 35.7 -> 36.7 us,
    // the per-occurrence form 38.5 -> 39.3: 28 KB of operands per wave in front of the head's second trip cost more than they hide)
    constexpr bool PREB = (SML_PREB != 0) && (MT == 1) && (D <= 32) && !TR;
    const f32x4* __restrict__ p2b = reinterpret_cast<const f32x4*>(sg.pk + sml_pk_p2b(D));
    const f32x4* __restrict__ p1b = reinterpret_cast<const f32x4*>(sg.pk + sml_pk_p1b(D));
    auto tileA2 = [wv](int t) { return wv * 4 + t; };
    auto tileA1 = [wv](int t) { return (wv / KSPL) * 5 + t; };
    auto nokofs = [](int) { return 0; };
    f32x4 ringb2[PREB ? KSD : 1][4], ringb1[PREB ? 4 : 1][5];
    float zpre[PREB ? MT : 1][4][4];
    // (issued BEHIND the head's first loads: loads return in issue order, and 28 KB of operands per wave ahead of the head's
    // few hundred bytes would hold the whole chain back)
    auto preload_operands = [&]() {
        if constexpr (PREB) {
            ring_preload<4, KSD>(ringb2, p2b, KSD, 0, lane, tileA2, nokofs);
            ring_preload<5, 4>(ringb1, p1b, 32, (wv % KSPL) * KPER, lane, tileA1, nokofs);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        zpre[mt][t][q] = sg.z1[(int64_t)(row0 + mt * SML_TM + 4 * g4 + q) * SML_HID + (wv * 4 + t) * 16 + l15];
        }
    };

    // ---- pair loss (model/conv_transfer.py:120-134) for this tile's rows: every row fetches the three
    // transferred rows of its triple, one thread per row forms the two scores and the loss terms,
    // then dOut is written element-wise.  User tiles own the loss value (each triple once).
    float lsum = 0.0f;
    const bool dense = DENSEOK && p_hdr != nullptr;               // (kernel-uniform)
    int live = R;
    if constexpr (DENSEOK) {
    if (dense) {
        // ---- distinct-row form: this tile's 16 rows are DISTINCT table rows; row r's dOut is the sum over the row's occurrences
        // (entries, in slot order) of coefficient x partner row.  One entry per group of D/4 lanes and round; the header, the
        // tile's first SML_TILE_ENT entries and then ALL their out rows are each one round trip.
        constexpr int LPR = D / 4, G = 512 / LPR, EPG = SML_TILE_ENT / G;
        const int grp = tid / LPR, sub = tid % LPR;
        const SmlTileHdr* hp = p_hdr + blockIdx.x;
        const uint4 h0 = *reinterpret_cast<const uint4*>(hp);
        uint2 en[EPG];
#pragma unroll
        for (int i = 0; i < EPG; ++i) en[i] = p_ent[(int64_t)blockIdx.x * SML_TILE_ENT + grp + i * G];
        uint32_t mylen = 0;
        if (tid < 16) mylen = hp->len[tid];
        __builtin_amdgcn_sched_barrier(0);
        preload_operands();
        const int count = (int)h0.x;
        live = (int)h0.y;
        if (live == 0) return;                                        // (a tile beyond the batch's distinct rows)
        if (wv == 0) {                                                // the rows' entry ranges: a prefix over the 16 lengths
            uint32_t inc = lane < 16 ? mylen : 0u;
#pragma unroll
            for (int off = 1; off < 16; off <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)inc, off, 64); if (lane >= off) inc += t; }
            if (lane < 16) { rlen[lane] = mylen; rstart[lane] = inc - mylen; }
            if (lane == 15) rstart[16] = inc;
        }
        const float inv_b = 1.0f / (float)p_B;
        float racc[EPT];
#pragma unroll
        for (int q = 0; q < EPT; ++q) racc[q] = 0.0f;
        const int nchunk = (count + SML_TILE_ENT - 1) / SML_TILE_ENT;
#pragma unroll 1
        for (int ch = 0; ch < nchunk; ++ch) {
            const int e0 = ch * SML_TILE_ENT;
            if (ch > 0) {                                             // (a row with hundreds of occurrences: the rest of the entries)
                __syncthreads();                                      // the previous chunk's sums have been read
#pragma unroll
                for (int i = 0; i < EPG; ++i) {
                    const int e = e0 + grp + i * G;
                    en[i] = e < count ? a.dn.spill[(int64_t)h0.z + (e - SML_TILE_ENT)] : make_uint2(0u, 0u);
                }
            }
            f32x4 U[EPG], I[EPG], N[EPG];
#pragma unroll
            for (int i = 0; i < EPG; ++i) {
                const bool valid = e0 + grp + i * G < count;
                const uint32_t du = valid ? (en[i].x & 0xffffu) : 0u, di = valid ? (en[i].x >> 16) : 0u, dn = valid ? (en[i].y & 0xffffu) : 0u;
                U[i] = *reinterpret_cast<const f32x4*>(p_out_all + (int64_t)du * D + sub * 4);
                I[i] = *reinterpret_cast<const f32x4*>(p_out_all + (int64_t)(a.ioff + di) * D + sub * 4);
                N[i] = *reinterpret_cast<const f32x4*>(p_out_all + (int64_t)(a.ioff + dn) * D + sub * 4);
            }
#pragma unroll
            for (int i = 0; i < EPG; ++i) {
                const int el = grp + i * G;                           // entry inside the chunk
                const bool valid = e0 + el < count;
                const uint32_t kind = (en[i].y >> 20) & 3u;
                float sp = 0.f, sn = 0.f, uu = 0.f;
#pragma unroll
                for (int c = 0; c < 4; ++c) { sp += U[i][c] * I[i][c]; sn += U[i][c] * N[i][c]; uu += U[i][c] * U[i][c]; }
#pragma unroll
                for (int off = 1; off < LPR; off <<= 1) {
                    sp += __shfl_xor(sp, off, 64); sn += __shfl_xor(sn, off, 64); uu += __shfl_xor(uu, off, 64);
                }
                float lt, d0, d1, inv_nu = 1.0f, cc = 0.0f;
                const bool nrm = a.kind == SML_LOSS_BPR_NORM || a.kind == SML_LOSS_BPR_UNIT;
                if (nrm) {
                    const float nu = sqrtf(uu);
                    inv_nu = 1.0f / nu;
                    cc = a.kind == SML_LOSS_BPR_NORM ? (sp - sn) / (nu * nu * nu) : 0.0f;
                    pair_terms(SML_LOSS_BPR, (sp - sn) * inv_nu, 0.0f, 1.0f, lt, d0, d1);
                    d1 = -d0;
                } else {
                    pair_terms(a.kind, sp, sn, inv_b, lt, d0, d1);
                }
                d0 *= a.scale; d1 *= a.scale;
                f32x4 c4;
                if (nrm) {
                    if (kind == 0u) c4 = d0 * ((I[i] - N[i]) * inv_nu - cc * U[i]);
                    else c4 = ((kind == 1u) ? d0 : d1) * inv_nu * U[i];
                } else {
                    if (kind == 0u) c4 = d0 * I[i] + d1 * N[i];
                    else c4 = ((kind == 1u) ? d0 : d1) * U[i];
                }
                if (valid) {
                    *reinterpret_cast<f32x4*>(Cs + el * D + sub * 4) = c4;
                    if (kind == 0u && sub == 0) lsum += lt * a.scale;           // every triple once: at its user's row
                }
            }
            __syncthreads();
            // row sums, in an order that is a function of the input alone.  Short rows: thread (row, w) adds its row's entries in
            // entry order, eight LDS reads in flight at a time (one dependent read per entry made a 140-occurrence row -- a Zipf head
            // user -- a 6 us chain on ONE tile).  Long rows (more than LONGROW entries in this chunk; workgroup-uniform): the
            // workgroup's 512 / D lane groups take strided shares, the shares are added in share order.
            constexpr int LONGROW = 32, NSH = 512 / D;
#pragma unroll
            for (int q = 0; q < EPT; ++q) {
                const int e = q * 512 + tid, r = e / D, w = e % D;
                const int lo = max((int)rstart[r], e0) - e0, hi = min((int)(rstart[r] + rlen[r]), e0 + SML_TILE_ENT) - e0;
                if (hi - lo > LONGROW) continue;
                float sacc = racc[q];
                int j = lo;
                for (; j + 8 <= hi; j += 8) {
                    float x[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) x[u] = Cs[(j + u) * D + w];
#pragma unroll
                    for (int u = 0; u < 8; ++u) sacc += x[u];
                }
                for (; j < hi; ++j) sacc += Cs[j * D + w];
                racc[q] = sacc;
            }
#pragma unroll 1
            for (int r = 0; r < 16; ++r) {
                const int lo = max((int)rstart[r], e0) - e0, hi = min((int)(rstart[r] + rlen[r]), e0 + SML_TILE_ENT) - e0;
                if (hi - lo <= LONGROW) continue;                      // (the same for every thread: LDS values)
                const int sh = tid / D, w = tid % D;
                float part = 0.0f;
                int j = lo + sh;
                for (; j + 3 * NSH < hi; j += 4 * NSH) {
                    const float x0 = Cs[j * D + w], x1 = Cs[(j + NSH) * D + w], x2 = Cs[(j + 2 * NSH) * D + w], x3 = Cs[(j + 3 * NSH) * D + w];
                    part += x0; part += x1; part += x2; part += x3;
                }
                for (; j < hi; j += NSH) part += Cs[j * D + w];
                Ps[sh * D + w] = part;
                __syncthreads();
                // thread (row, w) of THIS row picks the shares up (its q-th element is row r when r == (q * 512 + tid) / D)
#pragma unroll
                for (int q = 0; q < EPT; ++q) {
                    const int e = q * 512 + tid;
                    if (e / D == r) {
                        float x[NSH];
#pragma unroll
                        for (int u = 0; u < NSH; ++u) x[u] = Ps[u * D + (e % D)];
                        float sacc = racc[q];
#pragma unroll
                        for (int u = 0; u < NSH; ++u) sacc += x[u];
                        racc[q] = sacc;
                    }
                }
                __syncthreads();                                       // (Ps is free again)
            }
        }
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const int e = q * 512 + tid, r = e / D, w = e % D;
            dOs[r * SD + w] = r < live ? racc[q] : 0.0f;
        }
    }
    }
    if (!dense) {
    float* O3 = smem;                     // [3][R][D+1], aliases dZs (not yet live)
    float ou[EPT], oi[EPT], on[EPT];
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
        const int e = q * 512 + tid, r = e / D, w = e % D;
        const int row = row0 + r;
        ou[q] = oi[q] = on[q] = 0.0f;
        if (row < sg.n_rows) {
            const int t = (sg.is_item && row >= p_B) ? row - p_B : row;
            const float* pu = p_out_all + (int64_t)t * D + w;
            const float* pi = p_out_all + (int64_t)(a.ioff + t) * D + w;
            const float* pn = p_out_all + (int64_t)(a.ioff + p_B + t) * D + w;
            // (this kernel runs behind the unsplit forward: one plane, as a rule)
            for (int p = 0; p < a.out_np; ++p) {
                ou[q] += pu[p * a.out_pstride];
                oi[q] += pi[p * a.out_pstride];
                on[q] += pn[p * a.out_pstride];
            }
        }
        O3[(0 * R + r) * (D + 1) + w] = ou[q];
        O3[(1 * R + r) * (D + 1) + w] = oi[q];
        O3[(2 * R + r) * (D + 1) + w] = on[q];
    }
    preload_operands();
    __syncthreads();
    if (tid < R) {
        float sp = 0.f, sn = 0.f, uu = 0.f;
#pragma unroll 8
        for (int w = 0; w < D; ++w) {
            const float u = O3[(0 * R + tid) * (D + 1) + w];
            sp += u * O3[(1 * R + tid) * (D + 1) + w];
            sn += u * O3[(2 * R + tid) * (D + 1) + w];
            uu += u * u;
        }
        float lt, d0, d1, inv_nu = 1.0f, cc = 0.0f;
        if (a.kind == SML_LOSS_BPR_NORM || a.kind == SML_LOSS_BPR_UNIT) {
            const float nu = sqrtf(uu);
            inv_nu = 1.0f / nu;
            // BPR_NORM differentiates through the norm; BPR_UNIT's norm is detached (ConvTransfer.forward)
            cc = a.kind == SML_LOSS_BPR_NORM ? (sp - sn) / (nu * nu * nu) : 0.0f;
            pair_terms(SML_LOSS_BPR, (sp - sn) * inv_nu, 0.0f, 1.0f, lt, d0, d1);
            d1 = -d0;
        } else {
            pair_terms(a.kind, sp, sn, 1.0f / (float)p_B, lt, d0, d1);
        }
        cf[0][tid] = d0 * a.scale; cf[1][tid] = d1 * a.scale; cf[2][tid] = inv_nu; cf[3][tid] = cc;
        if (!sg.is_item && row0 + tid < sg.n_rows) lsum = lt * a.scale;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
        const int e = q * 512 + tid, r = e / D, w = e % D;
        const int row = row0 + r;
        const float d0 = cf[0][r], d1 = cf[1][r];
        float g;
        if (a.kind == SML_LOSS_BPR_NORM || a.kind == SML_LOSS_BPR_UNIT) {
            const float inv_nu = cf[2][r], cc = cf[3][r];
            if (!sg.is_item) g = d0 * ((oi[q] - on[q]) * inv_nu - cc * ou[q]);
            else g = ((row < p_B) ? d0 : d1) * ou[q] * inv_nu;
        } else {
            if (!sg.is_item) g = d0 * oi[q] + d1 * on[q];
            else g = ((row < p_B) ? d0 : d1) * ou[q];
        }
        if (row >= sg.n_rows) g = 0.0f;
        dOs[r * SD + w] = g;
        if (TR) sg.dout[(int64_t)row * D + w] = g;
    }
    }   // (!dense)
    // the (x_t, x_hat, x_com) rows of the tail: with one element per thread issue the loads now and
    // use them after both GEMMs; with more (d > 32) load them in the tail to keep registers free
    constexpr bool PRELOAD = (EPT == 1);
    float x0p = 0.f, x1p = 0.f, x2p = 0.f;
    if (PRELOAD) {
        const float* x = sg.xin + (int64_t)(row0 + tid / D) * 3 * D;
        x0p = x[tid % D]; x1p = x[D + tid % D]; x2p = x[2 * D + tid % D];
    }
    // MF stage, fused row update: what the step needs besides the gradient is on its way from here
    constexpr bool FUSABLE = !TR && D <= 64;
    const bool fused = FUSABLE && a.fu.slot_info != nullptr;           // (kernel-uniform)
    FusedPre fpre;
    fpre.info = SML_SLOT_ONCE; fpre.trow = 0; fpre.m = fpre.v = 0.0f; fpre.r0 = make_uint4(0u, 0u, 0u, 0u); fpre.r1 = fpre.r0;
    if constexpr (FUSABLE && PRELOAD) {
        if (fused) fused_prefetch<D>(a, sidx, row0 + tid / D, tid % D, row0 + tid / D < sg.n_rows, fpre);
        if (dense) dense_prefetch<D>(a, sidx, row0 + tid / D, tid % D, tid / D < live, fpre);
    }
    __syncthreads();
    TL(2);
    // ---- dA2[R x 512] = dOut[R x D] * W2 ; dZ1 = dA2 * Gelu'(z1) ; wave wv owns column tiles 4wv..4wv+3
    {
        f32x4 acc[MT][4];
        zero_acc(acc);
        float z[MT][4][4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    z[mt][t][q] = PREB ? zpre[mt][t][q] : sg.z1[(int64_t)(row0 + mt * SML_TM + 4 * g4 + q) * SML_HID + (wv * 4 + t) * 16 + l15];
        if constexpr (PREB) mma16_ring<MT, 4, KSD, KSD, false>(acc, ringb2, dOs + l15 * SD + 4 * g4, SML_TM * SD, p2b, KSD, 0, lane, tileA2, nokofs);
        else mma16_rows<MT, 4, KSD, KSD>(acc, dOs + l15 * SD + 4 * g4, SML_TM * SD, p2b, KSD, 0, lane, tileA2);
        float* dz1 = sg.dz1;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int n = (wv * 4 + t) * 16 + l15;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int r = mt * SML_TM + 4 * g4 + q;
                    const float dz = acc[mt][t][q] * sml_gelu_grad(z[mt][t][q]);
                    dZs[r * S2 + n] = dz;
                    if (TR) dz1[(int64_t)(row0 + r) * SML_HID + n] = dz;
                }
        }
    }
    if constexpr (FUSABLE && PRELOAD) {
        if (fused) fused_prefetch_record(a, sidx, row0 + tid / D < sg.n_rows, fpre);      // (slot_info arrived under the first product)
    }
    __syncthreads();
    TL(3);
    // ---- dA1[R x 5D] = dZ1[R x 512] * W1 ; waves = KSPL (reduction) x TSPL (5 column tiles each)
    {
        const int kq = wv % KSPL, tq = wv / KSPL;
        f32x4 acc[MT][5];
        zero_acc(acc);
        if constexpr (PREB) mma16_ring<MT, 5, KPER, 4, false>(acc, ringb1, dZs + l15 * S2 + 4 * g4, SML_TM * S2, p1b, 32, kq * KPER, lane, tileA1, nokofs);
        else mma16_rows<MT, 5, KPER, 4>(acc, dZs + l15 * S2 + 4 * g4, SML_TM * S2, p1b, 32, kq * KPER, lane,
                                        [tq](int t) { return tq * 5 + t; });
        __syncthreads();                        // every wave is done reading dZs
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int t = 0; t < 5; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    part[(kq * R + mt * SML_TM + 4 * g4 + q) * PSTR + (tq * 5 + t) * 16 + l15] = acc[mt][t][q];
    }
    __syncthreads();
    TL(4);
    // ---- per-coordinate tail: Gelu'(h2) -> conv2^T -> Gelu'(h1) -> conv1^T (row 1 = x_hat)
    // TR: the conv1/conv2 parameter gradients are ONE small matrix product over the tile's elements (see
    // k_transfer_bwd): G = sum_e A[e]^T B[e], accumulated on MFMA across the EPT passes in four registers per
    // wave -- not 96 per-thread accumulators (which spilled at d = 64 / 128)
    f32x4 cgacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int q = 0; q < EPT; ++q) {
        const int e = q * 512 + tid, r = e / D, w = e % D;
        const int row = row0 + r;
        const bool ok = row < sg.n_rows;
        float x0 = x0p, x1 = x1p, x2 = x2p;
        if (!PRELOAD) {
            const float* x = sg.xin + (int64_t)row * 3 * D;
            x0 = x[w]; x1 = x[D + w]; x2 = x[2 * D + w];
        }
        Pro p;
        conv_prologue(cws, x0, x1, x2, p);
        float dh2p[SML_C2];
#pragma unroll
        for (int c = 0; c < SML_C2; ++c) {
            float s = 0.0f;
#pragma unroll
            for (int k = 0; k < KSPL; ++k) s += part[(k * R + r) * PSTR + c * D + w];
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
        if constexpr (!TR) {
            if constexpr (DENSEOK) {
                if (dense) {
                    // every row of the tile is a distinct table row: its whole gradient is here -- l2 once per occurrence
                    // (model/transfer.py:486-488 sums over the gathered rows) -- and the lazy-Adam step runs in place
                    const bool okd = r < live;
                    const float nocc = (float)rlen[r];
                    if constexpr (!PRELOAD) dense_prefetch<D>(a, sidx, row, w, okd, fpre);
                    if (okd) {
                        float pnew = x1, mm = fpre.m, vv = fpre.v;
                        adam_apply(pnew, mm, vv, dxh + nocc * a.l2 * x1, a.fu.sched[a.fu.cur_step]);
                        const int64_t o = fpre.trow * D + w;
                        a.fu.w[sidx][o] = pnew; a.fu.m[sidx][o] = mm; a.fu.v[sidx][o] = vv;
                        if (w == 0) a.fu.last[sidx][fpre.trow] = a.fu.cur_step;
                        lsum += nocc * 0.5f * a.l2 * x1 * x1;
                    }
                    continue;
                }
            }
            if constexpr (D <= 64) {
                if (fused) {
                    if constexpr (!PRELOAD) { fused_prefetch<D>(a, sidx, row, w, ok, fpre); fused_prefetch_record(a, sidx, ok, fpre); }
                    fused_row_update<D>(a, sidx, row, w, ok, x1, dxh + a.l2 * x1, fpre);
                } else st_out<SML_WT_MFB>(&sg.dx[(int64_t)row * D + w], dxh + a.l2 * x1);
            } else {
                st_out<SML_WT_MFB>(&sg.dx[(int64_t)row * D + w], dxh + a.l2 * x1);
            }
            if (p_world > 0 && sidx && ok) {                    // several GPUs: the row also goes into every rank's inbox
                for (int q = 0; q < p_world; ++q) peer_store(a.push.dst[q] + (int64_t)row * D + w, dxh + a.l2 * x1);
            }
            if (ok) lsum += 0.5f * a.l2 * x1 * x1;      // + l2 * 0.5 * sum(x_hat^2), model/transfer.py:486-488
        }
        if constexpr (TR) {
            // A[e] = (dh1p[0..9], dh2p[0..4], 0), B[e] = (x0, x1, x2, 1, h1[0..9], 0, 0); 256 elements per round
            float* cgA = cgst;
            float* cgB = cgst + 256 * CGS;
#pragma unroll 1
            for (int half = 0; half < 2; ++half) {
                __syncthreads();               // the previous round's MFMA operands have been read
                if ((tid >> 8) == half) {
                    const int t8 = tid & 255;
                    float cga[16], cgb[16];
#pragma unroll
                    for (int c = 0; c < SML_C1; ++c) { cga[c] = ok ? dh1p[c] : 0.0f; cgb[4 + c] = p.h1[c]; }
#pragma unroll
                    for (int o = 0; o < SML_C2; ++o) cga[10 + o] = ok ? dh2p[o] : 0.0f;
                    cga[15] = 0.0f;
                    cgb[0] = x0; cgb[1] = x1; cgb[2] = x2; cgb[3] = 1.0f; cgb[14] = 0.0f; cgb[15] = 0.0f;
#pragma unroll
                    for (int i4 = 0; i4 < 4; ++i4) {
                        f32x4 va, vb;
#pragma unroll
                        for (int e4 = 0; e4 < 4; ++e4) { va[e4] = cga[i4 * 4 + e4]; vb[e4] = cgb[i4 * 4 + e4]; }
                        *reinterpret_cast<f32x4*>(cgA + t8 * CGS + i4 * 4) = va;
                        *reinterpret_cast<f32x4*>(cgB + t8 * CGS + i4 * 4) = vb;
                    }
                }
                __syncthreads();
                // wave wv: elements 64*(wv&3) + 32*(wv>>2) .. +31, 4 elements per MFMA
                const int e0 = 64 * (wv & 3) + 32 * (wv >> 2);
                float av8[8], bv8[8];
#pragma unroll
                for (int s8 = 0; s8 < 8; ++s8) {
                    av8[s8] = cgA[(e0 + 4 * s8 + g4) * CGS + l15];
                    bv8[s8] = cgB[(e0 + 4 * s8 + g4) * CGS + l15];
                }
#pragma unroll
                for (int s8 = 0; s8 < 8; ++s8) cgacc = mfma16(av8[s8], bv8[s8], cgacc);
            }
        }
    }
    {   // this workgroup's share of the batch loss: lanes, then waves in index order (deterministic)
        float v = lsum;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0) lred[wv] = v;
    }
    if constexpr (TR) {
        // G[c][0..2] = dW_conv1[c], G[c][3] = db_conv1[c], G[10+o][4+c] = dW_conv2[o][c], G[10+o][3] = db_conv2[o]
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) red[wv][(4 * g4 + q4) * 16 + l15] = cgacc[q4];
        __syncthreads();
        if (tid < 95) {
            int i, j;
            if (tid < 30) { i = tid / 3; j = tid % 3; }
            else if (tid < 40) { i = tid - 30; j = 3; }
            else if (tid < 90) { i = 10 + (tid - 40) / 10; j = 4 + (tid - 40) % 10; }
            else { i = 10 + (tid - 90); j = 3; }
            float sacc = 0.0f;
#pragma unroll
            for (int w8 = 0; w8 < 8; ++w8) sacc += red[w8][i * 16 + j];
            a.convg_part[(int64_t)blockIdx.x * SML_CG + tid] = sacc;
        }
    } else {
        __syncthreads();
    }
    if (tid == 0) {
        float s = 0.0f;
#pragma unroll
        for (int w8 = 0; w8 < 8; ++w8) s += lred[w8];
        a.loss_part[blockIdx.x] = s;
        TL(7);
    }
    if constexpr (!TR) { if (p_world > 0) peer_signal(a.push); }       // this workgroup's rows are acknowledged: +1 on every rank's counter
    TL_DONE();
}

// ------------------------------------------------------------------------------------
// backward: dOut -> (MF stage) dx_hat + l2*x_hat, or (TR stage) dZ1 rows + conv-grad partials.
// dx / dz1 scratch is padded to whole tiles (unconditional stores).
//
// CS = D/16 workgroups share a 16-row tile: workgroup (tile, cs) owns the 16 coordinates
// w in [16cs, 16cs+16) of every conv channel.  The flatten is channel-major (column c*D + w), so
// those are the five dA1 column tiles {c*D/16 + cs}: the big GEMM dA1 = dZ1 * W1 splits over the
// CS workgroups by OUTPUT columns -- no cross-workgroup sum -- while the small one (dA2 = dOut * W2,
// K = D) and the pair loss are simply repeated.  A 768-row TR batch is 96 workgroups at d = 32.
// ------------------------------------------------------------------------------------
template <int D, bool TR, bool PRE>
__global__ __launch_bounds__(512) void k_transfer_bwd(SmlBwdArgs a) {
    constexpr int R = SML_TM;
    constexpr int CS = D / 16;
    constexpr int S2 = SML_HID + 4;
    constexpr int SD = D + 4;
    constexpr int KSD = D / 16;
    constexpr int EPT = R * D / 512;
    constexpr int PSTR = SML_C2 * 16 + 1;
    constexpr int CGS = 20;                                        // conv-grad operand row stride (16-byte aligned)
    constexpr int SZ = cmax(cmax(R * S2 + R * SD, 8 * R * PSTR), 2 * 256 * CGS);
    static_assert(3 * R * (D + 1) <= SZ, "pair-loss staging fits");
    __shared__ __attribute__((aligned(16))) float smem[SZ + 104];
    __shared__ float red[TR ? 8 : 1][256];
    __shared__ float cf[4][SML_TM];
    __shared__ float lred[8];
    float* dZs = smem;                    // [R][516]
    float* dOs = smem + R * S2;           // [R][D+4]
    float* part = smem;                   // [8][R][81], aliases dZs after the second GEMM
    float* cws = smem + SZ;

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, g4 = lane >> 4;
    TL_BEGIN(2); TL_PREV();
    const int tile = (int)blockIdx.x / CS, cs = (int)blockIdx.x % CS;
    const int sidx = tile >= a.tiles0;
    // (both segments' fields come in with the kernel arguments in one scalar-load burst and are selected here: a
    // dynamically indexed a.seg[sidx] is a chain of dependent scalar loads at the very start of the kernel)
    const SmlBwdSeg sg = sidx ? a.seg[1] : a.seg[0];
    const int row0 = (tile - (sidx ? a.tiles0 : 0)) * R;
    float cw_reg = 0.0f;
    if (tid < 104) cw_reg = sg.theta[tid];      // parked in LDS below, once the other loads are on their way
    // Both GEMMs' operand images fit a register ring whole at d <= 64 (dA2: D/16 k-steps x 4 tiles, dA1: 4 k-steps
    // x 5 tiles per wave): they are fetched right after the pair-loss inputs below (loads return in issue order,
    // so the loss stage does not wait for them) and neither GEMM starts with an L2 round trip
    // (PRE is the launcher's choice: d = 32 always; d = 64 by default -- 220 registers, one workgroup per CU, which is
    // all a small batch's grid asks for -- with the on-demand form (<= 128 registers, two workgroups per CU) kept
    // for grids larger than the chip; d = 128: the rings do not fit)
    static_assert(!PRE || D <= 64, "operand rings fit the register file at d <= 64 only");
    const f32x4* __restrict__ p2b = reinterpret_cast<const f32x4*>(sg.pk + sml_pk_p2b(D));
    const f32x4* __restrict__ p1b = reinterpret_cast<const f32x4*>(sg.pk + sml_pk_p1b(D));
    auto tile2 = [wv](int t) { return wv * 4 + t; };
    auto tile1 = [cs](int t) { return t * (D / 16) + cs; };
    auto nokofs = [](int) { return 0; };
    f32x4 ring2[PRE ? KSD : 1][4], ring1[PRE ? 4 : 1][5];

    // ---- pair loss (model/conv_transfer.py:120-134) for this tile's rows: every row fetches the three
    // transferred rows of its triple (each the sum of the forward's out_np hidden-slice planes, added in
    // plane order), one thread per row forms the two scores and the loss terms, then dOut is written
    // element-wise.  User tiles own the loss value (each triple once, coordinate slice 0).
    float* O3 = smem;                     // [3][R][D+1], aliases dZs (not yet live)
    float ou[EPT], oi[EPT], on[EPT];
    // every global load of this kernel is issued before the first value is consumed (one memory round trip, not
    // three): the planes of the three out rows here, then the tail's inputs, z1 and the operand rings
    float vu[EPT][SML_FWD_NS], vi[EPT][SML_FWD_NS], vn[EPT][SML_FWD_NS];
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
        const int e = q * 512 + tid, r = e / D, w = e % D;
        const int row = row0 + r;
        const bool inr = row < sg.n_rows;
        const int t = !inr ? 0 : ((sg.is_item && row >= a.B) ? row - a.B : row);
        const float* pu = a.out_all + (int64_t)t * D + w;
        const float* pi = a.out_all + (int64_t)(a.ioff + t) * D + w;
        const float* pn = a.out_all + (int64_t)(a.ioff + a.B + t) * D + w;
#pragma unroll
        for (int p = 0; p < SML_FWD_NS; ++p) {
            // unconditional loads (a dead plane or an out-of-range row re-reads a live address and is weighted by zero)
            const int64_t po = (int64_t)min(p, a.out_np - 1) * a.out_pstride;
            vu[q][p] = pu[po]; vi[q][p] = pi[po]; vn[q][p] = pn[po];
        }
    }
    // the tail's (x_t, x_hat, x_com) and this wave's z1 fragment: issue the loads now, use them later
    const int tr_ = tid >> 4, twl = tid & 15, tw = cs * 16 + twl;       // tail element of threads 0..255
    float x0 = 0.f, x1 = 0.f, x2 = 0.f;
    if (tid < 256) {
        const float* x = sg.xin + (int64_t)(row0 + tr_) * 3 * D;
        x0 = x[tw]; x1 = x[D + tw]; x2 = x[2 * D + tw];
    }
    float z[4][4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            z[t][q] = sg.z1[(int64_t)(row0 + 4 * g4 + q) * SML_HID + (wv * 4 + t) * 16 + l15];
    if constexpr (PRE) {
        ring_preload<4, KSD>(ring2, p2b, KSD, 0, lane, tile2, nokofs);
        ring_preload<5, 4>(ring1, p1b, 32, wv * 4, lane, tile1, nokofs);
    }
    if (tid < 104) cws[tid] = cw_reg;
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
        const int e = q * 512 + tid, r = e / D, w = e % D;
        const float inr = (row0 + r < sg.n_rows) ? 1.0f : 0.0f;
        ou[q] = oi[q] = on[q] = 0.0f;
#pragma unroll
        for (int p = 0; p < SML_FWD_NS; ++p) {      // planes added in index order
            const float live = p < a.out_np ? inr : 0.0f;
            ou[q] += live * vu[q][p]; oi[q] += live * vi[q][p]; on[q] += live * vn[q][p];
        }
        if constexpr (D > 64) {           // a row spans two wavefronts: the scores go through LDS
            O3[(0 * R + r) * (D + 1) + w] = ou[q];
            O3[(1 * R + r) * (D + 1) + w] = oi[q];
            O3[(2 * R + r) * (D + 1) + w] = on[q];
        }
    }
    float lsum = 0.0f;
    if constexpr (D <= 64) {
        // a row's D elements sit in D adjacent lanes of one wavefront (d = 32: half a wave, d = 64: a wave):
        // the three dot products are xor-shuffle sums, every lane of the row forms the loss terms itself --
        // no LDS round trip, no barrier before dOut
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const int e = q * 512 + tid, r = e / D, w = e % D;
            const int row = row0 + r;
            float sp = ou[q] * oi[q], sn = ou[q] * on[q], uu = ou[q] * ou[q];
#pragma unroll
            for (int off = D / 2; off >= 1; off >>= 1) {
                sp += __shfl_xor(sp, off, 64); sn += __shfl_xor(sn, off, 64); uu += __shfl_xor(uu, off, 64);
            }
            float lt, d0, d1, inv_nu = 1.0f, cc = 0.0f;
            if (a.kind == SML_LOSS_BPR_NORM || a.kind == SML_LOSS_BPR_UNIT) {
                const float nu = sqrtf(uu);
                inv_nu = 1.0f / nu;
                // BPR_NORM differentiates through the norm; BPR_UNIT's norm is detached (ConvTransfer.forward)
                cc = a.kind == SML_LOSS_BPR_NORM ? (sp - sn) / (nu * nu * nu) : 0.0f;
                pair_terms(SML_LOSS_BPR, (sp - sn) * inv_nu, 0.0f, 1.0f, lt, d0, d1);
                d1 = -d0;
            } else {
                pair_terms(a.kind, sp, sn, 1.0f / (float)a.B, lt, d0, d1);
            }
            d0 *= a.scale; d1 *= a.scale;
            if (!sg.is_item && cs == 0 && w == 0 && row < sg.n_rows) lsum += lt * a.scale;
            float g;
            if (a.kind == SML_LOSS_BPR_NORM || a.kind == SML_LOSS_BPR_UNIT) {
                if (!sg.is_item) g = d0 * ((oi[q] - on[q]) * inv_nu - cc * ou[q]);
                else g = ((row < a.B) ? d0 : d1) * ou[q] * inv_nu;
            } else {
                if (!sg.is_item) g = d0 * oi[q] + d1 * on[q];
                else g = ((row < a.B) ? d0 : d1) * ou[q];
            }
            if (row >= sg.n_rows) g = 0.0f;
            dOs[r * SD + w] = g;
            if (TR && cs == 0) st_out<SML_WT_BWD>(&sg.dout[(int64_t)row * D + w], g);
        }
    } else {
        __syncthreads();
        if (tid < R) {
            float sp = 0.f, sn = 0.f, uu = 0.f;
    #pragma unroll 8
            for (int w = 0; w < D; ++w) {
                const float u = O3[(0 * R + tid) * (D + 1) + w];
                sp += u * O3[(1 * R + tid) * (D + 1) + w];
                sn += u * O3[(2 * R + tid) * (D + 1) + w];
                uu += u * u;
            }
            float lt, d0, d1, inv_nu = 1.0f, cc = 0.0f;
            if (a.kind == SML_LOSS_BPR_NORM || a.kind == SML_LOSS_BPR_UNIT) {
                const float nu = sqrtf(uu);
                inv_nu = 1.0f / nu;
                // BPR_NORM differentiates through the norm; BPR_UNIT's norm is detached (ConvTransfer.forward)
                cc = a.kind == SML_LOSS_BPR_NORM ? (sp - sn) / (nu * nu * nu) : 0.0f;
                pair_terms(SML_LOSS_BPR, (sp - sn) * inv_nu, 0.0f, 1.0f, lt, d0, d1);
                d1 = -d0;
            } else {
                pair_terms(a.kind, sp, sn, 1.0f / (float)a.B, lt, d0, d1);
            }
            cf[0][tid] = d0 * a.scale; cf[1][tid] = d1 * a.scale; cf[2][tid] = inv_nu; cf[3][tid] = cc;
            if (!sg.is_item && cs == 0 && row0 + tid < sg.n_rows) lsum = lt * a.scale;
        }
        __syncthreads();
    #pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const int e = q * 512 + tid, r = e / D, w = e % D;
            const int row = row0 + r;
            const float d0 = cf[0][r], d1 = cf[1][r];
            float g;
            if (a.kind == SML_LOSS_BPR_NORM || a.kind == SML_LOSS_BPR_UNIT) {
                const float inv_nu = cf[2][r], cc = cf[3][r];
                if (!sg.is_item) g = d0 * ((oi[q] - on[q]) * inv_nu - cc * ou[q]);
                else g = ((row < a.B) ? d0 : d1) * ou[q] * inv_nu;
            } else {
                if (!sg.is_item) g = d0 * oi[q] + d1 * on[q];
                else g = ((row < a.B) ? d0 : d1) * ou[q];
            }
            if (row >= sg.n_rows) g = 0.0f;
            dOs[r * SD + w] = g;
            if (TR && cs == 0) st_out<SML_WT_BWD>(&sg.dout[(int64_t)row * D + w], g);
        }
    }
    __syncthreads();
    TL(2);
    // The tail's forward recomputation (conv prologue of this thread's element and the Gelu' factors) depends only on
    // the inputs loaded at the top: done HERE by waves 0..3, on the VALU, while the other waves already feed the MFMA
    // pipe with dA2 -- not after both GEMMs, where everybody would wait for it
    Pro pt;
    float gg2[SML_C2], gg1[SML_C1];
    if (tid < 256) {
        conv_prologue(cws, x0, x1, x2, pt);
#pragma unroll
        for (int c = 0; c < SML_C2; ++c) gg2[c] = sml_gelu_grad(pt.h2p[c]);
#pragma unroll
        for (int c = 0; c < SML_C1; ++c) gg1[c] = sml_gelu_grad(pt.h1p[c]);
    }
    // ---- dA2[R x 512] = dOut[R x D] * W2 ; dZ1 = dA2 * Gelu'(z1) ; wave wv owns column tiles 4wv..4wv+3
    {
        f32x4 acc[1][4];
        zero_acc(acc);
        if constexpr (PRE) mma16_ring<1, 4, KSD, KSD, false>(acc, ring2, dOs + l15 * SD + 4 * g4, 0, p2b, KSD, 0, lane, tile2, nokofs);
        else mma16_rows<1, 4, KSD, KSD>(acc, dOs + l15 * SD + 4 * g4, 0, p2b, KSD, 0, lane, tile2);
        float* dz1 = sg.dz1;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int n = (wv * 4 + t) * 16 + l15;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = 4 * g4 + q;
                const float dz = acc[0][t][q] * sml_gelu_grad(z[t][q]);
                dZs[r * S2 + n] = dz;
                if (TR && (t % CS) == cs) st_out<SML_WT_BWD>(&dz1[(int64_t)(row0 + r) * SML_HID + n], dz);   // the CS workgroups share the save
            }
        }
    }
    __syncthreads();
    TL(3);
    // ---- dA1[R x 5*16] = dZ1[R x 512] * W1[:, this slice] ; every wave takes 4 of the 32 k-steps, all 5 channels
    {
        f32x4 acc[1][5];
        zero_acc(acc);
        if constexpr (PRE) mma16_ring<1, 5, 4, 4, false>(acc, ring1, dZs + l15 * S2 + 4 * g4, 0, p1b, 32, wv * 4, lane, tile1, nokofs);
        else mma16_rows<1, 5, 4, 4>(acc, dZs + l15 * S2 + 4 * g4, 0, p1b, 32, wv * 4, lane, tile1);
        __syncthreads();                        // every wave is done reading dZs
#pragma unroll
        for (int t = 0; t < 5; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) part[(wv * R + 4 * g4 + q) * PSTR + t * 16 + l15] = acc[0][t][q];
    }
    __syncthreads();
    TL(4);
    // ---- per-coordinate tail (threads 0..255, one element each): Gelu'(h2) -> conv2^T -> Gelu'(h1) -> conv1^T (row 1 = x_hat)
    float cga[16], cgb[16];
    if (tid < 256) {
        const int row = row0 + tr_;
        const bool ok = row < sg.n_rows;
        const Pro& p = pt;
        float dh2p[SML_C2];
#pragma unroll
        for (int c = 0; c < SML_C2; ++c) {
            float s = 0.0f;
#pragma unroll
            for (int k = 0; k < 8; ++k) s += part[(k * R + tr_) * PSTR + c * 16 + twl];
            dh2p[c] = s * gg2[c];
        }
        float dxh = 0.0f;
        float dh1p[SML_C1];
#pragma unroll
        for (int c = 0; c < SML_C1; ++c) {
            float s = 0.0f;
#pragma unroll
            for (int o = 0; o < SML_C2; ++o) s += dh2p[o] * cws[SML_OFF_C2W + o * SML_C1 + c];
            dh1p[c] = s * gg1[c];
            dxh += dh1p[c] * cws[SML_OFF_C1W + c * 3 + 1];
        }
        if (!TR) {
            sg.dx[(int64_t)row * D + tw] = dxh + a.l2 * x1;
            if (ok) lsum += 0.5f * a.l2 * x1 * x1;      // + l2 * 0.5 * sum(x_hat^2), model/transfer.py:486-488
        }
        if constexpr (TR) {
            // conv1/conv2 parameter gradients are one small matrix product over the tile's elements e:
            //   G[i][j] = sum_e A[e][i] * B[e][j],  A[e] = (dh1p[0..9], dh2p[0..4], 0),  B[e] = (x0, x1, x2, 1, h1[0..9], 0, 0)
            // G[c][0..2] = dW_conv1[c], G[c][3] = db_conv1[c], G[10+o][4+c] = dW_conv2[o][c], G[10+o][3] = db_conv2[o]
#pragma unroll
            for (int c = 0; c < SML_C1; ++c) { cga[c] = ok ? dh1p[c] : 0.0f; cgb[4 + c] = p.h1[c]; }
#pragma unroll
            for (int o = 0; o < SML_C2; ++o) cga[10 + o] = ok ? dh2p[o] : 0.0f;
            cga[15] = 0.0f;
            cgb[0] = x0; cgb[1] = x1; cgb[2] = x2; cgb[3] = 1.0f; cgb[14] = 0.0f; cgb[15] = 0.0f;
        }
    }
    {   // this workgroup's share of the batch loss: lanes, then waves in index order (deterministic)
        float v = lsum;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0) lred[wv] = v;
    }
    if constexpr (TR) {
        float* cgA = smem;                 // [256][CGS]  (the dA1 partials are in registers by now)
        float* cgB = smem + 256 * CGS;
        __syncthreads();
        if (tid < 256) {
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4) {
                f32x4 va, vb;
#pragma unroll
                for (int e = 0; e < 4; ++e) { va[e] = cga[i4 * 4 + e]; vb[e] = cgb[i4 * 4 + e]; }
                *reinterpret_cast<f32x4*>(cgA + tid * CGS + i4 * 4) = va;
                *reinterpret_cast<f32x4*>(cgB + tid * CGS + i4 * 4) = vb;
            }
        }
        __syncthreads();
        {   // wave wv: elements 64*(wv&3) .. +63, k-steps 8*(wv>>2) .. +7 (4 elements per MFMA)
            const int e0 = 64 * (wv & 3) + 32 * (wv >> 2);
            float av[8], bv[8];
#pragma unroll
            for (int s8 = 0; s8 < 8; ++s8) {
                av[s8] = cgA[(e0 + 4 * s8 + g4) * CGS + l15];
                bv[s8] = cgB[(e0 + 4 * s8 + g4) * CGS + l15];
            }
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s8 = 0; s8 < 8; ++s8) acc = mfma16(av[s8], bv[s8], acc);
#pragma unroll
            for (int q = 0; q < 4; ++q) red[wv][(4 * g4 + q) * 16 + l15] = acc[q];
        }
        __syncthreads();
        if (tid < 95) {
            int i, j;
            if (tid < 30) { i = tid / 3; j = tid % 3; }
            else if (tid < 40) { i = tid - 30; j = 3; }
            else if (tid < 90) { i = 10 + (tid - 40) / 10; j = 4 + (tid - 40) % 10; }
            else { i = 10 + (tid - 90); j = 3; }
            float sacc = 0.0f;
#pragma unroll
            for (int w8 = 0; w8 < 8; ++w8) sacc += red[w8][i * 16 + j];
            a.convg_part[(int64_t)blockIdx.x * SML_CG + tid] = sacc;
        }
    } else {
        __syncthreads();
    }
    if (tid == 0) {
        float sacc = 0.0f;
#pragma unroll
        for (int w8 = 0; w8 < 8; ++w8) sacc += lred[w8];
        a.loss_part[blockIdx.x] = sacc;
        TL(7);
    }
    TL_DONE();
}

// ------------------------------------------------------------------------------------
// TR stage, restructured step (v2): the backward is cut where the weight gradients become computable.
//   k_tr_bwd_head   pair loss -> dOut -> dA2 = dOut * W2 -> dZ1 = dA2 * Gelu'(z1): everything the weight-gradient
//                   tiles need (dOut, dZ1), and nothing else.  HS = 4 workgroups per 16-row tile, one 128-column slice
//                   of the hidden layer each (pair loss repeated, 16 KB of W2 per workgroup instead of 64 KB + 160 KB
//                   of W1): 192 light workgroups for a 768-row batch.
//   k_tr_wgrad2     ONE launch holds the weight-gradient tiles AND the rest of the backward -- dA1 = dZ1 * W1, the
//                   per-coordinate tail and the conv1/conv2 parameter gradients, which in the TR stage feed nothing but
//                   190 conv parameters -- as extra workgroups dispatched first.  The 5 us of dA1 + tail leave the
//                   step's critical path: they run BESIDE the dW1 / dW2 tiles (two 512-thread workgroups fit a CU:
//                   <= 128 VGPRs, 50 KB of LDS each).  The last tail workgroup to arrive sums the partials in index
//                   order (deterministic) and takes the conv parameters' Adam step.
// ------------------------------------------------------------------------------------
// The leading scalar parameters are what the kernel's first round of loads needs; built with -amdgpu-kernarg-preload-count=16
// (sml_amd/build.py) they arrive in SGPRs WITH the wavefront instead of behind a scalar-load round trip of the argument
// segment (tools/launch_boundary_probe.hip: 0.22-0.24 us per dependent kernel on this part); the by-value struct that follows
// carries the rest, which nothing waits for before the first loads are out.  (13 SGPRs: three pointers, one int64, five ints.)
template <int D>
__global__ __launch_bounds__(512) void k_tr_bwd_head(const float* __restrict__ p_out_all, const float* __restrict__ p_z1, const float* __restrict__ p_pk,
                                                     long long p_out_pstride, int p_B, int p_ioff, int p_tiles0, int p_tiles_total, int p_out_np, SmlBwdArgs a) {
    constexpr int R = SML_TM;
    constexpr int HS = 4;                 // hidden slices (workgroups) per row tile
    constexpr int SD = D + 4;
    constexpr int KSD = D / 16;
    constexpr int EPT = R * D / 512;
    static_assert(SML_HID / 16 / HS == 8, "one dA2 column tile per wave");
    __shared__ __attribute__((aligned(16))) float smem[cmax(R * SD, 3 * R * (D + 1))];
    __shared__ float cf[4][SML_TM];
    __shared__ float lred[8];
    float* dOs = smem;                    // [R][D+4]
    float* O3 = smem;                     // d = 128: [3][R][D+1] score staging (dead before dOs is written)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, g4 = lane >> 4;
    TL_BEGIN(2); TL_PREV();
    // same block -> (tile, slice) map as the hidden-split forward: slice hs on XCDs {2 hs, 2 hs + 1}
    const int x8 = (int)blockIdx.x % 8;
    const int hs = x8 / 2, tile = 2 * ((int)blockIdx.x / 8) + (x8 % 2);
    if (tile >= p_tiles_total) return;
    const int sidx = tile >= p_tiles0;
    const SmlBwdSeg sg = sidx ? a.seg[1] : a.seg[0];      // (the stores' pointers: needed at the end)
    const int row0 = (tile - (sidx ? p_tiles0 : 0)) * R;
    const int n_rows = sidx ? 2 * p_B : p_B;               // = n_rows
    const float* __restrict__ z1_seg = p_z1 + (sidx ? (int64_t)p_ioff * SML_HID : 0);       // = sg.z1 (slot0 = ioff)
    const f32x4* __restrict__ p2b = reinterpret_cast<const f32x4*>(p_pk + (sidx ? sml_pk_size(D) : 0) + sml_pk_p2b(D));   // = sg.pk + ...
    const int ct = hs * 8 + wv;           // this wave's dA2 column tile
    // ---- every global load first: the three out rows' planes, this wave's z1 fragment, its W2 operand image
    float vu[EPT][SML_FWD_NS], vi[EPT][SML_FWD_NS], vn[EPT][SML_FWD_NS];
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
        const int e = q * 512 + tid, r = e / D, w = e % D;
        const int row = row0 + r;
        const bool inr = row < n_rows;
        const int t = !inr ? 0 : ((sidx && row >= p_B) ? row - p_B : row);
        const float* pu = p_out_all + (int64_t)t * D + w;
        const float* pi = p_out_all + (int64_t)(p_ioff + t) * D + w;
        const float* pn = p_out_all + (int64_t)(p_ioff + p_B + t) * D + w;
#pragma unroll
        for (int p = 0; p < SML_FWD_NS; ++p) {
            const int64_t po = (int64_t)min(p, p_out_np - 1) * p_out_pstride;
            vu[q][p] = pu[po]; vi[q][p] = pi[po]; vn[q][p] = pn[po];
        }
    }
    float z[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) z[q] = z1_seg[(int64_t)(row0 + 4 * g4 + q) * SML_HID + ct * 16 + l15];
    f32x4 ring[KSD];
#pragma unroll
    for (int ks = 0; ks < KSD; ++ks) ring[ks] = p2b[(ct * KSD + ks) * 64 + lane];
    float ou[EPT], oi[EPT], on[EPT];
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
        const int e = q * 512 + tid, r = e / D, w = e % D;
        const float inr = (row0 + r < n_rows) ? 1.0f : 0.0f;
        ou[q] = oi[q] = on[q] = 0.0f;
#pragma unroll
        for (int p = 0; p < SML_FWD_NS; ++p) {      // planes added in index order
            const float live = p < p_out_np ? inr : 0.0f;
            ou[q] += live * vu[q][p]; oi[q] += live * vi[q][p]; on[q] += live * vn[q][p];
        }
        if constexpr (D > 64) {
            O3[(0 * R + r) * (D + 1) + w] = ou[q];
            O3[(1 * R + r) * (D + 1) + w] = oi[q];
            O3[(2 * R + r) * (D + 1) + w] = on[q];
        }
    }
    float lsum = 0.0f;
    auto coeffs = [&](float sp, float sn, float uu, float& lt, float& d0, float& d1, float& inv_nu, float& cc) {
        inv_nu = 1.0f; cc = 0.0f;
        if (a.kind == SML_LOSS_BPR_NORM || a.kind == SML_LOSS_BPR_UNIT) {
            const float nu = sqrtf(uu);
            inv_nu = 1.0f / nu;
            cc = a.kind == SML_LOSS_BPR_NORM ? (sp - sn) / (nu * nu * nu) : 0.0f;
            pair_terms(SML_LOSS_BPR, (sp - sn) * inv_nu, 0.0f, 1.0f, lt, d0, d1);
            d1 = -d0;
        } else {
            pair_terms(a.kind, sp, sn, 1.0f / (float)p_B, lt, d0, d1);
        }
    };
    auto dout_of = [&](int row, float d0, float d1, float inv_nu, float cc, float u, float i, float n) {
        float g;
        if (a.kind == SML_LOSS_BPR_NORM || a.kind == SML_LOSS_BPR_UNIT) {
            if (!sidx) g = d0 * ((i - n) * inv_nu - cc * u);
            else g = ((row < p_B) ? d0 : d1) * u * inv_nu;
        } else {
            if (!sidx) g = d0 * i + d1 * n;
            else g = ((row < p_B) ? d0 : d1) * u;
        }
        return row >= n_rows ? 0.0f : g;
    };
    if constexpr (D <= 64) {
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const int e = q * 512 + tid, r = e / D, w = e % D;
            const int row = row0 + r;
            float sp = ou[q] * oi[q], sn = ou[q] * on[q], uu = ou[q] * ou[q];
#pragma unroll
            for (int off = D / 2; off >= 1; off >>= 1) {
                sp += __shfl_xor(sp, off, 64); sn += __shfl_xor(sn, off, 64); uu += __shfl_xor(uu, off, 64);
            }
            float lt, d0, d1, inv_nu, cc;
            coeffs(sp, sn, uu, lt, d0, d1, inv_nu, cc);
            d0 *= a.scale; d1 *= a.scale;
            if (!sidx && hs == 0 && w == 0 && row < n_rows) lsum += lt * a.scale;
            const float g = dout_of(row, d0, d1, inv_nu, cc, ou[q], oi[q], on[q]);
            dOs[r * SD + w] = g;
            if (hs == 0) st_out<SML_WT_BWD>(&sg.dout[(int64_t)row * D + w], g);
        }
    } else {
        __syncthreads();
        if (tid < R) {
            float sp = 0.f, sn = 0.f, uu = 0.f;
#pragma unroll 8
            for (int w = 0; w < D; ++w) {
                const float u = O3[(0 * R + tid) * (D + 1) + w];
                sp += u * O3[(1 * R + tid) * (D + 1) + w];
                sn += u * O3[(2 * R + tid) * (D + 1) + w];
                uu += u * u;
            }
            float lt, d0, d1, inv_nu, cc;
            coeffs(sp, sn, uu, lt, d0, d1, inv_nu, cc);
            cf[0][tid] = d0 * a.scale; cf[1][tid] = d1 * a.scale; cf[2][tid] = inv_nu; cf[3][tid] = cc;
            if (!sidx && hs == 0 && row0 + tid < n_rows) lsum = lt * a.scale;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const int e = q * 512 + tid, r = e / D, w = e % D;
            const int row = row0 + r;
            const float g = dout_of(row, cf[0][r], cf[1][r], cf[2][r], cf[3][r], ou[q], oi[q], on[q]);
            dOs[r * SD + w] = g;
            if (hs == 0) st_out<SML_WT_BWD>(&sg.dout[(int64_t)row * D + w], g);
        }
    }
    __syncthreads();
    TL(2);
    // ---- dA2 tile = dOut[R x D] * W2[:, this tile] ; dZ1 = dA2 * Gelu'(z1)
    {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KSD; ++ks) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(dOs + l15 * SD + 4 * g4 + ks * 16);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = mfma16(av[e], ring[ks][e], acc);
        }
        const int n = ct * 16 + l15;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            st_out<SML_WT_BWD>(&sg.dz1[(int64_t)(row0 + 4 * g4 + q) * SML_HID + n], acc[q] * sml_gelu_grad(z[q]));
    }
    TL(3);
    {
        float v = lsum;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0) lred[wv] = v;
    }
    __syncthreads();
    if (tid == 0) {
        float sacc = 0.0f;
#pragma unroll
        for (int w8 = 0; w8 < 8; ++w8) sacc += lred[w8];
        a.loss_part[tile * HS + hs] = sacc;
        TL(7);
    }
    TL_DONE();
}

template <int D>
__device__ __forceinline__ void pack_store(float* __restrict__ pk, int off, float p) {
    constexpr int K1 = SML_C2 * D;
    if (off >= SML_OFF_F1W && off < sml_off_f1b(D)) {
        const int n = (off - SML_OFF_F1W) / K1, k = (off - SML_OFF_F1W) % K1;
        pk[sml_pk_p1(D) + pk_pos(K1 / 16, n, k)] = p;
        pk[sml_pk_p1b(D) + pk_pos(SML_HID / 16, k, n)] = p;
    } else if (off >= sml_off_f2w(D) && off < sml_off_f2b(D)) {
        const int j = (off - sml_off_f2w(D)) / SML_HID, n = (off - sml_off_f2w(D)) % SML_HID;
        pk[sml_pk_p2(D) + pk_pos(SML_HID / 16, j, n)] = p;
        pk[sml_pk_p2b(D) + pk_pos(D / 16, n, j)] = p;
    }
}

// ------------------------------------------------------------------------------------
// weight gradients (TR stage): dW1 = dZ1^T A1, db1, dW2 = dOut^T Gelu(z1), db2
// one workgroup per 32x32 output tile (v_mfma_f32_32x32x2_f32); the eight waves split the batch
// rows, each with up to 64 rows of operands (64 loads) in flight before its MFMAs.  The Adam state
// of the tile's weights does not depend on the reduction, so it is fetched first and is in
// registers by the time the gradient is complete.
// ------------------------------------------------------------------------------------
#define SML_WG_TILE_SMEM (8 * 32 * 33 + 8 * 32 + 32 * 33)     // floats of LDS one weight-gradient tile workgroup uses
// one 32x32 weight-gradient tile (+ fused Adam + operand-image refresh): shared by k_transfer_wgrad and k_tr_wgrad2.
// (xcd, kk): the XCD this workgroup runs on and its index among that XCD's tiles.
template <int D>
__device__ __forceinline__ void wgrad_tile(const SmlWgArgs& a, const SmlWgSeg& sg, int xcd, int kk, float* smem_wg, long long* tl_rec) {
    constexpr int K1 = SML_C2 * D;
    constexpr int KT = K1 / 32;
    constexpr int JT = D / 32;
    constexpr int T1 = 16 * KT, T2 = JT * 16, TN = T1 + T2;
    constexpr int NS = sml_net_size(D);
    float* part = smem_wg;                 // [8][32][33]
    float* csum = smem_wg + 8 * 32 * 33;   // [8][32]
    float* wt = csum + 8 * 32;             // [32][33] the tile's updated weights, for the operand-image refresh
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const bool fuse = a.theta != nullptr;
    SmlSched sc; sc.step_size = a.step_size; sc.bc2_sqrt = a.bc2_sqrt;
    // XCD-aware tile map.  Every operand of this kernel was written by the previous launches on OTHER XCDs, so a
    // tile's first read of each operand line crosses the fabric; workgroup b runs on XCD b % 8 (observed
    // dispatch order: affinity only), so XCD x takes, for both nets, the dW1 tiles of hidden blocks {2x, 2x+1}
    // (one eighth of dZ1's columns, all of A1) and the dW2 tiles of the same hidden blocks (one eighth of z1):
    // each XCD pulls an eighth of the two big operands instead of nearly all of them
    static_assert(TN % 8 == 0 && T2 == 16 * JT, "tile map");
    constexpr int PER = 2 * (KT + JT);                 // tiles per XCD per net
    const int net = 1 - kk / PER, rr = kk % PER;
    // (sg: this tile's net's segment = a.seg[net], handed in by the caller -- k_tr_wgrad2 patches its copies from preloaded parameters)
    const bool is_w1 = rr < 2 * KT;
    int ti, tj;                       // tile along output rows / cols
    if (is_w1) { ti = 2 * xcd + rr / KT; tj = rr % KT; } else { tj = 2 * xcd + (rr - 2 * KT) / JT; ti = (rr - 2 * KT) % JT; }
    const float* __restrict__ Asrc = is_w1 ? sg.dz1 : sg.dout;   // A[i][r] = Asrc[r][ti*32 + i]
    const int lda = is_w1 ? SML_HID : D;
    const float* __restrict__ Bsrc = is_w1 ? sg.a1 : sg.a2;      // B[r][j] = Bsrc[r][tj*32 + j]  (a2 = Gelu(z1), saved by the forward)
    const int ldb = is_w1 ? K1 : SML_HID;
    const bool gelu_b = !is_w1 && a.gelu_b != 0;                 // (workgroup-uniform)
    // this thread's two weights of the tile (+ a bias for 32 threads of the tj = 0 workgroups)
    int woff[2];
    float wp[2] = {0.f, 0.f}, wm[2] = {0.f, 0.f}, wvv[2] = {0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int e = q * 512 + tid, i = e >> 5, j = e & 31;
        woff[q] = is_w1 ? SML_OFF_F1W + (ti * 32 + i) * K1 + tj * 32 + j
                        : sml_off_f2w(D) + (ti * 32 + i) * SML_HID + tj * 32 + j;
    }
    const bool has_bias = (tj == 0 && tid < 32);
    const int boff = is_w1 ? sml_off_f1b(D) + ti * 32 + tid : sml_off_f2b(D) + ti * 32 + tid;
    float bp = 0.f, bm = 0.f, bv2 = 0.f;
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.0f;
    float colsum = 0.0f;
    // each wave takes a contiguous eighth of the batch rows, 64 rows (8 k-steps of 2x4 rows) per trip
    const int rows_per_wave = ((sg.n_rows + 255) / 256) * 32;
    const int r_begin = wv * rows_per_wave;
    const int r_end = min(sg.n_rows, r_begin + rows_per_wave);
    for (int rb = r_begin; rb < r_end; rb += 64) {
        float av[8][4], bv[8][4];
        const bool two = rb + 32 < r_end;          // wave-uniform
#pragma unroll
        for (int s4 = 0; s4 < 8; ++s4) {
            if (s4 >= 4 && !two) break;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = rb + s4 * 8 + 4 * hi + e;
                const bool ok = r < r_end;
                av[s4][e] = ok ? Asrc[(int64_t)r * lda + ti * 32 + l31] : 0.0f;
                bv[s4][e] = ok ? Bsrc[(int64_t)r * ldb + tj * 32 + l31] : 0.0f;
            }
        }
#pragma unroll
        for (int s4 = 0; s4 < 8; ++s4) {
            if (s4 >= 4 && !two) break;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float b = gelu_b ? sml_gelu(bv[s4][e]) : bv[s4][e];
                colsum += av[s4][e];
                acc = mfma32(av[s4][e], b, acc);
            }
        }
    }
    if (fuse) {
        // the tile's Adam state, BEHIND the operand loads and the products' issue (round 6): the operands' pointers arrive with the
        // wavefront (k_tr_wgrad2's preloaded parameters), theta / m / v come with the argument segment -- issued first, as before,
        // these loads held the operands back by that round trip; here their latency runs under the matrix pipe's tail and the
        // LDS reduction (the state was written by this very tile's workgroup of the previous step: the same XCD's L2)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int64_t i = (int64_t)net * NS + woff[q];
            wp[q] = a.theta[i]; wm[q] = a.m[i]; wvv[q] = a.v[i];
        }
        if (has_bias) { const int64_t i = (int64_t)net * NS + boff; bp = a.theta[i]; bm = a.m[i]; bv2 = a.v[i]; }
    }
    TL(2);
#pragma unroll
    for (int q = 0; q < 16; ++q) part[(wv * 32 + mfma32_row(q, lane)) * 33 + l31] = acc[q];
    colsum += __shfl_xor(colsum, 32, 64);
    if (lane < 32) csum[wv * 32 + lane] = colsum;
    __syncthreads();
    TL(3);
    float* __restrict__ g = sg.grad;
    auto finish = [&](int off, float gsum, float p, float m, float v) -> float {
        if (g) g[off] = gsum;
        if (fuse) {
            const int64_t i = (int64_t)net * NS + off;
            adam_apply(p, m, v, adam_wd(gsum, a.weight_decay, p), sc);
            a.theta[i] = p; a.m[i] = m; a.v[i] = v;
        }
        return p;
    };
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int e = q * 512 + tid, i = e >> 5, j = e & 31;
        float s = 0.0f;
#pragma unroll
        for (int w8 = 0; w8 < 8; ++w8) s += part[(w8 * 32 + i) * 33 + j];
        const float pnew = finish(woff[q], s, wp[q], wm[q], wvv[q]);
        wt[i * 33 + j] = a.peer.world > 0 ? s : pnew;      // several GPUs: the tile's GRADIENT is staged for the push
    }
    if (has_bias) {
        float s = 0.0f;
#pragma unroll
        for (int w8 = 0; w8 < 8; ++w8) s += csum[w8 * 32 + tid];
        finish(boff, s, bp, bm, bv2);
        for (int q = 0; q < a.peer.world; ++q) peer_store(a.peer.dst[q] + (int64_t)net * NS + boff, s);
    }
    if (a.peer.world > 0) {
        // several GPUs: the finished tile goes straight into slot [parity][this rank] of EVERY rank's inbox, 16 bytes per
        // lane (a tile row is 128 contiguous bytes of the flat gradient), then one counter increment per destination
        __syncthreads();
        if (tid < 256) {
            const int i = tid >> 3, c4 = tid & 7;
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = wt[i * 33 + c4 * 4 + e];
            const int64_t o = (int64_t)net * NS + (is_w1 ? SML_OFF_F1W + (ti * 32 + i) * K1 + tj * 32 + c4 * 4
                                                         : sml_off_f2w(D) + (ti * 32 + i) * SML_HID + tj * 32 + c4 * 4);
            for (int q = 0; q < a.peer.world; ++q) peer_store16(a.peer.dst[q] + o, v);
        }
        peer_signal(a.peer);
    }
    if (!fuse) return;
    // Operand-image refresh.  A 32x32 weight tile is four whole (column tile, k-step) blocks of 64 lanes x 4 floats
    // in EACH of its two images (forward and backward GEMM), i.e. eight contiguous 1 KB runs: one coalesced
    // 16-byte store per thread instead of four scattered 4-byte ones.
    __syncthreads();
    {
        const int img = tid >> 8, blk = (tid >> 6) & 3, ct = blk >> 1, ks = blk & 1, l = tid & 63;
        const int r0 = ti * 32, c0 = tj * 32;          // the tile's first weight row / column
        float* __restrict__ pkn = a.pk + (int64_t)net * sml_pk_size(D);
        f32x4 v;
        int64_t base;
        if (img == 0) {
            // forward image: columns = weight rows, reduction = weight columns
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = wt[(ct * 16 + (l & 15)) * 33 + ks * 16 + 4 * (l >> 4) + e];
            const int ksteps = is_w1 ? K1 / 16 : SML_HID / 16;
            base = (is_w1 ? sml_pk_p1(D) : sml_pk_p2(D)) + ((int64_t)((r0 >> 4) + ct) * ksteps + ((c0 >> 4) + ks)) * 256;
        } else {
            // backward image: columns = weight columns, reduction = weight rows
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = wt[(ks * 16 + 4 * (l >> 4) + e) * 33 + ct * 16 + (l & 15)];
            const int ksteps = is_w1 ? SML_HID / 16 : D / 16;
            base = (is_w1 ? sml_pk_p1b(D) : sml_pk_p2b(D)) + ((int64_t)((c0 >> 4) + ct) * ksteps + ((r0 >> 4) + ks)) * 256;
        }
        st_out16_wt(pkn + base + l * 4, v);      // (write-through: the next forward reads these from other XCDs)
    }
    TL(7);
}

template <int D>
__global__ __launch_bounds__(512) void k_transfer_wgrad(SmlWgArgs a) {
    __shared__ __attribute__((aligned(16))) float smem_wg[SML_WG_TILE_SMEM];
    TL_BEGIN(3); TL_PREV();
    constexpr int NS = sml_net_size(D);
    const int tid = threadIdx.x;
    const bool fuse = a.theta != nullptr;
    SmlSched sc; sc.step_size = a.step_size; sc.bc2_sqrt = a.bc2_sqrt;
    // Dispatch order = start order (the 194 workgroups start over ~1.3 us) and the work is uneven: the item net has
    // twice the user net's rows, so its tiles run ~2 us longer, and the two conv-parameter workgroups are a serial
    // chain of partial sums.  The long ones take the LOW block indices: conv workgroups first, then the item net's
    // tiles, the user net's last (per-workgroup end times from the in-kernel timeline: median 6.4, max 8.7 us before).
    if ((int)blockIdx.x < 2) {
        // conv1/conv2 parameters of one net: sum the backward workgroups' partials in order, then Adam
        const int net = (int)blockIdx.x;
        if (tid < 95) {
            const int off = tid < 30 ? tid : tid < 40 ? tid + 2 : tid < 90 ? tid + 4 : tid + 6;
            const int t0 = net ? a.tiles0 : 0, t1 = net ? a.tiles_total : a.tiles0;
            const int64_t i = (int64_t)net * NS + off;
            float p = 0.f, m = 0.f, v = 0.f;
            if (fuse) { p = a.theta[i]; m = a.m[i]; v = a.v[i]; }
            float g = 0.0f;
            int t = t0;
            for (; t + 8 <= t1; t += 8) {
                float x[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) x[u] = a.convg_part[(int64_t)(t + u) * SML_CG + tid];
#pragma unroll
                for (int u = 0; u < 8; ++u) g += x[u];
            }
            for (; t < t1; ++t) g += a.convg_part[(int64_t)t * SML_CG + tid];
            if (a.seg[net].grad) a.seg[net].grad[off] = g;    // the flat gradient is complete after this launch (null: nobody reads it)
            for (int q = 0; q < a.peer.world; ++q) peer_store(a.peer.dst[q] + i, g);
            if (fuse) {
                adam_apply(p, m, v, adam_wd(g, a.weight_decay, p), sc);
                a.theta[i] = p; a.m[i] = m; a.v[i] = v;
            }
        }
        if (a.peer.world > 0) peer_signal(a.peer);
        TL(7);
        TL_DONE();
        return;
    }
    {
        const int kk = ((int)blockIdx.x - 2) / 8;
        const SmlWgSeg sg = (kk / (2 * (SML_C2 * D / 32 + D / 32))) ? a.seg[0] : a.seg[1];      // net = 1 - kk / PER (wgrad_tile's map)
        wgrad_tile<D>(a, sg, (int)blockIdx.x % 8, kk, smem_wg, tl_rec);    // (8 consecutive blocks: one per XCD)
    }
    TL_DONE();
}

// The merged launch of the restructured TR step (see k_tr_bwd_head).  Workgroups [0, n_tail) are the backward's
// tail -- they have the longest chain and are dispatched first --, the rest are the weight-gradient tiles.
// (round 6: the first round of loads' operands as 14 preloaded dwords, as k_transfer_fwd -- both segments' operand arrays are one
// allocation each with the item segment at slot0 = B rounded up to SML_R rows; the launcher checks exactly that)
template <int D>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_tr_wgrad2(const float* __restrict__ p_dz1, const float* __restrict__ p_a1,
        const float* __restrict__ p_dout, const float* __restrict__ p_a2, const float* __restrict__ p_pk_net, const float* __restrict__ p_xin, int p_B, int p_n_tail,
        SmlWgArgs a) {
    // (only the two segment descriptors are copied and patched -- plain structs the compiler keeps in SGPRs; the argument struct itself
    // holds arrays that are indexed dynamically (peer.dst[q]): a local copy of IT would live in scratch memory)
    SmlWgSeg seg0 = a.seg[0], seg1 = a.seg[1];
    {
        const int64_t slot1 = (int64_t)SML_R * ((p_B + SML_R - 1) / SML_R);
        seg0.dz1 = p_dz1; seg1.dz1 = p_dz1 + slot1 * SML_HID;
        seg0.a1 = p_a1; seg1.a1 = p_a1 + slot1 * SML_C2 * D;
        seg0.dout = p_dout; seg1.dout = p_dout + slot1 * D;
        seg0.a2 = p_a2; seg1.a2 = p_a2 + slot1 * SML_HID;
        seg0.pk_net = p_pk_net; seg1.pk_net = p_pk_net + sml_pk_size(D);
        seg0.xin = p_xin; seg1.xin = p_xin + slot1 * 3 * D;
        seg0.n_rows = p_B; seg1.n_rows = 2 * p_B;
    }
    const int h_n_tail = p_n_tail, h_tiles0 = (p_B + SML_TM - 1) / SML_TM, h_tiles_total = h_tiles0 + (2 * p_B + SML_TM - 1) / SML_TM;
    constexpr int R = SML_TM;
    constexpr int CS = D / 16;
    constexpr int S2 = SML_HID + 4;
    constexpr int PSTR = SML_C2 * 16 + 1;
    constexpr int CGS = 20;
    constexpr int SZ0 = cmax(cmax(R * S2, 8 * R * PSTR), 2 * 256 * CGS);
    constexpr int TAIL_SMEM = SZ0 + 104 + 8 * 256;
    constexpr int NS = sml_net_size(D);
    __shared__ __attribute__((aligned(16))) float smem[cmax(TAIL_SMEM, SML_WG_TILE_SMEM)];
    __shared__ int s_last;
    TL_BEGIN(3); TL_PREV();
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, l15 = lane & 15, g4 = lane >> 4;
    if ((int)blockIdx.x >= h_n_tail) {
        const int vb = (int)blockIdx.x - h_n_tail;
        const SmlWgSeg sgt = ((vb / 8) / (2 * (SML_C2 * D / 32 + D / 32))) ? SmlWgSeg(seg0) : SmlWgSeg(seg1);      // (by value: net = 1 - kk / PER, wgrad_tile's map)
        wgrad_tile<D>(a, sgt, (int)blockIdx.x % 8, vb / 8, smem, tl_rec);
        TL_DONE();
        return;
    }
    // ---------------- tail role: dA1 = dZ1 * W1[:, slice], per-coordinate tail, conv-gradient partial
    float* part = smem;                   // [8][R][81]
    float* cgA = smem;                    // [256][CGS], aliases part after the tail
    float* cgB = smem + 256 * CGS;
    float* cws = smem + SZ0;
    float* red = smem + SZ0 + 104;        // [8][256]
    const int tb = (int)blockIdx.x;
    const int tile = tb / CS, cs = tb % CS;
    const bool live = tile < h_tiles_total;         // (an empty batch still has one tail workgroup: it elects itself)
    if (live) {
        const int sidx = tile >= h_tiles0;
        const SmlWgSeg sg = sidx ? seg1 : seg0;
        const int row0 = (tile - (sidx ? h_tiles0 : 0)) * R;
        const int tr_ = tid >> 4, twl = tid & 15, tw = cs * 16 + twl;       // tail element of threads 0..255
        float x0 = 0.f, x1 = 0.f, x2 = 0.f;
        if (tid < 256) {
            const float* x = sg.xin + (int64_t)(row0 + tr_) * 3 * D;
            x0 = x[tw]; x1 = x[D + tw]; x2 = x[2 * D + tw];
        }
        // dA1[R x 5*16] = dZ1[R x 512] * W1[:, slice]: every wave takes 4 of the 32 k-steps, all 5 channels.  Each wave's
        // dZ1 fragment is a DISJOINT k-range of the tile (read once per workgroup overall): it goes from global memory
        // straight into the MFMA A operand (lane: row l&15, 16 bytes at k = 16 ks + 4 (l>>4)) -- no LDS staging, no
        // barrier before the products -- and the first two k-steps of the W1 image are fetched in the same round trip.
        const f32x4* __restrict__ p1b = reinterpret_cast<const f32x4*>(sg.pk_net + sml_pk_p1b(D));
        auto tile1 = [cs](int t) { return t * (D / 16) + cs; };
        f32x4 av[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            av[ks] = *reinterpret_cast<const f32x4*>(sg.dz1 + (int64_t)(row0 + l15) * SML_HID + (wv * 4 + ks) * 16 + 4 * g4);
        f32x4 ring[2][5];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int t = 0; t < 5; ++t) ring[i][t] = p1b[(tile1(t) * 32 + wv * 4 + i) * 64 + lane];
        // (the conv weights' pointer is the one operand of this role that comes with the argument segment, not with the wavefront:
        // its load goes out behind the preloaded ones)
        __builtin_amdgcn_sched_barrier(0);
        if (tid < 104) cws[tid] = sg.theta_net[tid];
        __builtin_amdgcn_sched_barrier(0x86);
        __syncthreads();                            // cws is in LDS
        Pro pt;
        float gg2[SML_C2], gg1[SML_C1];
        {
            f32x4 acc[5];
#pragma unroll
            for (int t = 0; t < 5; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
                for (int t = 0; t < 5; ++t)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[t] = mfma16(av[ks][e], ring[ks & 1][t][e], acc[t]);
                if (ks + 2 < 4) {
#pragma unroll
                    for (int t = 0; t < 5; ++t) ring[ks & 1][t] = p1b[(tile1(t) * 32 + wv * 4 + ks + 2) * 64 + lane];
                }
                __builtin_amdgcn_sched_barrier(0x86);
            }
            // the tail's forward recomputation runs on the VALU while the matrix pipe drains the products above (the operand
            // registers are free by now: computed earlier, these 25 values pushed the kernel past 128 VGPRs into spills)
            if (tid < 256) {
                conv_prologue(cws, x0, x1, x2, pt);
#pragma unroll
                for (int c = 0; c < SML_C2; ++c) gg2[c] = sml_gelu_grad(pt.h2p[c]);
#pragma unroll
                for (int c = 0; c < SML_C1; ++c) gg1[c] = sml_gelu_grad(pt.h1p[c]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 5; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) part[(wv * R + 4 * g4 + q) * PSTR + t * 16 + l15] = acc[t][q];
        }
        __syncthreads();
        float cga[16], cgb[16];
        if (tid < 256) {
            const bool ok = row0 + tr_ < sg.n_rows;
            float dh2p[SML_C2];
#pragma unroll
            for (int c = 0; c < SML_C2; ++c) {
                float s2 = 0.0f;
#pragma unroll
                for (int k = 0; k < 8; ++k) s2 += part[(k * R + tr_) * PSTR + c * 16 + twl];
                dh2p[c] = s2 * gg2[c];
            }
            float dh1p[SML_C1];
#pragma unroll
            for (int c = 0; c < SML_C1; ++c) {
                float s2 = 0.0f;
#pragma unroll
                for (int o = 0; o < SML_C2; ++o) s2 += dh2p[o] * cws[SML_OFF_C2W + o * SML_C1 + c];
                dh1p[c] = s2 * gg1[c];
            }
            // G[i][j] = sum_e A[e][i] * B[e][j],  A[e] = (dh1p[0..9], dh2p[0..4], 0),  B[e] = (x0, x1, x2, 1, h1[0..9], 0, 0)
#pragma unroll
            for (int c = 0; c < SML_C1; ++c) { cga[c] = ok ? dh1p[c] : 0.0f; cgb[4 + c] = pt.h1[c]; }
#pragma unroll
            for (int o = 0; o < SML_C2; ++o) cga[10 + o] = ok ? dh2p[o] : 0.0f;
            cga[15] = 0.0f;
            cgb[0] = x0; cgb[1] = x1; cgb[2] = x2; cgb[3] = 1.0f; cgb[14] = 0.0f; cgb[15] = 0.0f;
        }
        __syncthreads();                            // the dA1 partials are in registers by now
        if (tid < 256) {
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4) {
                f32x4 va, vb4;
#pragma unroll
                for (int e = 0; e < 4; ++e) { va[e] = cga[i4 * 4 + e]; vb4[e] = cgb[i4 * 4 + e]; }
                *reinterpret_cast<f32x4*>(cgA + tid * CGS + i4 * 4) = va;
                *reinterpret_cast<f32x4*>(cgB + tid * CGS + i4 * 4) = vb4;
            }
        }
        __syncthreads();
        {
            const int e0 = 64 * (wv & 3) + 32 * (wv >> 2);
            float av[8], bv[8];
#pragma unroll
            for (int s8 = 0; s8 < 8; ++s8) {
                av[s8] = cgA[(e0 + 4 * s8 + g4) * CGS + l15];
                bv[s8] = cgB[(e0 + 4 * s8 + g4) * CGS + l15];
            }
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s8 = 0; s8 < 8; ++s8) acc = mfma16(av[s8], bv[s8], acc);
#pragma unroll
            for (int q = 0; q < 4; ++q) red[wv * 256 + (4 * g4 + q) * 16 + l15] = acc[q];
        }
        __syncthreads();
        if (tid < 95) {
            int i, j;
            if (tid < 30) { i = tid / 3; j = tid % 3; }
            else if (tid < 40) { i = tid - 30; j = 3; }
            else if (tid < 90) { i = 10 + (tid - 40) / 10; j = 4 + (tid - 40) % 10; }
            else { i = 10 + (tid - 90); j = 3; }
            float sacc = 0.0f;
#pragma unroll
            for (int w8 = 0; w8 < 8; ++w8) sacc += red[w8 * 256 + i * 16 + j];
            st_out<2>(&a.convg_out[(int64_t)tb * SML_CG + tid], sacc);
        }
    }
    TL(4);
    // Deferred form (every batch of an epoch but the last, one GPU): the partials are all this launch owes the conv
    // parameters -- the next batch's forward adds them (same order as below) and takes the Adam step in its prologue,
    // under its gather's round trips.  The election, the last arriver's re-read and its serial tail (3 us at the END
    // of this launch's chain) are gone.
    if (a.defer_conv) { TL_DONE(); return; }
    // ---- elect the last tail workgroup: it adds the partials in index order (deterministic) and finishes the conv parameters.
    // NO agent-scope fence here: __threadfence() is a write-back + invalidate of this XCD's whole L2, executed by every
    // wave of 96 workgroups underneath the weight-gradient tiles that live out of that L2 (measured: the launch took
    // 28 us).  The partials left as write-through (sc1) stores: once they are acknowledged (vmcnt = 0) they are in
    // memory, the arrival counter is an L2-bypassing atomic, and the last arriver reads the partials with sc1 loads.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) s_last = (__hip_atomic_fetch_add(a.arrive, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == h_n_tail - 1) ? 1 : 0;
    __syncthreads();
    if (!s_last) { TL_DONE(); return; }
    {
        const bool fuse = a.theta != nullptr;
        SmlSched sc; sc.step_size = a.step_size; sc.bc2_sqrt = a.bc2_sqrt;
        // All partial rows in ONE memory round trip: thread (rg, c4) adds the 16-byte column group c4 of the rows
        // t = rg, rg + 21, ... (user-net rows and item-net rows apart), then the 21 row groups meet in LDS and are added
        // in index order -- the summation order is a fixed function of the launch geometry (deterministic).
        constexpr int RG = 21, C4 = SML_CG / 4;          // 21 x 24 = 504 threads
        const int split = h_tiles0 * CS, total = h_tiles_total * CS;
        float* P = smem;                                 // [2][RG][SML_CG]
        if (tid < RG * C4) {
            const int rg = tid / C4, c4 = tid % C4;
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            for (int t0 = rg; t0 < total; t0 += 4 * RG) {
                f32x4 x[4];
                const float* srcp[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int t = t0 + u * RG;
                    // (clamped address, weighted below: the L2-bypassing loads are unconditional)
                    srcp[u] = a.convg_out + (int64_t)(t < total ? t : 0) * SML_CG + c4 * 4;
                }
                peer_load16x4(x, srcp);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int t = t0 + u * RG;
                    if (t < split) acc0 += x[u]; else if (t < total) acc1 += x[u];
                }
            }
            *reinterpret_cast<f32x4*>(P + (0 * RG + rg) * SML_CG + c4 * 4) = acc0;
            *reinterpret_cast<f32x4*>(P + (1 * RG + rg) * SML_CG + c4 * 4) = acc1;
        }
        const int net = tid >> 7, k = tid & 127;        // threads 0..94: user net, 128..222: item net
        const bool mine = tid < 256 && k < 95;
        const int off = k < 30 ? k : k < 40 ? k + 2 : k < 90 ? k + 4 : k + 6;
        const int64_t i = (int64_t)net * NS + off;
        float p = 0.f, m = 0.f, v = 0.f;
        if (mine && fuse) { p = a.theta[i]; m = a.m[i]; v = a.v[i]; }
        __syncthreads();
        if (mine) {
            float g = 0.0f;
#pragma unroll
            for (int rg = 0; rg < RG; ++rg) g += P[(net * RG + rg) * SML_CG + k];
            { float* gp = net ? seg1.grad : seg0.grad; if (gp) gp[off] = g; }
            for (int q = 0; q < a.peer.world; ++q) peer_store(a.peer.dst[q] + i, g);
            if (fuse) {
                adam_apply(p, m, v, adam_wd(g, a.weight_decay, p), sc);
                a.theta[i] = p; a.m[i] = m; a.v[i] = v;
            }
        }
        if (tid == 0) __hip_atomic_store(a.arrive, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
        if (a.peer.world > 0) peer_signal(a.peer);
    }
    TL(7);
    TL_DONE();
}

// ------------------------------------------------------------------------------------
// theta Adam (torch.optim.Adam with weight_decay added to the gradient,
// model/transfer.py:393, 728) + refresh of the MFMA operand images
// ------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void k_theta_pack(const float* __restrict__ theta, float* __restrict__ pk) {
    constexpr int NS = sml_net_size(D);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 2 * NS) return;
    const int net = i / NS, off = i % NS;
    pack_store<D>(pk + (int64_t)net * sml_pk_size(D), off, theta[i]);
}

template <int D, bool PEER>
__global__ __launch_bounds__(256) void k_theta_adam(SmlThetaAdamArgs a) {
    constexpr int NS = sml_net_size(D);
    static_assert(NS % 4 == 0, "a thread's four floats belong to one net");
    // several GPUs, one-shot exchange: wait until every rank's gradient tiles have landed in this rank's inbox
    if constexpr (PEER) { if (!a.peer.waited) peer_wait(a.peer); }
    // four consecutive parameters per thread: 16-byte loads of gradient / theta / m / v, 16-byte stores
    const int i0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (i0 >= 2 * NS) return;
    const int net = i0 / NS, off0 = i0 % NS;
    f32x4 g;
    if constexpr (PEER) {
        // the slots are added IN RANK ORDER on every rank: bit-identical sums, hence bit-identical replicas
        static_assert(SML_MAX_PEERS == 8, "peer_load16x8");
        f32x4 x[8];
        const float* src[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) src[q] = a.peer.slot0 + (q < a.peer.world ? q : 0) * a.peer.slot_stride + i0;   // (unused slots re-read slot 0)
        peer_load16x8(x, src);
        g = x[0];
#pragma unroll
        for (int q = 1; q < SML_MAX_PEERS; ++q) if (q < a.peer.world) g += x[q];
    } else {
        g = *reinterpret_cast<const f32x4*>(a.grad + i0);             // complete (and all-reduced) flat gradient
    }
    if (a.clip_sumsq != nullptr) {
        // torch.nn.utils.clip_grad_norm_(transfer.parameters(), max_norm, 2) -- model/transfer.py:725-727: the whole
        // gradient is scaled by min(1, max_norm / (||g||_2 + 1e-6)) before the optimiser (and its weight decay) sees it
        const float coef = fminf(a.clip_max_norm / (sqrtf(*a.clip_sumsq) + 1e-6f), 1.0f);
        g *= coef;
    }
    f32x4 p = *reinterpret_cast<const f32x4*>(a.theta + i0), m = *reinterpret_cast<const f32x4*>(a.m + i0),
          v = *reinterpret_cast<const f32x4*>(a.v + i0);
    SmlSched s; s.step_size = a.step_size; s.bc2_sqrt = a.bc2_sqrt;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int off = off0 + e;
        if (off < SML_OFF_F1W && !conv_slot_used_host(off)) continue;   // alignment padding of the conv block: untouched
        float pe = p[e], me = m[e], ve = v[e];
        adam_apply(pe, me, ve, adam_wd(g[e], a.weight_decay, pe), s);
        p[e] = pe; m[e] = me; v[e] = ve;
        pack_store<D>(a.pk + (int64_t)net * sml_pk_size(D), off, pe);
    }
    *reinterpret_cast<f32x4*>(a.theta + i0) = p;
    *reinterpret_cast<f32x4*>(a.m + i0) = m;
    *reinterpret_cast<f32x4*>(a.v + i0) = v;
}

// sum of squares of the flat gradient, one workgroup, fixed order (the conv block's alignment padding holds zeros)
__global__ __launch_bounds__(1024) void k_grad_sumsq(const float* __restrict__ g, long long n, float* __restrict__ out) {
    __shared__ float part[16];
    float acc = 0.0f;
    for (long long i = 4ll * threadIdx.x; i < n; i += 4096) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(g + i);
        acc += x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) { float t = 0.0f; for (int w = 0; w < 16; ++w) t += part[w]; *out = t; }
}

// ------------------------------------------------------------------------------------
// lane-map self test: D = A(16 x 32) * W(16 cols x 32)^T through the same operand paths
// ------------------------------------------------------------------------------------
__global__ void k_selftest(const float* __restrict__ A, const float* __restrict__ W, float* __restrict__ pk,
                           float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float As[16 * 36];
    const int lane = threadIdx.x, l15 = lane & 15, g4 = lane >> 4;
    for (int e = lane; e < 16 * 32; e += 64) {
        As[(e / 32) * 36 + (e % 32)] = A[e];
        pk[pk_pos(2, e / 32, e % 32)] = W[e];       // W[col][red]
    }
    __syncthreads();
    f32x4 acc[1][1];
    zero_acc(acc);
    mma16_rows<1, 1, 2, 2>(acc, As + l15 * 36 + 4 * g4, 0, reinterpret_cast<const f32x4*>(pk), 2, 0, lane,
                           [](int) { return 0; });
    for (int q = 0; q < 4; ++q) out[(4 * g4 + q) * 16 + l15] = acc[0][0][q];
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

#ifdef SML_TIMELINE
hipError_t sml_debug_set_timeline(long long* buf) { return hipMemcpyToSymbol(HIP_SYMBOL(g_timeline), &buf, sizeof(buf)); }
#else
hipError_t sml_debug_set_timeline(long long*) { return hipErrorNotSupported; }
#endif
hipError_t sml_launch_fwd(int d, int mt, int ns, const SmlFwdArgs& a, int tiles_total, hipStream_t st, bool side) {
    if (tiles_total <= 0) return hipSuccess;
    if (side && mt == 2 && ns == 1 && (d == 32 || d == 64)) {     // (the evaluation stream's table-sized forward: same code, own name)
        if (d == 32) k_side_transfer_fwd<32><<<dim3(tiles_total), dim3(512), 0, st>>>(a);
        else k_side_transfer_fwd<64><<<dim3(tiles_total), dim3(512), 0, st>>>(a);
        return hipGetLastError();
    }
    // the layout the kernel's preloaded leading parameters stand for (k_transfer_fwd): refuse anything else loudly
    const SmlSeg& s0 = a.seg[0]; const SmlSeg& s1 = a.seg[1];
    const bool two = tiles_total > a.tiles0 && s1.n_rows > 0;
    const int tt = a.tiles_total > a.tiles0 ? a.tiles_total : a.tiles0;          // (table-sized calls leave tiles_total unset: nothing reads it there)
    const int n1 = s0.n_rows == s0.B ? 2 * s0.B : ((2 * s0.B + SML_TM - 1) / SML_TM) * SML_TM;
    if ((s0.tri != nullptr && s0.is_item != 0) || a.cg_split < 0 || a.cg_split > 0x7fff || a.cg_total < 0 || a.cg_total > 0xffff ||
        a.tiles0 < 0 || (tt != a.tiles0 && tt != a.tiles0 + (2 * s0.B + SML_TM - 1) / SML_TM) ||
        (two && (s1.pk != s0.pk + sml_pk_size(d) || s1.theta != s0.theta + sml_net_size(d) || s1.tri != s0.tri || s1.B != s0.B || s1.n_rows != n1 ||
                 (s1.tri != nullptr && s1.is_item != 1))))
        return hipErrorInvalidValue;
#define SML_FWD_HOT_ARGS s0.pk, s0.theta, s0.tri, a.cs_in, a.cg_part, s0.B, s0.n_rows, (int)((unsigned)a.tiles0 | (tt != a.tiles0 ? 0x80000000u : 0u)), \
                         (a.cg_split | (a.cg_total << 15)), a
    if (mt == 1 && ns == 1) { SML_DISPATCH_D(d, k_transfer_fwd<DD, 1, 1><<<dim3(tiles_total), dim3(512), 0, st>>>(SML_FWD_HOT_ARGS)); }
    else if (mt == 1 && ns == 4) { SML_DISPATCH_D(d, k_transfer_fwd<DD, 1, 4><<<dim3(((tiles_total + 1) / 2) * 8), dim3(512), 0, st>>>(SML_FWD_HOT_ARGS)); }
    else if (mt == 1 && ns == 2) { SML_DISPATCH_D(d, k_transfer_fwd<DD, 1, 2><<<dim3(tiles_total * 2), dim3(512), 0, st>>>(SML_FWD_HOT_ARGS)); }
    else if (mt == 2 && ns == 1) {
        // table-sized calls: two hidden passes, two workgroups per CU (SML_FWD_HSEQ=1: the one-pass form)
        static const bool hseq = !(getenv("SML_FWD_HSEQ") && atoi(getenv("SML_FWD_HSEQ")) == 1);
        // (d = 128: the two-pass form still needs 118 KB of LDS -- one workgroup per CU either way -- so it keeps one pass)
        if (hseq && d == 32) k_transfer_fwd<32, 2, 1, 2><<<dim3(tiles_total), dim3(512), 0, st>>>(SML_FWD_HOT_ARGS);
        else if (hseq && d == 64) k_transfer_fwd<64, 2, 1, 2><<<dim3(tiles_total), dim3(512), 0, st>>>(SML_FWD_HOT_ARGS);
        else { SML_DISPATCH_D(d, k_transfer_fwd<DD, 2, 1><<<dim3(tiles_total), dim3(512), 0, st>>>(SML_FWD_HOT_ARGS)); }
    }
    else if (mt == 3 && ns == 1 && d == 32) { k_transfer_fwd<32, 3, 1><<<dim3(tiles_total), dim3(512), 0, st>>>(SML_FWD_HOT_ARGS); }
    else return hipErrorInvalidValue;
#undef SML_FWD_HOT_ARGS
    return hipGetLastError();
}
hipError_t sml_launch_bwd(int d, int split, const SmlBwdArgs& a, int tiles_total, hipStream_t st) {
    if (tiles_total <= 0) return hipSuccess;
    if (split) {      // d/16 workgroups per row tile
        // operand rings preloaded at d = 32 (one memory round trip for the whole kernel); fetched on demand at d >= 64
        const char* fp = d == 64 ? getenv("SML_BWD_PRE") : nullptr;           // (measurement / test override, d = 64 only)
        const int force_pre = fp && *fp ? atoi(fp) : -1;
        const int grid = tiles_total * (d / 16);
        // (d = 64 measured both ways on MI355X, TR batch 256: 39.09 us/batch preloaded vs 39.12 on demand -- no
        // difference, so the form with 124 registers instead of 224 is the default)
        const bool pre = d == 32 || (d == 64 && force_pre > 0);
        const bool tr = a.convg_part != nullptr;
#define SML_BWD_LAUNCH(DD, TRV, PREV) k_transfer_bwd<DD, TRV, PREV><<<dim3(grid), dim3(512), 0, st>>>(a)
        if (d == 32) { if (tr) SML_BWD_LAUNCH(32, true, true); else SML_BWD_LAUNCH(32, false, true); }
        else if (d == 64 && pre) { if (tr) SML_BWD_LAUNCH(64, true, true); else SML_BWD_LAUNCH(64, false, true); }
        else if (d == 64) { if (tr) SML_BWD_LAUNCH(64, true, false); else SML_BWD_LAUNCH(64, false, false); }
        else if (d == 128) { if (tr) SML_BWD_LAUNCH(128, true, false); else SML_BWD_LAUNCH(128, false, false); }
        else return hipErrorInvalidValue;
#undef SML_BWD_LAUNCH
    } else {
        if (a.seg[1].theta != a.seg[0].theta + sml_net_size(d)) return hipErrorInvalidValue;      // (what the kernel's preloaded parameters assume)
#define SML_BWDF_HOT a.dn.hdr, a.dn.ent, a.seg[0].theta, a.out_all, a.tiles0, a.push.world, a.tiles_live, a.B, a
        if (a.convg_part != nullptr) { SML_DISPATCH_D(d, k_transfer_bwd_full<DD, 1, true><<<dim3(tiles_total), dim3(512), 0, st>>>(SML_BWDF_HOT)); }
        else { SML_DISPATCH_D(d, k_transfer_bwd_full<DD, 1, false><<<dim3(tiles_total), dim3(512), 0, st>>>(SML_BWDF_HOT)); }
#undef SML_BWDF_HOT
    }
    return hipGetLastError();
}
int sml_wgrad_grid(int d);
hipError_t sml_launch_tr_bwd_head(int d, const SmlBwdArgs& a, int tiles_total, hipStream_t st) {
    if (tiles_total <= 0) return hipSuccess;
    // (leading scalars = the struct's own fields; the kernel derives each segment's z1 / pk / n_rows from them: slot0 = ioff, pk + net * size)
    SML_DISPATCH_D(d, k_tr_bwd_head<DD><<<dim3(((tiles_total + 1) / 2) * 8), dim3(512), 0, st>>>(a.out_all, a.seg[0].z1, a.seg[0].pk, (long long)a.out_pstride,
                                                                                                   a.B, a.ioff, a.tiles0, a.tiles_total, a.out_np, a));
    return hipGetLastError();
}
int sml_wgrad2_pushers(int d) { return sml_wgrad_grid(d) - 2 + 1; }      // every tile workgroup + the last tail workgroup
hipError_t sml_launch_tr_wgrad2(int d, const SmlWgArgs& a, hipStream_t st) {
    const int tiles = sml_wgrad_grid(d) - 2;
    // the layout the kernel's preloaded leading parameters stand for: refuse anything else loudly
    const SmlWgSeg& s0 = a.seg[0]; const SmlWgSeg& s1 = a.seg[1];
    const int B = s0.n_rows;
    const int64_t slot1 = (int64_t)SML_R * ((B + SML_R - 1) / SML_R);
    if (s1.n_rows != 2 * B || s1.dz1 != s0.dz1 + slot1 * SML_HID || s1.a1 != s0.a1 + slot1 * SML_C2 * d || s1.dout != s0.dout + slot1 * d ||
        s1.a2 != s0.a2 + slot1 * SML_HID || s1.pk_net != s0.pk_net + sml_pk_size(d) || s1.xin != s0.xin + slot1 * 3 * d ||
        a.tiles0 != (B + SML_TM - 1) / SML_TM || a.tiles_total != a.tiles0 + (2 * B + SML_TM - 1) / SML_TM)
        return hipErrorInvalidValue;
    SML_DISPATCH_D(d, k_tr_wgrad2<DD><<<dim3(a.n_tail + tiles), dim3(512), 0, st>>>(s0.dz1, s0.a1, s0.dout, s0.a2, s0.pk_net, s0.xin, B, a.n_tail, a));
    return hipGetLastError();
}
int sml_wgrad_grid(int d) {
    const int tn = 16 * (SML_C2 * d / 32) + (d / 32) * 16;
    const int extra = 2;        // + one conv-parameter workgroup per net (sums the backward's partials; Adam when fused)
    return 2 * tn + extra;
}
hipError_t sml_launch_wgrad(int d, const SmlWgArgs& a, hipStream_t st) {
    SML_DISPATCH_D(d, k_transfer_wgrad<DD><<<dim3(sml_wgrad_grid(d)), dim3(512), 0, st>>>(a));
    return hipGetLastError();
}
hipError_t sml_launch_theta_adam(int d, const SmlThetaAdamArgs& a, hipStream_t st) {
    const int n = 2 * sml_net_size(d) / 4;         // four parameters per thread
    if (a.peer.world > 0) { SML_DISPATCH_D(d, k_theta_adam<DD, true><<<dim3((n + 255) / 256), dim3(256), 0, st>>>(a)); }
    else { SML_DISPATCH_D(d, k_theta_adam<DD, false><<<dim3((n + 255) / 256), dim3(256), 0, st>>>(a)); }
    return hipGetLastError();
}
// compact copy of both nets' conv parameters and their Adam moments: cs[net][3: p, m, v][SML_CG] (k_transfer_fwd's conv step)
__global__ __launch_bounds__(256) void k_conv_state_init(const float* __restrict__ theta, const float* __restrict__ m, const float* __restrict__ v,
                                                         int net_size, float* __restrict__ cs) {
    const int net = threadIdx.x >> 7, k = threadIdx.x & 127;
    if (k >= 95) return;
    const int off = k < 30 ? k : k < 40 ? k + 2 : k < 90 ? k + 4 : k + 6;
    const long long i = (long long)net * net_size + off;
    float* o = cs + net * 3 * SML_CG;
    o[k] = theta[i]; o[SML_CG + k] = m[i]; o[2 * SML_CG + k] = v[i];
}
hipError_t sml_launch_conv_state_init(int d, const float* theta, const float* m, const float* v, float* cs, hipStream_t st) {
    k_conv_state_init<<<dim3(1), dim3(256), 0, st>>>(theta, m, v, sml_net_size(d), cs);
    return hipGetLastError();
}
hipError_t sml_launch_grad_sumsq(const float* grad, int64_t n, float* out, hipStream_t st) {
    k_grad_sumsq<<<dim3(1), dim3(1024), 0, st>>>(grad, (long long)n, out);       // n is a multiple of four (two nets)
    return hipGetLastError();
}
// bf16x3 table-sized forward (d = 32): image size in bytes, the pack launch, the forward launch (one 32-row tile per workgroup)
size_t sml_bx3_bytes(int d) { return d == 32 ? (size_t)2 * sml_bx3_size(32) * sizeof(unsigned short) : 0; }
hipError_t sml_launch_theta_pack_bx3(int d, const float* theta, void* pkx, hipStream_t st) {
    if (d != 32) return hipErrorInvalidValue;
    const int n = 2 * (SML_HID * SML_C2 * 32 + 32 * SML_HID);
    k_theta_pack_bx3<32><<<dim3((n + 255) / 256), dim3(256), 0, st>>>(theta, (unsigned short*)pkx);
    return hipGetLastError();
}
hipError_t sml_launch_fwd_bx3(int d, const SmlFwdArgs& a, const void* pkx_net, int tiles, hipStream_t st, bool side) {
    if (d != 32) return hipErrorInvalidValue;
    if (side) k_transfer_fwd_bx3<32, true><<<dim3(tiles), dim3(512), 0, st>>>(a, (const unsigned short*)pkx_net);
    else k_transfer_fwd_bx3<32, false><<<dim3(tiles), dim3(512), 0, st>>>(a, (const unsigned short*)pkx_net);
    return hipGetLastError();
}
hipError_t sml_launch_mf_fwd_bx3(int d, const SmlFwdArgs& a, const void* pkx, int tiles, hipStream_t st) {
    if (d != 32) return hipErrorInvalidValue;
    if (tiles <= 0) return hipSuccess;           // (several GPUs: a rank whose share of a global batch is empty still takes part in the step)
    // the layout the kernel's preloaded leading parameters stand for: refuse anything else loudly
    const SmlSeg& s0 = a.seg[0]; const SmlSeg& s1 = a.seg[1];
    const bool dn = s0.drec != nullptr;
    const int B = s0.B, t0 = (B + SML_TM - 1) / SML_TM;
    if (a.tiles0 != t0 || s1.theta != s0.theta + sml_net_size(d) || s1.B != B || s0.is_item != 0 || s1.is_item != 1 ||
        (dn ? (s0.hdr == nullptr || s1.hdr != s0.hdr + t0 || s1.drec != s0.drec + (int64_t)SML_R * ((B + SML_R - 1) / SML_R) ||
               s0.n_rows != SML_TM * t0 || s1.n_rows != SML_TM * ((2 * B + SML_TM - 1) / SML_TM))
            : (s0.hdr != nullptr || s1.drec != nullptr || s1.tri != s0.tri || s0.n_rows != B || s1.n_rows != 2 * B)))
        return hipErrorInvalidValue;
    k_mf_fwd_bx3<32><<<dim3(tiles), dim3(512), 0, st>>>((const unsigned short*)pkx, s0.theta, a.sched, dn ? (const void*)s0.drec : (const void*)s0.tri, s0.hdr,
                                                         B, a.cur_step, a.sched_len, a);
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
