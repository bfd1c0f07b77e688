// Index preparation of an epoch, by hand (replaces the library radix sort + select + four helper kernels).
//
// What the step kernels need, per batch b and table (users / items) -- a "list" of N occurrences (row, value):
//   * for every row that occurs more than once: its values (slots) in ascending occurrence order, contiguous in `vals`,
//     and one SmlRun record (row, first position, length, first slots);
//   * for every occurrence: the "row occurs once in this batch" mark (bare step), or one record per sorted position
//     (MF stage);
//   * the batch's list of hot runs (longer than SML_HOT).
// Semantics of the reference path this serves: torch's sparse-gradient embedding update, model/baseline.py:188-201
// (duplicate indices of a batch are summed before the step).
//
// Algorithm.  The batch is implicit in the position, so nothing is sorted across batches:
//   k_prep_hist     tile histograms of the bucket id (row & (nbk-1)) -- the triples are read once for all three columns;
//   k_prep_scan     per list: bucket offsets, the tiles' first positions per bucket, the list of oversized buckets;
//   k_prep_scatter  stable partition: every occurrence goes to its bucket as ONE packed entry (row_hi << vb | value),
//                   in occurrence order (ranks by wavefront ballots, no atomics: the order is a function of the input);
//   k_prep_wave     lists cut into small buckets: one WAVEFRONT per bucket -- duplicate filter (two LDS bitmaps), the few
//                   candidates ranked in registers, the outputs leave directly; fuller buckets go onto a list;
//   k_prep_bucket   one workgroup per bucket (or per listed bucket; or per list, straight from the triples: MF stage): the
//                   filter, then the candidates are sorted by row_hi in LDS (stable radix passes of <= 9 bits), runs are
//                   read off the sorted bucket and the outputs leave directly;
//   k_prep_large    buckets above SML_PREP_SMALL entries (a row with thousands of occurrences): the same passes in LDS
//                   + registers up to 96 KB of entries, beyond that through global memory, chunk by chunk;
// Run records (bare step) go straight into the batch's run list: a bucket counts its records, takes their place with ONE
// returning atomicAdd on the batch's counter (a cache line per batch) and writes them -- round 3's staging array and its
// compaction pass (k_prep_compact: 37-46 us and 160 MB per epoch) are gone; the order of the buckets inside a list is
// arrival order, which nothing depends on.
// No atomics on the path of every occurrence (one per bucket with duplicated rows, one per oversized bucket, one per hot run).
// Several GPUs: the same kernels over other occurrence streams (occ_of, k_prep_hist_x / k_prep_scatter_x).
// HBM traffic per triple: 24 B (triples) x 2 + 12 B written + 12 B read + marks / records of duplicated rows -- against
// 4 radix passes over 8-byte pairs per table before.  Occurrences of one row always meet in one bucket; buckets are cut
// by the LOW row bits so that a popular block of neighbouring ids spreads over all of them.
#include "sml_kernels.h"

namespace {

__device__ __forceinline__ uint64_t lanes_below() { return (1ull << (threadIdx.x & 63)) - 1ull; }

// lanes (among the valid ones) holding the same `key` in its low `bits` bits
__device__ __forceinline__ uint64_t match_any(uint32_t key, bool valid, int bits) {
    const uint64_t v = __ballot(valid);
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    for (int b = 0; b < bits; ++b) {
        const uint32_t bit = (key >> b) & 1u;
        const uint64_t bb = __ballot(bit != 0u);
        const uint32_t flip = bit - 1u;                    // all ones for a clear bit: the lanes whose bit is clear too
        lo &= (uint32_t)bb ^ flip; hi &= (uint32_t)(bb >> 32) ^ flip;
    }
    return ((uint64_t)hi << 32) | lo;
}

// One round of the stable in-wavefront ranking: the wavefront's counter row `cnt` holds, per key, the occurrences of
// earlier rounds; returns this occurrence's rank among the wavefront's equal keys so far.
//
// fast (round 4): ONE returning LDS atomic per occurrence.  The counter row is 16-bit counts, two per dword; a lane adds
// 1 << 16 * (key & 1) to its key's dword and reads its rank out of the OLD value.  That is a STABLE rank only if the lanes of
// one instruction that hit the same dword are served in ascending lane order -- which the hardware does, but no manual says
// so: every context measures it before its first preparation (k_rank_probe: collisions of every multiplicity, 1 to 16
// wavefronts per workgroup on separate rows; tools/lds_atomic_order_probe.hip is the long form: 2.1e9 atomics, none out of
// order) and the kernels take this path only while the probe's violation count is zero -- otherwise the ballot ranking
// below (about 130 instructions per round instead of 6: the partition and the bucket sorts were bound by exactly this).
// AGG (the oversized buckets: a hot row's occurrences are most of such a bucket, and 64 atomics on one address are served one
// after the other): the lanes that share the first lane's key -- and then the first remaining lane's -- take ONE add of their
// number each, the rank being the lane's place among them; what is left goes lane by lane.
template <bool AGG = false>
__device__ __forceinline__ uint32_t wave_rank(unsigned short* cnt, uint32_t key, bool valid, int bits, bool fast) {
    if (fast) {
        uint32_t rank = 0;
        bool todo = valid;
        if constexpr (AGG) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const uint64_t vm = __ballot(todo);
                if (vm == 0ull) break;
                const int first = __ffsll((long long)vm) - 1;
                const uint32_t k0 = (uint32_t)__builtin_amdgcn_readlane((int)key, first);
                const uint64_t m0 = __ballot(todo && key == k0);
                uint32_t base = 0;
                if ((int)(threadIdx.x & 63) == first)
                    base = atomicAdd(reinterpret_cast<uint32_t*>(cnt + (k0 & ~1u)), (uint32_t)__popcll(m0) << (16 * (k0 & 1u)));
                base = (uint32_t)__builtin_amdgcn_readlane((int)base, first);
                if (todo && key == k0) { rank = ((base >> (16 * (k0 & 1u))) & 0xffffu) + (uint32_t)__popcll(m0 & lanes_below()); todo = false; }
            }
        }
        if (todo) {
            const uint32_t old = atomicAdd(reinterpret_cast<uint32_t*>(cnt + (key & ~1u)), 1u << (16 * (key & 1u)));
            rank = (old >> (16 * (key & 1u))) & 0xffffu;
        }
        return rank;
    }
    const int lane = threadIdx.x & 63;
    const uint64_t m = match_any(key, valid, bits);
    const int leader = valid ? (__ffsll((long long)m) - 1) : lane;
    uint32_t prev = 0;
    if (valid && lane == leader) { prev = cnt[key]; cnt[key] = (unsigned short)(prev + __popcll(m)); }
    prev = (uint32_t)__shfl((int)prev, leader, 64);
    return prev + (uint32_t)__popcll(m & lanes_below());
}

struct BatchGeo { int64_t start; int Bb; uint32_t ioff; };
__device__ __forceinline__ BatchGeo batch_geo(const SmlPrepArgs& a, int b) {
    BatchGeo g;
    if (a.boff != nullptr) { g.start = a.boff[b]; g.Bb = a.boff[b + 1] - a.boff[b]; }
    else { g.start = (int64_t)b * a.batch; const int64_t rem = a.n - g.start; g.Bb = (int)(rem < a.batch ? rem : a.batch); }
    g.ioff = a.pad_tiles ? (uint32_t)((g.Bb + SML_R - 1) / SML_R) * SML_R : (uint32_t)g.Bb;
    return g;
}

// XCD-aware block maps.  Workgroups go to the eight XCDs round-robin by their linear index, and each XCD has its own
// L2: the scattered 4-byte entry stores and 1-byte mark stores of one batch only merge into whole lines before they
// leave for HBM if they all pass through ONE L2 (a batch's entry arrays and marks are a few MB).  So the linear index
// is laid out as ((batch group * items + item) * 8 + x) with batch = group * 8 + x: batch b's workgroups all run on XCD
// b % 8, one batch of the group after the other.  Measured (FETCH_SIZE / WRITE_SIZE): see DESIGN.md.
struct XcdMap { int b, item; };
__device__ __forceinline__ XcdMap xcd_map(unsigned id, int items) {
    const int x = (int)(id & 7u);
    const unsigned rest = id >> 3;
    XcdMap m; m.item = (int)(rest % (unsigned)items); m.b = (int)(rest / (unsigned)items) * 8 + x;
    return m;
}
static inline unsigned xcd_grid(int nb, int items) { return (unsigned)(((nb + 7) / 8) * items * 8); }

// ------------------------------------------------------------------------------------
// k_prep_hist: grid (tpb, nb), 1024 threads, SML_PREP_IPT triples per thread.
// Streams of a tile: 0 = users, 1 = positives, 2 = negatives (the item list's order is positives then negatives, so the
// two halves are separate tiles of that list: half * tpb + k).
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_prep_hist(SmlPrepArgs a) {
    __shared__ uint32_t h[3][SML_PREP_MAXBK];
    const int k = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const BatchGeo g = batch_geo(a, b);
    const int nbu = a.t[0].nbk, nbi = a.t[1].nbk;
    for (int i = tid; i < 3 * SML_PREP_MAXBK; i += 1024) (&h[0][0])[i] = 0u;
    if (k == 0 && b == 0 && tid < 4) a.n_medium[tid] = 0;          // n_medium, -, longest run, n_large (one int4 of counters)
    __syncthreads();
    // A histogram does not care which thread counts which occurrence: the tile's 3 * 4,096 indices are read as ONE flat
    // array, consecutive lanes consecutive 8-byte elements (a wavefront's load is 512 contiguous bytes: four lines, not the
    // twelve a column load at a 24-byte stride touches) -- element e of the tile belongs to stream e % 3.
    const int nel = 3 * min(SML_PREP_TT, g.Bb - k * SML_PREP_TT);
    const int64_t* flat = a.tri + (g.start + (int64_t)k * SML_PREP_TT) * 3;
    const int s0 = tid % 3;                                        // (1024 = 1 mod 3: load j of this thread is stream (s0 + j) % 3)
    uint32_t v[3 * SML_PREP_IPT];
#pragma unroll
    for (int j = 0; j < 3 * SML_PREP_IPT; ++j) { const int e = tid + 1024 * j; v[j] = e < nel ? (uint32_t)flat[e] : 0u; }
#pragma unroll
    for (int j = 0; j < 3 * SML_PREP_IPT; ++j) {
        const bool valid = tid + 1024 * j < nel;
        const int sj = (s0 + j) % 3;
        // (a list that is one bucket: one add per wavefront instead of 64 on one address)
        if (nbu == 1) { const uint64_t m = __ballot(valid && sj == 0); if ((tid & 63) == 0 && m) atomicAdd(&h[0][0], (uint32_t)__popcll(m)); }
        else if (valid && sj == 0) atomicAdd(&h[0][v[j] & (nbu - 1)], 1u);
        if (nbi == 1) {
            const uint64_t m1 = __ballot(valid && sj == 1), m2 = __ballot(valid && sj == 2);
            if ((tid & 63) == 0) { if (m1) atomicAdd(&h[1][0], (uint32_t)__popcll(m1)); if (m2) atomicAdd(&h[2][0], (uint32_t)__popcll(m2)); }
        } else if (valid && sj != 0) atomicAdd(&h[sj][v[j] & (nbi - 1)], 1u);
    }
    __syncthreads();
    uint32_t* hu = a.t[0].hist + ((int64_t)b * a.tpb + k) * nbu;
    for (int i = tid; i < nbu; i += 1024) hu[i] = h[0][i];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        uint32_t* hi = a.t[1].hist + (((int64_t)b * 2 + half) * a.tpb + k) * nbi;
        for (int i = tid; i < nbi; i += 1024) hi[i] = h[1 + half][i];
    }
}

// ------------------------------------------------------------------------------------
// Other occurrence sources (mode != 0): the job-wide item lists of the bare step on several GPUs.
//   mode 1  replicated items: users = this rank's; items = every rank's (rank q, element e, column c) from items_all;
//   mode 2  sharded items, list A: users = this rank's; items = the JOB's occurrences of the tail rows this rank owns
//           (row index local to the shard); occurrences of other owners / of head rows are left out;
//   mode 3  sharded items, list B: no users; items = this rank's own occurrences of head rows.
// Streams of a tile: [users] then (q, c) for q in ranks, c in {positive, negative}; the item list's order is stream-major.
// ------------------------------------------------------------------------------------
struct Occ { uint32_t row, val; bool valid; };
__device__ __forceinline__ Occ occ_of(const SmlPrepArgs& a, const BatchGeo& g, int s, int t) {
    Occ o; o.valid = t < g.Bb; o.row = 0u; o.val = 0u;
    if (!o.valid) return o;
    if (a.mode == 4) { o.row = (uint32_t)a.x_keys[g.start + t]; o.val = a.x_vals[g.start + t]; return o; }
    if (a.has_users && s == 0) { o.row = (uint32_t)a.tri[(g.start + t) * 3]; o.val = (uint32_t)t; return o; }
    const int si = s - a.has_users, q = si >> 1, c = si & 1;
    if (a.mode == 3) {
        const int64_t raw = a.tri[(g.start + t) * 3 + 1 + c];
        o.valid = raw < a.head_rows; o.row = (uint32_t)raw; o.val = (uint32_t)((c ? 2 * g.Bb : g.Bb) + t);
        return o;
    }
    const int64_t raw = a.items_all[((int64_t)q * a.n + g.start + t) * 2 + c];
    o.val = (uint32_t)((int64_t)q * a.val_q + (c ? g.Bb : 0) + t);
    if (a.mode == 1) { o.row = (uint32_t)raw; return o; }
    const int64_t r = raw - a.head_rows;
    o.valid = r >= 0 && r / a.shard_rows == a.shard_rank;
    o.row = (uint32_t)(r - (int64_t)a.shard_rank * a.shard_rows);
    return o;
}

__global__ __launch_bounds__(1024) void k_prep_hist_x(SmlPrepArgs a) {
    __shared__ uint32_t h[SML_PREP_MAXBK];
    const int k = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const BatchGeo g = batch_geo(a, b);
    if (k == 0 && b == 0 && tid < 4) a.n_medium[tid] = 0;
    const int t0 = k * SML_PREP_TT + (tid >> 6) * (64 * SML_PREP_IPT) + (tid & 63);
    const int ns = a.has_users + a.nis;
    for (int s = 0; s < ns; ++s) {
        const int T = (a.has_users && s == 0) ? 0 : 1;
        const SmlPrepTable& tb = a.t[T];
        const int nbk = tb.nbk;
        __syncthreads();
        for (int i = tid; i < nbk; i += 1024) h[i] = 0u;
        __syncthreads();
#pragma unroll
        for (int r = 0; r < SML_PREP_IPT; ++r) {
            const Occ o = occ_of(a, g, s, t0 + r * 64);
            if (nbk == 1) { const uint64_t m = __ballot(o.valid); if ((tid & 63) == 0 && m) atomicAdd(&h[0], (uint32_t)__popcll(m)); }
            else if (o.valid) atomicAdd(&h[o.row & (uint32_t)(nbk - 1)], 1u);
        }
        __syncthreads();
        const int tile = T == 0 ? k : (s - a.has_users) * a.tpb + k;
        uint32_t* out = tb.hist + ((int64_t)b * tb.ntile + tile) * nbk;
        for (int i = tid; i < nbk; i += 1024) out[i] = h[i];
    }
}

template <typename E>
__global__ __launch_bounds__(1024) void k_prep_scatter_x(SmlPrepArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned short cnt[16][SML_PREP_MAXBK];
    __shared__ uint32_t tbase[SML_PREP_MAXBK];
    const XcdMap xm = xcd_map(blockIdx.x, a.tpb);
    const int k = xm.item, b = xm.b, tid = threadIdx.x, wv = tid >> 6;
    if (b >= a.nb) return;
    const BatchGeo g = batch_geo(a, b);
    if (k * SML_PREP_TT >= g.Bb) return;
    constexpr int IPT = SML_PREP_IPT;
    const bool fast = a.rank_viol != nullptr && *a.rank_viol == 0;
    const int t0 = k * SML_PREP_TT + wv * (64 * IPT) + (tid & 63);
    const int ns = a.has_users + a.nis;
    for (int s = 0; s < ns; ++s) {
        const int T = (a.has_users && s == 0) ? 0 : 1;
        const SmlPrepTable& tb = a.t[T];
        const int nbk = tb.nbk, lb = tb.lb;
        const int tile = T == 0 ? k : (s - a.has_users) * a.tpb + k;
        const uint32_t* base = tb.hist + ((int64_t)b * tb.ntile + tile) * nbk;
        __syncthreads();                                    // the previous stream's readers of cnt / tbase are done
        // (all 16 counter rows in two 16-byte stores per thread: the kernel is bound by instruction issue, and clearing only the
        // nbk counters in use of every row took 8 trips of index arithmetic)
        reinterpret_cast<uint4*>(&cnt[0][0])[tid] = make_uint4(0u, 0u, 0u, 0u);
        reinterpret_cast<uint4*>(&cnt[0][0])[tid + 1024] = make_uint4(0u, 0u, 0u, 0u);
        for (int i = tid; i < nbk; i += 1024) tbase[i] = base[i];
        __syncthreads();
        Occ o[IPT]; uint32_t wr[IPT];
#pragma unroll
        for (int r = 0; r < IPT; ++r) {
            o[r] = occ_of(a, g, s, t0 + r * 64);
            wr[r] = wave_rank(cnt[wv], o[r].row & (uint32_t)(nbk - 1), o[r].valid, lb, fast);
        }
        __syncthreads();
        for (int i = tid; i < nbk; i += 1024) {
            uint32_t run = 0;
#pragma unroll
            for (int w = 0; w < 16; ++w) { const uint32_t c = cnt[w][i]; cnt[w][i] = (unsigned short)run; run += c; }
        }
        __syncthreads();
        E* ent = reinterpret_cast<E*>(tb.ent);
        // marks (this rank's own occurrences): users start at "once"; the item rows are never updated in place here
        uint8_t* uniq = (a.uniq && T == 0) ? a.uniq + (int64_t)b * a.uniq_stride : nullptr;
#pragma unroll
        for (int r = 0; r < IPT; ++r) {
            const int t = t0 + r * 64;
            if (uniq && t < g.Bb) { uniq[t] = 1; uniq[g.Bb + t] = 0; uniq[2 * g.Bb + t] = 0; }
            if (o[r].valid) {
                const uint32_t bin = o[r].row & (uint32_t)(nbk - 1);
                const uint32_t dest = tbase[bin] + cnt[wv][bin] + wr[r];
                const E hi = (E)(o[r].row >> lb);
                ent[dest] = sizeof(E) == 8 ? (E)(((uint64_t)hi << 32) | (uint64_t)o[r].val) : (E)((hi << tb.vb) | (E)o[r].val);
            }
        }
    }
}

// exclusive scan of one value per thread over a 1024-thread block (wave shuffles + 16 wave totals)
__device__ __forceinline__ uint32_t block_excl_scan_1024(uint32_t v, uint32_t* wsum /*[17]*/) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)inc, off, 64); if (lane >= off) inc += t; }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    uint32_t before = 0;
    for (int j = 0; j < w; ++j) before += wsum[j];
    __syncthreads();
    return before + inc - v;
}

// ------------------------------------------------------------------------------------
// k_prep_scan: grid (nb, 2): one workgroup per list.  hist[k][bin] becomes the position (in the table's occurrence
// array) where tile k's first occurrence of that bucket goes.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_prep_scan(SmlPrepArgs a) {
    __shared__ uint32_t wsum[17];
    __shared__ uint32_t part[1024];                                  // [tile group][bin]
    const int b = blockIdx.x, T = blockIdx.y, tid = threadIdx.x;
    const SmlPrepTable& tb = a.t[T];
    const BatchGeo g = batch_geo(a, b);
    const int nbk = tb.nbk, lb = tb.lb, ntile = tb.ntile;
    const int64_t list_start = (int64_t)tb.lmul * g.start;
    uint32_t* H = tb.hist + (int64_t)b * ntile * nbk;
    if (ntile == 0) {                                                // a table without occurrences (the head list has no users)
        if (tid < nbk) tb.bk[(int64_t)b * nbk + tid] = make_uint2(0u, 0u);
        if (tid == 0 && tb.run_off != nullptr) { tb.run_off[b] = 0; tb.run_cnt[(int64_t)b * SML_PREP_CNT_STRIDE] = 0; }
        return;
    }
    // all 1024 threads: thread (group, bin) owns a contiguous share of the tiles
    const int ngrp = 1024 >> lb, grp = tid >> lb, bin = tid & (nbk - 1);
    const int per = (ntile + ngrp - 1) / ngrp, k0 = min(ntile, grp * per), k1 = min(ntile, k0 + per);
    // (a thread's share of the tiles -- 64 at the bare step's table shapes -- stays in registers: ONE round trip of loads
    // instead of two passes of four dependent trips each; this kernel is 32 workgroups of pure latency)
    constexpr int KEEP = 64;
    const bool keep = per <= KEEP;                                   // (block-uniform)
    uint32_t cnt_k[KEEP];
    uint32_t mine = 0;
    if (keep) {
#pragma unroll
        for (int i = 0; i < KEEP; ++i) cnt_k[i] = k0 + i < k1 ? H[(int64_t)(k0 + i) * nbk + bin] : 0u;
#pragma unroll
        for (int i = 0; i < KEEP; ++i) mine += cnt_k[i];
    } else {
#pragma unroll 16
        for (int k = k0; k < k1; ++k) mine += H[(int64_t)k * nbk + bin];
    }
    part[tid] = mine;
    __syncthreads();
    uint32_t tot = 0, before = 0;
    for (int j = 0; j < ngrp; ++j) { const uint32_t c = part[(j << lb) + bin]; tot += c; before += j < grp ? c : 0u; }
    const uint32_t off = block_excl_scan_1024(tid < nbk ? tot : 0u, wsum);      // (threads of group 0: one per bin, in bin order)
    if (tid < nbk) part[tid] = off;
    __syncthreads();
    {
        uint32_t run = (uint32_t)list_start + part[bin] + before;
        if (keep) {
#pragma unroll
            for (int i = 0; i < KEEP; ++i) { if (k0 + i < k1) H[(int64_t)(k0 + i) * nbk + bin] = run; run += cnt_k[i]; }
        } else {
#pragma unroll 16
            for (int k = k0; k < k1; ++k) { const uint32_t c = H[(int64_t)k * nbk + bin]; H[(int64_t)k * nbk + bin] = run; run += c; }
        }
    }
    if (tid < nbk) {
        tb.bk[(int64_t)b * nbk + tid] = make_uint2(off, tot);
        if (tot > SML_PREP_SMALL) {
            const int slot = atomicAdd(a.n_large, 1);
            if (slot < a.large_cap) { a.large[2 * slot] = ((uint32_t)T << 31) | (uint32_t)b; a.large[2 * slot + 1] = (uint32_t)tid; }
        }
    }
    if (tid == 0) {
        if (tb.run_off != nullptr) { tb.run_off[b] = (int)(list_start >> tb.rshift); tb.run_cnt[(int64_t)b * SML_PREP_CNT_STRIDE] = 0; }
        if (T == 0 && a.hot_count != nullptr) a.hot_count[b] = 0;
    }
}

// ------------------------------------------------------------------------------------
// k_prep_scatter: one workgroup per tile as in k_prep_hist, XCD-aware linear grid.  Occurrence order inside a tile: wavefront-major, then round, then lane --
// i.e. ascending triple index; the wavefronts' counts per bucket are prefixed in wavefront order.
// ------------------------------------------------------------------------------------
// Round 4: the entries do not leave one 4-byte store at a time.  By ablation the 12.6 M scattered dword stores of an epoch were
// 44 of the kernel's 77 us (ranking 3, marks 1): a tile's entries are first put in bucket order in LDS (their position inside
// the tile = the tile-local prefix of their bucket + their rank), then written out by consecutive threads -- a bucket's four to
// eight entries of a tile are neighbours there and leave as one 16- to 32-byte piece of a line.
template <typename E>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_prep_scatter(SmlPrepArgs a) {     // (64 VGPRs: two workgroups per CU)
    __shared__ __attribute__((aligned(16))) unsigned short cnt[16][SML_PREP_MAXBK];
    __shared__ uint32_t tbase[SML_PREP_MAXBK];
    __shared__ uint32_t lbase[SML_PREP_MAXBK];              // first position of the bucket inside the tile's staged entries
    __shared__ E stage[SML_PREP_TT];                        // the tile's entries of one stream in bucket order
    __shared__ unsigned short sbin[SML_PREP_TT];            // ... and their buckets
    __shared__ uint32_t wsum[17];
    const XcdMap xm = xcd_map(blockIdx.x, a.tpb);
    const int k = xm.item, b = xm.b, tid = threadIdx.x, wv = tid >> 6;
    if (b >= a.nb) return;
    const BatchGeo g = batch_geo(a, b);
    if (k * SML_PREP_TT >= g.Bb) return;
    constexpr int IPT = SML_PREP_IPT;
    const bool fast = a.rank_viol != nullptr && *a.rank_viol == 0;
    uint32_t row[3][IPT];
    const int t0 = k * SML_PREP_TT + wv * (64 * IPT) + (tid & 63);
#pragma unroll
    for (int r = 0; r < IPT; ++r) {
        const int t = t0 + r * 64;
        if (t < g.Bb) {
            const int64_t* p = a.tri + (g.start + t) * 3;
            row[0][r] = (uint32_t)p[0]; row[1][r] = (uint32_t)p[1]; row[2][r] = (uint32_t)p[2];
        } else { row[0][r] = row[1][r] = row[2][r] = 0u; }
    }
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const SmlPrepTable& tb = a.t[s ? 1 : 0];
        const int nbk = tb.nbk, lb = tb.lb;
        const uint32_t* base = s == 0 ? tb.hist + ((int64_t)b * a.tpb + k) * nbk
                                      : tb.hist + (((int64_t)b * 2 + (s - 1)) * a.tpb + k) * nbk;
        if (s) __syncthreads();                         // the previous stream's readers of cnt / tbase are done
        if (nbk >= 2) { for (int i = tid; i < 8 * nbk; i += 1024) reinterpret_cast<uint32_t*>(&cnt[(2 * i) >> lb][0])[((2 * i) & (nbk - 1)) >> 1] = 0u; }
        else if (tid < 16) cnt[tid][0] = 0;
        for (int i = tid; i < nbk; i += 1024) tbase[i] = base[i];
        __syncthreads();
        uint32_t wr[IPT];
#pragma unroll
        for (int r = 0; r < IPT; ++r) wr[r] = wave_rank(cnt[wv], row[s][r] & (uint32_t)(nbk - 1), t0 + r * 64 < g.Bb, lb, fast);
        __syncthreads();
        if (nbk >= 2) {
            // thread i < nbk / 2: buckets 2i and 2i + 1 as ONE dword of two 16-bit counts (a tile holds 4,096 entries: no carry
            // between the halves) -- half the threads, half the instructions of a 16-bit walk per bucket
            uint32_t tot2 = 0;
            if (tid < (nbk >> 1)) {
#pragma unroll
                for (int w = 0; w < 16; ++w) {
                    uint32_t* p2 = reinterpret_cast<uint32_t*>(&cnt[w][0]) + tid;
                    const uint32_t c2 = *p2; *p2 = tot2; tot2 += c2;
                }
            }
            const uint32_t lo = tot2 & 0xffffu;
            const uint32_t ex = block_excl_scan_1024(lo + (tot2 >> 16), wsum);       // (two barriers inside)
            if (tid < (nbk >> 1)) { lbase[2 * tid] = ex; lbase[2 * tid + 1] = ex + lo; }
        } else {
            uint32_t tot = 0;                              // (one bucket: thread 0)
            if (tid < nbk) {
#pragma unroll
                for (int w = 0; w < 16; ++w) { const uint32_t c = cnt[w][tid]; cnt[w][tid] = (unsigned short)tot; tot += c; }
            }
            const uint32_t ex = block_excl_scan_1024(tot, wsum);
            if (tid < nbk) lbase[tid] = ex;
        }
        __syncthreads();
        const uint32_t vbase = s == 0 ? 0u : (s == 1 ? g.ioff : g.ioff + (uint32_t)g.Bb);
        uint8_t* uniq = a.uniq ? a.uniq + (int64_t)b * a.uniq_stride + vbase : nullptr;     // every mark starts at "once"
        int n_tile = 0;
#pragma unroll
        for (int r = 0; r < IPT; ++r) {
            const int t = t0 + r * 64;
            if (t < g.Bb) {
                const uint32_t bin = row[s][r] & (uint32_t)(nbk - 1);
                const uint32_t lpos = lbase[bin] + cnt[wv][bin] + wr[r];
                const E hi = (E)(row[s][r] >> lb);
                if (uniq) uniq[t] = 1;
                stage[lpos] = sizeof(E) == 8 ? (E)(((uint64_t)hi << 32) | (uint64_t)(vbase + (uint32_t)t))
                                             : (E)((hi << tb.vb) | (E)(vbase + (uint32_t)t));
                sbin[lpos] = (unsigned short)bin;
            }
        }
        n_tile = min(SML_PREP_TT, g.Bb - k * SML_PREP_TT);        // valid occurrences of this tile (block-uniform)
        __syncthreads();
        E* ent = reinterpret_cast<E*>(tb.ent);
        for (int i = tid; i < n_tile; i += 1024) {
            const uint32_t bin = sbin[i];
            ent[tbase[bin] + ((uint32_t)i - lbase[bin])] = stage[i];
        }
    }
}

// ------------------------------------------------------------------------------------
// What leaves a sorted bucket.  `get(q)` reads sorted entry q of the bucket (LDS or global), S entries, `pos0` = the
// bucket's first position in the table's occurrence array.
// ------------------------------------------------------------------------------------
template <typename E>
__device__ __forceinline__ uint32_t ent_hi(E e, int vb) { return sizeof(E) == 8 ? (uint32_t)((uint64_t)e >> 32) : (uint32_t)(e >> vb); }
template <typename E>
__device__ __forceinline__ uint32_t ent_val(E e, int vb) { return sizeof(E) == 8 ? (uint32_t)e : (uint32_t)(e & (((E)1 << vb) - 1)); }

template <typename E, int NT, typename Get>
__device__ __forceinline__ uint32_t emit_bucket(const SmlPrepArgs& a, const SmlPrepTable& tb, int T, int b, uint32_t bin, uint32_t pos0, int S,
                                                Get get, uint32_t* scratch /*[NT/64]*/) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int vb = tb.vb;
    uint8_t* uniq = a.uniq ? a.uniq + (int64_t)b * a.uniq_stride : nullptr;
    // compact mode (round 4): the bucket first COUNTS its records (a run's last occurrence: next entry differs; duplicated:
    // previous one equal), takes their place in the batch's run list with ONE returning atomicAdd on the batch's counter
    // (a cache line per batch: 16 + 16 counters in one line was what made a shared counter cost 600 us in round 3) and
    // writes them there -- no staging array, no compaction pass.  The ORDER of the buckets inside a batch's list is then
    // whatever order they arrive in: nothing depends on it (every run is one row, summed in its own fixed slot order).
    SmlRun* out = tb.runs;
    const uint32_t list0 = (uint32_t)(tb.lmul * batch_geo(a, b).start);      // first position of the batch's list
    uint32_t done = 0;                                            // compact records of earlier trips (block-uniform)
    if (!a.records) {
        uint32_t mine = 0;
        for (int q = tid; q < S; q += NT) {
            const uint32_t rh = ent_hi<E>(get(q), vb);
            const bool tail = q + 1 >= S || ent_hi<E>(get(q + 1), vb) != rh;
            const bool head = q == 0 || ent_hi<E>(get(q - 1), vb) != rh;
            mine += (tail && (tb.allruns || !head)) ? 1u : 0u;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) mine += (uint32_t)__shfl_xor((int)mine, off, 64);
        if (lane == 0) scratch[wv] = mine;
        __syncthreads();
        uint32_t total = 0;
#pragma unroll
        for (int j = 0; j < NT / 64; ++j) total += scratch[j];
        __syncthreads();
        if (tid == 0) scratch[0] = total ? (uint32_t)atomicAdd(tb.run_cnt + (int64_t)b * SML_PREP_CNT_STRIDE, (int)total) : 0u;
        __syncthreads();
        out = tb.runs + tb.run_off[b] + scratch[0];
        __syncthreads();
    }
    if (a.records && a.dense) {
        // Distinct-row numbering (MF stage, SmlDense; the list is this ONE bucket): the runs are counted, run k -- in row
        // order -- gets scratch row (k % ntiles) * 16 + k / ntiles of its list (tile k % ntiles: neighbouring rows, e.g. a Zipf
        // head with consecutive ids, go to different tiles), every occurrence learns its row's scratch row (slot_info), and
        // the run's record goes there.  No per-position records.
        uint32_t mine = 0;
        for (int q = tid; q < S; q += NT) mine += (q + 1 >= S || ent_hi<E>(get(q + 1), vb) != ent_hi<E>(get(q), vb)) ? 1u : 0u;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) mine += (uint32_t)__shfl_xor((int)mine, off, 64);
        if (lane == 0) scratch[wv] = mine;
        __syncthreads();
        uint32_t nd = 0;
#pragma unroll
        for (int j = 0; j < NT / 64; ++j) nd += scratch[j];
        __syncthreads();
        const uint32_t ntiles = (nd + 15u) >> 4;
        const BatchGeo g = batch_geo(a, b);
        const int64_t sbase = (int64_t)b * a.dense_stride;
        SmlRun* drec = a.dense_rec + sbase + (T ? (int64_t)g.ioff : 0);
        if (tid == 0) a.dense_n[2 * b + T] = (int)nd;
#pragma unroll 1
        for (int q0 = 0; q0 < S; q0 += NT) {
            const int q = q0 + tid;
            const bool in = q < S;
            E e = 0; uint32_t rh = 0;
            bool head = false, tail = false;
            if (in) {
                e = get(q); rh = ent_hi<E>(e, vb);
                head = q == 0 || ent_hi<E>(get(q - 1), vb) != rh;
                tail = q + 1 >= S || ent_hi<E>(get(q + 1), vb) != rh;
                if (a.order_viol != nullptr && q + 1 < S) {
                    const E en = get(q + 1);
                    const uint32_t nh = ent_hi<E>(en, vb);
                    if (nh < rh || (a.vals_ascend && nh == rh && ent_val<E>(en, vb) <= ent_val<E>(e, vb))) {
                        if (atomicAdd(a.order_viol, 1) == 0) { a.order_viol[1] = (a.mode << 28) | (T << 27) | b; a.order_viol[2] = (q << 12) | (S & 0xfff); }
                    }
                }
            }
            const uint32_t val = ent_val<E>(e, vb);
            if (in) tb.vals[pos0 + q] = val;
            const uint64_t wm = __ballot(tail);
            if (lane == 0) scratch[wv] = (uint32_t)__popcll(wm);
            __syncthreads();
            uint32_t before = 0, tot = 0;
#pragma unroll
            for (int j = 0; j < NT / 64; ++j) { const uint32_t c = scratch[j]; before += j < wv ? c : 0u; tot += c; }
            const uint32_t k = done + before + (uint32_t)__popcll(wm & lanes_below());       // runs that END before q = the index of q's run
            done += tot;
            __syncthreads();
            const uint32_t kd = (k % ntiles) * 16u + k / ntiles;
            if (in) a.slot_info[sbase + val] = kd;
            if (tail) {
                int hq = q;
                if (!head) {
                    int back = 1;
                    while (back <= 4 && q - back >= 0 && ent_hi<E>(get(q - back), vb) == rh) ++back;
                    if (back <= 4) hq = q - back + 1;
                    else {
                        int lo = 0, hi2 = q - 4;
                        while (lo < hi2) { const int mid = (lo + hi2) >> 1; if (ent_hi<E>(get(mid), vb) < rh) lo = mid + 1; else hi2 = mid; }
                        hq = lo;
                    }
                }
                const int len = q - hq + 1;
                SmlRun r;
                r.row = (rh << tb.lb) | bin; r.pos = pos0 + (uint32_t)hq; r.len = (uint32_t)len; r.pad = 0;
#pragma unroll
                for (int j = 0; j < SML_RUN_INL; ++j) r.slot[j] = j < len ? ent_val<E>(get(hq + j), vb) : 0u;
                drec[kd] = r;
            }
        }
        return done;
    }
#pragma unroll 1
    for (int q0 = 0; q0 < S; q0 += NT) {                          // block-uniform trip count
        const int q = q0 + tid;
        const bool in = q < S;
        E e = 0; uint32_t rh = 0, prev = 0, next = 0;
        if (in) {
            e = get(q); rh = ent_hi<E>(e, vb);
            prev = q > 0 ? ent_hi<E>(get(q - 1), vb) : ~rh;
            next = q + 1 < S ? ent_hi<E>(get(q + 1), vb) : ~rh;
            // the sorted bucket's invariant (SmlPrepArgs.order_viol): (row, value) ascending -- an unstable rank would break it
            if (a.order_viol != nullptr && q + 1 < S && (next < rh || (a.vals_ascend && next == rh && ent_val<E>(get(q + 1), vb) <= ent_val<E>(e, vb)))) {
                // (the first violation leaves its coordinates: occurrence source, table, batch; position and bucket size)
                if (atomicAdd(a.order_viol, 1) == 0) { a.order_viol[1] = (a.mode << 28) | (T << 27) | b | (a.records ? (1 << 26) : 0); a.order_viol[2] = (q << 12) | (S & 0xfff); }
            }
        }
        const bool head = in && prev != rh, tail = in && next != rh;
        const bool dup = in && !(head && tail);
        const uint32_t val = ent_val<E>(e, vb);
        if (a.records || tb.allruns) {                              // every occurrence is listed (and no mark is this list's)
            if (in) tb.vals[pos0 + q] = val;
        } else if (dup) {
            tb.vals[pos0 + q] = val;
            if (uniq) uniq[val] = 0;
        }
        // the run's record is written by its LAST occurrence, which looks its first one up in the sorted bucket
        // (records mode with slot_info: EVERY occurrence looks it up -- its slot is told where the run's record is)
        const bool tell = a.records && a.slot_info != nullptr && in;
        int hq = q, len = 0;
        if (tail || tell) {
            if (!head) {
                int back = 1;
                while (back <= 4 && q - back >= 0 && ent_hi<E>(get(q - back), vb) == rh) ++back;
                if (back <= 4) hq = q - back + 1;
                else {                                            // lower bound of rh in [0, q - 4]
                    int lo = 0, hi2 = q - 4;
                    while (lo < hi2) { const int mid = (lo + hi2) >> 1; if (ent_hi<E>(get(mid), vb) < rh) lo = mid + 1; else hi2 = mid; }
                    hq = lo;
                }
            }
            if (tail) len = q - hq + 1;
        }
        if (tell) a.slot_info[(int64_t)b * a.slot_stride + val] = (head && tail) ? SML_SLOT_ONCE : (pos0 + (uint32_t)hq - list0);
        const bool want = (a.records || tb.allruns) ? tail : (tail && len >= 2);
        uint32_t idx;
        if (!a.records) {
            const uint64_t wm = __ballot(want);
            if (lane == 0) scratch[wv] = (uint32_t)__popcll(wm);
            __syncthreads();
            uint32_t before = 0, tot = 0;
#pragma unroll
            for (int j = 0; j < NT / 64; ++j) { const uint32_t c = scratch[j]; before += j < wv ? c : 0u; tot += c; }
            idx = done + before + (uint32_t)__popcll(wm & lanes_below());
            done += tot;
            __syncthreads();
        } else {
            idx = pos0 + (uint32_t)hq;
            if (in && !head) {                                     // a position inside a run: an empty record
                SmlRun z; z.row = (rh << tb.lb) | bin; z.pos = pos0 + (uint32_t)q; z.len = 0; z.pad = 0;
#pragma unroll
                for (int j = 0; j < SML_RUN_INL; ++j) z.slot[j] = 0u;
                out[pos0 + q] = z;
            }
        }
        if (want) {
            SmlRun r;
            r.row = (rh << tb.lb) | bin; r.pos = pos0 + (uint32_t)hq; r.len = (uint32_t)len; r.pad = 0;
#pragma unroll
            for (int j = 0; j < SML_RUN_INL; ++j) r.slot[j] = j < len ? ent_val<E>(get(hq + j), vb) : 0u;
            out[idx] = r;
            if (!a.records && len > SML_HOT) {
                atomicMax(a.max_len, len);
                if (a.hot_list != nullptr) {
                    const int slot = atomicAdd(a.hot_count + b, 1);
                    if (slot < a.hot_cap) {
                        uint32_t* h = a.hot_list + ((int64_t)b * a.hot_cap + slot) * 3;
                        h[0] = r.pos | ((uint32_t)T << 31); h[1] = (uint32_t)len; h[2] = r.row;
                    }
                }
            }
        }
    }
    return done;
}

// ------------------------------------------------------------------------------------
// k_prep_wave: lists cut into SMALL buckets (a table in wave mode: about 256 occurrences per bucket -- uniform users at
// a 262,144 batch): one WAVEFRONT per bucket, four per workgroup, no barrier anywhere.  The duplicate filter (bitmaps
// over a 13-bit hash of row_hi), the candidates compacted in occurrence order, and -- when at most 64 remain, which is
// the rule where duplicates are rare -- ranked against each other in registers; the records leave from there.  A bucket
// with more candidates is left to k_prep_bucket (listed in `medium`).
// ------------------------------------------------------------------------------------
__device__ __forceinline__ void prep_punt(const SmlPrepArgs& a, int T, int b, uint32_t bin) {
    const int slot = atomicAdd(a.n_medium, 1);
    a.medium[2 * slot] = ((uint32_t)T << 31) | (uint32_t)b; a.medium[2 * slot + 1] = bin;
}
template <typename E>
__global__ __launch_bounds__(256) void k_prep_wave(SmlPrepArgs a, int T) {
    __shared__ uint32_t bm_all[4][2][256];
    __shared__ E cand_all[4][64];
    __shared__ E sorted_all[4][64];
    __shared__ int wcnt[4];
    __shared__ int wbase;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const SmlPrepTable& tb = a.t[T];
    const int quads = (tb.nbk + 3) >> 2;                             // four buckets of one list per workgroup
    const XcdMap xm = xcd_map(blockIdx.x, quads);
    const int b = xm.b;
    const uint32_t bin = (uint32_t)(xm.item * 4 + wv);
    if (b >= a.nb) return;                                           // (the whole workgroup: b is block-uniform)
    // every wavefront handles its bucket without a barrier; what it has to emit (`want`, the record, the count) meets the other
    // three at ONE pair of barriers at the end, where the workgroup takes the four buckets' place in the batch's run list with
    // one returning atomicAdd (one per bucket: the wave kernel went from 15 to 38 us -- 16,384 buckets queue on 16 counters)
    bool want = false;
    uint64_t wm = 0ull;
    SmlRun rec;
    rec.row = rec.pos = rec.len = rec.pad = 0u;
#pragma unroll
    for (int jj = 0; jj < SML_RUN_INL; ++jj) rec.slot[jj] = 0u;
    do {
        if (bin >= (uint32_t)tb.nbk) break;
        const uint2 oc = tb.bk[(int64_t)b * tb.nbk + bin];
        const int S = (int)oc.y;
        if (S == 0) break;
        if (S > SML_PREP_SMALL) break;                               // k_prep_large's (listed by k_prep_scan)
        if (S > 512) { if (lane == 0) prep_punt(a, T, b, bin); break; }
        const BatchGeo g = batch_geo(a, b);
        const uint32_t pos0 = (uint32_t)(tb.lmul * g.start) + oc.x;
        const E* src = reinterpret_cast<const E*>(tb.ent) + pos0;
        uint32_t (*bm)[256] = bm_all[wv];
        E* cand = cand_all[wv];
        E* sorted = sorted_all[wv];
        const int R0 = (S + 63) >> 6;
        const int vb = tb.vb;
        E e0[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) e0[r] = (r < R0 && r * 64 + lane < S) ? src[r * 64 + lane] : (E)0;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) (&bm[0][0])[jj * 64 + lane] = 0u;
        const uint32_t hmask = (1u << min(tb.hb, 13)) - 1u;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (r < R0 && r * 64 + lane < S) {
                const uint32_t h = ent_hi<E>(e0[r], vb) & hmask, bit = 1u << (h & 31);
                const uint32_t old = atomicOr(&bm[0][h >> 5], bit);
                if (old & bit) atomicOr(&bm[1][h >> 5], bit);
            }
        }
        uint32_t nc = 0;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (r < R0) {
                const uint32_t h = ent_hi<E>(e0[r], vb) & hmask;
                const bool isd = r * 64 + lane < S && ((bm[1][h >> 5] >> (h & 31)) & 1u);
                const uint64_t m = __ballot(isd);
                const uint32_t idx = nc + (uint32_t)__popcll(m & lanes_below());
                if (isd && idx < 64) cand[idx] = e0[r];
                nc += (uint32_t)__popcll(m);
            }
        }
        if (nc == 0) break;
        if (nc > 64) { if (lane == 0) prep_punt(a, T, b, bin); break; }
        const int n = (int)nc;
        {   // rank among the candidates: by row_hi, equal ones by position (stable)
            const E e = lane < n ? cand[lane] : (E)0;
            const uint32_t key = ent_hi<E>(e, vb);
            uint32_t rank = 0;
            for (int jj = 0; jj < n; ++jj) {
                const uint32_t kj = (uint32_t)__builtin_amdgcn_readlane((int)key, jj);
                rank += (kj < key || (kj == key && jj < lane)) ? 1u : 0u;
            }
            if (lane < n) sorted[rank] = e;
        }
        const int q = lane;
        const bool in = q < n;
        const E e = in ? sorted[q] : (E)0;
        const uint32_t rh = ent_hi<E>(e, vb);
        const uint32_t prev = (in && q > 0) ? ent_hi<E>(sorted[q - 1], vb) : ~rh;
        const uint32_t next = (in && q + 1 < n) ? ent_hi<E>(sorted[q + 1], vb) : ~rh;
        const bool head = in && prev != rh, tail = in && next != rh, dup = in && !(head && tail);
        const uint32_t val = ent_val<E>(e, vb);
        if (dup) {
            tb.vals[pos0 + q] = val;
            if (a.uniq) a.uniq[(int64_t)b * a.uniq_stride + val] = 0;
        }
        const uint64_t heads = __ballot(head);
        const uint64_t upto = heads & ((lanes_below() << 1) | 1ull);           // heads at or below this lane
        const int hq = 63 - __clzll((long long)(upto | 1ull));
        const int len = q - hq + 1;
        want = tail && len >= 2;
        wm = __ballot(want);
        if (want) {
            rec.row = (rh << tb.lb) | bin; rec.pos = pos0 + (uint32_t)hq; rec.len = (uint32_t)len;
#pragma unroll
            for (int jj = 0; jj < SML_RUN_INL; ++jj) rec.slot[jj] = jj < len ? ent_val<E>(sorted[hq + jj], vb) : 0u;
        }
    } while (false);
    if (lane == 0) wcnt[wv] = (int)__popcll(wm);
    __syncthreads();
    if (tid == 0) {
        const int tot = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
        wbase = tot ? atomicAdd(tb.run_cnt + (int64_t)b * SML_PREP_CNT_STRIDE, tot) : 0;
    }
    __syncthreads();
    if (want) {
        int before = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) before += w < wv ? wcnt[w] : 0;
        tb.runs[tb.run_off[b] + wbase + before + (int)__popcll(wm & lanes_below())] = rec;
    }
}

// ------------------------------------------------------------------------------------
// k_prep_count (round 4): lists whose buckets leave FEW row bits (hb <= 11: a 1M-row item table cut into 512 buckets) -- one
// WAVEFRONT per bucket, four per workgroup, and no sort at all.  Every possible row_hi of the bucket has a counter in
// LDS.  Pass 1: each occurrence takes ONE returning LDS atomic on its row's counter: the old value is its rank inside
// the row's run (lanes of an instruction are served in lane order -- the measured property the `fast` ranking rests on --
// and a lane's rounds are issued in occurrence order: the rank is the stable one).  Pass 2: the counters of duplicated rows
// (count >= 2) are scanned -- lane-major, four neighbouring counters per lane and trip: the order of the RUNS inside a bucket
// is free, only the order inside a run is not -- and every counter becomes (count | base << 16).  Pass 3: a duplicated
// occurrence knows its place, base + rank: its slot goes there (LDS stage, then one coalesced store of the bucket's
// values), its unique mark is cleared, and the occurrence of rank count - 1 writes the run's record.  That replaces the
// duplicate filter's two bitmaps + compaction + two radix passes + head / tail search of k_prep_bucket (70 us per epoch
// for the items of a 10M x 1M job) by about 6 LDS operations per occurrence.  Buckets it cannot take (more than
// SML_PREP_CCAP entries or SML_PREP_CDUP duplicated ones; a device whose LDS atomics failed the order probe) go onto
// the `medium` list: k_prep_bucket's.
// ------------------------------------------------------------------------------------
// The three passes of one bucket with a COMPILE-TIME round count RR >= ceil(S / 64): straight-line code.  A lane beyond the
// bucket's end loads a clamped address and counts on a spare counter (cleared again before pass 3) instead of being masked
// off: with `if (valid)` around every round the compiler built an exec-mask block with its own vmcnt(0) per round -- one store
// round trip per round -- and spilled two dozen masks.  Returns the number of runs (records), 0 if none / punted.
template <typename E, int RR>
__device__ __forceinline__ int count_bucket(const SmlPrepArgs& a, const SmlPrepTable& tb, int T, int b, uint32_t bin, const E* src, int S,
                                            uint32_t pos0, uint32_t* cw, uint32_t* sval, unsigned short* rlist) {
    constexpr int QT = SML_PREP_CROWS / 256;                         // counter quads per lane
    const int lane = threadIdx.x & 63;
    const int vb = tb.vb;
    const int lim = S - lane;                                        // round r of this lane is inside the bucket iff r * 64 < lim
    E e[RR];
#pragma unroll
    for (int r = 0; r < RR; ++r) e[r] = src[min(r * 64 + lane, S - 1)];
    const int nd = 1 << tb.hb;
    const int nq = (nd + 3) >> 2;                                    // counter quads in use
    const uint32_t spare = (uint32_t)(4 * nq);
#pragma unroll
    for (int k = 0; k < QT; ++k) {
        const int q = k * 64 + lane;
        if (q < nq) *reinterpret_cast<uint4*>(cw + 4 * q) = make_uint4(0u, 0u, 0u, 0u);
    }
    if (lane == 0) cw[spare] = 0u;
    __builtin_amdgcn_wave_barrier();
    // pass 1: rank inside the row's run
    uint32_t rk[RR];
#pragma unroll
    for (int r = 0; r < RR; ++r) rk[r] = atomicAdd(&cw[r * 64 < lim ? ent_hi<E>(e[r], vb) : spare], 1u);
    __builtin_amdgcn_wave_barrier();
    // pass 2: the duplicated rows' runs -- their bases (exclusive scan of the counts, lane-major) and their list
    uint4 c4[QT];
    uint32_t mine = 0;                                               // occurrences | runs << 16 of this lane's counters
#pragma unroll
    for (int k = 0; k < QT; ++k) {
        const int q = k * 64 + lane;
        c4[k] = q < nq ? *reinterpret_cast<const uint4*>(cw + 4 * q) : make_uint4(0u, 0u, 0u, 0u);
        const uint32_t c[4] = {c4[k].x, c4[k].y, c4[k].z, c4[k].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) mine += c[j] >= 2u ? (c[j] | 0x10000u) : 0u;
    }
    uint32_t inc = mine;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)inc, off, 64); if (lane >= off) inc += t; }
    const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
    const int ndup = (int)(tot & 0xffffu);
    if (ndup == 0) return 0;
    if (ndup > SML_PREP_CDUP) { if (lane == 0) prep_punt(a, T, b, bin); return 0; }
    uint32_t run = (inc - mine) & 0xffffu, ri = (inc - mine) >> 16;
#pragma unroll
    for (int k = 0; k < QT; ++k) {
        const int q = k * 64 + lane;
        uint32_t c[4] = {c4[k].x, c4[k].y, c4[k].z, c4[k].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool dupl = c[j] >= 2u;
            if (dupl) { rlist[ri] = (unsigned short)(4 * q + j); ++ri; }
            const uint32_t n = dupl ? c[j] : 0u;
            c[j] |= run << 16; run += n;
        }
        if (q < nq) *reinterpret_cast<uint4*>(cw + 4 * q) = make_uint4(c[0], c[1], c[2], c[3]);
    }
    if (lane == 0) cw[spare] = 0u;                                   // (lanes beyond the end are no run)
    __builtin_amdgcn_wave_barrier();
    // pass 3: duplicated occurrences to their places (LDS), then their unique marks (global stores, back to back)
    uint32_t dmask = 0;
#pragma unroll
    for (int r = 0; r < RR; ++r) {
        const uint32_t w = cw[r * 64 < lim ? ent_hi<E>(e[r], vb) : spare];
        if ((w & 0xffffu) >= 2u) {
            sval[(w >> 16) + rk[r]] = ent_val<E>(e[r], vb);
            dmask |= 1u << r;
        }
    }
    uint8_t* uniq = a.uniq ? a.uniq + (int64_t)b * a.uniq_stride : nullptr;
    if (uniq) {
#pragma unroll
        for (int r = 0; r < RR; ++r) if ((dmask >> r) & 1u) uniq[ent_val<E>(e[r], vb)] = 0;
    }
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < ndup; i += 64) tb.vals[pos0 + i] = sval[i];
    return (int)(tot >> 16);
}

template <typename E>
__global__ __launch_bounds__(256) void k_prep_count(SmlPrepArgs a, int T) {
    __shared__ __attribute__((aligned(16))) uint32_t cw_all[4][SML_PREP_CROWS + 4];        // (+ the spare counter of lanes beyond the bucket's end)
    __shared__ uint32_t sval_all[4][SML_PREP_CDUP];
    __shared__ unsigned short rlist_all[4][SML_PREP_CDUP / 2];       // the bucket's duplicated rows (row_hi), in run order
    __shared__ int wcnt[4];
    __shared__ int wbase;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const SmlPrepTable& tb = a.t[T];
    const int quads = (tb.nbk + 3) >> 2;                             // four buckets of one list per workgroup
    const XcdMap xm = xcd_map(blockIdx.x, quads);
    const int b = xm.b;
    const uint32_t bin = (uint32_t)(xm.item * 4 + wv);
    if (b >= a.nb) return;                                           // (the whole workgroup: b is block-uniform)
    uint32_t* cw = cw_all[wv];
    uint32_t* sval = sval_all[wv];
    unsigned short* rlist = rlist_all[wv];
    uint32_t pos0 = 0;
    int nrec = 0;
    do {
        if (bin >= (uint32_t)tb.nbk) break;
        const uint2 oc = tb.bk[(int64_t)b * tb.nbk + bin];
        const int S = (int)oc.y;
        if (S == 0) break;
        if (S > SML_PREP_SMALL) break;                               // k_prep_large's (listed by k_prep_scan)
        const bool fast = a.rank_viol != nullptr && *a.rank_viol == 0;
        if (S > SML_PREP_CCAP || !fast) { if (lane == 0) prep_punt(a, T, b, bin); break; }
        const BatchGeo g = batch_geo(a, b);
        pos0 = (uint32_t)(tb.lmul * g.start) + oc.x;
        const E* src = reinterpret_cast<const E*>(tb.ent) + pos0;
        // (round counts in steps: a bucket of a 1M-row table's 512 holds 1,024 +- 32 entries -- 16 or 17 rounds)
        if (S <= 4 * 64) nrec = count_bucket<E, 4>(a, tb, T, b, bin, src, S, pos0, cw, sval, rlist);
        else if (S <= 12 * 64) nrec = count_bucket<E, 12>(a, tb, T, b, bin, src, S, pos0, cw, sval, rlist);
        else if (S <= 16 * 64) nrec = count_bucket<E, 16>(a, tb, T, b, bin, src, S, pos0, cw, sval, rlist);
        else if (S <= 18 * 64) nrec = count_bucket<E, 18>(a, tb, T, b, bin, src, S, pos0, cw, sval, rlist);
        else nrec = count_bucket<E, SML_PREP_CCAP / 64>(a, tb, T, b, bin, src, S, pos0, cw, sval, rlist);
    } while (false);
    // the four buckets take their place in the batch's run list with ONE returning atomicAdd
    if (lane == 0) wcnt[wv] = nrec;
    __syncthreads();
    if (tid == 0) {
        const int tot = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
        wbase = tot ? atomicAdd(tb.run_cnt + (int64_t)b * SML_PREP_CNT_STRIDE, tot) : 0;
    }
    __syncthreads();
    if (nrec == 0) return;
    int before = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) before += w < wv ? wcnt[w] : 0;
    SmlRun* out = tb.runs + tb.run_off[b] + wbase + before;
    for (int i = lane; i < nrec; i += 64) {                          // one record per run: neighbouring lanes, neighbouring records
        const uint32_t h = rlist[i], w = cw[h], len = w & 0xffffu, bs = w >> 16;
        SmlRun rec;
        rec.row = (h << tb.lb) | bin; rec.pos = pos0 + bs; rec.len = len; rec.pad = 0;
#pragma unroll
        for (int j = 0; j < SML_RUN_INL; ++j) rec.slot[j] = (uint32_t)j < len ? sval[bs + j] : 0u;
        out[i] = rec;
        if (len > SML_HOT) {
            atomicMax(a.max_len, (int)len);
            if (a.hot_list != nullptr) {
                const int slot = atomicAdd(a.hot_count + b, 1);
                if (slot < a.hot_cap) {
                    uint32_t* hl = a.hot_list + ((int64_t)b * a.hot_cap + slot) * 3;
                    hl[0] = rec.pos | ((uint32_t)T << 31); hl[1] = len; hl[2] = rec.row;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------
// bucket_body / k_prep_bucket: 256 threads per bucket; buckets of at most SML_PREP_SMALL entries.
// ------------------------------------------------------------------------------------
template <typename E, bool DIRECT = false>
__device__ __forceinline__ void bucket_body(const SmlPrepArgs& a, int T, int b, uint32_t bin, E (*buf)[SML_PREP_SMALL],
                                            unsigned short (*cnt)[512], uint32_t* dbase, uint32_t* scratch) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const SmlPrepTable& tb = a.t[T];
    const BatchGeo g = batch_geo(a, b);
    const bool fast = a.rank_viol != nullptr && *a.rank_viol == 0;
    // DIRECT (records mode, every list one bucket of at most SML_PREP_SMALL occurrences -- the MF stage's batches): the
    // list is read straight from the triples, there is no partition
    const uint2 oc = DIRECT ? make_uint2(0u, (uint32_t)(T ? 2 * g.Bb : g.Bb)) : tb.bk[(int64_t)b * tb.nbk + bin];   // (first position inside the list, entries)
    int S = (int)oc.y;
    if (S == 0 || S > SML_PREP_SMALL) return;
    const uint32_t pos0 = (uint32_t)(tb.lmul * g.start) + oc.x;
    const E* src = reinterpret_cast<const E*>(tb.ent) + pos0;
    if (DIRECT) {
        for (int i = tid; i < S; i += 256) {
            const int64_t t = T == 0 ? i : (i < g.Bb ? i : i - g.Bb);
            const uint32_t row = (uint32_t)a.tri[(g.start + t) * 3 + (T == 0 ? 0 : (i < g.Bb ? 1 : 2))];
            const uint32_t val = T == 0 ? (uint32_t)i : g.ioff + (uint32_t)i;
            buf[0][i] = sizeof(E) == 8 ? (E)(((uint64_t)row << 32) | val) : (E)(((E)row << tb.vb) | (E)val);
        }
    } else if (a.records || tb.allruns) {
        for (int i = tid; i < S; i += 256) buf[0][i] = src[i];
    } else {
        // Duplicate filter: only occurrences of rows that occur at least twice need sorting (with uniform users that is
        // 3 % of them).  Two bitmaps over row_hi (hashed to 15 bits when it is wider: false positives only cost sorting
        // work): "seen" and "seen again"; the candidates are then compacted in occurrence order.
        uint32_t (*bm)[1024] = reinterpret_cast<uint32_t (*)[1024]>(&buf[1][0]);     // (the second buffer is idle until the sort)
        const int R0 = (S + 255) >> 8;
        const uint32_t hmask = (1u << min(tb.hb, 15)) - 1u;
        E e0[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int i = wv * (R0 * 64) + r * 64 + lane;
            e0[r] = (r < R0 && i < S) ? src[i] : (E)0;
        }
        for (int i = tid; i < 2048; i += 256) (&bm[0][0])[i] = 0u;
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int i = wv * (R0 * 64) + r * 64 + lane;
            if (r < R0 && i < S) {
                const uint32_t h = ent_hi<E>(e0[r], tb.vb) & hmask, bit = 1u << (h & 31);
                const uint32_t old = atomicOr(&bm[0][h >> 5], bit);
                if (old & bit) atomicOr(&bm[1][h >> 5], bit);
            }
        }
        __syncthreads();
        uint32_t wtot = 0, lidx[8];
        bool isd[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            isd[r] = false; lidx[r] = 0;
            if (r < R0) {
                const int i = wv * (R0 * 64) + r * 64 + lane;
                const uint32_t h = ent_hi<E>(e0[r], tb.vb) & hmask;
                isd[r] = i < S && ((bm[1][h >> 5] >> (h & 31)) & 1u);
                const uint64_t m = __ballot(isd[r]);
                lidx[r] = wtot + (uint32_t)__popcll(m & lanes_below());
                wtot += (uint32_t)__popcll(m);
            }
        }
        if (lane == 0) scratch[wv] = wtot;
        __syncthreads();
        uint32_t before = 0, total = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) { const uint32_t c = scratch[j]; before += j < wv ? c : 0u; total += c; }
#pragma unroll
        for (int r = 0; r < 8; ++r) if (isd[r]) buf[0][before + lidx[r]] = e0[r];
        S = (int)total;
        if (S == 0) return;
    }
    int cur = 0;
    const int R = (S + 255) >> 8;                     // rounds: every wavefront owns a contiguous stripe of R * 64 entries
    if (S <= 64 && tb.npass > 0) {
        // a handful of entries: the first wavefront ranks them against each other (stable: equal keys by position)
        __syncthreads();
        if (wv == 0) {
            const E e = lane < S ? buf[0][lane] : (E)0;
            const uint32_t key = ent_hi<E>(e, tb.vb);
            uint32_t rank = 0;
            for (int j = 0; j < S; ++j) {
                const uint32_t kj = (uint32_t)__builtin_amdgcn_readlane((int)key, j);
                rank += (kj < key || (kj == key && j < lane)) ? 1u : 0u;
            }
            if (lane < S) buf[1][rank] = e;
        }
        cur = 1;
    } else
    for (int p = 0; p < tb.npass; ++p) {
        const int lo = p * tb.pbits, bits = min(tb.pbits, tb.hb - lo), nd = 1 << bits;
        // every wavefront clears ITS counter row: a slower wavefront may still be reading its own row for the previous
        // pass's scatter (the rows of others are only touched between the two barriers below)
        for (int i = lane; i < nd; i += 64) cnt[wv][i] = 0;
        __syncthreads();                                 // (also: the bucket is loaded / the previous pass has landed)
        E ev[8]; uint32_t wr[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (r < R) {
                const int i = wv * (R * 64) + r * 64 + lane;
                const bool valid = i < S;
                ev[r] = valid ? buf[cur][i] : (E)0;
                const uint32_t dg = (ent_hi<E>(ev[r], tb.vb) >> lo) & (uint32_t)(nd - 1);
                wr[r] = wave_rank(cnt[wv], dg, valid, bits, fast);
            }
        }
        __syncthreads();
        // per digit: the wavefronts' counts become their prefix; the digit totals are scanned over the block
        {
            uint32_t tot2[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int dgt = 2 * tid + j;
                uint32_t run = 0;
                if (dgt < nd) {
#pragma unroll
                    for (int w = 0; w < 4; ++w) { const uint32_t c = cnt[w][dgt]; cnt[w][dgt] = (unsigned short)run; run += c; }
                }
                tot2[j] = run;
            }
            uint32_t inc = tot2[0] + tot2[1];
            const uint32_t mine = inc;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)inc, off, 64); if (lane >= off) inc += t; }
            if (lane == 63) scratch[wv] = inc;
            __syncthreads();
            uint32_t before = 0;
            for (int j = 0; j < wv; ++j) before += scratch[j];
            const uint32_t ex = before + inc - mine;
            if (2 * tid < nd) dbase[2 * tid] = ex;
            if (2 * tid + 1 < nd) dbase[2 * tid + 1] = ex + tot2[0];
            __syncthreads();
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (r < R) {
                const int i = wv * (R * 64) + r * 64 + lane;
                if (i < S) {
                    const uint32_t dg = (ent_hi<E>(ev[r], tb.vb) >> lo) & (uint32_t)(nd - 1);
                    buf[cur ^ 1][dbase[dg] + cnt[wv][dg] + wr[r]] = ev[r];
                }
            }
        }
        cur ^= 1;
    }
    __syncthreads();
    const E* sorted = buf[cur];
    emit_bucket<E, 256>(a, tb, T, b, bin, pos0, S, [&](int q) { return sorted[q]; }, scratch);
}

// listed == 0: one workgroup per bucket of table T (grid nb * nbk).  listed == 1: the buckets k_prep_wave left
// (grid-stride over `medium`).  listed == 2: one workgroup per list, read straight from the triples.
template <typename E>
__global__ __launch_bounds__(256, sizeof(E) == 4 ? 7 : 4) void k_prep_bucket(SmlPrepArgs a, int T, int listed) {
    __shared__ E buf[2][SML_PREP_SMALL];
    __shared__ __attribute__((aligned(16))) unsigned short cnt[4][512];
    __shared__ uint32_t dbase[512];
    __shared__ uint32_t scratch[8];
    if (listed == 2) {                                           // straight from the triples: workgroup = (batch, table)
        bucket_body<E, true>(a, (int)(blockIdx.x & 1u), (int)(blockIdx.x >> 1), 0u, buf, cnt, dbase, scratch);
        return;
    }
    if (!listed) {
        const XcdMap xm = xcd_map(blockIdx.x, a.t[T].nbk);
        if (xm.b < a.nb) bucket_body<E>(a, T, xm.b, (uint32_t)xm.item, buf, cnt, dbase, scratch);
        return;
    }
    const int n_medium = *a.n_medium;
    for (int w = blockIdx.x; w < n_medium; w += gridDim.x) {
        const uint32_t tl = a.medium[2 * w];
        __syncthreads();                                     // the previous bucket's readers of the LDS arrays are done
        bucket_body<E>(a, (int)(tl >> 31), (int)(tl & 0x7fffffffu), a.medium[2 * w + 1], buf, cnt, dbase, scratch);
    }
}

// ------------------------------------------------------------------------------------
// k_prep_large: the oversized buckets, one 1024-thread workgroup each (grid-stride over the list k_prep_scan built).
// The same stable passes, 4096 entries at a time, between the two entry arrays.
// ------------------------------------------------------------------------------------
// buckets of up to SML_PREP_LDSCAP bytes of entries stay in LDS between the passes (the wavefronts hold their stripes in
// registers while the buffer is rewritten); larger ones go chunk-wise between the two global entry arrays
#define SML_PREP_LDSCAP 98304
template <typename E>
__global__ __launch_bounds__(1024) void k_prep_large(SmlPrepArgs a) {
    constexpr int CAP = SML_PREP_LDSCAP / (int)sizeof(E);            // 24,576 four-byte entries: 24 per thread
    constexpr int RMAX = CAP / 1024;
    __shared__ E lbuf[CAP];
    __shared__ __attribute__((aligned(16))) unsigned short cnt[16][512];
    __shared__ uint32_t dbase[512];
    __shared__ uint32_t wsum[17];
    __shared__ uint32_t scratch[18];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const bool fast = a.rank_viol != nullptr && *a.rank_viol == 0;
    const int n_large = min(*a.n_large, a.large_cap);
    for (int w = blockIdx.x; w < n_large; w += gridDim.x) {
        const uint32_t tl = a.large[2 * w], bin = a.large[2 * w + 1];
        const int T = (int)(tl >> 31), b = (int)(tl & 0x7fffffffu);
        const SmlPrepTable& tb = a.t[T];
        const uint2 oc = tb.bk[(int64_t)b * tb.nbk + bin];
        const int S = (int)oc.y;
        const BatchGeo g = batch_geo(a, b);
        const uint32_t pos0 = (uint32_t)(tb.lmul * g.start) + oc.x;
        E* src = reinterpret_cast<E*>(tb.ent) + pos0;
        E* dst = reinterpret_cast<E*>(tb.ent2) + pos0;
        __syncthreads();                                         // the previous bucket's readers of the LDS arrays are done
        if (S <= CAP) {
            // ---- in LDS: wavefront w owns the stripe [w * R * 64, (w + 1) * R * 64) of the bucket, R rounds of 64
            const int R = (S + 1023) >> 10;
            for (int i = tid; i < S; i += 1024) lbuf[i] = src[i];
            for (int p = 0; p < tb.npass; ++p) {
                const int lo = p * tb.pbits, bits = min(tb.pbits, tb.hb - lo), nd = 1 << bits;
                for (int i = lane; i < nd; i += 64) cnt[wv][i] = 0;
                __syncthreads();                                 // (the bucket is loaded / the previous pass has landed)
                E ev[RMAX]; unsigned short wr[RMAX];
#pragma unroll
                for (int r = 0; r < RMAX; ++r) {
                    if (r < R) {
                        const int i = wv * (R * 64) + r * 64 + lane;
                        const bool valid = i < S;
                        ev[r] = valid ? lbuf[i] : (E)0;
                        wr[r] = (unsigned short)wave_rank<true>(cnt[wv], (ent_hi<E>(ev[r], tb.vb) >> lo) & (uint32_t)(nd - 1), valid, bits, fast);
                    }
                }
                __syncthreads();                                 // every stripe is in registers, every count is in
                uint32_t tot = 0;
                if (tid < nd) {
#pragma unroll
                    for (int w2 = 0; w2 < 16; ++w2) { const uint32_t c = cnt[w2][tid]; cnt[w2][tid] = (unsigned short)tot; tot += c; }
                }
                const uint32_t ex = block_excl_scan_1024(tot, wsum);
                if (tid < nd) dbase[tid] = ex;
                __syncthreads();
#pragma unroll
                for (int r = 0; r < RMAX; ++r) {
                    if (r < R) {
                        const int i = wv * (R * 64) + r * 64 + lane;
                        if (i < S) {
                            const uint32_t dg = (ent_hi<E>(ev[r], tb.vb) >> lo) & (uint32_t)(nd - 1);
                            lbuf[dbase[dg] + cnt[wv][dg] + wr[r]] = ev[r];
                        }
                    }
                }
            }
            __syncthreads();
            const E* sorted = lbuf;
            emit_bucket<E, 1024>(a, tb, T, b, bin, pos0, S, [&](int q) { return sorted[q]; }, scratch);
            continue;
        }
        for (int p = 0; p < tb.npass; ++p) {
            const int lo = p * tb.pbits, bits = min(tb.pbits, tb.hb - lo), nd = 1 << bits;
            __syncthreads();
            if (tid < 512) dbase[tid] = 0;
            __syncthreads();
            for (int c0 = 0; c0 < S; c0 += 4096) {                 // digit totals (one LDS atomic per group of equal lanes)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = c0 + wv * 256 + r * 64 + lane;
                    const bool valid = i < S;
                    const uint32_t dg = valid ? (ent_hi<E>(src[i], tb.vb) >> lo) & (uint32_t)(nd - 1) : 0u;
                    const uint64_t m = match_any(dg, valid, bits);
                    if (valid && lane == __ffsll((long long)m) - 1) atomicAdd(&dbase[dg], (uint32_t)__popcll(m));
                }
            }
            __syncthreads();
            {
                const uint32_t v = tid < nd ? dbase[tid] : 0u;
                const uint32_t ex = block_excl_scan_1024(v, wsum);
                if (tid < nd) dbase[tid] = ex;
            }
            __syncthreads();
            for (int c0 = 0; c0 < S; c0 += 4096) {
                for (int i = lane; i < nd; i += 64) cnt[wv][i] = 0;
                __syncthreads();
                E ev[4]; uint32_t wr[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = c0 + wv * 256 + r * 64 + lane;
                    const bool valid = i < S;
                    ev[r] = valid ? src[i] : (E)0;
                    wr[r] = wave_rank<true>(cnt[wv], (ent_hi<E>(ev[r], tb.vb) >> lo) & (uint32_t)(nd - 1), valid, bits, fast);
                }
                __syncthreads();
                uint32_t ctot = 0;
                if (tid < nd) {
#pragma unroll
                    for (int w2 = 0; w2 < 16; ++w2) { const uint32_t c = cnt[w2][tid]; cnt[w2][tid] = (unsigned short)ctot; ctot += c; }
                }
                __syncthreads();
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = c0 + wv * 256 + r * 64 + lane;
                    if (i < S) {
                        const uint32_t dg = (ent_hi<E>(ev[r], tb.vb) >> lo) & (uint32_t)(nd - 1);
                        dst[dbase[dg] + cnt[wv][dg] + wr[r]] = ev[r];
                    }
                }
                __syncthreads();
                if (tid < nd) dbase[tid] += ctot;
            }
            __threadfence_block();
            __syncthreads();
            E* t2 = src; src = dst; dst = t2;
        }
        const E* sorted = src;
        emit_bucket<E, 1024>(a, tb, T, b, bin, pos0, S, [&](int q) { return sorted[q]; }, scratch);
    }
}


// ------------------------------------------------------------------------------------
// k_mf_tiles (MF stage, distinct-row form): grid (tiles_cap, nb), one workgroup per 16-row tile of a batch -- the same tile
// numbering as the forward / backward launches of that batch (user tiles first).  Writes the tile's header and, per occurrence
// of its rows in (row, slot) order, one entry naming the scratch rows of the occurrence's triple: what the backward needs to
// form the tile's summed dOut rows in one round trip (SmlDense / SmlTileHdr).  Runs behind the list kernel (it reads both
// tables' slot_info).
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_mf_tiles(SmlPrepArgs a) {
    __shared__ uint32_t s_pos[16], s_start[17], s_spill;
    const int j = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const BatchGeo g = batch_geo(a, b);
    const int tu = (g.Bb + 15) >> 4, ti = (2 * g.Bb + 15) >> 4;
    SmlTileHdr* hdr = a.tile_hdr + (int64_t)b * a.tiles_cap + j;
    const int T = j >= tu ? 1 : 0, t = j - (T ? tu : 0);
    int nrows = 0;
    if (j < tu + ti) {
        const int nd = a.dense_n[2 * b + T], ntiles = (nd + 15) >> 4;
        if (t < ntiles) nrows = (nd - t + ntiles - 1) / ntiles;          // rows r with r * ntiles + t < nd
    }
    if (nrows == 0) {
        if (tid < 3) reinterpret_cast<uint4*>(hdr)[tid] = make_uint4(0u, 0u, 0u, 0u);
        return;
    }
    const int64_t sbase = (int64_t)b * a.dense_stride;
    const SmlRun* drec = a.dense_rec + sbase + (T ? (int64_t)g.ioff : 0) + t * 16;
    if (tid < 64) {                                                  // (wavefront 0: the 16 records in one round trip, prefix by shuffles)
        uint32_t len = 0, pos = 0;
        if (tid < nrows) { const uint4 r0 = *reinterpret_cast<const uint4*>(drec + tid); pos = r0.y; len = r0.z; }
        uint32_t inc = len;
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) { const uint32_t t2 = (uint32_t)__shfl_up((int)inc, off, 64); if (tid >= off) inc += t2; }
        if (tid < 16) { s_pos[tid] = pos; s_start[tid] = inc - len; }
        const uint32_t run = (uint32_t)__shfl((int)inc, 15, 64);
        // the header: 16 bytes of counts + 16 lengths (two per dword)
        const uint32_t lnext = (uint32_t)__shfl_down((int)len, 1, 64);
        uint32_t spill = 0;
        if (tid == 0) {
            spill = run > SML_TILE_ENT ? (uint32_t)atomicAdd(a.spill_cnt + b, (int)(run - SML_TILE_ENT)) : 0u;
            s_start[16] = run; s_spill = spill;
            *reinterpret_cast<uint4*>(hdr) = make_uint4(run, (uint32_t)nrows, spill, 0u);
        }
        if (tid < 16 && (tid & 1) == 0) reinterpret_cast<uint32_t*>(hdr->len)[tid >> 1] = (len & 0xffffu) | (lnext << 16);
    }
    __syncthreads();
    const uint32_t count = s_start[16];
    const uint32_t* vals = a.t[T].vals;
    const uint32_t* s2d = a.slot_info + sbase;
    uint2* ent = a.tile_ent + ((int64_t)b * a.tiles_cap + j) * SML_TILE_ENT;
    uint2* spill = a.tile_spill + (int64_t)b * 3 * a.batch + s_spill;
    for (uint32_t e = (uint32_t)tid; e < count; e += 256u) {
        int r = 0;
#pragma unroll
        for (int q = 1; q < 16; ++q) r += (q < nrows && s_start[q] <= e) ? 1 : 0;
        const uint32_t slot = vals[s_pos[r] + (e - s_start[r])];
        uint32_t tt = slot, kind = 0u;
        if (T) { const uint32_t si = slot - g.ioff; kind = si < (uint32_t)g.Bb ? 1u : 2u; tt = kind == 1u ? si : si - (uint32_t)g.Bb; }
        const uint32_t du = s2d[tt], di = s2d[g.ioff + tt], dn = s2d[g.ioff + (uint32_t)g.Bb + tt];
        const uint2 v = make_uint2(du | (di << 16), dn | ((uint32_t)r << 16) | (kind << 20));
        if (e < SML_TILE_ENT) ent[e] = v; else spill[e - SML_TILE_ENT] = v;
    }
}

// ------------------------------------------------------------------------------------
// k_rank_probe: is a returning LDS atomic a stable rank on this device?  (See wave_rank.)  Every wavefront draws keys from
// ranges of 1 .. 512 (every collision multiplicity), some lanes sit a round out, and compares the old half-word it got
// back with the number of lower lanes on the same key: whatever the counter held before the instruction must come out the
// same for every lane of a key.  *viol counts the lanes for which it did not.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_rank_probe(int* viol) {
    __shared__ __attribute__((aligned(16))) unsigned short cnt[16][512];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t s = 0x9e3779b9u * ((uint32_t)blockIdx.x * 1024u + threadIdx.x + 1u);
    auto rng = [&]() { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; };
    int bad = 0;
    const int ranges[8] = {1, 2, 3, 7, 32, 100, 317, 512};
    for (int rr = 0; rr < 8; ++rr) {
        for (int i = lane; i < 512; i += 64) cnt[wv][i] = 0;
        for (int sub = 0; sub < 6; ++sub) {
            const uint32_t key = rng() % (uint32_t)ranges[rr];
            const bool valid = (rng() & 7u) != 0u;
            const uint32_t old = wave_rank(cnt[wv], key, valid, 0, true);
            const uint64_t m = match_any(key, valid, 9);
            const uint32_t base = old - (uint32_t)__popcll(m & lanes_below());
            const int leader = valid ? (__ffsll((long long)m) - 1) : lane;
            const uint32_t base0 = (uint32_t)__shfl((int)base, leader, 64);
            if (valid && base != base0) ++bad;
        }
    }
    if (bad) atomicAdd(viol, bad);
}

template <typename E>
hipError_t launch_prep(const SmlPrepArgs& a, hipStream_t st) {
    if (a.mode == 0 && a.records && a.t[0].nbk == 1 && a.t[1].nbk == 1 && 2 * (int64_t)a.batch <= SML_PREP_SMALL) {
        k_prep_bucket<E><<<dim3((unsigned)(2 * a.nb)), dim3(256), 0, st>>>(a, 0, 2);       // ONE launch: no partition
        if (a.dense) k_mf_tiles<<<dim3((unsigned)a.tiles_cap, (unsigned)a.nb), dim3(256), 0, st>>>(a);
        return hipGetLastError();
    }
    const dim3 tiles((unsigned)a.tpb, (unsigned)a.nb);
    if (a.mode == 0) k_prep_hist<<<tiles, dim3(1024), 0, st>>>(a); else k_prep_hist_x<<<tiles, dim3(1024), 0, st>>>(a);
    k_prep_scan<<<dim3((unsigned)a.nb, 2), dim3(1024), 0, st>>>(a);
    if (a.mode == 0) k_prep_scatter<E><<<dim3(xcd_grid(a.nb, a.tpb)), dim3(1024), 0, st>>>(a);
    else k_prep_scatter_x<E><<<dim3(xcd_grid(a.nb, a.tpb)), dim3(1024), 0, st>>>(a);
    for (int T = 0; T < 2; ++T) {
        if (a.t[T].wave == 1) k_prep_wave<E><<<dim3(xcd_grid(a.nb, (a.t[T].nbk + 3) / 4)), dim3(256), 0, st>>>(a, T);
        if (a.t[T].wave == 2) k_prep_count<E><<<dim3(xcd_grid(a.nb, (a.t[T].nbk + 3) / 4)), dim3(256), 0, st>>>(a, T);
    }
    if (a.t[0].wave || a.t[1].wave) k_prep_bucket<E><<<dim3(256), dim3(256), 0, st>>>(a, 0, 1);
    for (int T = 0; T < 2; ++T)
        if (!a.t[T].wave) k_prep_bucket<E><<<dim3(xcd_grid(a.nb, a.t[T].nbk)), dim3(256), 0, st>>>(a, T, 0);
    k_prep_large<E><<<dim3(256), dim3(1024), 0, st>>>(a);
    return hipGetLastError();
}

}  // namespace

hipError_t sml_launch_rank_probe(int* viol, hipStream_t st) {
    k_rank_probe<<<dim3(512), dim3(1024), 0, st>>>(viol);
    return hipGetLastError();
}
hipError_t sml_launch_prep(const SmlPrepArgs& a, int ent_bytes, hipStream_t st) {
    if (a.n <= 0 || a.nb <= 0) return hipSuccess;
    return ent_bytes == 8 ? launch_prep<uint64_t>(a, st) : launch_prep<uint32_t>(a, st);
}
