// Host-visible kernel argument blocks and launcher prototypes (internal to libsml_hip.so).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "sml_dev.h"

// One run of equal keys in a sorted occurrence list: `len` occurrences of table row `row` at
// sorted positions pos .. pos+len-1 (len = 0: this position does not start a run).  The slots
// (gradient-row indices) of the first SML_RUN_INL occurrences ride in the record, so the common
// short run needs no second index load.
#define SML_RUN_INL 4
struct __attribute__((aligned(16))) SmlRun { uint32_t row, pos, len, pad; uint32_t slot[SML_RUN_INL]; };

// ---- MF stage, distinct-row form (round 5): the transfer net runs once per DISTINCT (table, row) of a batch ----------------
// The reference gathers one embedding row per OCCURRENCE (model/transfer.py:466-472) and autograd sums the duplicates'
// gradients in the embedding's backward; the net is a function of the row alone, so an occurrence's output equals its row's,
// and d loss / d x_hat[row] = J(row)^T * (sum over the row's occurrences of d loss / d out[occurrence]).  The index
// preparation numbers a batch's distinct rows per table (sorted by row; distinct row k goes to tile k % ntiles, position
// k / ntiles: rows with neighbouring ids -- a Zipf head -- land in different tiles), and k_mf_tiles lays out, per 16-row
// tile, what its backward needs to form the SUM of its rows' dOut before the matrix products: one entry per occurrence, in
// (row, slot) order -- the summation order is a function of the input alone.
#define SML_TILE_ENT 256          // entries of a tile held in its fixed block; the rest (a row with hundreds of occurrences) spill
struct __attribute__((aligned(16))) SmlTileHdr {
    uint32_t count;               // occurrences of the tile's rows (entries)
    uint32_t nrows;               // live rows of the tile (0: the tile is not in use)
    uint32_t spill;               // entries [SML_TILE_ENT, count) are at spill[spill ...] of the batch
    uint32_t pad;
    unsigned short len[16];       // occurrences of row r (entries of a row are contiguous, rows ascending)
};
// entry: x = scratch row of the triple's user | scratch row (inside the item run) of its positive << 16;
//        y = ... of its negative | tile row r << 16 | kind << 20 (0: the row is the triple's user, 1: its positive, 2: its negative)

// One contiguous run of rows that goes through one net.
struct SmlSeg {
    const float* theta;      // this net's flat parameter block
    const float* pk;         // this net's MFMA operand images
    const float* xt_tab;     // x_t source  (table when tri != null, else contiguous rows)
    const float* xh_tab;     // x_hat source
    const float* m_tab;      // lazy Adam state of the x_hat table (null: rows are current)
    const float* v_tab;
    const int32_t* last_tab;
    const int64_t* tri;      // [B,3] (u,i,j) or null for identity indexing
    int B;                   // triples in this batch
    int is_item;             // 0: rows = tri[:,0]; 1: rows = tri[:,1] then tri[:,2]
    int n_rows;
    float* out;              // [n_rows, d]
    float* z1;               // optional saves for backward
    float* xin;              // [n_rows, 3, d]  (x_t, x_hat, x_com)
    float* a1;               // [n_rows, 5d]
    float* a2;               // TR stage: [n_rows, 512] Gelu(z1), the B operand of dW2 (the weight-gradient kernel then needs no Gelu)
    float* mrep; float* vrep;   // lazy gather: [n_rows, d] the rows' Adam moments after the replay (for the row update)
    // MF stage, distinct-row form (SmlDense; null: off): scratch row k of this run is DISTINCT table row drec[k].row, tile j's
    // live rows are the first hdr[j].nrows of its 16 (a tile without rows exits)
    const SmlRun* drec; const SmlTileHdr* hdr;
};
struct SmlFwdArgs {
    SmlSeg seg[2];
    int tiles0;              // tiles of seg[0]; the rest belong to seg[1]
    int cur_step;            // Adam step about to be applied (replay runs to cur_step-1)
    const SmlSched* sched;
    int64_t out_pstride;     // hidden-split forward (NS > 1): floats between the NS partial planes of `out`
    int tiles_total;
    int k2;                  // ConvTransfer nets: kernel (2,1), the x_com row is zero
    int unit_rows;           // NS = 1 only: rows of seg[0] leave divided by their norm (ConvTransfer's user output)
    int sched_len;           // > 0: the schedule table is followed by the closed-form replay tables (sml_dev.h) and this launch uses them
    // TR stage, hidden-split form (NS = 4), one GPU: the PREVIOUS batch's conv-parameter Adam step is taken HERE (cs_in !=
    // null).  The merged launch of batch b - 1 left its tail workgroups' compact conv-gradient partials in cg_part
    // (rows [0, cg_split): user net, [cg_split, cg_total): item net; null: nothing pending, the parameters pass through);
    // every workgroup adds its net's partials in the fixed order of k_tr_wgrad2's last arriver and steps its own copy of the
    // 95 parameters from cs_in ([2 nets][3: p, m, v][SML_CG]); the net's first workgroup also stores the result to cs_out
    // (the other parity: nobody of this launch reads it) and to the flat theta / m / v for the launches that follow.
    const float* cg_part; int cg_split, cg_total;
    const float* cs_in; float* cs_out;
    float* cs_theta; float* cs_m; float* cs_v;       // flat buffers (both nets)
    float cs_wd, cs_step_size, cs_bc2_sqrt;
};
#define SML_FWD_NS 4         // largest hidden-dimension split of the training-batch forward (planes of `out`)

// ---- one-shot exchange over peer mappings (include/sml_hip.h, sml_peer_*) -------------------------------------
// destinations of one push: rank q's slot for THIS rank's contribution and its arrival counter, for this step's parity
struct SmlPeerPush {
    int world;                                   // 0: off
    float* dst[SML_MAX_PEERS];
    unsigned long long* flag[SML_MAX_PEERS];
};
// sources of one poll: this rank's own slots / counters of this step's parity
struct SmlPeerPoll {
    int world;                                   // 0: off
    const float* slot0; long long slot_stride;   // slot q at slot0 + q * slot_stride (floats)
    const unsigned long long* flag0;             // [world]
    unsigned long long expect;                   // counter value that completes this step
    long long timeout;                           // 100 MHz ticks
    int* err;                                    // incidents (consumers that gave up)
    int waited;                                  // 1: a k_peer_wait launch ahead of this kernel did the waiting
};

struct SmlBwdSeg {
    const float* theta;
    const float* pk;
    float* dout;             // [n_rows, d] this run's dOut rows (written in the TR stage for the weight gradients)
    int is_item;             // 0: rows are the batch's users; 1: positives then negatives
    const float* z1;         // [n_rows, 512]
    const float* xin;        // [n_rows, 3, d]
    float* dx;               // MF stage: [n_rows, d] gradient w.r.t. x_hat (+ l2*x_hat); null in TR stage
    float* dz1;              // TR stage: [n_rows, 512]; null in MF stage
    int n_rows;
};
#define SML_SLOT_ONCE 0x80000000u
// MF stage, one GPU, index lists built by index_prep.hip: the row update (k_run_update<Adam>) is taken by the backward itself.
// A row that occurs ONCE in the batch (slot_info) is stepped by the threads that hold its gradient, from the forward's replayed
// copies (xin row 1, mrep, vrep).  An occurrence of a duplicated row leaves its gradient row (write-through), bumps the run's
// arrival counter, and the LAST arriver adds the run's rows -- in slot order, with k_run_update's own summation order, so the
// result is the same bits whichever kernel takes the step -- and steps the row.  slot_info == null: off (dx rows are left
// for k_run_update).
struct SmlFusedUpdate {
    const uint32_t* slot_info;                 // this batch's [ioff + 2B]
    const SmlRun* rec[2];                      // this batch's records (users, items), one per sorted position
    const uint32_t* val[2];                    // whole-epoch sorted value lists (SmlRun.pos indexes them)
    int* arrive;                               // [B] + [2B] arrival counters (zero between launches)
    float* dx_all;                             // gradient rows by slot (users at 0, items at ioff)
    const int64_t* tri;                        // this batch's triples
    float* w[2]; float* m[2]; float* v[2]; int32_t* last[2];
    const float* mrep; const float* vrep;      // by slot, like dx_all
    const SmlSched* sched; int cur_step;
};
// distinct-row form of the MF backward (hdr == null: off): this batch's tile headers / entry blocks / spill area (workgroup =
// tile index), the dense records (users at 0, items at ioff) and the per-batch triple count
struct SmlDense {
    const SmlTileHdr* hdr; const uint2* ent; const uint2* spill; const SmlRun* drec;
};
struct SmlBwdArgs {
    SmlBwdSeg seg[2];
    SmlFusedUpdate fu;
    SmlDense dn;
    // MF stage on several GPUs, one-shot exchange (k_transfer_bwd_full): the item tiles store their gradient rows straight into
    // every rank's inbox slot (row r of this rank's 2B item rows at dst[q] + r * d) and EVERY workgroup of the launch signals --
    // the grid is cut for the epoch's batch cap, the same on every rank, so every rank's counters grow alike; workgroups
    // beyond tiles_live (this batch is shorter than the cap) only signal.  world 0: off (k_peer_push does it).
    SmlPeerPush push; int tiles_live;
    int tiles0;
    float l2;
    // the pair loss is evaluated here: out rows of the whole batch (u' at t, i' at ioff+t, n' at ioff+B+t)
    const float* out_all; int B; int ioff; int kind; float scale;
    int out_np; int64_t out_pstride;   // `out_all` is the sum of out_np planes, out_pstride floats apart
    float* loss_part;        // [tiles] this batch's per-workgroup loss partials
    float* convg_part;       // TR stage: [tiles, SML_CG] per-tile compact conv1/conv2 gradient partials; else null
    int tiles_total;         // k_tr_bwd_head: row tiles of the batch (its grid is padded to the XCD map)
};

struct SmlWgSeg {
    const float* dz1; const float* a1; const float* dout; const float* a2;    // a2 = Gelu(z1), saved by the forward
    float* grad;             // this net's flat gradient block
    int n_rows;
    const float* theta_net; const float* pk_net; const float* xin;   // k_tr_wgrad2's tail workgroups (dA1, tail, conv gradients)
};
struct SmlWgArgs {
    SmlWgSeg seg[2];
    // fused Adam (single-GPU path): every workgroup owns a fully reduced tile of a weight
    // gradient, so it can take the Adam step for those weights (and refresh their operand-image
    // entries) on the spot; two extra workgroups finish the conv parameters.  null theta: off.
    float* theta; float* m; float* v; float* pk;
    const float* convg_part; int tiles0, tiles_total;
    float weight_decay, step_size, bc2_sqrt;
    SmlPeerPush peer;        // several GPUs: every finished gradient tile is also stored into the peers' inboxes
    // k_tr_wgrad2: the first n_tail workgroups are the backward's tail (row tile tb / (d/16), coordinate slice tb % (d/16));
    // tiles0 / tiles_total then count ROW tiles; `arrive` is a device counter (0 between launches) the tail
    // workgroups use to elect the last arriver, which finishes the conv parameters
    int n_tail; float* convg_out; int* arrive;
    int defer_conv;          // k_tr_wgrad2: 1 = the tail workgroups only leave their partials; the NEXT forward finishes the conv parameters
    int gelu_b;              // 1 = seg.a2 holds z1: the dW2 tiles apply Gelu to their B operand themselves (the forward saved z1 only)
};

struct SmlThetaAdamArgs {
    float* theta; float* m; float* v; float* grad; float* pk;
    float weight_decay, step_size, bc2_sqrt;
    SmlPeerPoll peer;        // several GPUs: the gradient is the rank-order sum of the inbox slots (grad is not read)
    const float* clip_sumsq; float clip_max_norm;   // --clip_grad: sum of squares of the whole gradient (device), the norm bound
};
hipError_t sml_launch_grad_sumsq(const float* grad, int64_t n, float* out, hipStream_t st);
hipError_t sml_launch_conv_state_init(int d, const float* theta, const float* m, const float* v, float* cs, hipStream_t st);

// mt row-tiles of 16 per workgroup; ns workgroups share a row tile (1, or SML_FWD_NS with mt = 1)
hipError_t sml_launch_fwd(int d, int mt, int ns, const SmlFwdArgs& a, int tiles_total, hipStream_t st, bool side = false);
// split != 0: d/16 workgroups per row tile (coordinate split); 0: one workgroup per row tile
hipError_t sml_launch_bwd(int d, int split, const SmlBwdArgs& a, int tiles_total, hipStream_t st);
hipError_t sml_launch_wgrad(int d, const SmlWgArgs& a, hipStream_t st);
// restructured TR step: backward head (loss -> dOut -> dZ1) and the merged weight-gradient + backward-tail launch
hipError_t sml_launch_tr_bwd_head(int d, const SmlBwdArgs& a, int tiles_total, hipStream_t st);
hipError_t sml_launch_tr_wgrad2(int d, const SmlWgArgs& a, hipStream_t st);
int sml_wgrad2_pushers(int d);                   // counter increments one merged launch adds per destination (fixed)
hipError_t sml_launch_theta_adam(int d, const SmlThetaAdamArgs& a, hipStream_t st);
hipError_t sml_launch_theta_pack(int d, const float* theta, float* pk, hipStream_t st);
// table-sized forward on bf16 products with fp32-grade results (transfer_net.hip, k_transfer_fwd_bx3): d = 32
size_t sml_bx3_bytes(int d);                     // operand images of both nets (0: not available at this d)
hipError_t sml_launch_theta_pack_bx3(int d, const float* theta, void* pkx, hipStream_t st);
hipError_t sml_launch_mf_fwd_bx3(int d, const SmlFwdArgs& a, const void* pkx, int tiles, hipStream_t st);      // the MF stage's 16-row forward (both nets' images)
hipError_t sml_launch_fwd_bx3(int d, const SmlFwdArgs& a, const void* pkx_net, int tiles, hipStream_t st, bool side);
int sml_wgrad_grid(int d);                       // workgroups (= pushers) of one weight-gradient launch
// generic push / wait / rank-order sum over peer mappings (mf_kernels.hip)
int sml_peer_push_blocks(long long n_floats);
hipError_t sml_launch_peer_push(const float* src, long long n_floats, const SmlPeerPush& p, hipStream_t st);
hipError_t sml_launch_peer_wait(const SmlPeerPoll& p, hipStream_t st);
hipError_t sml_launch_peer_sum(float* dst, long long n_floats, const SmlPeerPoll& p, hipStream_t st, int theta_net = 0);
hipError_t sml_launch_peer_read(const void* src, void* dst, long long n16, hipStream_t st);
hipError_t sml_launch_selftest(const float* A, const float* W, float* pk, float* out, hipStream_t st);

// sharded bare step: which occurrences a list holds
struct SmlShardKeys {
    int mode;                // 1: the job's occurrences of tail rows THIS rank owns (key row = row local to the shard, value =
                             //    src rank * rows_cap + slot in that rank's batch); 2: this rank's own occurrences of head rows
                             //    (key row = row, value = slot in the local dx)
    int rank; long long head_rows, shard_rows, rows_cap;
};
// ---- mf_kernels.hip ------------------------------------------------------------------
hipError_t sml_launch_adaptive_users(int d, const float* xin, float* dx, int B, float beta, float* loss_slot, hipStream_t st);
hipError_t sml_launch_loss_finalize(const float* part, int n_batches, int stride, const int* counts,
                                    float* out, hipStream_t st);


struct SmlRunArgs {
    // run records of this batch.  off_* == null: one record per sorted position (n_* of them, len 0 =
    // not a head).  off_* != null: compacted whole-epoch lists; this batch's range is off[b]..off[b+1].
    const SmlRun* run_u; int n_u;
    const SmlRun* run_i; int n_i;
    const int* off_u; const int* off_i; int batch_index;
    const int* cnt_u; const int* cnt_i;             // not null: this batch's records are off[b] .. off[b] + cnt[b * SML_PREP_CNT_STRIDE] (lists built by index_prep.hip)
    int known;                                      // 1: run_u / run_i / n_u / n_i ARE this batch's slice of the compacted lists (the host has read the counts back:
                                                    // no dependent offset / count loads at the head of the kernel); off_* / cnt_* are null then
    const uint32_t* val_u; const uint32_t* val_i;   // whole sorted value lists (SmlRun.pos indexes them): slot of each occurrence
    const float* dx;         // per-occurrence gradient rows the user values index
    const float* dx_i;       // ... and the item values index (the all-gathered buffer on several GPUs)
    void* w_user; void* w_item;
    float* m_user; float* v_user; float* m_item; float* v_item;   // Adam only
    int32_t* last_user; int32_t* last_item;                       // Adam only
    const SmlSched* sched; int cur_step;                          // Adam only
    int sched_len;                                                // Adam only, > 0: closed-form replay tables behind the schedule (sml_dev.h)
    // Adam, MF stage: the forward already replayed every gathered row's pending zero-gradient steps; it left the
    // replayed row in xin[slot][1] and the moments in rep_m / rep_v [slot]: the update starts from those (no second
    // replay, no table read).  rep_u / rep_i: which of the two lists' slots index that local scratch.
    const float* rep_x; const float* rep_m; const float* rep_v; int rep_u, rep_i;
    int rep_x_stride, rep_x_off;   // the replayed row of slot s starts at rep_x[s * rep_x_stride + rep_x_off]
    float lr;                                                     // SGD only
    SmlPeerPoll wait;        // several GPUs, one-shot exchange: every workgroup first waits for the ranks' gradient rows (world 0: off)
    // hot rows (SGD, large batches): runs longer than SML_HOT are listed here (by the index preparation)
    // instead of being summed by one wavefront; the first hot_blocks workgroups of the run kernel reduce
    // their chunks and k_hot_apply finishes them.  null: off.
    const uint32_t* hot_list;// this batch's [hot_cap][3]: (pos | is_item << 31), len, row
    const int* hot_count;    // this batch's number of hot runs
    int hot_blocks;
    int* hot_first;          // [hot_cap] first chunk index of each hot run
    float* hot_part;         // [hot_chunks][d] chunk partial sums
    int hot_cap;
};
#define SML_HOT 128          // runs longer than this take the hot path
#define SML_HOT_CHUNK 512    // occurrences per workgroup in the hot-row partial sums
#define SML_HOT_MAXCAP 8192
// ------------------------------------------------------------------------------------
// Index preparation by hand (index_prep.hip): per batch and table ("list"), the occurrences are partitioned
// by the LOW bits of their row into buckets (stable: slot order survives), each bucket is sorted by the
// remaining row bits in LDS, and the run records / unique marks / hot-row list leave from there.
// ------------------------------------------------------------------------------------
#define SML_PREP_CNT_STRIDE 32   // ints between the batches' run counters: one cache line each (every bucket of a batch bumps its batch's)
#define SML_PREP_IPT 4           // triples per thread of a partition tile
#define SML_PREP_TT (1024 * SML_PREP_IPT)   // triples per partition tile
#define SML_PREP_MAXBK 1024      // most buckets per list
#define SML_PREP_CG 8            // buckets per workgroup of the record compaction
#define SML_PREP_SMALL 2048      // entries a bucket may hold to be sorted in LDS by the small-bucket kernel
#define SML_PREP_CROWS 2048      // k_prep_count: most distinct row_hi values of a bucket (one LDS counter each: hb <= 11)
#define SML_PREP_CCAP 1536       // ... most entries of a bucket it takes (24 per lane), of which at most
#define SML_PREP_CDUP 1024       // ... this many may belong to duplicated rows (their slots are staged in LDS)
struct SmlPrepTable {
    int nbk, lb;                 // buckets per list (a power of two) and its log2
    int hb;                      // row bits above the bucket bits (sorted inside the bucket)
    int vb;                      // value bits inside an entry (32: 64-bit entries)
    int npass, pbits;            // LDS radix passes over those bits, bits per pass (<= 9)
    int wave;                    // compact mode, one WAVEFRONT per bucket first: 1 = k_prep_wave (small buckets, few duplicates),
                                 // 2 = k_prep_count (few row_hi bits: a counter per row)
    int ntile;                   // partition tiles per list (tiles per batch x streams of this table; 0: the table has no occurrences)
    int lmul;                    // a list starts at lmul * (the batch's first triple) in the table's occurrence arrays
    int allruns;                 // compact mode: EVERY run gets a record (not only duplicated ones), every occurrence its value; no marks
    int rshift;                  // a bucket's record stretch starts at position >> rshift (1: at most every second occurrence heads a record)
    void* ent; void* ent2;       // [occurrences] partitioned entries (row_hi << vb | value); ent2: ping-pong for large buckets
    uint32_t* hist;              // [nb][tiles][nbk] tile histograms, turned into the tiles' first positions
    uint2* bk;                   // [nb][nbk] (first position of the bucket inside its list, entries)
    uint32_t* brc;               // (unused since round 4: a bucket takes its records' place from the batch's counter)
    SmlRun* runs_tmp;            // (unused since round 4)
    uint32_t* vals;              // [occurrences] out: values (slots) in sorted order -- written for duplicated runs (all, in records mode)
    SmlRun* runs;                // compact mode: the table's run records, list b's at run_off[b]; records mode: one per position
    int* run_off; int* run_cnt;  // [nb] / [nb * SML_PREP_CNT_STRIDE] (compact mode): list b's records are runs[run_off[b] .. + run_cnt[b * stride]),
                                 // placed bucket by bucket through one returning atomicAdd on the batch's counter
};
struct SmlPrepArgs {
    const int64_t* tri; int64_t n; int batch; int nb; int tpb;   // tpb: partition tiles per batch
    const int* boff;             // planned batches of unequal size (or null)
    int pad_tiles;               // the batch's first item value is rounded up to a multiple of SML_R
    int records;                 // 1: one record per sorted position (MF stage); 0: compact records of duplicated runs + unique marks
    // occurrence source (index_prep.hip, occ_of): 0 this rank's triples; 1 / 2 / 3 the multi-GPU item lists
    int mode, has_users, nis;    // nis: item streams per tile (2, or 2 * world)
    const int64_t* items_all; int64_t val_q;        // [world][n][2] gathered item columns; value stride per rank
    // mode 4 (the MF stage's job-wide item lists on several GPUs): ONE item stream of explicit (key, value) occurrences, batch-major
    // -- "triple" t of batch b is occurrence boff[b] + t (or the uniform layout), row = the key's low 32 bits; no users
    const uint64_t* x_keys; const uint32_t* x_vals;
    int64_t head_rows, shard_rows; int shard_rank;
    SmlPrepTable t[2];           // users, items
    uint8_t* uniq; int64_t uniq_stride;
    // records mode (MF stage), optional: what every slot needs to take its row's update on its own (the fused row update of
    // k_transfer_bwd_full): slot_info[b * slot_stride + value] = SML_SLOT_ONCE if the row occurs once in the batch, else the
    // position (inside the batch's list) of the record of its run
    uint32_t* slot_info; int64_t slot_stride;
    // records mode, both lists one bucket (the MF stage's batches): distinct-row numbering instead of the per-position records
    // (SmlDense).  slot_info then holds, per slot, the scratch row of its table row (inside its run); dense_rec the records by
    // scratch row ([nb][dense_stride], dense_stride = whole tiles of both lists: distinct row k sits at scratch row
    // (k % ntiles) * 16 + k / ntiles, i.e. anywhere below ntiles * 16 -- past 2 * batch when batch % 8 != 0; users at 0, items at
    // the batch's ioff); dense_n [nb][2] the distinct rows per list;
    // k_mf_tiles writes tile_hdr [nb][tiles_cap], tile_ent [nb][tiles_cap][SML_TILE_ENT], spill [nb][3 * batch] / spill_cnt [nb]
    int dense; SmlRun* dense_rec; int* dense_n; int64_t dense_stride;
    SmlTileHdr* tile_hdr; uint2* tile_ent; uint2* tile_spill; int* spill_cnt; int tiles_cap;
    uint32_t* hot_list; int* hot_count; int hot_cap; int* max_len;
    uint32_t* medium; int* n_medium;                  // same pairs: buckets k_prep_wave leaves to k_prep_bucket
    uint32_t* large; int* n_large; int large_cap;     // (table << 31 | list), bucket -- buckets the small kernel leaves
    // stable ranks from ONE returning LDS atomic per occurrence, if *rank_viol == 0: the context's start-up probe
    // (sml_launch_rank_probe) counted no returning atomic that was served out of lane order; null / non-zero: ballot ranking
    const int* rank_viol;
    // always-on invariant of every sorted bucket that leaves through emit_bucket: neighbouring entries must be ordered by
    // (row, value) -- a stable sort of occurrences whose values ascend in occurrence order.  Violations are COUNTED here (never reset:
    // any non-zero value means a list of this index set was built wrong at some point) and the next epoch call that finds the count
    // on the host fails with SML_ESTATE.  null: off.
    int* order_viol;
    int vals_ascend;             // 1: a list's values ascend in occurrence order (every source but the owner-split job-wide MF lists, whose values
                                 // are slots by OWNER): only then does the invariant's value half apply; the row half always does
};
hipError_t sml_launch_rank_probe(int* viol, hipStream_t st);
hipError_t sml_launch_prep(const SmlPrepArgs& a, int ent_bytes, hipStream_t st);
hipError_t sml_launch_hot_apply(int d, int dtype_bytes, const SmlRunArgs& a, hipStream_t st);
hipError_t sml_launch_run_adam(int d, const SmlRunArgs& a, int64_t max_records, hipStream_t st);
hipError_t sml_launch_run_sgd(int d, int dtype_bytes, const SmlRunArgs& a, int64_t max_records, hipStream_t st);
hipError_t sml_launch_adam_flush(int d, float* w, float* m, float* v, int32_t* last, int64_t rows,
                                 const SmlSched* sched, int cur_step, int sched_len, hipStream_t st);
// key_bytes 4: keys (batch << row_bits) | row in 32 bits; 8: (batch << 32) | row
hipError_t sml_launch_mark_runs(int key_bytes, const void* keys, const uint32_t* vals, int64_t n, int row_bits, SmlRun* rec,
                                uint8_t* flag_dup, uint8_t* uniq, int64_t uniq_stride, int64_t uniq_item_base, hipStream_t st);
hipError_t sml_launch_mark_unique(int key_bytes, const void* keys, const uint32_t* vals, int64_t n, int row_bits, uint8_t* uniq,
                                  int64_t uniq_stride, hipStream_t st);
hipError_t sml_launch_make_runs(int key_bytes, const void* keys, const uint32_t* vals, int64_t n, int row_bits, const uint32_t* heads,
                                const int* n_heads, int64_t max_heads, SmlRun* runs, int* max_len, int64_t seg, int is_item,
                                uint32_t* hot_list, int* hot_count, int hot_cap, hipStream_t st);
hipError_t sml_launch_batch_offsets(const SmlRun* runs, const int* n_sel, int nb, int64_t seg, int* off, hipStream_t st);
// several GPUs, bare step: keys / slots of the job's item occurrences from the gathered item columns
// items_all [world][n][2]; slot of rank q's element t of batch b: q*2*batch + t (positive), q*2*batch + B_b + t (negative)
hipError_t sml_launch_build_item_keys_x(int key_bytes, const int64_t* items_all, int world, int64_t n, int batch, int row_bits_i,
                                        void* key_i, uint32_t* val_i, hipStream_t st);
// the same for the item-sharded step: occurrences that do not belong to the list get the batch's sentinel row (all
// ones in row_bits_i bits), which sorts behind every real row of the batch and is never selected as a run
hipError_t sml_launch_build_item_keys_sh(int key_bytes, const int64_t* items_all, int world, int64_t n, int batch, int row_bits_i,
                                         const SmlShardKeys& sk, void* key_i, uint32_t* val_i, hipStream_t st);
// dense all-reduced head update: w_head[row] -= lr * (slot 0 + slot 1 + ... in rank order)[row]
hipError_t sml_launch_head_apply(int d, int dtype_bytes, void* w_head, long long head_rows, float lr, const SmlPeerPoll& p, hipStream_t st);
hipError_t sml_launch_peer_signal(const SmlPeerPush& p, hipStream_t st);
hipError_t sml_launch_set_ptr_tab(void** dst, void* const* src, int n, hipStream_t st);
// uniq[b][B_b .. 3*B_b) = 0: item occurrences are never updated in place
hipError_t sml_launch_zero_item_marks(uint8_t* uniq, int64_t n, int batch, hipStream_t st);
// boff: device [nb+1] offsets of the batches inside tri (null: batches of `batch`, the last one ragged)
hipError_t sml_launch_build_keys(int key_bytes, const int64_t* tri, int64_t n, int batch, int pad_tiles, int row_bits_u,
                                 int row_bits_i, void* key_u, uint32_t* val_u, void* key_i, uint32_t* val_i, const int* boff,
                                 int nb, hipStream_t st);

struct SmlBareArgs {
    void* w_user; void* w_item;
    const uint8_t* uniq;     // [3B] 1: this occurrence's row occurs once in the batch -> updated in place here
    float lr;
    const int64_t* tri; int B;
    float* dx;               // [3B, d] per-occurrence gradients (fp32)
    float* loss_part;
    int kind; float lam_user, lam_item;
    float scale;             // multiplies the pair loss and its gradients (B_local / B_global for a split BCE batch; else 1)
    // lazy dense-Adam form (sched != null; fp32 tables): the rows' pending zero-gradient steps are replayed
    // before use; uniq is ignored (every occurrence emits its gradient row)
    const float* m_user; const float* v_user; const float* m_item; const float* v_item;
    const int32_t* last_user; const int32_t* last_item;
    const SmlSched* sched; int cur_step;
    float* xrep; float* mrep; float* vrep;   // lazy form: [3B, d] replayed rows / moments per occurrence, for the row update
    // item-SHARDED form (several GPUs over peer mappings; shard_tab != null): item rows [0, head_rows) are read from the
    // local replica `w_item`, row r >= head_rows from rank q = (r - head_rows) / shard_rows's shard (shard_tab[q], peer
    // memory).  A head occurrence emits its gradient row into the local dx as usual; a tail occurrence stores it
    // straight into its OWNER's inbox (inbox_tab[q] + push_off + slot * d), and every workgroup then bumps every
    // rank's arrival counter once (peer).
    const void* const* shard_tab; float* const* inbox_tab;
    long long head_rows, shard_rows, push_off;
    SmlPeerPush peer;
};

hipError_t sml_launch_bare_grad(int d, int dtype_bytes, const SmlBareArgs& a, int* n_blocks, hipStream_t st);
hipError_t sml_launch_mf_forward(int d, const float* wu, const float* wi, const int64_t* user, const int64_t* item,
                                 int64_t n, int norm, float* uemb, float* iemb, float* score, hipStream_t st);
hipError_t sml_launch_eval_ranks(int d, const float* wu, const float* wi, const int64_t* rows, int64_t n,
                                 int n_cols, int32_t* rank, hipStream_t st);
hipError_t sml_launch_eval_bucketize(const int64_t* rows, int64_t n, int n_cols, int64_t n_item, int32_t* rows_out,
                                     int32_t* bucket_off, hipStream_t st);
hipError_t sml_launch_copy_tables(int n_jobs, void* const* dst, const void* const* src, const long long* bytes, hipStream_t st);
hipError_t sml_debug_set_timeline(long long* buf);
hipError_t sml_launch_flag_set(int* flag, int value, hipStream_t st);
hipError_t sml_launch_flag_wait(int* flag, int value, long long timeout_ticks, hipStream_t st);
hipError_t sml_launch_eval_ranks_bucketed(int d, const float* wu, const float* wi, const int32_t* rows_b,
                                          const int32_t* bucket_off, int64_t n, int n_cols, int32_t* rank, int max_blocks,
                                          hipStream_t st);
int sml_evs_slices(int d, int64_t n_item);
hipError_t sml_launch_evs_prepare(const int64_t* rows, int64_t n, int n_cols, int ns, int32_t* whist, int32_t* mbcnt, int32_t* seg_off,
                                  uint32_t* entries, hipStream_t st);
hipError_t sml_launch_evs_ranks(int d, const float* wu, const float* wi, const int64_t* rows, const uint32_t* entries, const int32_t* seg_off,
                                int64_t n, int n_cols, int64_t n_item, int ns, float* ug, float* s0, uint16_t* partial, int32_t* rank,
                                int max_blocks, hipStream_t st);
hipError_t sml_launch_sample_negatives(const int64_t* users, int64_t n, const int64_t* item_all, int64_t pop, const int64_t* user_ptr,
                                       int64_t n_users, const int64_t* user_items, uint64_t seed, int64_t* negs, int* failed,
                                       hipStream_t st);
hipError_t sml_launch_device_epoch(const int64_t* ui, const void* mat, int elem_bytes, int64_t stride, int64_t col, int64_t n, uint64_t seed,
                                   int64_t* out3, hipStream_t st);
hipError_t sml_launch_eval_metrics(const int32_t* rank, int64_t n, int topk, float* out, hipStream_t st);
