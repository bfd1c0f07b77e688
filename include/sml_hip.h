/*
 * sml_hip.h -- C ABI of libsml_hip.so, the MI355X (gfx950) implementation of the
 * SML per-period retraining hot path.
 *
 * The reference (zyang1580/SML) has no FFI: the path sits behind a Python module
 * surface.  Each entry point below names the reference code it replaces
 * (paths relative to the reference checkout).  The Python host side
 * (sml_amd/engine.py) binds these with ctypes; INTEGRATION.md shows the stub a
 * reference maintainer would add.
 *
 * Conventions
 *   - every function returns 0 on success or a negative SML_E* code and never
 *     throws; sml_last_error() returns a thread-local message for the last failure;
 *   - every device buffer is a caller-owned raw device pointer (e.g. a torch
 *     tensor's data_ptr()); the library never frees or retains it past the call;
 *   - every launch goes to the caller's hipStream_t (passed as void*), is
 *     asynchronous and performs no hidden synchronisation, except where noted;
 *   - scratch lives in an opaque sml_ctx; scalars (losses) are written to device
 *     memory supplied by the caller;
 *   - tables are row-major fp32 [rows, d]; d in {32, 64, 128}; indices are int64
 *     exactly as the reference's DataLoader yields them (model/transfer.py:466-468);
 *   - theta (the transfer net) is ONE flat fp32 buffer holding the user net then the
 *     item net, each laid out as sml_theta_offset() reports (16-byte aligned tensors
 *     in the reference's state_dict order: conv1.weight, conv1.bias, conv2.weight,
 *     conv2.bias, fc1.weight, fc1.bias, fc2.weight, fc2.bias; model/conv_transfer.py:23-34).
 */
#ifndef SML_HIP_H
#define SML_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SML_OK 0
#define SML_EINVAL (-1)   /* bad argument (unsupported d, null pointer, size mismatch) */
#define SML_EHIP (-2)     /* a HIP runtime call failed; see sml_last_error() */
#define SML_ENOMEM (-3)
#define SML_ESTATE (-4)   /* call sequence error (e.g. packed weights stale) */

/* loss selection: ConvTransfer_com.run_MF(BCE=True) is the reference default
 * (model/conv_transfer.py:113-126); BPR is its BCE=False branch (:128-134) */
#define SML_LOSS_BCE 0
#define SML_LOSS_BPR 1
#define SML_LOSS_BPR_NORM 2   /* BPR with score / ||u'|| (norm=True, :130-132) */
#define SML_LOSS_BPR_UNIT 3   /* ConvTransfer.run_MF (:71-85): BPR over u' / ||u'||.detach() */

typedef struct sml_ctx sml_ctx;

const char* sml_last_error(void);
int sml_version(void);

/* ---- context ------------------------------------------------------------------ */
/* Scratch for batches of up to max_batch triples at width d on `device`. */
int sml_ctx_create(sml_ctx** out, int device, int d, int max_batch);
int sml_ctx_destroy(sml_ctx* ctx);
/* Which of the reference's two convolutional transfers the following calls run
 * (model/transfer.py:377-382, --transfer_type):
 *   0  ConvTransfer_com (conv_com, the default; model/conv_transfer.py:87-135)
 *   1  ConvTransfer     (conv; :52-85): conv1 kernel (2,1) over (x_t, x_hat) -- theta keeps the [10][3]
 *      conv1 block with a zero third column -- no x_com row, user-net output divided by its detached norm
 *      (sml_transfer_forward with net 0 returns the normalised rows), loss SML_LOSS_BPR_UNIT -- or SML_LOSS_BPR_NORM for
 *      ConvTransfer.run_MF(norm=True), model/conv_transfer.py:79-81: the same value, the norm differentiated through.
 * + 2 (variant 2 / 3): the context is an EVALUATION-STREAM context -- a second context whose sml_transfer_forward calls over
 *      whole tables run beside the training context's on another stream (forwards that only an evaluation reads); nothing
 *      changes in what they compute, but those launches carry a kernel name (k_side_transfer_fwd) and a timing class of
 *      their own, so that traces and sml_prof_get tell them from the training stream's. */
int sml_ctx_set_variant(sml_ctx* ctx, int variant);
/* --clip_grad (reference model/transfer.py:725-727, torch.nn.utils.clip_grad_norm_(transfer.parameters(), max_norm, 2)
 * between backward and optimizer.step() in the TR loop): max_norm > 0 makes the TR stage epoch finish the flat theta
 * gradient (after the exchange on several GPUs; on the one-shot peer exchange the rank-order sum of the inbox slots is
 * materialised first), take its 2-norm and scale it by min(1, max_norm / (norm + 1e-6)) before Adam; the step then runs
 * un-fused (one reduction launch + one Adam launch more per batch).  0 switches it off. */
int sml_ctx_set_grad_clip(sml_ctx* ctx, float max_norm);
/* --need_adaptive (reference model/transfer.py:490-499, beta = 0.1 there): the MF stage's loss gains
 * sum over the batch's unique users of beta * count_u / ||w_u|| (detached) * ||w_u||^2; beta > 0 adds it (one small launch per
 * batch between the backward and the row update), 0 switches it off. */
int sml_ctx_set_adaptive(sml_ctx* ctx, float beta);

/* ---- theta layout ------------------------------------------------------------- */
/* Floats in one net's flat block / offset of tensor `which` (0..7 in state_dict
 * order) inside it.  The full theta buffer is 2 * sml_theta_net_size(d) floats. */
int64_t sml_theta_net_size(int d);
int64_t sml_theta_offset(int d, int which);
/* Rebuild the MFMA operand images of fc1/fc2 after theta was written by the host
 * (construction, load_state_dict).  The TR-stage Adam keeps them current itself. */
int sml_theta_pack(sml_ctx* ctx, const float* theta, void* stream);

/* ---- native RCCL exchange (multi-GPU) ------------------------------------------------ */
/* The library can issue the two exchange collectives itself, on the compute stream: no host
 * callback per batch.  It binds the RCCL that is ALREADY loaded in the process (the host passes
 * the path of that librccl.so; no link-time dependency), creates its own communicator from a
 * 128-byte unique id the host broadcasts (rank 0: sml_comm_unique_id), and then
 *   - sml_tr_stage_epoch with grad_hook == NULL all-reduces the flat theta gradient,
 *   - sml_mf_stage_epoch with xchg->hook == NULL all-gathers the item-gradient rows.
 * sml_comm_allreduce / sml_comm_allgather are exported for the host's start-up self-check. */
int sml_comm_load(const char* librccl_path);
int sml_comm_unique_id(void* out128);
int sml_comm_init(sml_ctx* ctx, int world, int rank, const void* id128);
int sml_comm_destroy(sml_ctx* ctx);
int sml_comm_allreduce(sml_ctx* ctx, float* buf, int64_t n, void* stream);
int sml_comm_allgather(sml_ctx* ctx, const float* src, float* dst, int64_t n_per_rank, void* stream);

/* ---- one-shot exchange over peer mappings (multi-GPU; no collective library on the data path) --------------
 * The reference is single-device (main_yelp.py:125, .cuda() throughout model/transfer.py:317-385): this wraps its
 * per-batch loops (theta-gradient of a 256-triple batch, model/transfer.py:701-728; item-gradient rows of a
 * 1,024-triple batch, :463-511) for one process per GPU.  At these sizes (0.79 MB of theta gradient, 262 KB of rows
 * per rank, every 26-42 us) a collective library's launch + protocol floor is longer than the step it sits in; with
 * 7 direct xGMI links per GPU every rank can instead WRITE its contribution straight into every peer's memory:
 *
 *   regions   every rank provides an `inbox` (theta slots [2 parities][world][G floats], then row slots
 *             [2 parities][world][rows_cap][d]) and a `flags` region (arrival counters), sized by
 *             sml_peer_region_bytes, allocated with sml_peer_alloc (uncached / fine-grained device memory: remote
 *             stores land in it without the owner's L2 holding stale lines) and ZEROED there.
 *   mapping   sml_peer_attach takes RAW POINTERS: inbox[q] / flags[q] = rank q's regions as addressable from this
 *             device.  One process per GPU: export with sml_peer_export (hipIpcGetMemHandle; dmabuf IPC:
 *             HSA_ENABLE_IPC_MODE_LEGACY=0), all-gather the 64-byte handles by any means, sml_peer_open them
 *             (hipIpcOpenMemHandle).  Several ranks inside one process (tests on one GPU): pass the allocations.
 *   push      TR stage: k_transfer_wgrad's epilogue stores every finished gradient tile into slot [step & 1][rank]
 *             of EVERY rank's inbox (system-scope write-through stores), then one system-scope counter increment
 *             per workgroup and destination.  MF stage / bare step: a copy kernel pushes the rank's item-gradient
 *             rows the same way.
 *   poll+sum  the consumer (theta Adam / the row update) polls its OWN counters for the step's value, then adds the
 *             world slots IN RANK ORDER: every rank forms bit-identical sums, replicas stay bit-identical without
 *             a broadcast.  Two parities are enough: a rank pushes step b+2 only after consuming step b+1, which
 *             needed every peer's b+1 push, issued after that peer consumed step b.
 *   errors    a consumer that is not released within the time-out (sml_peer_attach) counts an incident and goes
 *             on with what it has: sml_peer_status reports the count (synchronous).
 * With peers attached, sml_tr_stage_epoch (grad_hook == NULL) and sml_mf_stage_epoch / sml_embed_loss_sgd_epoch
 * (exchange hook == NULL) use this path instead of the RCCL communicator. */
#define SML_MAX_PEERS 8
int sml_peer_region_bytes(sml_ctx* ctx, int world, int64_t rows_cap, int64_t* inbox_bytes, int64_t* flags_bytes);
int sml_peer_alloc(int device, int64_t bytes, void** ptr);      /* zeroed; synchronous */
int sml_peer_free(int device, void* ptr);
/* What sml_peer_alloc got for `ptr`: 0 uncached, 1 fine-grained, 2 plain (coarse-grained) device memory, -1 not one of its
 * allocations.  Inboxes and flags that other DEVICES write must not be plain: sml_amd.dist refuses that combination. */
int sml_peer_mem_kind(void* ptr);
/* dst <- src (bytes, a multiple of 16; both 16-byte aligned) through system-scope loads that bypass this device's caches:
 * how a rank reads memory another device rewrites (an item shard of sml_embed_loss_sgd_epoch_sharded).  The start-up
 * visibility check of sml_amd.dist uses it on a probe allocation of the same kind as the shards. */
int sml_peer_read(int device, const void* src, void* dst, int64_t bytes, void* stream);
int sml_peer_export(void* ptr, void* handle64);
int sml_peer_open(int device, const void* handle64, void** ptr);
int sml_peer_close(int device, void* ptr);
int sml_peer_attach(sml_ctx* ctx, int world, int rank, void* const* inbox, void* const* flags, int64_t rows_cap,
                    double timeout_s);
int sml_peer_detach(sml_ctx* ctx);
int sml_peer_status(sml_ctx* ctx, int* timeouts);
/* start-up self-check: every rank pushes n floats of `src` into all inboxes' theta slots and reads back the rank-order
 * sum of all ranks' pushes into dst (device, n floats; n <= 2 * sml_theta_net_size).  Consumes one theta exchange step:
 * every rank must call it the same number of times.  timeout_s > 0 replaces the attach-time hang guard for this call. */
int sml_peer_allreduce_check(sml_ctx* ctx, const float* src, float* dst, int64_t n, double timeout_s, void* stream);

/* ---- a5/a6/a10: transfer net forward ------------------------------------------- */
/* ConvTransfer_com.forward (model/conv_transfer.py:92-110) for `net` (0 = user
 * transfer, 1 = item transfer): out[n,:] = net(x_t[n,:], x_hat[n,:]).  Rows are
 * contiguous; out may alias neither input.  Used for meta_train.updata
 * (model/transfer.py:884-902) over whole tables. */
int sml_transfer_forward(sml_ctx* ctx, const float* theta, int net, const float* x_t,
                         const float* x_hat, float* out, int64_t n_rows, void* stream);

/* ---- a8: MF stage (meta_train.MF_train_onestage inner loop, model/transfer.py:463-511) */
typedef struct {
    float* w_user;         /* MFbase.user_laten.weight [U,d] (trainable W_hat) */
    float* w_item;         /* MFbase.item_laten.weight [I,d] */
    const float* last_user;/* last_user_weight  W_{t-1} [U,d] */
    const float* last_item;/* last_item_weight  W_{t-1} [I,d] */
    float* m_user; float* v_user;   /* Adam exp_avg / exp_avg_sq, same shape as the tables */
    float* m_item; float* v_item;
    int32_t* step_user;    /* [U] last Adam step applied to each row (lazy replay); -1 = never touched (m = v = 0) */
    int32_t* step_item;    /* [I] */
    int64_t n_user, n_item;
} sml_mf_tables;

/* Multi-GPU exchange of the MF stage (NULL on one GPU).  Users are row-sharded: a rank only
 * sees triples of users it owns, so user rows never leave the rank.  Item tables are
 * replicated; every batch, after the backward pass has been queued on `stream`, `hook` is
 * called on the host to all-gather each rank's per-occurrence item-gradient rows
 * (dx_local[item_off .. item_off + 2*batch) -> dx_items_all[world][2*batch][d]); the item
 * update then runs over the GLOBAL occurrence list (key_items/val_items: every batch's
 * world*2*B_b occurrences as (batch << 32 | item row), sorted by the caller or -- lists_unsorted -- by the library,
 * value = slot in dx_items_all),
 * so all replicas apply the identical summed update.  loss_scale = B_local / B_global.
 * hook == NULL: the library's own RCCL communicator (sml_comm_init) does the all-gather. */
typedef int (*sml_mf_hook)(void* user, int64_t batch_index);
typedef struct {
    int world;
    const uint64_t* key_items;
    const uint32_t* val_items;
    float* dx_local;        /* caller-owned scratch, >= (3*batch + 64 + slot_stride) * d floats */
    float* dx_items_all;    /* caller-owned, world * slot_stride * d floats */
    sml_mf_hook hook;
    void* hook_user;
    float loss_scale;       /* used when no batch plan gives per-batch scales */
    int64_t slot_stride;    /* rows every rank contributes per batch to dx_items_all (0: 2*batch) */
    const int64_t* item_off;/* host [n_batches+1]: batch b's global item occurrences are key/val_items[item_off[b] .. item_off[b+1])
                               (NULL: world*2*batch per full batch, the uniform layout) */
    int64_t push_rows;      /* peer path only (sml_peer_attach, hook == NULL): rows a rank actually pushes per batch, the same
                               on every rank, 2*batch <= push_rows <= slot_stride (0: slot_stride).  There slot_stride must
                               equal the inboxes' rows_cap: the gathered buffer IS one parity of this rank's row slots, and
                               dx_items_all is not used. */
    int lists_unsorted;     /* 1: key_items / val_items are batch-major (batch b's occurrences at item_off[b] .. / the uniform layout) but
                               NOT sorted inside a batch: the library builds every batch's run list itself (index_prep.hip: stable, i.e.
                               a row's occurrences keep the caller's order -- the same on every rank).  0: sorted (stably) by the caller. */
} sml_mf_exchange;

/* Batches of unequal size.  A global batch split over ranks by user owner leaves every rank a DIFFERENT number of
 * triples per batch (possibly none).  With a plan, `triples` holds the rank's batches back to back, batch b =
 * rows [batch_off[b], batch_off[b+1]), each at most `batch` long (the size the scratch was made for), and
 * loss_scale[b] multiplies batch b's loss and gradients (B_local/B_global for the mean-type BCE, 1 for the
 * sum-type BPR kinds).  An empty batch still takes part in the exchange and in the optimiser step.
 * NULL plan: consecutive batches of `batch` triples, the last one ragged. */
typedef struct {
    int64_t n_batches;
    const int64_t* batch_off;       /* host [n_batches + 1] */
    const int32_t* batch_off_dev;   /* the same offsets as int32 on the device (the index preparation reads them there) */
    const float* loss_scale;        /* host [n_batches], or NULL for the call's scalar */
} sml_batch_plan;

/* One epoch over n pre-drawn triples (u,i,j) int64 [n,3], in batches of `batch`:
 * 6 gathers -> run_MF -> + l2*0.5*sum(x_hat^2) -> backward to the W_hat rows ->
 * Adam(lr, betas 0.9/0.999, eps 1e-8, wd 0) with the DENSE semantics of
 * torch.optim.Adam reproduced lazily per row (rows not in a batch still take their
 * zero-gradient steps; they are replayed when the row is next touched or flushed).
 * *step is the optimiser's global step counter (in: steps done; out: + n_batches).
 * batch_loss[n_batches] (device) receives each batch's loss as the reference's
 * loss_batch (model/transfer.py:488).  Asynchronous. */
int sml_mf_stage_epoch(sml_ctx* ctx, const float* theta, const sml_mf_tables* t,
                       const int64_t* triples, int64_t n, int batch, float lr, float l2,
                       int loss_kind, int64_t* step, float* batch_loss, const sml_mf_exchange* xchg,
                       const sml_batch_plan* plan, void* stream);
/* Replay every pending zero-gradient Adam step so the tables can be read out
 * (before save_MF_weight / updata / evaluation; model/transfer.py:518, 777, 832). */
int sml_mf_adam_flush(sml_ctx* ctx, const sml_mf_tables* t, float lr, int64_t step, void* stream);

/* ---- a9: TR stage (meta_train.transfer_train_onestage inner loop, model/transfer.py:701-728) */
typedef struct {
    const float* last_user; const float* last_item;   /* W_{t-1} */
    const float* hat_user;  const float* hat_item;    /* user_weight_hat / item_weight_hat */
    int64_t n_user, n_item;
} sml_tr_tables;

/* One epoch: per batch, run_MF(norm=False) forward, backward to theta, then
 * Adam(lr, weight_decay added to the gradient) on theta (and m, v: flat buffers of
 * the same layout).  theta_grad (2*net_size floats) receives each batch's flat
 * gradient; NULL: internal scratch on the exchange paths, and NOT WRITTEN AT ALL by the
 * single-GPU step (Adam is fused into the weight-gradient kernel there and nothing
 * reads the gradient).  With grad_hook == NULL the whole
 * epoch is queued asynchronously.  A non-NULL grad_hook is called on the host after
 * each batch's backward has been queued on `stream` and before its Adam step is
 * queued, with the flat theta-gradient buffer: the multi-GPU path all-reduces it
 * there (on the same stream).  loss_scale multiplies loss and gradients
 * (B_local / B_global when a global batch is split over ranks; 1 otherwise). */
typedef int (*sml_grad_hook)(void* user, float* grad, int64_t n_floats, int64_t batch_index);
int sml_tr_stage_epoch(sml_ctx* ctx, float* theta, float* adam_m, float* adam_v, float* theta_grad,
                       const sml_tr_tables* t, const int64_t* triples, int64_t n, int batch,
                       float lr, float weight_decay, int loss_kind, float loss_scale,
                       int64_t* step, float* batch_loss, sml_grad_hook grad_hook, void* hook_user,
                       const sml_batch_plan* plan, void* stream);

/* ---- a7 with gradients: ConvTransfer_com.run_MF for a caller that backpropagates it itself ------------------------
 * (reference model/conv_transfer.py:113-135 as called from model/transfer.py:476-502 and :714-723: the caller does
 * zero_grad -> run_MF -> loss.backward() -> optimizer.step()).  One batch of B triples given as ROW BLOCKS, not tables:
 * user_last / user_hat [B,d]; item_last / item_hat [2B,d] = the positives' rows then the negatives' rows.  Writes the
 * loss (device scalar), d loss / d user_hat [B,d], d loss / d item_hat [2B,d] (gradient reaches x_hat through the
 * stack's second row only: x_com is built from a detached x_hat, :93-99) and the flat theta gradient
 * (2 * sml_theta_net_size floats, theta layout); any of the three gradient outputs may be NULL.  No optimiser step, no
 * l2 term: the caller adds what its loop adds.  B <= the context's max_batch.  Asynchronous. */
int sml_run_mf_grad(sml_ctx* ctx, const float* theta, const float* user_last, const float* user_hat,
                    const float* item_last, const float* item_hat, int B, int loss_kind, float* loss,
                    float* d_user_hat, float* d_item_hat, float* theta_grad, void* stream);

/* ---- a3: bare fused embed + loss + SGD write-back ------------------------------ */
/* gather 3 rows, 2 dot products, BCE (model/baseline.py:188-201) or BPR
 * (model/MF.py:139-144 without biases) loss, gradients, and synchronous minibatch
 * SGD: W -= lr * dL/dW with duplicate rows' gradients summed before the write.
 * dtype_bytes: 4 (fp32 tables) or 2 (fp16 tables, fp32 arithmetic).
 * batch_loss[n_batches] as above.
 * prepared_slot: -1 builds the epoch's index lists (per batch: a stable partition of the occurrences
 * by low row bits, every bucket sorted in LDS -- packed 4-byte entries when n_user / n_item and the
 * batch size allow -- unique marks, compacted run records of the duplicated rows; index_prep.hip)
 * inline on `stream`; 0/1 uses the lists a previous
 * sml_embed_loss_sgd_prepare call built for the SAME triples/n/batch -- so a caller can
 * prepare epoch e+1 on a side stream while epoch e runs.  No host wait either way: the call QUERIES whether
 * the lists it uses are complete; if they are (prepared ahead) and hold no hot run, the hot-row kernels are
 * skipped; if they are still in flight the hot-row kernels are launched and find their lists empty on the device.
 * (The caller orders `stream` behind the preparation stream, as for any two streams.) */
/* Several GPUs (NULL on one): users are row-sharded -- w_user holds this rank's rows, `triples` its users' triples
 * (local user index), every rank brings the SAME n and batch -- and the item table is replicated.  The ranks'
 * global batch b is the union of their local batches b.  Item rows are then never updated in place: every item
 * occurrence emits its gradient row, the ranks all-gather those rows after the gradient pass (dx[B .. B + 2*batch)
 * -> dx_items_all[world][2*batch][d]; `hook` on the host, or the library's own communicator when hook == NULL),
 * and every rank applies the identical update over the job's global item-occurrence list, which the index
 * preparation builds from `items_all` (the item columns of every rank's triples, gathered once per epoch by the
 * caller): replicas stay bit-identical, the step is the synchronous SGD step of the global batch.  BCE's mean is
 * over the global batch: loss_scale = B_local / B_global (1 for BPR). */
typedef struct {
    int world;
    const int64_t* items_all;   /* device [world][n][2]: (positive, negative) of every rank's triples, rank-major */
    float* dx_local;            /* caller-owned, 3*batch * d floats: this rank's per-occurrence gradient rows (what a host-side
                                   hook gathers from) */
    float* dx_items_all;        /* caller-owned, world * 2*batch * d floats */
    sml_mf_hook hook;
    void* hook_user;
    float loss_scale;
} sml_bare_exchange;

int sml_embed_loss_sgd_prepare(sml_ctx* ctx, const int64_t* triples, int64_t n, int batch,
                               int64_t n_user, int64_t n_item, int slot, const sml_bare_exchange* xchg, void* stream);
/* Test hook: copies one of the prepared lists of `slot` to the host (blocking; the caller has synchronised the
 * preparation stream).  which: 0/1 run records of users/items (8 x uint32: row, pos, len, pad, slot[4]), 2/3 first run
 * of each batch, 4/5 runs per batch (-1 per batch when the lists carry nb+1 offsets instead), 6/7 sorted values,
 * 8 unique marks, 9 hot lists, 10 hot counts, 11 (max run length, hot_cap).  Returns the bytes copied (<= bytes), < 0 on error. */
int64_t sml_index_lists_read(sml_ctx* ctx, int slot, int which, void* host, int64_t bytes);
int sml_embed_loss_sgd_epoch(sml_ctx* ctx, void* w_user, void* w_item, int64_t n_user,
                             int64_t n_item, int dtype_bytes, const int64_t* triples, int64_t n,
                             int batch, float lr, float lam_user, float lam_item, int loss_kind,
                             float* batch_loss, int prepared_slot, const sml_bare_exchange* xchg, void* stream);

/* The bare step with the ITEM TABLE SHARDED over the ranks (configs 4 / 5: 10M x 1M and 50M x 5M over 8 GPUs), on the
 * one-shot peer exchange (sml_peer_attach; rows_cap >= 2*batch, head_rows * d <= 2 * sml_theta_net_size(d)).
 * Replicating the items makes the step exchange-bound by a factor of ten (every rank would receive every rank's
 * 2*batch gradient rows); here
 *   - the head rows [0, head_rows) -- the popular items of a Zipf catalogue, whose occurrences would otherwise all
 *     land on one owner -- are replicated (w_item_head): every rank sums its OWN occurrences' gradient rows into a
 *     dense [head_rows, d] partial, the partials are all-reduced one-shot (pushed into every rank's inbox, added in
 *     rank order) and every replica applies the identical update;
 *   - a tail row r >= head_rows lives on rank (r - head_rows) / shard_rows only (item_shard[q]: every rank's shard
 *     as addressable from this device): the gradient pass READS it from its owner over the peer mapping and STORES
 *     the occurrence's gradient row straight into the owner's inbox; the owner adds the rows of all ranks in a fixed
 *     order (job-wide occurrence list built from items_all) and writes its shard -- owner-computes, an all-to-all whose
 *     volume per rank does not grow with the world size.
 * Users are row-sharded as for the sml_bare_exchange path: w_user, triples with local user indices, the same n and batch on
 * every rank).  Per batch two counter rounds order the ranks: "all gradient rows of batch b have landed" before an
 * owner updates, "every owner has updated" before anybody reads rows for batch b + 1.  Exact synchronous SGD of the
 * GLOBAL batch; replicas of the head stay bit-identical.  dx scratch and index lists live in the context. */
typedef struct {
    int world, rank;
    int64_t head_rows, shard_rows;
    void* const* item_shard;    /* host array [world] of device pointers: rank q's tail shard [shard_rows, d] */
    void* w_item_head;          /* [head_rows, d] (NULL when head_rows == 0) */
    const int64_t* items_all;   /* device [world][n][2], as in sml_bare_exchange */
    float loss_scale;
} sml_bare_shard;
int sml_embed_loss_sgd_epoch_sharded(sml_ctx* ctx, void* w_user, int64_t n_user, int64_t n_item, int dtype_bytes,
                                     const int64_t* triples, int64_t n, int batch, float lr, float lam_user, float lam_item,
                                     int loss_kind, float* batch_loss, const sml_bare_shard* sh, void* stream);

/* The same step as the reference's baselines run it (model/baseline.py:188-201 in base_train, :343-361 in
 * run_one_stage2: fine-tune / full-retrain MF): BCE + L2 loss and torch.optim.Adam(lr, wd 0) over the DENSE
 * tables, reproduced lazily per row exactly as in sml_mf_stage_epoch (same sml_mf_tables; last_* are not
 * read).  Flush with sml_mf_adam_flush before the tables are read out.  fp32 tables. */
int sml_embed_loss_adam_epoch(sml_ctx* ctx, const sml_mf_tables* t, const int64_t* triples, int64_t n,
                              int batch, float lr, float lam_user, float lam_item, int loss_kind,
                              int64_t* step, float* batch_loss, void* stream);

/* ---- a2: MFbasemode.forward (model/MF.py:34-43) --------------------------------- */
int sml_mf_forward(sml_ctx* ctx, const float* w_user, const float* w_item, const int64_t* user,
                   const int64_t* item, int64_t n, int norm, float* uemb, float* iemb, float* score,
                   void* stream);

/* ---- a13: evaluation (MFbasemode.test, model/MF.py:45-80) ------------------------ */
/* rows int64 [n, n_cols]: col 0 user, col 1 the positive, cols 2.. negatives.
 * rank[r] = #{candidates scoring strictly above the positive} (= the position
 * torch.topk assigns it for tie-free scores). */
int sml_eval_ranks(sml_ctx* ctx, const float* w_user, const float* w_item, const int64_t* rows,
                   int64_t n, int n_cols, int32_t* rank, void* stream);
/* L2-blocked form for test sets that are evaluated repeatedly (the validation rows of a period):
 * sml_eval_prepare groups each row's candidates by item range ONCE into rows_b int32 [n, n_cols]
 * (half the index bytes of the int64 input; n_item < 2^31) and bucket_off int32 [n, 9];
 * sml_eval_ranks_blocked then yields the same ranks as sml_eval_ranks with every XCD gathering from
 * its own eighth of the item table.  max_workgroups > 0 caps the (persistent) grid: an evaluation
 * queued on a side stream underneath training kernels leaves them most of each CU's wave slots. */
int sml_eval_prepare(sml_ctx* ctx, const int64_t* rows, int64_t n, int n_cols, int64_t n_item,
                     int32_t* rows_b, int32_t* bucket_off, void* stream);
int sml_eval_ranks_blocked(sml_ctx* ctx, const float* w_user, const float* w_item, const int32_t* rows_b,
                           const int32_t* bucket_off, int64_t n, int n_cols, int32_t* rank,
                           int max_workgroups, void* stream);
/* LDS-sliced form (d = 32, n_item <= 2^20, at most 32767 candidates per row, fewer than 2^31 entries in all): the
 * candidates are re-ordered ONCE per test set slice-major (slice = 1,024 consecutive item rows, staged in LDS; inside a
 * slice by mini-block of 64 test rows, then by test row, every (row, slice) unit padded to an even count), and the rank pass
 * reads item rows from LDS instead of gathering them through L1 -- same ranks as sml_eval_ranks (same rounding of every
 * score), bit for bit.
 *   sml_eval_sliced_slices        number of slices, 0 when the shape is outside the range above (use the forms above)
 *   sml_eval_sliced_entries       uint32 words of `entries` (candidates + padding, an upper bound)
 *   sml_eval_sliced_work_ints     int32 words of `work` the preparation needs (contents not needed afterwards)
 *   sml_eval_sliced_scratch_bytes bytes of `scratch` one rank pass needs (contents not needed afterwards; 16-byte aligned)
 *   sml_eval_prepare_sliced       entries uint32 [sml_eval_sliced_entries], seg_off int32 [slices * ceil(n / 64) + 1]
 *   sml_eval_ranks_sliced         rows is the ORIGINAL int64 table (user and positive columns are read from it);
 *                                 max_workgroups > 0: grid size wanted (one workgroup occupies a whole CU's LDS). */
int sml_eval_sliced_slices(sml_ctx* ctx, int64_t n, int n_cols, int64_t n_item);
int64_t sml_eval_sliced_entries(sml_ctx* ctx, int64_t n, int n_cols, int64_t n_item);
int64_t sml_eval_sliced_work_ints(sml_ctx* ctx, int64_t n, int n_cols, int64_t n_item);
int64_t sml_eval_sliced_scratch_bytes(sml_ctx* ctx, int64_t n, int n_cols, int64_t n_item);
int sml_eval_prepare_sliced(sml_ctx* ctx, const int64_t* rows, int64_t n, int n_cols, int64_t n_item,
                            uint32_t* entries, int32_t* seg_off, int32_t* work, void* stream);
int sml_eval_ranks_sliced(sml_ctx* ctx, const float* w_user, const float* w_item, const int64_t* rows,
                          const uint32_t* entries, const int32_t* seg_off, int64_t n, int n_cols, int64_t n_item,
                          void* scratch, int32_t* rank, int max_workgroups, void* stream);
/* ---- a11: save_MF_weight (model/transfer.py:911-943) and evaluation snapshots ------- */
/* n <= 4 device-to-device copies (dst[q] <- src[q], bytes[q]; all multiples of 16) in one launch. */
int sml_copy_tables(int n, void* const* dst, const void* const* src, const int64_t* bytes, void* stream);

/* A HIP stream restricted to the compute units [cu_lo, cu_hi) of the device's CU mask (on MI355X mask bit i is a
 * CU of XCD i % 8, so a contiguous range takes the same number of CUs from every XCD: each range keeps all eight
 * L2s and an eighth of its CUs on each).  Evaluations queued on such a stream run beside the training kernels on
 * CUs of their own instead of sharing SIMDs and L2 ports with them.  The caller owns the stream. */
int sml_stream_create_cu_range(void** stream, int device, int cu_lo, int cu_hi);
int sml_stream_destroy(void* stream);
/* Orders `waiter` after everything queued on `signaler` so far, through a device-scope event (no system fence). */
int sml_stream_wait_stream(void* waiter, void* signaler);
/* The same ordering WITHOUT a cross-queue barrier packet: sml_flag_set (a one-thread kernel on the signalling stream,
 * after the work to be waited for) stores `value` into the device word flag[0]; sml_flag_wait (a kernel on the waiting
 * stream, before the dependent work) polls until flag[0] >= value.  `flag` points at TWO int32 words, both starting at
 * 0: flag[0] the sequence word (values must grow monotonically), flag[1] a counter of waiters that were not released
 * within timeout_s and gave up -- their dependent work then ran unordered.  The time-out is a hang guard (make it
 * long): it never touches the sequence word, so later waiters stay ordered; the host reads flag[1] when it collects
 * (or drops) results and treats a non-zero count as an error. */
int sml_flag_set(int32_t* flag, int value, void* stream);
int sml_flag_wait(int32_t* flag, int value, double timeout_s, void* stream);
/* hits = #{rank < topk}, ndcg = sum 1/log2(rank+2) over hits; out[0]=hits, out[1]=ndcg (device). */
int sml_eval_metrics(sml_ctx* ctx, const int32_t* rank, int64_t n, int topk, float* out, void* stream);

/* ---- measurement ------------------------------------------------------------------ */
/* Optional HIP-event timing of every kernel launch on the caller's stream, by kernel class
 * (bench.py's roofline leg).  Off by default; when on, each launch is bracketed by two
 * event records.  sml_prof_get synchronises on the last recorded event. */
int sml_prof_enable(sml_ctx* ctx, int on);
/* In-kernel timeline of the training kernels (measurement builds only: -DSML_TIMELINE, see tools/timeline_probe.py;
 * returns -1 otherwise).  buf: device int64, [0] = record counter (zero it), 16-stamp records from [16]; NULL: off. */
int sml_debug_timeline(long long* buf);
int sml_prof_reset(sml_ctx* ctx);
/* Average reading (microseconds) of n EMPTY event pairs on `stream`: the cost of the bracket itself, which a caller
 * subtracts from per-launch averages of short kernels.  Synchronous. */
int sml_prof_pair_overhead(sml_ctx* ctx, int n, void* stream, double* avg_us);
int sml_prof_classes(void);
const char* sml_prof_name(int cls);
int sml_prof_get(sml_ctx* ctx, int cls, int64_t* count, double* total_ms);

/* ---- device batch supply (fast mode) --------------------------------------------------- */
/* offlineDataset_withsample's negatives (reference data/dataset.py:63-71) drawn ON THE DEVICE: for every
 * element e, candidates uniform over item_all[0..pop) are redrawn while they are one of users[e]'s own
 * items (CSR: user_ptr [n_users+1] into ascending user_items; all device pointers).  Same distribution as
 * the reference's loop, NOT its numpy random stream: the driver uses it only when asked
 * (--device_batches); the stream-exact host path (sml_host_resolve_negatives_csr) is the default.
 * Counter-based generator keyed by (seed, e): reproducible for a given seed.  *failed (device int32) counts
 * elements left at -1 after 4096 rejected draws. */
int sml_sample_negatives(sml_ctx* ctx, const int64_t* users, int64_t n, const int64_t* item_all, int64_t pop,
                         const int64_t* user_ptr, int64_t n_users, const int64_t* user_items, uint64_t seed,
                         int64_t* negs, int32_t* failed, void* stream);

/* One shuffled pass over a period's rows assembled on the device -- the device form of a DataLoader(shuffle=True) pass over
 * trainDataset_withPreSample (reference data/dataset2.py:172-201: rows [user, item, c2, c3, ...], ONE pre-sampled column per
 * pass is the negative) and of the (user, item) half of offlineDataset_withsample (data/dataset.py:41-71).  ui: device int64
 * [n,2]; mat (or NULL): device integer matrix (elem_bytes 4 or 8) with row_stride elements per row -- out3[e] = (ui[r][0],
 * ui[r][1], mat[r*row_stride + col]) with r = perm(e); NULL leaves column 2 alone (sml_sample_negatives fills it).  perm is a
 * counter-based permutation of [0, n) keyed by `seed` (a Feistel network, cycle-walked): the same distribution as the
 * reference's shuffle, NOT torch's randperm stream -- used only under --device_batches; every rank of a job derives the
 * same epoch from the same seed without communication.  Asynchronous. */
int sml_device_epoch(sml_ctx* ctx, const int64_t* ui, const void* mat, int elem_bytes, int64_t row_stride, int64_t col, int64_t n,
                     uint64_t seed, int64_t* out3, void* stream);

/* ---- host helper: batch supply ------------------------------------------------------- */
/* Sequential rejection sampling of offlineDataset_withsample.__getitem__ (reference
 * data/dataset.py:63-71) over a pre-drawn candidate stream, on the HOST (no GPU involved):
 * element e takes candidates cand[ptr], cand[ptr+1], ... until one is not an item of user[e]
 * ((user,item) pairs given sorted as user*stride+item), exactly as the per-item loop consumes the
 * random stream.  *consumed = candidates used, *resolved = elements that got their negative; when
 * the stream runs out first (*resolved < n, *consumed == m) the caller continues from element
 * *resolved with the next draws.  Drawing exactly as many candidates as there are unresolved
 * elements per call therefore consumes the generator exactly as the per-item loop does. */
int sml_host_resolve_negatives(const int64_t* users, int64_t n, const int64_t* cand, int64_t m,
                               const int64_t* pairs_sorted, int64_t n_pairs, int64_t stride,
                               int64_t* negs, int64_t* consumed, int64_t* resolved);

/* The same walk with the users' items in CSR form (user_ptr int64 [n_users + 1] into user_items, each user's
 * items ascending): two memory touches per candidate instead of a 17-step bisection of the pair list. */
/* Host-side gathers of the same batch supply: out3 is the epoch's int64 [n,3] triple array.  gather_pairs: columns 0, 1 <-
 * ui[order[e]] (ui = contiguous int64 [n_rows,2]); gather_column: column out_col <- element `col` of row order[e] of a
 * row-major integer matrix (elem_bytes 4 or 8).  data/dataset2.py:172-201 (pre-sampled column), data/dataset.py:41-71. */
int sml_host_gather_pairs(const int64_t* ui, int64_t n_rows, const int64_t* order, int64_t n, int64_t* out3);
int sml_host_gather_column(const void* mat, int64_t n_rows, int64_t row_stride_bytes, int elem_bytes, int64_t col,
                           const int64_t* order, int64_t n, int64_t* out3, int out_col);
int sml_host_resolve_negatives_csr(const int64_t* users, int64_t n, const int64_t* cand, int64_t m,
                                   const int64_t* user_ptr, int64_t n_users, const int64_t* user_items,
                                   int64_t* negs, int64_t* consumed, int64_t* resolved);

/* ---- self test ------------------------------------------------------------------ */
/* Checks the MFMA operand/accumulator lane maps this library assumes against a
 * scalar loop on the device.  Synchronous.  Returns 0 if they hold. */
int sml_selftest(int device);

#ifdef __cplusplus
}
#endif
#endif /* SML_HIP_H */
