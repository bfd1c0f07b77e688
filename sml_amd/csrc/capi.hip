// C ABI of libsml_hip.so (see include/sml_hip.h): context, scratch, per-epoch launch loops.
#include <hip/hip_runtime.h>
#ifdef SML_TEST_PREP_REFERENCE     // (test build only, tests/build_reference.py: the library-sort reference of the index preparation)
#include <hipcub/hipcub.hpp>
#endif
#include <rccl/rccl.h>      // types only: the functions are bound at run time from the loaded librccl
#include <dlfcn.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/sml_hip.h"
#include "sml_kernels.h"

namespace {

thread_local std::string g_err;

int fail(int code, const char* what, const char* detail) {
    g_err = std::string(what) + ": " + (detail ? detail : "");
    return code;
}
#define HIPCHK(expr)                                                                  \
    do {                                                                              \
        hipError_t _e = (expr);                                                       \
        if (_e != hipSuccess) return fail(SML_EHIP, #expr, hipGetErrorString(_e));    \
    } while (0)

struct DevGuard {
    int prev = -1;
    bool changed = false;
    explicit DevGuard(int dev) {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) changed = (hipSetDevice(dev) == hipSuccess);
    }
    ~DevGuard() { if (changed) (void)hipSetDevice(prev); }
};

// Buffers that were outgrown are NOT freed on the spot: hipFree waits for the whole device, and on a device that
// another rank's kernel is polling on (peer exchange: a consumer spinning for THIS host thread's next launch) that wait
// never ends before the poller's time-out.  They are parked here and freed when a context is destroyed.
// The same holds for destroying a whole context (Python's garbage collector may finalise an old engine on ANY thread,
// e.g. inside a rank thread whose peer is polling -- found with a stack dump of a stalled thread-rank test): everything a
// context owns is parked, and the yard is emptied only while no context of the process has peer mappings attached.
struct Graveyard {
    std::vector<void*> dead, dead_host;
    std::vector<sml_ctx*> zombies;   // whole contexts whose destruction (RCCL communicator, events, frees) has to wait
    std::mutex mu;
    bool defer(sml_ctx* c) { std::lock_guard<std::mutex> g(mu); if (peers_live <= 0) return false; zombies.push_back(c); return true; }
    std::vector<sml_ctx*> take_zombies() { std::lock_guard<std::mutex> g(mu); std::vector<sml_ctx*> z; if (peers_live <= 0) z.swap(zombies); return z; }
    int peers_live = 0;          // contexts with peer mappings attached (their consumers may be polling on the device)
    void park(void* p) { if (p) { std::lock_guard<std::mutex> g(mu); dead.push_back(p); } }
    void park_host(void* p) { if (p) { std::lock_guard<std::mutex> g(mu); dead_host.push_back(p); } }
    void peers(int delta) { std::lock_guard<std::mutex> g(mu); peers_live += delta; }
    void reap() {
        std::vector<void*> d, h;
        { std::lock_guard<std::mutex> g(mu); if (peers_live > 0) return; d.swap(dead); h.swap(dead_host); }
        for (void* p : d) (void)hipFree(p);
        for (void* p : h) (void)hipHostFree(p);
    }
} g_graveyard;

template <typename T>
struct Buf {
    T* p = nullptr;
    size_t cap = 0;   // elements
    hipError_t ensure(size_t need) {
        if (need <= cap) return hipSuccess;
        if (p) { g_graveyard.park(p); p = nullptr; cap = 0; }
        size_t want = need + need / 8;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), want * sizeof(T));
        if (e != hipSuccess) return e;
        cap = want;
        return hipSuccess;
    }
    void release() { g_graveyard.park(p); p = nullptr; cap = 0; }
};

bool d_ok(int d) { return d == 32 || d == 64 || d == 128; }

// RCCL entry points, resolved from the librccl.so the process already uses (torch's)
struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok() const { return handle != nullptr; }
} g_rccl;
#define NCCLCHK(expr)                                                                                       \
    do {                                                                                                    \
        ncclResult_t _r = (expr);                                                                           \
        if (_r != ncclSuccess) return fail(SML_EHIP, #expr, g_rccl.GetErrorString ? g_rccl.GetErrorString(_r) : "rccl error"); \
    } while (0)

// ---- optional per-kernel-class timing with HIP events on the caller's stream -------------
enum ProfClass { PC_FWD = 0, PC_BWD, PC_WGRAD, PC_THETA_ADAM, PC_PAIR_LOSS /* hot-row kernels */, PC_SEG_ADAM, PC_SEG_SGD, PC_BARE_GRAD,
                 PC_EVAL_RANKS, PC_FLUSH, PC_PACK, PC_SORT, PC_MISC, PC_FWD_SIDE, PC_COUNT };
const char* const kProfNames[PC_COUNT] = {"k_transfer_fwd", "k_transfer_bwd", "k_transfer_wgrad", "k_theta_adam",
                                          "k_hot_rows", "k_seg_update_adam", "k_seg_update_sgd", "k_bare_grad",
                                          "k_eval_ranks", "k_adam_flush", "k_theta_pack", "sort_epoch", "misc",
                                          "k_side_transfer_fwd"};
struct Prof {
    bool on = false;
    std::vector<hipEvent_t> ev;     // pairs
    std::vector<int> cls;
    size_t used = 0;                // pairs in flight
    double total_ms[PC_COUNT] = {0};
    int64_t count[PC_COUNT] = {0};
    static constexpr size_t kPairs = 4096;
    int64_t lost = 0;               // pairs whose elapsed time could not be read (reported by sml_prof_get's caller as missing)
    void drain() {
        if (!used) return;
        // the ring is shared by every stream the context is used on (training stream, side-stream evaluations):
        // wait for EACH pair's end event, not just the newest one
        for (size_t i = 0; i < used; ++i) {
            float ms = 0.f;
            (void)hipEventSynchronize(ev[2 * i + 1]);
            if (hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]) == hipSuccess) { total_ms[cls[i]] += ms; count[cls[i]]++; }
            else ++lost;
        }
        used = 0;
    }
    void begin(int c, hipStream_t st) {
        if (!on) return;
        if (ev.empty()) { ev.resize(2 * kPairs); cls.resize(kPairs); for (auto& e : ev) (void)hipEventCreate(&e); }
        if (used == kPairs) drain();
        cls[used] = c;
        (void)hipEventRecord(ev[2 * used], st);
    }
    void end(hipStream_t st) {
        if (!on) return;
        (void)hipEventRecord(ev[2 * used + 1], st);
        ++used;
    }
    void release() { drain(); for (auto& e : ev) (void)hipEventDestroy(e); ev.clear(); }
};

}  // namespace

// sorted occurrence lists and run records of one epoch (two sets: one may be prepared on a side
// stream while the other is in use).  Keys are uint32 ((batch << row_bits) | row) when that fits,
// else uint64 ((batch << 32) | row); the key buffers are sized for the wider form.
struct IndexSet {
    Buf<uint64_t> key_u, key_u2, key_i, key_i2;
    Buf<uint32_t> val_u, val_u2, val_i, val_i2;
    Buf<SmlRun> rec_u, rec_i;          // one record per sorted position
    Buf<SmlRun> runs_u, runs_i;        // compacted: duplicated runs only (bare step)
    Buf<uint32_t> heads_u, heads_i;    // sorted positions of the duplicated-run heads
    Buf<uint32_t> hot_list;            // [nb][hot_cap][3] hot runs of every batch (bare step, large batches)
    Buf<int> hot_count;                // [nb]
    int hot_cap = 0;                   // 0: the hot-row path is off for this epoch's batch size
    Buf<uint8_t> uniq;
    Buf<uint32_t> slot_info;           // records mode, by hand: per slot, "once" or the position of its run's record (SmlFusedUpdate)
    int64_t slot_stride = 0;           // ... slots per batch; 0: this set has none
    int64_t dense_stride = 0;          // distinct-row records per batch (whole tiles of both lists)
    // MF stage, distinct-row form (SmlDense): records by scratch row, distinct rows per (batch, table), per-tile headers / entry blocks / spill
    Buf<SmlRun> dense_rec; Buf<int> dense_n; Buf<SmlTileHdr> tile_hdr; Buf<uint2> tile_ent, tile_spill; Buf<int> spill_cnt;
    bool dense = false; int tiles_cap = 0;
    Buf<int> off_u, off_i, n_sel;
    // index_prep.hip (the by-hand preparation): tile histograms, bucket offsets / counts, run counts, oversized buckets
    Buf<uint32_t> hist_u, hist_i, bko_u, bko_i, bkc_u, bkc_i, large, medium;
    Buf<SmlRun> stage_u, stage_i;      // per-bucket stretches of run records before the compaction
    Buf<int> cnt_u, cnt_i;
    Buf<int> rank_viol;      // [0]: lanes the start-up probe found served out of lane order by a returning LDS atomic (wave_rank)
    bool rank_probed = false;
    bool by_hand = false;              // the lists were built by index_prep.hip: a batch's runs are off[b] .. off[b] + cnt[b]
    Buf<char> cub_tmp;
    int key_bytes = 8, row_bits_u = 32, row_bits_i = 32;
    int* viol_host = nullptr;          // pinned: entries emit_bucket found out of (row, value) order, ever (SmlPrepArgs.order_viol)
    hipEvent_t viol_ready = nullptr;
    int* max_len_host = nullptr;       // pinned: longest duplicated run of the prepared epoch (0 if none exceeds SML_HOT)
    int* lists_host = nullptr;         // pinned, compact lists by hand: [off_u (nb+1)][off_i (nb+1)][cnt_u (nb*STRIDE)][cnt_i (nb*STRIDE)] -- the
    int64_t lists_host_nb = 0, lists_nb = 0;   // (capacity; batches of the prepared epoch, 0: not read back) batches' places in the run lists, read back with max_len (same event): the run kernel then needs no offset / count loads
    hipEvent_t ready = nullptr;        // recorded after the copy into max_len_host
    int64_t n = -1; int batch = 0; const void* triples = nullptr; int world = 1;   // what was prepared here
    void release() {
        key_u.release(); key_u2.release(); key_i.release(); key_i2.release();
        val_u.release(); val_u2.release(); val_i.release(); val_i2.release();
        rec_u.release(); rec_i.release(); runs_u.release(); runs_i.release();
        hist_u.release(); hist_i.release(); bko_u.release(); bko_i.release(); bkc_u.release(); bkc_i.release(); large.release(); medium.release();
        cnt_u.release(); cnt_i.release(); stage_u.release(); stage_i.release(); rank_viol.release();
        dense_rec.release(); dense_n.release(); tile_hdr.release(); tile_ent.release(); tile_spill.release(); spill_cnt.release();
        uniq.release(); slot_info.release(); heads_u.release(); heads_i.release(); hot_list.release(); hot_count.release(); off_u.release(); off_i.release(); n_sel.release();
        cub_tmp.release();
        if (max_len_host) { g_graveyard.park_host(max_len_host); max_len_host = nullptr; }
        if (viol_host) { g_graveyard.park_host(viol_host); viol_host = nullptr; }
        if (viol_ready) { (void)hipEventDestroy(viol_ready); viol_ready = nullptr; }
        if (lists_host) { g_graveyard.park_host(lists_host); lists_host = nullptr; lists_host_nb = 0; }
        if (ready) { (void)hipEventDestroy(ready); ready = nullptr; }
    }
};

struct SchedRetired { SmlSched* dev; SmlSched* host; hipEvent_t done; };

struct sml_ctx {
    int device = 0, d = 32, max_batch = 0;
    int variant = 0;         // 0: ConvTransfer_com, 1: ConvTransfer (sml_ctx_set_variant)
    bool side = false;       // an evaluation-stream context (sml_ctx_set_variant, bit 1): its table-sized forwards run under their own name / class
    float clip_max_norm = 0.0f;   // > 0: the TR stage clips the theta gradient's norm (sml_ctx_set_grad_clip)
    float adaptive_beta = 0.0f;   // > 0: the MF stage adds the reference's --need_adaptive user-norm term (sml_ctx_set_adaptive)
    Buf<float> clip_sumsq;
    IndexSet ix[2];
    // transfer-net workspaces, 3*B slots each
    Buf<float> out, dout, dx, xin, z1, a1, a2, dz1, mrep, vrep;
    Buf<float> pk, grad, convg, loss_part;
    Buf<float> pkx;          // bf16x3 operand images of the table-sized forward (both nets)
    Buf<float> cstate;       // [2 parities][2 nets][3][SML_CG]: the conv parameters' working copy of a TR epoch (deferred conv step)
    int pk_set = 0;          // which of the two operand-image sets is current
    Buf<int> arrive;         // k_tr_wgrad2's tail-workgroup arrival counter (0 between launches)
    Buf<int> run_arrive;     // MF stage, fused row update: per-run arrival counters of a batch (0 between launches)
    // Adam schedule of the MF optimiser
    Buf<SmlSched> sched;
    int sched_len = 0;
    float sched_lr = -1.0f;
    std::vector<SchedRetired> sched_retired;     // replaced schedule tables / their pinned sources, freed once idle
    hipEvent_t sched_ready = nullptr;            // behind the newest table's upload
    hipStream_t sched_stream = nullptr;
    Buf<int32_t> dummy;
    Buf<SmlRun> rec_x;       // run records of the multi-GPU global item list (caller-sorted keys)
    Buf<int32_t> xoff;       // ... and, for lists the library builds itself, the batches' offsets in the occurrence stream
    ncclComm_t comm = nullptr;
    int comm_world = 1, comm_rank = 0;
    // one-shot exchange over peer mappings (sml_peer_attach): world == 0 means detached
    struct Peer {
        int world = 0, rank = 0;
        char* inbox[SML_MAX_PEERS] = {nullptr};
        unsigned long long* flags[SML_MAX_PEERS] = {nullptr};
        int64_t theta_slot = 0;          // floats per (parity, source) theta slot
        int64_t rows_cap = 0;            // rows per (parity, source) row slot
        int64_t tick[3] = {0, 0, 0};     // exchanges done, per kind (0: theta / dense head, 1: rows, 2: "owner has updated")
        unsigned long long expect[3][2] = {{0, 0}, {0, 0}, {0, 0}};   // [kind][parity]: counter value after the last push
        bool done_pending = false;       // sharded bare step: the last batch's "owner done" round has not been waited for yet
        SmlPeerPoll done_poll;
        long long timeout = 0;           // 100 MHz ticks
        int* err = nullptr;              // device: incidents
    } peer;
    Buf<int> hot_first;
    Buf<float> hot_part;
    Buf<float> head_part;    // sharded bare step: this rank's dense [head_rows, d] gradient partial
    Buf<void*> ptr_tab;      // sharded bare step: [0..8) shard pointers, [8..16) inbox bases (device table, per-lane indexed)
    Prof prof;

    void release_all() {
        prof.release();
        out.release(); dout.release(); dx.release(); xin.release(); z1.release(); a1.release(); a2.release(); dz1.release();
        mrep.release(); vrep.release();
        pk.release(); pkx.release(); grad.release(); convg.release(); loss_part.release(); arrive.release(); run_arrive.release(); cstate.release();
        ix[0].release(); ix[1].release();
        sched.release(); dummy.release(); rec_x.release(); xoff.release();
        for (auto& r : sched_retired) { g_graveyard.park(r.dev); g_graveyard.park_host(r.host); (void)hipEventDestroy(r.done); }
        sched_retired.clear();
        if (sched_ready) { (void)hipEventDestroy(sched_ready); sched_ready = nullptr; }
        hot_first.release(); hot_part.release(); head_part.release(); ptr_tab.release();
        if (peer.err) { g_graveyard.park(peer.err); peer.err = nullptr; }
    }
};

namespace {

// step_size / bc2_sqrt exactly as torch.optim.Adam computes them on the host (double), rounded once
SmlSched sched_entry(double lr, int64_t k) {
    SmlSched s;
    if (k <= 0) { s.step_size = 0.f; s.bc2_sqrt = 1.f; return s; }
    const double bc1 = 1.0 - std::pow((double)0.9, (double)k);
    const double bc2 = 1.0 - std::pow((double)0.999, (double)k);
    s.step_size = (float)(lr / bc1);
    s.bc2_sqrt = (float)std::sqrt(bc2);
    return s;
}

// Grow (or rebuild for another lr) the Adam schedule table WITHOUT a host wait: the new table is filled from a pinned
// host buffer by an asynchronous copy on the caller's stream; the table it replaces may still be read by kernels in
// flight, so it is retired with an event behind them and freed by a later call once that event has completed (or with
// the context).  A call on ANOTHER stream before the copy has landed is ordered behind it by an event wait.
void sched_reap(sml_ctx* c, bool all) {
    for (size_t i = 0; i < c->sched_retired.size();) {
        SchedRetired& r = c->sched_retired[i];
        if (all || hipEventQuery(r.done) == hipSuccess) {
            g_graveyard.park(r.dev);
            g_graveyard.park_host(r.host);
            (void)hipEventDestroy(r.done);
            c->sched_retired[i] = c->sched_retired.back();
            c->sched_retired.pop_back();
        } else ++i;
    }
}
int ensure_sched(sml_ctx* c, float lr, int64_t upto, hipStream_t st) {
    if (c->sched_lr == lr && upto < c->sched_len) {
        if (c->sched_ready && st != c->sched_stream && hipEventQuery(c->sched_ready) != hipSuccess)
            HIPCHK(hipStreamWaitEvent(st, c->sched_ready, 0));
        return SML_OK;
    }
    int64_t len = c->sched_len > 0 && c->sched_lr == lr ? c->sched_len : 0;
    while (len <= upto) len = len ? len * 2 : 65536;
    if (len > (int64_t)1 << 30) return fail(SML_EINVAL, "adam schedule", "step counter too large");
    sched_reap(c, false);
    SmlSched* h = nullptr;
    SmlSched* dnew = nullptr;
    // behind the schedule: the closed-form replay tables (sml_dev.h) -- H[len + 1][4], R[SML_RP_N + 1][4] in double, B[SML_RP_N + 1][2] in float
    const size_t bytes = (size_t)len * sizeof(SmlSched) + ((size_t)len + 1) * 4 * sizeof(double) + (size_t)(SML_RP_N + 1) * (4 * sizeof(double) + 2 * sizeof(float));
    HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&h), bytes, hipHostMallocDefault));
    for (int64_t k = 0; k < len; ++k) h[(size_t)k] = sched_entry((double)lr, k);
    {
        double* H = reinterpret_cast<double*>(h + len);
        double* R = H + 4 * ((size_t)len + 1);
        float* B = reinterpret_cast<float*>(R + 4 * (SML_RP_N + 1));
        // the constants as the loop form rounds them: m <- m - (1 - beta1) m in float, v <- v * beta2, eps * bc2 in float
        const double b1 = 1.0 - (double)(1.0f - SML_BETA1), b2 = (double)SML_BETA2, sg = std::sqrt(b2), eps = (double)SML_EPS;
        double rho[4];
        for (int q = 0; q < 4; ++q) rho[q] = b1 / std::pow(sg, 1.0 + q);
        // H_q[k0] = sum_{k >= k0} c_k E_k^q rho_q^(k - k0 + 1), backwards; beyond the table the summands are taken as constant (never reached:
        // every reader stays below `len`, and what the tail contributes to an entry 100 steps below the end is under 1e-5 of it)
        {
            const SmlSched z = h[(size_t)len - 1];
            const double c = (double)z.step_size * (double)z.bc2_sqrt, E = eps * (double)z.bc2_sqrt;
            double Eq = 1.0;
            for (int q = 0; q < 4; ++q) { H[4 * (size_t)len + q] = c * Eq * rho[q] / (1.0 - rho[q]); Eq *= E; }
        }
        for (int64_t k = len - 1; k >= 0; --k) {
            const SmlSched z = h[(size_t)k];
            const double c = k >= 1 ? (double)z.step_size * (double)z.bc2_sqrt : 0.0, E = eps * (double)z.bc2_sqrt;
            double Eq = 1.0;
            for (int q = 0; q < 4; ++q) { H[4 * (size_t)k + q] = rho[q] * (c * Eq + H[4 * ((size_t)k + 1) + q]); Eq *= E; }
        }
        for (int n = 0; n <= SML_RP_N; ++n) {
            for (int q = 0; q < 4; ++q) R[4 * n + q] = std::pow(rho[q], (double)n);
            B[2 * n] = (float)std::pow(b1, (double)n); B[2 * n + 1] = (float)std::pow(b2, (double)n);
        }
    }
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&dnew), bytes);
    if (e == hipSuccess) e = hipMemcpyAsync(dnew, h, bytes, hipMemcpyHostToDevice, st);
    SchedRetired r;
    r.dev = c->sched.p; r.host = h; r.done = nullptr;
    if (e == hipSuccess) e = hipEventCreateWithFlags(&r.done, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventRecord(r.done, st);          // behind the copy and behind every earlier reader on `st`
    if (e != hipSuccess) {
        if (dnew) (void)hipFree(dnew);
        (void)hipHostFree(h);
        if (r.done) (void)hipEventDestroy(r.done);
        return fail(SML_EHIP, "adam schedule", hipGetErrorString(e));
    }
    c->sched_retired.push_back(r);
    c->sched.p = dnew; c->sched.cap = (size_t)len;
    c->sched_len = (int)len;
    c->sched_lr = lr;
    if (!c->sched_ready) HIPCHK(hipEventCreateWithFlags(&c->sched_ready, hipEventDisableTiming));
    HIPCHK(hipEventRecord(c->sched_ready, st));
    c->sched_stream = st;
    return SML_OK;
}

int tiles_of(int rows) { return (rows + SML_R - 1) / SML_R; }          // 32-row padding units
int wg_tiles(int rows, int mt) { const int r = SML_TM * mt; return (rows + r - 1) / r; }   // workgroups

// Geometry policy of the training kernels.  A batch whose row tiles cannot fill the 256 CUs is
// latency-bound on the per-tile chain: split the hidden dimension of the forward over 4 (or 2)
// workgroups per row tile and the backward over d/16 coordinate slices.  Once the row tiles alone
// fill the chip the splits only repeat the gather/prologue work, so large batches keep one
// workgroup per tile.  (SML_FWD_NS / SML_BWD_SPLIT override the policy for measurements.)
int env_int(const char* name, int dflt) { const char* v = getenv(name); return v && *v ? atoi(v) : dflt; }
// closed-form replay (sml_dev.h): a launch that replays rows up to step `to` uses the tables behind the schedule when the
// bias-correction term has flattened (to >= SML_RP_K0; SML_REPLAY_K0 overrides for tests) -- SML_REPLAY_CLOSED=0: the loop everywhere
int replay_len(const sml_ctx* c, int64_t to) {
    const int on = env_int("SML_REPLAY_CLOSED", 1), k0 = env_int("SML_REPLAY_K0", SML_RP_K0);
    return (on != 0 && to >= k0 && to - SML_RP_N >= 1 && to + 1 < c->sched_len) ? c->sched_len : 0;
}
int fwd_split(int row_tiles) {
    const int forced = env_int("SML_FWD_NS", 0);
    if (forced == 1 || forced == 2 || forced == 4) return forced;
    return row_tiles <= 64 ? 4 : row_tiles <= 128 ? 2 : 1;
}
int bwd_split(int row_tiles) {
    const int forced = env_int("SML_BWD_SPLIT", -1);
    if (forced == 0 || forced == 1) return forced;
    return row_tiles <= 128 ? 1 : 0;
}

// slot layout of a batch: users at [0, B), items at [ioff, ioff + 2B), ioff = B rounded up to a
// tile; both runs padded to whole tiles so the kernels store tile rows unconditionally
int ensure_transfer_ws(sml_ctx* c, int B, bool tr_stage, hipStream_t st) {
    const size_t slots = (size_t)SML_R * (tiles_of(B) + tiles_of(2 * B)), d = (size_t)c->d;
    HIPCHK(c->out.ensure(slots * d * SML_FWD_NS));     // the training forward writes SML_FWD_NS partial planes
    HIPCHK(c->dout.ensure(slots * d));
    HIPCHK(c->xin.ensure(slots * 3 * d));
    HIPCHK(c->z1.ensure(slots * SML_HID));
    if (tr_stage) {
        HIPCHK(c->a1.ensure(slots * SML_C2 * d));
        HIPCHK(c->a2.ensure(slots * SML_HID));
        HIPCHK(c->dz1.ensure(slots * SML_HID));
        HIPCHK(c->convg.ensure((slots / SML_TM + 4) * (d / 16) * SML_CG));   // one partial per backward workgroup
        HIPCHK(c->grad.ensure((size_t)2 * sml_net_size(c->d)));
        // (on the caller's stream, never the null stream: a null-stream operation orders itself behind EVERY blocking
        // stream of the process -- with another rank's kernel of the same process polling for this rank's next launch,
        // that is a dead-lock until the poll's time-out: seen in the thread-rank tests)
        if (!c->arrive.p) { HIPCHK(c->arrive.ensure(4)); HIPCHK(hipMemsetAsync(c->arrive.p, 0, 4 * sizeof(int), st)); }
    } else {
        HIPCHK(c->dx.ensure(slots * d));
        HIPCHK(c->mrep.ensure(slots * d)); HIPCHK(c->vrep.ensure(slots * d));
    }
    return SML_OK;
}

// TWO sets of MFMA operand images (each: user net, item net).  The restructured TR step's merged launch rewrites the
// images (fused Adam of the weight-gradient tiles) WHILE its tail workgroups still read W1's image for dA1: the new
// images go to the other set, and the sets swap after the launch.  Everybody else reads pk_cur().
int ensure_pk(sml_ctx* c) {
    HIPCHK(c->pk.ensure((size_t)4 * sml_pk_size(c->d)));
    return SML_OK;
}
float* pk_cur(const sml_ctx* c) { return c->pk.p + (size_t)c->pk_set * 2 * sml_pk_size(c->d); }
float* pk_other(const sml_ctx* c) { return c->pk.p + (size_t)(1 - c->pk_set) * 2 * sml_pk_size(c->d); }

int ceil_log2(int64_t x) { int b = 0; while (((int64_t)1 << b) < x) ++b; return b; }

// ---- peer regions: inbox = [theta: 2 parities][world][theta_slot floats] then [rows: 2][world][rows_cap][d] floats;
// flags = uint64 [kind 0: theta, 1: rows][parity][world]
int64_t peer_theta_slot(int d) { return ((int64_t)2 * sml_net_size(d) + 63) / 64 * 64; }
float* peer_theta_at(const sml_ctx* c, int owner, int parity, int src) {
    return reinterpret_cast<float*>(c->peer.inbox[owner]) + ((int64_t)parity * c->peer.world + src) * c->peer.theta_slot;
}
float* peer_rows_at(const sml_ctx* c, int owner, int parity, int src) {
    float* base = reinterpret_cast<float*>(c->peer.inbox[owner]) + (int64_t)2 * c->peer.world * c->peer.theta_slot;
    return base + ((int64_t)parity * c->peer.world + src) * c->peer.rows_cap * c->d;
}
unsigned long long* peer_flag_at(const sml_ctx* c, int owner, int kind, int parity, int src) {
    return c->peer.flags[owner] + ((int64_t)kind * 2 + parity) * c->peer.world + src;
}
// descriptors of the next exchange step of `kind` (0 theta, 1 rows) in which every rank's counters grow by `incr`
void peer_step(sml_ctx* c, int kind, int incr, SmlPeerPush* push, SmlPeerPoll* poll) {
    const int W = c->peer.world, me = c->peer.rank;
    const int parity = (int)(c->peer.tick[kind] & 1);
    c->peer.tick[kind] += 1;
    c->peer.expect[kind][parity] += (unsigned long long)incr;
    memset(push, 0, sizeof(*push)); memset(poll, 0, sizeof(*poll));
    push->world = W;
    for (int q = 0; q < W; ++q) {
        push->dst[q] = kind == 0 ? peer_theta_at(c, q, parity, me) : kind == 1 ? peer_rows_at(c, q, parity, me) : nullptr;
        push->flag[q] = peer_flag_at(c, q, kind, parity, me);
    }
    poll->world = W;
    poll->slot0 = kind == 0 ? peer_theta_at(c, me, parity, 0) : kind == 1 ? peer_rows_at(c, me, parity, 0) : nullptr;
    poll->slot_stride = kind == 0 ? c->peer.theta_slot : c->peer.rows_cap * c->d;
    poll->flag0 = peer_flag_at(c, me, kind, parity, 0);
    poll->expect = c->peer.expect[kind][parity];
    poll->timeout = c->peer.timeout;
    poll->err = c->peer.err;
}

// SML_PREP=cub keeps the library sort (A/B tests); default: index_prep.hip
static bool prep_by_hand() {
    const char* e = getenv("SML_PREP");          // (read per call: the A/B test flips it inside one process)
    return !(e && !strcmp(e, "cub"));
}

// The same lists as sort_epoch below, built by index_prep.hip (one GPU's own occurrences; no library code).
// mode 0: this rank's own occurrences.  1: replicated items on several GPUs (bx: the item lists are the JOB's).
// 2 / 3: the item-sharded step's lists A (users + the job's occurrences of the tail rows this rank owns) and B (this
// rank's own occurrences of head rows; no users) -- sh, rows_cap.  See occ_of in index_prep.hip.
// a completed read-back of the sorted-order violation count that is not zero: some earlier list of this index set was wrong
int sort_order_check(IndexSet* c) {
    if (c->viol_host && c->viol_ready && hipEventQuery(c->viol_ready) == hipSuccess && *c->viol_host != 0) {
        char msg[320];
        const int w1 = c->viol_host[1], w2 = c->viol_host[2];
        snprintf(msg, sizeof(msg), "a sorted bucket left out of (row, value) order (%d entries; first: occurrence source %d, table %d, batch %d, %s mode, "
                                   "position %d of a %d-entry bucket): the stable ranking failed on this device (SML_PREP_RANK=ballot selects the ballot ranking)",
                 c->viol_host[0], (w1 >> 28) & 7, (w1 >> 27) & 1, w1 & 0x3ffffff, (w1 >> 26) & 1 ? "records" : "compact / dense", w2 >> 12, w2 & 0xfff);
        return fail(SML_ESTATE, "index preparation", msg);
    }
    return SML_OK;
}
int prep_epoch(IndexSet* c, const int64_t* tri, int64_t n, int batch, int pad_tiles, int64_t n_user, int64_t n_item,
               bool dups, hipStream_t st, const sml_batch_plan* plan, int mode = 0, const sml_bare_exchange* bx = nullptr,
               const sml_bare_shard* sh = nullptr, int64_t rows_cap = 0, bool want_dense = false, const uint32_t* x_vals = nullptr) {
    const int64_t nb = plan ? plan->n_batches : (n + batch - 1) / batch;
    { const int vrc = sort_order_check(c); if (vrc) return vrc; }
    c->by_hand = true; c->slot_stride = 0; c->dense_stride = 0; c->dense = false;
    if (n == 0) { c->n = 0; c->batch = batch; c->triples = tri; return SML_OK; }
    const int W = mode == 1 ? bx->world : (mode == 2 ? sh->world : 1);
    const int nis = mode == 4 ? 1 : 2 * W;                  // item streams per tile (mode 4: one stream of explicit occurrences)
    const int64_t n_items = (int64_t)nis * n;               // item occurrences of the epoch's lists (upper bound in modes 2 / 3)
    if (n_items > 0x7fffffff) return fail(SML_EINVAL, "index preparation", "too many item occurrences in one epoch");
    const bool has_users = mode != 3 && mode != 4, allruns_i = mode != 0;
    SmlPrepArgs a;
    memset(&a, 0, sizeof(a));
    a.tri = tri; a.n = n; a.batch = batch; a.nb = (int)nb; a.tpb = (batch + SML_PREP_TT - 1) / SML_PREP_TT;
    a.boff = plan ? plan->batch_off_dev : nullptr; a.pad_tiles = pad_tiles; a.records = dups ? 0 : 1;
    a.mode = mode; a.has_users = has_users ? 1 : 0; a.nis = nis;
    if (mode == 1) { a.items_all = bx->items_all; a.val_q = (int64_t)2 * batch; }
    if (mode == 2) { a.items_all = sh->items_all; a.val_q = rows_cap; }
    if (mode == 4) { a.x_keys = reinterpret_cast<const uint64_t*>(tri); a.x_vals = x_vals; a.tri = nullptr; }
    if (mode == 2 || mode == 3) { a.head_rows = sh->head_rows; a.shard_rows = sh->shard_rows; a.shard_rank = sh->rank; }
    const int64_t ioff_max = pad_tiles ? ((int64_t)(batch + SML_R - 1) / SML_R) * SML_R : batch;
    // values: users < batch; items < ioff + 2 * batch (own), < world * 2 * batch (mode 1), < world * rows_cap (mode 2), < 3 * batch (mode 3)
    const int64_t max_val_i = mode == 1 ? (int64_t)W * 2 * batch : mode == 2 ? (int64_t)W * rows_cap : mode == 3 ? (int64_t)3 * batch
                              : mode == 4 ? rows_cap : ioff_max + 2 * (int64_t)batch;        // (mode 4: rows_cap carries the largest value + 1)
    const int64_t rows_i = mode == 2 ? sh->shard_rows : mode == 3 ? (sh->head_rows > 0 ? sh->head_rows : 1) : n_item;
    int vb[2] = {ceil_log2(batch), ceil_log2(max_val_i)};
    int lb[2], rb[2] = {n_user > 0 ? ceil_log2(n_user) : 32, rows_i > 0 ? ceil_log2(rows_i) : 32};
    for (int T = 0; T < 2; ++T) {
        const int64_t max_n = (int64_t)(T ? nis : 1) * batch;
        lb[T] = max_n <= SML_PREP_SMALL ? 0 : ceil_log2((max_n + 1023) / 1024);
        if (lb[T] > 10) lb[T] = 10;
    }
    // packed 4-byte entries (row_hi << vb | value) when both tables fit -- more buckets per list buy row bits
    bool narrow = true;
    for (int T = 0; T < 2; ++T) if (rb[T] + vb[T] - 32 > 10) narrow = false;
    if (narrow) { for (int T = 0; T < 2; ++T) if (rb[T] + vb[T] - 32 > lb[T]) lb[T] = rb[T] + vb[T] - 32; }
    else vb[0] = vb[1] = 32;
    Buf<uint32_t>* hist[2] = {&c->hist_u, &c->hist_i};
    Buf<uint32_t>* bko[2] = {&c->bko_u, &c->bko_i};
    HIPCHK(c->key_u.ensure((size_t)n + 1)); HIPCHK(c->key_u2.ensure((size_t)n + 1));
    HIPCHK(c->key_i.ensure((size_t)n_items + 1)); HIPCHK(c->key_i2.ensure((size_t)n_items + 1));
    HIPCHK(c->val_u2.ensure((size_t)n + 1)); HIPCHK(c->val_i2.ensure((size_t)n_items + 1));
    HIPCHK(c->n_sel.ensure(4));
    const size_t large_cap = (size_t)((n + n_items) / SML_PREP_SMALL) + 8;
    HIPCHK(c->large.ensure(2 * large_cap));
    const size_t runs_i_cap = (size_t)(allruns_i ? n_items : n) + 8;       // duplicated runs: at most every second occurrence
    if (dups) {
        HIPCHK(c->uniq.ensure((size_t)3 * nb * batch));
        HIPCHK(c->runs_u.ensure((size_t)n / 2 + 8)); HIPCHK(c->runs_i.ensure(runs_i_cap));
        HIPCHK(c->off_u.ensure((size_t)nb + 1)); HIPCHK(c->off_i.ensure((size_t)nb + 1));
        HIPCHK(c->cnt_u.ensure((size_t)(nb + 1) * SML_PREP_CNT_STRIDE)); HIPCHK(c->cnt_i.ensure((size_t)(nb + 1) * SML_PREP_CNT_STRIDE));
    } else {
        HIPCHK(c->rec_u.ensure((size_t)n)); HIPCHK(c->rec_i.ensure((size_t)(n_items > 2 * n ? n_items : 2 * n)));
        if (mode == 0) {          // every slot learns where its run's record is (the MF stage's fused row update)
            c->slot_stride = ioff_max + 2 * (int64_t)batch;
            HIPCHK(c->slot_info.ensure((size_t)nb * c->slot_stride));
            a.slot_info = c->slot_info.p; a.slot_stride = c->slot_stride;
            // distinct-row form: both lists must be ONE bucket each (the numbering is the bucket's), i.e. 2 * batch <= SML_PREP_SMALL
            if (want_dense && pad_tiles && 2 * (int64_t)batch <= SML_PREP_SMALL) {
                c->tiles_cap = wg_tiles(batch, 1) + wg_tiles(2 * batch, 1);
                // records are addressed by SCRATCH ROW, which reaches the end of a list's last tile: whole tiles per list
                c->dense_stride = ioff_max + (int64_t)SML_TM * wg_tiles(2 * batch, 1);
                HIPCHK(c->dense_rec.ensure((size_t)nb * c->dense_stride)); HIPCHK(c->dense_n.ensure((size_t)2 * nb));
                HIPCHK(c->tile_hdr.ensure((size_t)nb * c->tiles_cap)); HIPCHK(c->tile_ent.ensure((size_t)nb * c->tiles_cap * SML_TILE_ENT));
                HIPCHK(c->tile_spill.ensure((size_t)nb * 3 * batch)); HIPCHK(c->spill_cnt.ensure((size_t)nb));
                HIPCHK(hipMemsetAsync(c->spill_cnt.p, 0, (size_t)nb * sizeof(int), st));
                a.dense = 1; a.dense_rec = c->dense_rec.p; a.dense_stride = c->dense_stride; a.dense_n = c->dense_n.p; a.tile_hdr = c->tile_hdr.p; a.tile_ent = c->tile_ent.p;
                a.tile_spill = c->tile_spill.p; a.spill_cnt = c->spill_cnt.p; a.tiles_cap = c->tiles_cap;
            }
        }
    }
    for (int T = 0; T < 2; ++T) {
        SmlPrepTable& t = a.t[T];
        t.lb = lb[T]; t.nbk = 1 << lb[T];
        t.hb = rb[T] > lb[T] ? rb[T] - lb[T] : 0; t.vb = vb[T];
        t.npass = (t.hb + 8) / 9; t.pbits = t.npass ? (t.hb + t.npass - 1) / t.npass : 0;
        t.allruns = (T == 1 && allruns_i) ? 1 : 0; t.rshift = t.allruns ? 0 : 1;
        t.lmul = T ? nis : 1;
        t.ntile = T ? nis * a.tpb : (has_users ? a.tpb : 0);
        t.wave = (dups && !t.allruns && ((int64_t)(T ? nis : 1) * batch >> t.lb) <= 256 && !getenv("SML_PREP_NOWAVE")) ? 1 : 0;
        // few row bits left inside a bucket: a counter per row instead of a sort (k_prep_count; SML_PREP_COUNT=0: A/B tests)
        if (dups && !t.allruns && t.nbk > 1 && (1 << t.hb) <= SML_PREP_CROWS && ((int64_t)(T ? nis : 1) * batch >> t.lb) <= 1024 &&
            env_int("SML_PREP_COUNT", 1) != 0 && !getenv("SML_PREP_NOWAVE")) t.wave = 2;
        HIPCHK(hist[T]->ensure((size_t)nb * (T ? nis : 1) * a.tpb * t.nbk));
        HIPCHK(bko[T]->ensure((size_t)2 * nb * t.nbk));
        t.hist = hist[T]->p; t.bk = reinterpret_cast<uint2*>(bko[T]->p); t.brc = nullptr;
        t.ent = T ? (void*)c->key_i.p : (void*)c->key_u.p; t.ent2 = T ? (void*)c->key_i2.p : (void*)c->key_u2.p;
        t.vals = T ? c->val_i2.p : c->val_u2.p;
        if (dups) { t.runs_tmp = nullptr; t.runs = T ? c->runs_i.p : c->runs_u.p; t.run_off = T ? c->off_i.p : c->off_u.p; t.run_cnt = T ? c->cnt_i.p : c->cnt_u.p; }
        else t.runs = T ? c->rec_i.p : c->rec_u.p;
    }
    a.large = c->large.p; a.n_large = c->n_sel.p + 3; a.large_cap = (int)large_cap;
    HIPCHK(c->medium.ensure((size_t)2 * nb * (a.t[0].nbk + a.t[1].nbk) + 2));
    a.medium = c->medium.p; a.n_medium = c->n_sel.p;
    c->hot_cap = 0;
    if (dups) {
        a.uniq = has_users ? c->uniq.p : nullptr; a.uniq_stride = (int64_t)3 * batch; a.max_len = c->n_sel.p + 2;
        const int64_t hot_cap64 = ((int64_t)batch + (int64_t)nis * batch) / SML_HOT + 8;
        const int hot_cap = (int)(hot_cap64 < 0x7fffffff ? hot_cap64 : 0x7fffffff);
        c->hot_cap = (mode <= 1 && batch >= 4096 && hot_cap <= SML_HOT_MAXCAP) ? hot_cap : 0;      // (the sharded step has no hot-row path)
        if (c->hot_cap) {
            HIPCHK(c->hot_list.ensure((size_t)nb * c->hot_cap * 3)); HIPCHK(c->hot_count.ensure((size_t)nb));
            a.hot_list = c->hot_list.p; a.hot_count = c->hot_count.p; a.hot_cap = c->hot_cap;
        }
    }
    // stable ranks from one returning LDS atomic per occurrence -- only where this device hands them out in lane order:
    // measured once per index set, on the stream, ahead of its first preparation (no host wait: the kernels read the count)
    if (!c->rank_probed) {
        HIPCHK(c->rank_viol.ensure(4));
        HIPCHK(hipMemsetAsync(c->rank_viol.p, 0, 4 * sizeof(int), st));
        HIPCHK(sml_launch_rank_probe(c->rank_viol.p, st));
        c->rank_probed = true;
    }
    { const char* rk = getenv("SML_PREP_RANK"); a.rank_viol = (rk && !strcmp(rk, "ballot")) ? nullptr : c->rank_viol.p; }
    a.order_viol = c->rank_viol.p + 1;             // (word 1 of the probe's block: zeroed with it, never reset)
    a.vals_ascend = (mode != 4 || plan == nullptr) ? 1 : 0;      // (source 4 with unequal batches: the driver's owner-split lists, values = slots by owner)
    if (a.dense && !(a.t[0].nbk == 1 && a.t[1].nbk == 1)) a.dense = 0;       // (wide rows forced more buckets: per-occurrence form)
    c->dense = a.dense != 0;
    HIPCHK(sml_launch_prep(a, narrow ? 4 : 8, st));
    // the sorted-order invariant's violation count travels to the host behind every preparation (4 bytes); the NEXT call that
    // finds a completed copy with a non-zero count fails: a list was built wrong (sort_order_check)
    if (!c->viol_host) { HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&c->viol_host), 4 * sizeof(int), hipHostMallocDefault)); memset(c->viol_host, 0, 4 * sizeof(int)); }
    if (!c->viol_ready) HIPCHK(hipEventCreateWithFlags(&c->viol_ready, hipEventDisableTiming));
    HIPCHK(hipMemcpyAsync(c->viol_host, c->rank_viol.p + 1, 3 * sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(hipEventRecord(c->viol_ready, st));
    if (dups) {
        if (!c->max_len_host) HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&c->max_len_host), sizeof(int), hipHostMallocDefault));
        if (!c->ready) HIPCHK(hipEventCreateWithFlags(&c->ready, hipEventDisableTiming));
        HIPCHK(hipMemcpyAsync(c->max_len_host, c->n_sel.p + 2, sizeof(int), hipMemcpyDeviceToHost, st));
        // the batches' run-list offsets and counts travel too (a few KB; epochs of up to 4,096 batches): an epoch prepared ahead
        // launches its run kernels with the slices as plain arguments
        if (mode == 0 && nb <= 4096) {
            if (c->lists_host_nb < nb) {
                if (c->lists_host) g_graveyard.park_host(c->lists_host);
                c->lists_host = nullptr; c->lists_host_nb = 0;
                HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&c->lists_host), (size_t)(2 * (nb + 1) + 2 * nb * SML_PREP_CNT_STRIDE) * sizeof(int), hipHostMallocDefault));
                c->lists_host_nb = nb;
            }
            int* h = c->lists_host;
            HIPCHK(hipMemcpyAsync(h, c->off_u.p, (size_t)(nb + 1) * sizeof(int), hipMemcpyDeviceToHost, st));
            HIPCHK(hipMemcpyAsync(h + (nb + 1), c->off_i.p, (size_t)(nb + 1) * sizeof(int), hipMemcpyDeviceToHost, st));
            HIPCHK(hipMemcpyAsync(h + 2 * (nb + 1), c->cnt_u.p, (size_t)nb * SML_PREP_CNT_STRIDE * sizeof(int), hipMemcpyDeviceToHost, st));
            HIPCHK(hipMemcpyAsync(h + 2 * (nb + 1) + nb * SML_PREP_CNT_STRIDE, c->cnt_i.p, (size_t)nb * SML_PREP_CNT_STRIDE * sizeof(int), hipMemcpyDeviceToHost, st));
            c->lists_nb = nb;
        } else c->lists_nb = 0;
        HIPCHK(hipEventRecord(c->ready, st));
    }
    c->n = mode >= 2 ? -1 : n; c->batch = batch; c->triples = tri; c->world = W;
    return SML_OK;
}

#ifdef SML_TEST_PREP_REFERENCE
#include "../../tests/csrc/prep_cub_reference.inc"
#endif

// The index lists of an epoch: per batch the occurrences sorted by row (stable), run records, "row occurs once" marks -- all by
// index_prep.hip (prep_epoch).  SML_PREP=cub asks for the library-sort REFERENCE implementation instead: that lives under
// tests/csrc and is compiled only into the test build of this library (tests/build_reference.py, -DSML_TEST_PREP_REFERENCE);
// the product has no library sort and says so.
int sort_epoch(IndexSet* c, const int64_t* tri, int64_t n, int batch, int pad_tiles, int64_t n_user, int64_t n_item,
               bool dups, hipStream_t st, const sml_batch_plan* plan = nullptr, const sml_bare_exchange* bx = nullptr, bool want_dense = false) {
    // bx (bare step on several GPUs): the item lists are the JOB's -- every rank's 2n item occurrences
    if (prep_by_hand()) return prep_epoch(c, tri, n, batch, pad_tiles, n_user, n_item, dups, st, plan, bx ? 1 : 0, bx, nullptr, 0, want_dense);
#ifdef SML_TEST_PREP_REFERENCE
    return sort_epoch_reference(c, tri, n, batch, pad_tiles, n_user, n_item, dups, st, plan, bx);
#else
    return fail(SML_ESTATE, "index preparation", "SML_PREP=cub: the library-sort reference is test infrastructure (tests/build_reference.py), not in this library");
#endif
}

// Index lists of the item-sharded bare step.  A: this rank's users (as in sort_epoch) and the JOB's occurrences of the
// tail rows this rank owns (every run is applied by the run kernel).  Bset: this rank's own occurrences of head rows
// (summed into the dense partial).
int sort_epoch_sharded(IndexSet* A, IndexSet* Bset, const int64_t* tri, int64_t n, int batch, int64_t n_user, const sml_bare_shard* sh,
                       int64_t rows_cap, hipStream_t st) {
    if (prep_by_hand()) {
        int rc = prep_epoch(A, tri, n, batch, 0, n_user, 0, true, st, nullptr, 2, nullptr, sh, rows_cap);
        if (!rc && sh->head_rows > 0) rc = prep_epoch(Bset, tri, n, batch, 0, n_user, 0, true, st, nullptr, 3, nullptr, sh, rows_cap);
        return rc;
    }
#ifdef SML_TEST_PREP_REFERENCE
    return sort_epoch_sharded_reference(A, Bset, tri, n, batch, n_user, sh, rows_cap, st);
#else
    return fail(SML_ESTATE, "index preparation", "SML_PREP=cub: the library-sort reference is test infrastructure (tests/build_reference.py), not in this library");
#endif
}

}  // namespace

extern "C" {

const char* sml_last_error(void) { return g_err.c_str(); }
int sml_version(void) { return 1; }

int64_t sml_theta_net_size(int d) { return d_ok(d) ? sml_net_size(d) : -1; }
int64_t sml_theta_offset(int d, int which) {
    if (!d_ok(d)) return -1;
    switch (which) {
        case 0: return SML_OFF_C1W;
        case 1: return SML_OFF_C1B;
        case 2: return SML_OFF_C2W;
        case 3: return SML_OFF_C2B;
        case 4: return SML_OFF_F1W;
        case 5: return sml_off_f1b(d);
        case 6: return sml_off_f2w(d);
        case 7: return sml_off_f2b(d);
        default: return -1;
    }
}

int sml_ctx_create(sml_ctx** out, int device, int d, int max_batch) {
    if (!out || !d_ok(d) || max_batch <= 0) return fail(SML_EINVAL, "sml_ctx_create", "d must be 32/64/128, max_batch > 0");
    int n_dev = 0;
    HIPCHK(hipGetDeviceCount(&n_dev));
    if (device < 0 || device >= n_dev) return fail(SML_EINVAL, "sml_ctx_create", "no such device");
    sml_ctx* c = new (std::nothrow) sml_ctx();
    if (!c) return fail(SML_ENOMEM, "sml_ctx_create", "host allocation");
    c->device = device; c->d = d; c->max_batch = max_batch;
    *out = c;
    return SML_OK;
}

int sml_ctx_set_variant(sml_ctx* ctx, int variant) {
    if (!ctx || variant < 0 || variant > 3) return fail(SML_EINVAL, "sml_ctx_set_variant", "variant must be 0 or 1 (+ 2: an evaluation-stream context)");
    ctx->variant = variant & 1; ctx->side = (variant & 2) != 0;
    return SML_OK;
}

int sml_ctx_set_adaptive(sml_ctx* ctx, float beta) {
    if (!ctx || !(beta >= 0.0f)) return fail(SML_EINVAL, "sml_ctx_set_adaptive", "beta must be >= 0 (0: off)");
    ctx->adaptive_beta = beta;
    return SML_OK;
}

int sml_ctx_set_grad_clip(sml_ctx* ctx, float max_norm) {
    if (!ctx || !(max_norm >= 0.0f)) return fail(SML_EINVAL, "sml_ctx_set_grad_clip", "max_norm must be >= 0 (0: off)");
    ctx->clip_max_norm = max_norm;
    return SML_OK;
}

static void ctx_destroy_now(sml_ctx* ctx) {
    (void)sml_comm_destroy(ctx);
    { DevGuard g(ctx->device); ctx->release_all(); }
    delete ctx;
}
int sml_ctx_destroy(sml_ctx* ctx) {
    if (!ctx) return SML_OK;
    (void)sml_peer_detach(ctx);
    // While ANY context of the process has peer mappings attached, a consumer kernel may be polling on the device for a
    // launch of some host thread -- possibly the very thread this destructor runs on (Python's garbage collector finalises
    // old engines wherever it happens to run).  Destroying an RCCL communicator, freeing device or pinned memory all wait
    // for the device: the context is kept and destroyed with the next one that goes while nobody is attached.
    if (g_graveyard.defer(ctx)) return SML_OK;
    ctx_destroy_now(ctx);
    for (sml_ctx* z : g_graveyard.take_zombies()) ctx_destroy_now(z);
    g_graveyard.reap();
    return SML_OK;
}

int sml_theta_pack(sml_ctx* ctx, const float* theta, void* stream) {
    if (!ctx || !theta) return fail(SML_EINVAL, "sml_theta_pack", "null argument");
    DevGuard g(ctx->device);
    int rc = ensure_pk(ctx); if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    ctx->prof.begin(PC_PACK, st); HIPCHK(sml_launch_theta_pack(ctx->d, theta, pk_cur(ctx), st)); ctx->prof.end(st);
    return SML_OK;
}

int sml_transfer_forward(sml_ctx* ctx, const float* theta, int net, const float* x_t, const float* x_hat, float* out,
                         int64_t n_rows, void* stream) {
    if (n_rows == 0) return SML_OK;
    if (!ctx || !theta || !x_t || !x_hat || !out || (net != 0 && net != 1) || n_rows < 0 || n_rows > 0x7fffffff)
        return fail(SML_EINVAL, "sml_transfer_forward", "bad argument");
    DevGuard g(ctx->device);
    hipStream_t st = (hipStream_t)stream;
    int rc = ensure_pk(ctx); if (rc) return rc;
    // table-sized calls at d = 32: the bf16x3 kernel (fp32-grade results on the bf16 matrix rate; SML_FWD_BX3=0: the fp32 products)
    const size_t bx3 = n_rows > 8192 && env_int("SML_FWD_BX3", 1) != 0 && env_int("SML_FWD_MT", 0) == 0 ? sml_bx3_bytes(ctx->d) : 0;
    if (bx3) {
        HIPCHK(ctx->pkx.ensure(bx3 / sizeof(float) + 4));
        ctx->prof.begin(PC_PACK, st); HIPCHK(sml_launch_theta_pack_bx3(ctx->d, theta, ctx->pkx.p, st)); ctx->prof.end(st);
        SmlFwdArgs a;
        memset(&a, 0, sizeof(a));
        SmlSeg& s = a.seg[0];
        s.theta = theta + (int64_t)net * sml_net_size(ctx->d);
        s.xt_tab = x_t; s.xh_tab = x_hat; s.n_rows = (int)n_rows; s.out = out;
        a.k2 = ctx->variant == 1; a.unit_rows = (ctx->variant == 1 && net == 0);
        a.tiles0 = wg_tiles((int)n_rows, 2);
        const char* pkx_net = reinterpret_cast<const char*>(ctx->pkx.p) + (size_t)net * (bx3 / 2);
        ctx->prof.begin(ctx->side ? PC_FWD_SIDE : PC_FWD, st);
        HIPCHK(sml_launch_fwd_bx3(ctx->d, a, pkx_net, a.tiles0, st, ctx->side));
        ctx->prof.end(st);
        return SML_OK;
    }
    ctx->prof.begin(PC_PACK, st); HIPCHK(sml_launch_theta_pack(ctx->d, theta, pk_cur(ctx), st)); ctx->prof.end(st);
    SmlFwdArgs a;
    memset(&a, 0, sizeof(a));
    SmlSeg& s = a.seg[0];
    s.theta = theta + (int64_t)net * sml_net_size(ctx->d);
    s.pk = pk_cur(ctx) + (int64_t)net * sml_pk_size(ctx->d);
    s.xt_tab = x_t; s.xh_tab = x_hat; s.n_rows = (int)n_rows; s.out = out;
    a.k2 = ctx->variant == 1; a.unit_rows = (ctx->variant == 1 && net == 0);
    // table-sized calls are bound by streaming the weights through L2 once per workgroup: 32 rows per workgroup
    // halve that traffic, 48 (d = 32: what the LDS tiles allow) cut it to a third
    // (SML_FWD_MT overrides for measurements: 3 = the 48-row one-workgroup-per-CU form at d = 32)
    const int mt_env = env_int("SML_FWD_MT", 0);
    const int mt = n_rows > 8192 ? ((mt_env == 3 && ctx->d == 32) ? 3 : 2) : 1;
    a.tiles0 = wg_tiles((int)n_rows, mt);
    a.seg[1] = s; a.seg[1].n_rows = 0;
    const bool side = ctx->side && mt == 2 && (ctx->d == 32 || ctx->d == 64);
    ctx->prof.begin(side ? PC_FWD_SIDE : PC_FWD, st); HIPCHK(sml_launch_fwd(ctx->d, mt, 1, a, a.tiles0, st, side)); ctx->prof.end(st);
    return SML_OK;
}

int sml_mf_stage_epoch(sml_ctx* ctx, const float* theta, const sml_mf_tables* t, const int64_t* triples, int64_t n,
                       int batch, float lr, float l2, int loss_kind, int64_t* step, float* batch_loss,
                       const sml_mf_exchange* xchg, const sml_batch_plan* plan, void* stream) {
    if (!ctx || !theta || !t || !step || !batch_loss || batch <= 0 || (plan ? n < 0 : n <= 0) || (n > 0 && !triples))
        return fail(SML_EINVAL, "sml_mf_stage_epoch", "bad argument");
    if (plan && (plan->n_batches <= 0 || !plan->batch_off || !plan->batch_off_dev || plan->batch_off[0] != 0 || plan->batch_off[plan->n_batches] != n))
        return fail(SML_EINVAL, "sml_mf_stage_epoch", "batch plan does not cover the triples");
    if (batch > ctx->max_batch) return fail(SML_EINVAL, "sml_mf_stage_epoch", "batch exceeds ctx max_batch");
    if (n > 0x3fffffff) return fail(SML_EINVAL, "sml_mf_stage_epoch", "epoch too long");
    if (loss_kind < 0 || loss_kind > 3 || (ctx->variant == 1 ? (loss_kind != SML_LOSS_BPR_UNIT && loss_kind != SML_LOSS_BPR_NORM) : loss_kind == SML_LOSS_BPR_UNIT))
        return fail(SML_EINVAL, "sml_mf_stage_epoch", "loss_kind (variant 1 takes SML_LOSS_BPR_UNIT, or SML_LOSS_BPR_NORM for run_MF(norm=True); SML_LOSS_BPR_UNIT goes with variant 1 only)");
    if (xchg && (xchg->world < 1 || !xchg->key_items || !xchg->val_items || !xchg->dx_local || !xchg->dx_items_all))
        return fail(SML_EINVAL, "sml_mf_stage_epoch", "incomplete exchange descriptor");
    const bool mf_peers = xchg && !xchg->hook && ctx->peer.world > 0;
    if (mf_peers && ctx->peer.world != xchg->world) return fail(SML_ESTATE, "sml_mf_stage_epoch", "peer mappings were attached for another world size");
    if (mf_peers && (xchg->slot_stride > 0 ? xchg->slot_stride : (int64_t)2 * batch) != ctx->peer.rows_cap)
        return fail(SML_EINVAL, "sml_mf_stage_epoch", "on the peer path the exchange's slot_stride must equal the rows_cap the inboxes were attached with");
    if (mf_peers && (xchg->push_rows < 0 || xchg->push_rows > ctx->peer.rows_cap || (xchg->push_rows > 0 && xchg->push_rows < 2 * (int64_t)batch)))
        return fail(SML_EINVAL, "sml_mf_stage_epoch", "push_rows must cover a batch's 2*batch item rows and fit a slot");
    if (xchg && !xchg->hook && !mf_peers && (!ctx->comm || ctx->comm_world != xchg->world))
        return fail(SML_ESTATE, "sml_mf_stage_epoch", "exchange without a hook needs sml_peer_attach or sml_comm_init with the same world size");
    DevGuard g(ctx->device);
    hipStream_t st = (hipStream_t)stream;
    const int d = ctx->d;
    const int64_t nb = plan ? plan->n_batches : (n + batch - 1) / batch;
    if (plan) for (int64_t b = 0; b < nb; ++b)
        if (plan->batch_off[b + 1] < plan->batch_off[b] || plan->batch_off[b + 1] - plan->batch_off[b] > batch)
            return fail(SML_EINVAL, "sml_mf_stage_epoch", "a planned batch is longer than `batch`");
    int rc;
    if ((rc = ensure_pk(ctx))) return rc;
    if ((rc = ensure_transfer_ws(ctx, batch, false, st))) return rc;
    if ((rc = ensure_sched(ctx, lr, *step + nb + 1, st))) return rc;
    const int lstride = (wg_tiles(batch, 1) + wg_tiles(2 * batch, 1)) * (d / 16);     // one loss partial per backward workgroup
    const int64_t out_pstride = (int64_t)SML_R * (tiles_of(batch) + tiles_of(2 * batch)) * d;
    const int fns = fwd_split(wg_tiles(batch, 1) + wg_tiles(2 * batch, 1)), bsplit = bwd_split(wg_tiles(batch, 1) + wg_tiles(2 * batch, 1));
    HIPCHK(ctx->loss_part.ensure((size_t)nb * lstride));
    ctx->prof.begin(PC_PACK, st); HIPCHK(sml_launch_theta_pack(d, theta, pk_cur(ctx), st)); ctx->prof.end(st);
    // d = 32, one workgroup per row tile: the forward's fc1 / fc2 on bf16x3 products (k_mf_fwd_bx3; SML_MF_BX3=0: fp32 products);
    // theta is fixed during the MF stage: its bf16 planes are packed once per epoch
    const bool mf_bx3 = fns == 1 && sml_bx3_bytes(d) > 0 && env_int("SML_MF_BX3", 1) != 0;
    if (mf_bx3) {
        HIPCHK(ctx->pkx.ensure(sml_bx3_bytes(d) / sizeof(float) + 4));
        ctx->prof.begin(PC_PACK, st); HIPCHK(sml_launch_theta_pack_bx3(d, theta, ctx->pkx.p, st)); ctx->prof.end(st);
    }
    // One GPU, the one-workgroup-per-tile backward, a row inside one wavefront (d <= 64): the backward takes the row update itself
    // (SmlFusedUpdate; SML_MF_FUSED_UPDATE=0: the k_run_update launch, A/B tests) -- and, round 5, the net runs once per DISTINCT
    // row of the batch (SmlDense; SML_MF_DISTINCT=0: one pass per occurrence, A/B tests) when the lists allow it
    const bool can_fuse = !xchg && !bsplit && d <= 64 && ctx->adaptive_beta <= 0.0f && env_int("SML_MF_FUSED_UPDATE", 1) != 0;
    const bool want_dense = can_fuse && fns == 1 && env_int("SML_MF_DISTINCT", 1) != 0;
    ctx->prof.begin(PC_SORT, st); rc = sort_epoch(&ctx->ix[0], triples, n, batch, 1, t->n_user, t->n_item, false, st, plan, nullptr, want_dense);
    const int64_t x_total = !xchg ? 0 : xchg->item_off ? xchg->item_off[nb] : (int64_t)xchg->world * 2 * n;
    const int64_t x_stride = !xchg ? 0 : xchg->slot_stride > 0 ? xchg->slot_stride : (int64_t)2 * batch;
    const bool x_by_hand = xchg && xchg->lists_unsorted != 0;
    if (!rc && xchg && x_total > 0 && x_by_hand) {
        // the job's item occurrences arrive batch-major and UNSORTED: every batch's run list by index_prep.hip (occurrence source 4:
        // one stream of explicit (key, value) pairs; records mode) -- no library sort anywhere on this path since round 5
        int64_t x_seg = (int64_t)xchg->world * 2 * batch;
        sml_batch_plan px;
        memset(&px, 0, sizeof(px));
        if (xchg->item_off) {
            std::vector<int32_t> off32((size_t)nb + 1);
            x_seg = 1;
            for (int64_t b = 0; b <= nb; ++b) {
                off32[(size_t)b] = (int32_t)xchg->item_off[b];
                if (b && xchg->item_off[b] - xchg->item_off[b - 1] > x_seg) x_seg = xchg->item_off[b] - xchg->item_off[b - 1];
            }
            HIPCHK(ctx->xoff.ensure((size_t)nb + 1));
            HIPCHK(hipMemcpyAsync(ctx->xoff.p, off32.data(), ((size_t)nb + 1) * sizeof(int32_t), hipMemcpyHostToDevice, st));   // (pageable source: staged before the call returns)
            px.n_batches = nb; px.batch_off_dev = ctx->xoff.p;
        }
        if (x_seg > 0x3fffffff) return fail(SML_EINVAL, "sml_mf_stage_epoch", "a batch's job-wide item list is too long");
        rc = prep_epoch(&ctx->ix[1], reinterpret_cast<const int64_t*>(xchg->key_items), x_total, (int)x_seg, 0, 1, t->n_item, false, st,
                        xchg->item_off ? &px : nullptr, 4, nullptr, nullptr, (int64_t)xchg->world * x_stride, false, xchg->val_items);
    } else if (!rc && xchg && x_total > 0) {   // ... or run records over the caller's sorted keys
        HIPCHK(ctx->rec_x.ensure((size_t)x_total));
        HIPCHK(sml_launch_mark_runs(8, xchg->key_items, xchg->val_items, x_total, 32, ctx->rec_x.p, nullptr, nullptr, 0, 0, st));
    }
    ctx->prof.end(st); if (rc) return rc;
    HIPCHK(hipMemsetAsync(ctx->loss_part.p, 0, (size_t)nb * lstride * sizeof(float), st));
    const int64_t ns = sml_net_size(d), ps = sml_pk_size(d);
    float* dx_buf = xchg ? xchg->dx_local : ctx->dx.p;
    const bool dense = want_dense && ctx->ix[0].by_hand && ctx->ix[0].dense;
    const bool fused = can_fuse && !dense && ctx->ix[0].by_hand && ctx->ix[0].slot_stride > 0;
    if (env_int("SML_TRACE", 0))        // (tests assert WHICH form ran: an A/B that silently compares a form with itself proves nothing)
        fprintf(stderr, "[sml] mf_stage_epoch: form=%s batches=%lld batch=%d d=%d\n", dense ? "distinct-rows" : fused ? "per-occurrence+fused-update" : "per-occurrence+run-update",
                (long long)nb, batch, d);
    if (fused) {
        // (the counters clear themselves batch by batch; they are zeroed per epoch all the same -- 12 KB -- so that an epoch
        // that was cut short, e.g. by a failed launch, cannot leave a count behind for the next one)
        HIPCHK(ctx->run_arrive.ensure((size_t)3 * ctx->max_batch + 8));
        HIPCHK(hipMemsetAsync(ctx->run_arrive.p, 0, ((size_t)3 * ctx->max_batch + 8) * sizeof(int), st));
    }
    for (int64_t b = 0; b < nb; ++b) {
        const int64_t off0 = plan ? plan->batch_off[b] : b * batch;
        const int B = plan ? (int)(plan->batch_off[b + 1] - off0) : (int)((n - off0) < batch ? (n - off0) : batch);
        const int64_t* tri = triples + off0 * 3;
        const int cur = (int)(*step + 1 + b);
        SmlFwdArgs f;
        memset(&f, 0, sizeof(f));
        for (int s = 0; s < 2; ++s) {
            SmlSeg& sg = f.seg[s];
            sg.theta = theta + s * ns; sg.pk = pk_cur(ctx) + s * ps;
            sg.xt_tab = s ? t->last_item : t->last_user;
            sg.xh_tab = s ? t->w_item : t->w_user;
            sg.m_tab = s ? t->m_item : t->m_user; sg.v_tab = s ? t->v_item : t->v_user;
            sg.last_tab = s ? t->step_item : t->step_user;
            sg.tri = tri; sg.B = B; sg.is_item = s; sg.n_rows = s ? 2 * B : B;
            const int64_t slot0 = s ? (int64_t)SML_R * tiles_of(B) : 0;
            sg.out = ctx->out.p + slot0 * d; sg.z1 = ctx->z1.p + slot0 * SML_HID; sg.xin = ctx->xin.p + slot0 * 3 * d;
            sg.a1 = nullptr;
            sg.mrep = ctx->mrep.p + slot0 * d; sg.vrep = ctx->vrep.p + slot0 * d;
            if (dense) {        // scratch row k = distinct row k of this list (every tile's rows named by its header)
                sg.n_rows = SML_TM * wg_tiles(sg.n_rows, 1);          // (whole tiles: a live row may sit past the batch's ragged end)
                sg.drec = ctx->ix[0].dense_rec.p + b * ctx->ix[0].dense_stride + slot0;
                sg.hdr = ctx->ix[0].tile_hdr.p + b * ctx->ix[0].tiles_cap + (s ? wg_tiles(B, 1) : 0);
            }
        }
        f.tiles0 = wg_tiles(B, 1); f.cur_step = cur; f.sched = ctx->sched.p; f.out_pstride = out_pstride; f.k2 = ctx->variant == 1;
        f.sched_len = replay_len(ctx, cur - 1);
        const int tiles = f.tiles0 + wg_tiles(2 * B, 1);
        f.tiles_total = tiles;
        ctx->prof.begin(PC_FWD, st);
        if (mf_bx3) HIPCHK(sml_launch_mf_fwd_bx3(d, f, ctx->pkx.p, tiles, st));
        else HIPCHK(sml_launch_fwd(d, 1, fns, f, tiles, st));
        ctx->prof.end(st);
        SmlBwdArgs w;
        memset(&w, 0, sizeof(w));
        for (int s = 0; s < 2; ++s) {
            SmlBwdSeg& sg = w.seg[s];
            const int64_t slot0 = s ? (int64_t)SML_R * tiles_of(B) : 0;
            sg.theta = theta + s * ns; sg.pk = pk_cur(ctx) + s * ps;
            sg.dout = ctx->dout.p + slot0 * d; sg.is_item = s; sg.z1 = ctx->z1.p + slot0 * SML_HID; sg.xin = ctx->xin.p + slot0 * 3 * d;
            sg.dx = dx_buf + slot0 * d; sg.dz1 = nullptr; sg.n_rows = s ? 2 * B : B;
        }
        w.tiles0 = f.tiles0; w.l2 = l2; w.convg_part = nullptr;
        w.out_all = ctx->out.p; w.B = B; w.ioff = SML_R * tiles_of(B); w.kind = loss_kind;
        w.scale = (plan && plan->loss_scale) ? plan->loss_scale[b] : xchg ? xchg->loss_scale : 1.0f;
        w.loss_part = ctx->loss_part.p + b * lstride;
        w.out_np = fns; w.out_pstride = out_pstride;
        if (dense) {
            w.dn.hdr = ctx->ix[0].tile_hdr.p + b * ctx->ix[0].tiles_cap;
            w.dn.ent = ctx->ix[0].tile_ent.p + b * ctx->ix[0].tiles_cap * SML_TILE_ENT;
            w.dn.spill = ctx->ix[0].tile_spill.p + b * 3 * batch;
            w.dn.drec = ctx->ix[0].dense_rec.p + b * ctx->ix[0].dense_stride;
        }
        if (fused || dense) {
            SmlFusedUpdate& fu = w.fu;
            if (fused) {
                fu.slot_info = ctx->ix[0].slot_info.p + b * ctx->ix[0].slot_stride;
                fu.rec[0] = ctx->ix[0].rec_u.p + off0; fu.rec[1] = ctx->ix[0].rec_i.p + 2 * off0;
                fu.val[0] = ctx->ix[0].val_u2.p; fu.val[1] = ctx->ix[0].val_i2.p;
                fu.arrive = ctx->run_arrive.p;
            }
            fu.dx_all = dx_buf; fu.tri = tri;
            fu.w[0] = (float*)t->w_user; fu.w[1] = (float*)t->w_item; fu.m[0] = t->m_user; fu.m[1] = t->m_item;
            fu.v[0] = t->v_user; fu.v[1] = t->v_item; fu.last[0] = t->step_user; fu.last[1] = t->step_item;
            fu.mrep = ctx->mrep.p; fu.vrep = ctx->vrep.p; fu.sched = ctx->sched.p; fu.cur_step = cur;
        }
        // several GPUs, one-shot exchange, the one-workgroup-per-tile backward: its item tiles push their rows themselves and every
        // workgroup signals -- the grid is cut for the batch CAP (the same on every rank).  SML_PEER_PUSH_LAUNCH=1: A/B (k_peer_push)
        const bool push_fused = mf_peers && !bsplit && !env_int("SML_PEER_PUSH_LAUNCH", 0) &&
                                (xchg->push_rows > 0 ? xchg->push_rows : x_stride) >= 2 * (int64_t)B;
        SmlPeerPush push_f; SmlPeerPoll poll_f;
        int bwd_grid = tiles;
        if (push_fused) {
            bwd_grid = wg_tiles(batch, 1) + wg_tiles(2 * batch, 1);
            if (bwd_grid < tiles) bwd_grid = tiles;
            peer_step(ctx, 1, bwd_grid, &push_f, &poll_f);
            w.push = push_f; w.tiles_live = tiles;
        }
        ctx->prof.begin(PC_BWD, st); HIPCHK(sml_launch_bwd(d, bsplit, w, bwd_grid, st)); ctx->prof.end(st);
        if (fused || dense) continue;         // (the backward stepped the rows)
        if (ctx->adaptive_beta > 0.0f) {      // --need_adaptive: the users' norm term joins their gradient rows and the batch's loss
            ctx->prof.begin(PC_MISC, st);
            HIPCHK(sml_launch_adaptive_users(d, ctx->xin.p, dx_buf, B, ctx->adaptive_beta, w.loss_part, st));
            ctx->prof.end(st);
        }
        SmlRunArgs u;
        memset(&u, 0, sizeof(u));
        u.run_u = ctx->ix[0].rec_u.p + off0; u.n_u = B; u.val_u = ctx->ix[0].val_u2.p;
        u.run_i = ctx->ix[0].rec_i.p + 2 * off0; u.n_i = 2 * B; u.val_i = ctx->ix[0].val_i2.p;
        u.dx = dx_buf; u.dx_i = dx_buf; u.w_user = t->w_user; u.w_item = t->w_item;
        if (xchg) {
            // every rank contributes x_stride rows per batch (its 2*B item-gradient rows first: B may differ from rank
            // to rank and be zero); an empty local batch still joins the collective
            const float* gathered = xchg->dx_items_all;
            if (xchg->hook) {
                if (xchg->hook(xchg->hook_user, b) != 0) return fail(SML_ESTATE, "sml_mf_stage_epoch", "exchange hook failed");
            } else if (mf_peers) {
                // one-shot: this rank's x_stride rows go straight into slot [parity][rank] of every rank's inbox; the row
                // update starts once every rank's rows have landed here (the slots of one parity are the gathered buffer)
                const int64_t ioff = (int64_t)SML_R * tiles_of(B);
                SmlPeerPush push; SmlPeerPoll poll;
                const int64_t x_push = xchg->push_rows > 0 ? xchg->push_rows : x_stride;     // (the same on every rank)
                ctx->prof.begin(PC_MISC, st);
                if (push_fused) poll = poll_f;                                               // (the backward pushed and signalled)
                else {
                    peer_step(ctx, 1, sml_peer_push_blocks(x_push * d), &push, &poll);
                    HIPCHK(sml_launch_peer_push(dx_buf + ioff * d, x_push * d, push, st));
                }
                if (env_int("SML_PEER_WAIT_LAUNCH", 0)) HIPCHK(sml_launch_peer_wait(poll, st));      // (A/B: the wait as its own launch)
                else u.wait = poll;                                                               // ... or at the head of the row update
                ctx->prof.end(st);
                gathered = poll.slot0;
            } else {
                const int64_t ioff = (int64_t)SML_R * tiles_of(B);
                NCCLCHK(g_rccl.AllGather(dx_buf + ioff * d, xchg->dx_items_all, (size_t)x_stride * d, ncclFloat, ctx->comm, st));
            }
            const int64_t x0 = xchg->item_off ? xchg->item_off[b] : (int64_t)xchg->world * 2 * b * batch;
            u.run_i = (x_by_hand ? ctx->ix[1].rec_i.p : ctx->rec_x.p) + x0;
            u.val_i = x_by_hand ? ctx->ix[1].val_i2.p : xchg->val_items;
            u.n_i = xchg->item_off ? (int)(xchg->item_off[b + 1] - x0) : xchg->world * 2 * B;
            u.dx_i = gathered;
        }
        u.m_user = t->m_user; u.v_user = t->v_user; u.m_item = t->m_item; u.v_item = t->v_item;
        u.last_user = t->step_user; u.last_item = t->step_item; u.sched = ctx->sched.p; u.cur_step = cur; u.lr = lr; u.sched_len = replay_len(ctx, cur - 1);
        // rows continue from the forward's replayed copies (local scratch; the multi-GPU item list's slots index the
        // all-gathered buffer instead, so item rows are replayed from the table there)
        u.rep_x = ctx->xin.p; u.rep_m = ctx->mrep.p; u.rep_v = ctx->vrep.p; u.rep_u = 1; u.rep_i = xchg ? 0 : 1;
        u.rep_x_stride = 3 * d; u.rep_x_off = d;
        ctx->prof.begin(PC_SEG_ADAM, st); HIPCHK(sml_launch_run_adam(d, u, (int64_t)u.n_u + u.n_i, st)); ctx->prof.end(st);
    }
    ctx->prof.begin(PC_MISC, st); HIPCHK(sml_launch_loss_finalize(ctx->loss_part.p, (int)nb, lstride, nullptr, batch_loss, st)); ctx->prof.end(st);
    *step += nb;
    return SML_OK;
}

int sml_mf_adam_flush(sml_ctx* ctx, const sml_mf_tables* t, float lr, int64_t step, void* stream) {
    if (!ctx || !t || step < 0) return fail(SML_EINVAL, "sml_mf_adam_flush", "bad argument");
    if (step == 0) return SML_OK;
    DevGuard g(ctx->device);
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if ((rc = ensure_sched(ctx, lr, step + 1, st))) return rc;
    ctx->prof.begin(PC_FLUSH, st); HIPCHK(sml_launch_adam_flush(ctx->d, t->w_user, t->m_user, t->v_user, t->step_user, t->n_user, ctx->sched.p, (int)step, replay_len(ctx, step), st));
    HIPCHK(sml_launch_adam_flush(ctx->d, t->w_item, t->m_item, t->v_item, t->step_item, t->n_item, ctx->sched.p, (int)step, replay_len(ctx, step), st)); ctx->prof.end(st);
    return SML_OK;
}

int sml_tr_stage_epoch(sml_ctx* ctx, float* theta, float* adam_m, float* adam_v, float* theta_grad,
                       const sml_tr_tables* t, const int64_t* triples, int64_t n, int batch, float lr,
                       float weight_decay, int loss_kind, float loss_scale, int64_t* step, float* batch_loss,
                       sml_grad_hook grad_hook, void* hook_user, const sml_batch_plan* plan, void* stream) {
    if (!ctx || !theta || !adam_m || !adam_v || !t || !step || !batch_loss || batch <= 0 || (plan ? n < 0 : n <= 0) || (n > 0 && !triples))
        return fail(SML_EINVAL, "sml_tr_stage_epoch", "bad argument");
    if (plan && (plan->n_batches <= 0 || !plan->batch_off || plan->batch_off[0] != 0 || plan->batch_off[plan->n_batches] != n))
        return fail(SML_EINVAL, "sml_tr_stage_epoch", "batch plan does not cover the triples");
    if (batch > ctx->max_batch) return fail(SML_EINVAL, "sml_tr_stage_epoch", "batch exceeds ctx max_batch");
    if (loss_kind < 0 || loss_kind > 3 || (ctx->variant == 1 ? (loss_kind != SML_LOSS_BPR_UNIT && loss_kind != SML_LOSS_BPR_NORM) : loss_kind == SML_LOSS_BPR_UNIT))
        return fail(SML_EINVAL, "sml_tr_stage_epoch", "loss_kind (variant 1 takes SML_LOSS_BPR_UNIT, or SML_LOSS_BPR_NORM for run_MF(norm=True); SML_LOSS_BPR_UNIT goes with variant 1 only)");
    DevGuard g(ctx->device);
    hipStream_t st = (hipStream_t)stream;
    const int d = ctx->d;
    const int64_t nb = plan ? plan->n_batches : (n + batch - 1) / batch;
    if (plan) for (int64_t b = 0; b < nb; ++b)
        if (plan->batch_off[b + 1] < plan->batch_off[b] || plan->batch_off[b + 1] - plan->batch_off[b] > batch)
            return fail(SML_EINVAL, "sml_tr_stage_epoch", "a planned batch is longer than `batch`");
    int rc;
    if ((rc = ensure_pk(ctx))) return rc;
    if ((rc = ensure_transfer_ws(ctx, batch, true, st))) return rc;
    // --clip_grad (model/transfer.py:725-727): the flat gradient is completed, its norm taken, and the Adam launch scales it
    const bool clip = ctx->clip_max_norm > 0.0f;
    if (clip) HIPCHK(ctx->clip_sumsq.ensure(4));
    float* grad = theta_grad ? theta_grad : ctx->grad.p;
    const int fns = fwd_split(wg_tiles(batch, 1) + wg_tiles(2 * batch, 1)), bsplit = bwd_split(wg_tiles(batch, 1) + wg_tiles(2 * batch, 1));
    const int cs = bsplit ? d / 16 : 1;                                 // backward workgroups per row tile
    // Restructured step (default): backward head (loss -> dOut -> dZ1) + ONE launch with the weight-gradient tiles and
    // the rest of the backward beside them.  SML_TR_V2=0 runs the round-2 kernels (k_transfer_bwd + k_transfer_wgrad).
    const bool v2 = env_int("SML_TR_V2", 1) != 0;
    const int lstride = (wg_tiles(batch, 1) + wg_tiles(2 * batch, 1)) * (d / 16 > 4 ? d / 16 : 4);
    const int64_t out_pstride = (int64_t)SML_R * (tiles_of(batch) + tiles_of(2 * batch)) * d;
    HIPCHK(ctx->loss_part.ensure((size_t)nb * lstride));
    ctx->prof.begin(PC_PACK, st); HIPCHK(sml_launch_theta_pack(d, theta, pk_cur(ctx), st)); ctx->prof.end(st);
    HIPCHK(hipMemsetAsync(ctx->loss_part.p, 0, (size_t)nb * lstride * sizeof(float), st));
    HIPCHK(hipMemsetAsync(grad, 0, (size_t)2 * sml_net_size(d) * sizeof(float), st));
    const int64_t ns = sml_net_size(d), ps = sml_pk_size(d);
    // Deferred conv step (one GPU, fused Adam, restructured step, hidden-split forward): the merged launch of every batch
    // but the epoch's last leaves the conv-gradient partials to the NEXT batch's forward, which adds them and steps the 190
    // conv parameters in its prologue (SmlFwdArgs).  SML_TR_DEFER=0: the merged launch's last tail workgroup does it.
    const bool fused_path = !clip && !grad_hook && ctx->peer.world <= 0 && ctx->comm == nullptr;
    const bool defer = v2 && fns == 4 && fused_path && !plan && nb > 1 && env_int("SML_TR_DEFER", 1) != 0;
    // Round 5: the forward saves z1 only; the dW2 tiles of the merged launch apply Gelu to their B operand themselves (x * sigmoid(1.702 x):
    // ten instructions on a VALU that is idle there) -- 1.5 MB less written per step, bit-identical.  SML_TR_A2_RECOMPUTE=0: the forward
    // saves Gelu(z1) as well (A/B tests)
    const bool a2_recompute = v2 && env_int("SML_TR_A2_RECOMPUTE", 1) != 0;
    int prev_split = 0, prev_total = 0;
    if (defer) {
        HIPCHK(ctx->cstate.ensure((size_t)2 * 2 * 3 * SML_CG));
        HIPCHK(sml_launch_conv_state_init(d, theta, adam_m, adam_v, ctx->cstate.p, st));
    }
    for (int64_t b = 0; b < nb; ++b) {
        const int64_t off0 = plan ? plan->batch_off[b] : b * batch;
        const int B = plan ? (int)(plan->batch_off[b + 1] - off0) : (int)((n - off0) < batch ? (n - off0) : batch);
        const int64_t* tri = triples + off0 * 3;
        SmlFwdArgs f;
        memset(&f, 0, sizeof(f));
        if (defer) {
            const SmlSched scp = sched_entry((double)lr, *step + b);      // the PREVIOUS batch's step (b >= 1)
            f.cg_part = b > 0 ? ctx->convg.p : nullptr; f.cg_split = prev_split; f.cg_total = prev_total;
            f.cs_in = ctx->cstate.p + (size_t)(b & 1) * 2 * 3 * SML_CG; f.cs_out = ctx->cstate.p + (size_t)((b + 1) & 1) * 2 * 3 * SML_CG;
            f.cs_theta = theta; f.cs_m = adam_m; f.cs_v = adam_v;
            f.cs_wd = weight_decay; f.cs_step_size = scp.step_size; f.cs_bc2_sqrt = scp.bc2_sqrt;
        }
        for (int s = 0; s < 2; ++s) {
            SmlSeg& sg = f.seg[s];
            sg.theta = theta + s * ns; sg.pk = pk_cur(ctx) + s * ps;
            sg.xt_tab = s ? t->last_item : t->last_user;
            sg.xh_tab = s ? t->hat_item : t->hat_user;
            sg.tri = tri; sg.B = B; sg.is_item = s; sg.n_rows = s ? 2 * B : B;
            const int64_t slot0 = s ? (int64_t)SML_R * tiles_of(B) : 0;
            sg.out = ctx->out.p + slot0 * d; sg.z1 = ctx->z1.p + slot0 * SML_HID; sg.xin = ctx->xin.p + slot0 * 3 * d;
            sg.a1 = ctx->a1.p + slot0 * SML_C2 * d; sg.a2 = a2_recompute ? nullptr : ctx->a2.p + slot0 * SML_HID;
        }
        f.tiles0 = wg_tiles(B, 1); f.cur_step = 0; f.sched = nullptr; f.out_pstride = out_pstride; f.k2 = ctx->variant == 1;
        const int tiles = f.tiles0 + wg_tiles(2 * B, 1);
        f.tiles_total = tiles;
        ctx->prof.begin(PC_FWD, st); HIPCHK(sml_launch_fwd(d, 1, fns, f, tiles, st)); ctx->prof.end(st);
        SmlBwdArgs w;
        memset(&w, 0, sizeof(w));
        SmlWgArgs wg;
        memset(&wg, 0, sizeof(wg));
        for (int s = 0; s < 2; ++s) {
            SmlBwdSeg& sg = w.seg[s];
            const int64_t slot0 = s ? (int64_t)SML_R * tiles_of(B) : 0;
            sg.theta = theta + s * ns; sg.pk = pk_cur(ctx) + s * ps;
            sg.dout = ctx->dout.p + slot0 * d; sg.is_item = s; sg.z1 = ctx->z1.p + slot0 * SML_HID; sg.xin = ctx->xin.p + slot0 * 3 * d;
            sg.dx = nullptr; sg.dz1 = ctx->dz1.p + slot0 * SML_HID; sg.n_rows = s ? 2 * B : B;
            SmlWgSeg& q = wg.seg[s];
            q.dz1 = sg.dz1; q.a1 = ctx->a1.p + slot0 * SML_C2 * d; q.dout = sg.dout; q.a2 = a2_recompute ? sg.z1 : ctx->a2.p + slot0 * SML_HID;
            // (one GPU, Adam fused into the weight-gradient kernel, no gradient buffer asked for: the flat gradient is
            // not written at all -- 0.8 MB less for the launch to leave dirty in L2)
            const bool fused_only = !clip && !grad_hook && (ctx->comm == nullptr || ctx->peer.world > 0) && theta_grad == nullptr;
            q.grad = fused_only ? nullptr : grad + s * ns; q.n_rows = sg.n_rows;
            q.theta_net = sg.theta; q.pk_net = sg.pk; q.xin = sg.xin;
        }
        w.tiles0 = f.tiles0; w.l2 = 0.0f; w.convg_part = ctx->convg.p;
        w.out_all = ctx->out.p; w.B = B; w.ioff = SML_R * tiles_of(B); w.kind = loss_kind;
        w.scale = (plan && plan->loss_scale) ? plan->loss_scale[b] : loss_scale; w.loss_part = ctx->loss_part.p + b * lstride;
        w.out_np = fns; w.out_pstride = out_pstride; w.tiles_total = tiles;
        ctx->prof.begin(PC_BWD, st);
        if (v2) HIPCHK(sml_launch_tr_bwd_head(d, w, tiles, st)); else HIPCHK(sml_launch_bwd(d, bsplit, w, tiles, st));
        ctx->prof.end(st);
        const SmlSched sc = sched_entry((double)lr, *step + 1 + b);
        // (v2: tiles0 / tiles_total count ROW tiles, the first n_tail workgroups are the backward's tail)
        const int wcs = v2 ? 1 : cs;
        auto launch_wgrad = [&](const SmlWgArgs& g) { return v2 ? sml_launch_tr_wgrad2(d, g, st) : sml_launch_wgrad(d, g, st); };
        wg.convg_part = ctx->convg.p; wg.tiles0 = f.tiles0 * wcs; wg.tiles_total = tiles * wcs;
        wg.n_tail = tiles * (d / 16) > 0 ? tiles * (d / 16) : 1; wg.convg_out = ctx->convg.p; wg.arrive = ctx->arrive.p;
        wg.defer_conv = (defer && b + 1 < nb) ? 1 : 0; wg.gelu_b = a2_recompute ? 1 : 0;
        prev_split = f.tiles0 * (d / 16); prev_total = tiles * (d / 16);
        const bool peers = !grad_hook && ctx->peer.world > 0;       // peer mappings attached: one-shot push / poll
        const bool native = !grad_hook && !peers && ctx->comm != nullptr;     // a communicator exists: exchange natively
        if (peers) {
            // the weight-gradient workgroups store their finished tiles into every rank's inbox; the Adam launch polls
            // this rank's counters and adds the slots in rank order
            SmlThetaAdamArgs ad;
            memset(&ad, 0, sizeof(ad));
            peer_step(ctx, 0, v2 ? sml_wgrad2_pushers(d) : sml_wgrad_grid(d), &wg.peer, &ad.peer);
            ctx->prof.begin(PC_WGRAD, st); HIPCHK(launch_wgrad(wg)); ctx->prof.end(st);
            ad.theta = theta; ad.m = adam_m; ad.v = adam_v; ad.grad = grad; ad.pk = pk_cur(ctx);
            ad.weight_decay = weight_decay; ad.step_size = sc.step_size; ad.bc2_sqrt = sc.bc2_sqrt;
            // every workgroup of the Adam kernel polls the counters itself (SML_PEER_POLL_IN_ADAM=0: a one-wavefront
            // k_peer_wait launch ahead of it instead -- one launch more; same results)
            static const bool poll_in_adam = env_int("SML_PEER_POLL_IN_ADAM", 1) != 0;
            ctx->prof.begin(PC_THETA_ADAM, st);
            if (clip) {
                // --clip_grad on the peer carrier: the rank-order sum of the slots is materialised once (k_peer_sum polls
                // and adds exactly as the fused form would), its norm taken, and the plain Adam launch scales it -- every
                // rank forms the same bits, so the replicas stay identical
                HIPCHK(sml_launch_peer_sum(grad, 2 * ns, ad.peer, st, (int)ns));
                HIPCHK(sml_launch_grad_sumsq(grad, 2 * ns, ctx->clip_sumsq.p, st));
                ad.peer.world = 0;
                ad.clip_sumsq = ctx->clip_sumsq.p; ad.clip_max_norm = ctx->clip_max_norm;
            } else if (!poll_in_adam) { HIPCHK(sml_launch_peer_wait(ad.peer, st)); ad.peer.waited = 1; }
            HIPCHK(sml_launch_theta_adam(d, ad, st));
            ctx->prof.end(st);
        } else if (!grad_hook && !native && !clip) {
            // one GPU: the weight-gradient workgroups take the Adam step for the tiles they own
            // (v2: the refreshed images go to the OTHER set -- the launch's tail workgroups are reading this one)
            wg.theta = theta; wg.m = adam_m; wg.v = adam_v; wg.pk = v2 ? pk_other(ctx) : pk_cur(ctx);
            wg.weight_decay = weight_decay; wg.step_size = sc.step_size; wg.bc2_sqrt = sc.bc2_sqrt;
            ctx->prof.begin(PC_WGRAD, st); HIPCHK(launch_wgrad(wg)); ctx->prof.end(st);
            if (v2) ctx->pk_set ^= 1;
        } else {
            // the weight-gradient launch leaves the flat gradient complete (its conv workgroups sum the backward's
            // partials): all-reduce it, then one Adam launch
            ctx->prof.begin(PC_WGRAD, st); HIPCHK(launch_wgrad(wg)); ctx->prof.end(st);
            SmlThetaAdamArgs ad;
            memset(&ad, 0, sizeof(ad));
            ad.theta = theta; ad.m = adam_m; ad.v = adam_v; ad.grad = grad; ad.pk = pk_cur(ctx);
            ad.weight_decay = weight_decay; ad.step_size = sc.step_size; ad.bc2_sqrt = sc.bc2_sqrt;
            if (native) {
                NCCLCHK(g_rccl.AllReduce(grad, grad, (size_t)(2 * ns), ncclFloat, ncclSum, ctx->comm, st));
            } else if (grad_hook) {
                const int hr = grad_hook(hook_user, grad, 2 * ns, b);
                if (hr != 0) return fail(SML_ESTATE, "sml_tr_stage_epoch", "grad_hook failed");
            }
            ctx->prof.begin(PC_THETA_ADAM, st);
            if (clip) {        // (the norm of the job's gradient: after the exchange)
                HIPCHK(sml_launch_grad_sumsq(grad, 2 * ns, ctx->clip_sumsq.p, st));
                ad.clip_sumsq = ctx->clip_sumsq.p; ad.clip_max_norm = ctx->clip_max_norm;
            }
            HIPCHK(sml_launch_theta_adam(d, ad, st)); ctx->prof.end(st);
        }
    }
    ctx->prof.begin(PC_MISC, st); HIPCHK(sml_launch_loss_finalize(ctx->loss_part.p, (int)nb, lstride, nullptr, batch_loss, st)); ctx->prof.end(st);
    *step += nb;
    return SML_OK;
}

int sml_run_mf_grad(sml_ctx* ctx, const float* theta, const float* user_last, const float* user_hat, const float* item_last,
                    const float* item_hat, int B, int loss_kind, float* loss, float* d_user_hat, float* d_item_hat,
                    float* theta_grad, void* stream) {
    if (!ctx || !theta || !user_last || !user_hat || !item_last || !item_hat || !loss || B <= 0)
        return fail(SML_EINVAL, "sml_run_mf_grad", "bad argument");
    if (B > ctx->max_batch) return fail(SML_EINVAL, "sml_run_mf_grad", "batch exceeds ctx max_batch");
    if (loss_kind < 0 || loss_kind > 3 || (ctx->variant == 1 ? (loss_kind != SML_LOSS_BPR_UNIT && loss_kind != SML_LOSS_BPR_NORM) : loss_kind == SML_LOSS_BPR_UNIT))
        return fail(SML_EINVAL, "sml_run_mf_grad", "loss_kind (variant 1 takes SML_LOSS_BPR_UNIT, or SML_LOSS_BPR_NORM for run_MF(norm=True); SML_LOSS_BPR_UNIT goes with variant 1 only)");
    DevGuard g(ctx->device);
    hipStream_t st = (hipStream_t)stream;
    const int d = ctx->d;
    int rc;
    if ((rc = ensure_pk(ctx))) return rc;
    if ((rc = ensure_transfer_ws(ctx, B, true, st))) return rc;       // TR-stage saves (a1, a2, dz1, conv partials, flat gradient)
    if ((rc = ensure_transfer_ws(ctx, B, false, st))) return rc;      // + the MF stage's dx rows
    const int64_t ns = sml_net_size(d), ps = sml_pk_size(d);
    const int tiles0 = wg_tiles(B, 1), tiles = tiles0 + wg_tiles(2 * B, 1);
    const int fns = fwd_split(tiles);
    const int lstride = tiles * (d / 16 > 4 ? d / 16 : 4);
    const int64_t out_pstride = (int64_t)SML_R * (tiles_of(B) + tiles_of(2 * B)) * d;
    HIPCHK(ctx->loss_part.ensure((size_t)2 * lstride));
    HIPCHK(hipMemsetAsync(ctx->loss_part.p, 0, (size_t)2 * lstride * sizeof(float), st));
    HIPCHK(sml_launch_theta_pack(d, theta, pk_cur(ctx), st));
    // ---- forward over contiguous row blocks (identity indexing), with every save the two backward forms need
    SmlFwdArgs f;
    memset(&f, 0, sizeof(f));
    SmlBwdArgs w;
    memset(&w, 0, sizeof(w));
    SmlWgArgs wg;
    memset(&wg, 0, sizeof(wg));
    float* grad = theta_grad ? theta_grad : ctx->grad.p;
    for (int s = 0; s < 2; ++s) {
        const int64_t slot0 = s ? (int64_t)SML_R * tiles_of(B) : 0;
        SmlSeg& sg = f.seg[s];
        sg.theta = theta + s * ns; sg.pk = pk_cur(ctx) + s * ps;
        sg.xt_tab = s ? item_last : user_last; sg.xh_tab = s ? item_hat : user_hat;
        sg.tri = nullptr; sg.B = B; sg.is_item = s; sg.n_rows = s ? 2 * B : B;
        sg.out = ctx->out.p + slot0 * d; sg.z1 = ctx->z1.p + slot0 * SML_HID; sg.xin = ctx->xin.p + slot0 * 3 * d;
        sg.a1 = ctx->a1.p + slot0 * SML_C2 * d; sg.a2 = ctx->a2.p + slot0 * SML_HID;
        SmlBwdSeg& bg = w.seg[s];
        bg.theta = sg.theta; bg.pk = sg.pk; bg.dout = ctx->dout.p + slot0 * d; bg.is_item = s; bg.z1 = sg.z1; bg.xin = sg.xin;
        bg.dx = ctx->dx.p + slot0 * d; bg.dz1 = ctx->dz1.p + slot0 * SML_HID; bg.n_rows = sg.n_rows;
        SmlWgSeg& q = wg.seg[s];
        q.dz1 = bg.dz1; q.a1 = sg.a1; q.dout = bg.dout; q.a2 = sg.a2; q.grad = grad + s * ns; q.n_rows = sg.n_rows;
        q.theta_net = sg.theta; q.pk_net = sg.pk; q.xin = sg.xin;
    }
    f.tiles0 = tiles0; f.out_pstride = out_pstride; f.k2 = ctx->variant == 1; f.tiles_total = tiles;
    HIPCHK(sml_launch_fwd(d, 1, fns, f, tiles, st));
    w.tiles0 = tiles0; w.l2 = 0.0f; w.out_all = ctx->out.p; w.B = B; w.ioff = SML_R * tiles_of(B); w.kind = loss_kind; w.scale = 1.0f;
    w.out_np = fns; w.out_pstride = out_pstride; w.tiles_total = tiles;
    // ---- gradient w.r.t. the x_hat rows: the MF-stage backward (loss partials of this launch are the ones reported)
    if (d_user_hat || d_item_hat || !theta_grad) {
        w.convg_part = nullptr; w.loss_part = ctx->loss_part.p;
        SmlBwdArgs wx = w;
        wx.seg[0].dz1 = nullptr; wx.seg[1].dz1 = nullptr;
        HIPCHK(sml_launch_bwd(d, bwd_split(tiles), wx, tiles, st));
        if (d_user_hat) HIPCHK(hipMemcpyAsync(d_user_hat, ctx->dx.p, (size_t)B * d * sizeof(float), hipMemcpyDeviceToDevice, st));
        if (d_item_hat) HIPCHK(hipMemcpyAsync(d_item_hat, ctx->dx.p + (size_t)SML_R * tiles_of(B) * d, (size_t)2 * B * d * sizeof(float),
                                              hipMemcpyDeviceToDevice, st));
        HIPCHK(sml_launch_loss_finalize(ctx->loss_part.p, 1, lstride, nullptr, loss, st));
    }
    // ---- gradient w.r.t. theta: backward head + weight-gradient launch WITHOUT the fused Adam (flat gradient only)
    if (theta_grad) {
        w.convg_part = ctx->convg.p; w.loss_part = ctx->loss_part.p + lstride;
        w.seg[0].dx = nullptr; w.seg[1].dx = nullptr;
        HIPCHK(sml_launch_tr_bwd_head(d, w, tiles, st));
        wg.convg_part = ctx->convg.p; wg.tiles0 = tiles0; wg.tiles_total = tiles;
        wg.n_tail = tiles * (d / 16); wg.convg_out = ctx->convg.p; wg.arrive = ctx->arrive.p;
        HIPCHK(hipMemsetAsync(grad, 0, (size_t)2 * ns * sizeof(float), st));       // (the conv block's alignment padding)
        HIPCHK(sml_launch_tr_wgrad2(d, wg, st));
        if (!(d_user_hat || d_item_hat)) HIPCHK(sml_launch_loss_finalize(ctx->loss_part.p + lstride, 1, lstride, nullptr, loss, st));
    }
    return SML_OK;
}

int sml_embed_loss_sgd_prepare(sml_ctx* ctx, const int64_t* triples, int64_t n, int batch, int64_t n_user,
                               int64_t n_item, int slot, const sml_bare_exchange* xchg, void* stream) {
    if (!ctx || !triples || n <= 0 || batch <= 0 || (slot != 0 && slot != 1))
        return fail(SML_EINVAL, "sml_embed_loss_sgd_prepare", "bad argument");
    if (xchg && (xchg->world < 1 || !xchg->items_all)) return fail(SML_EINVAL, "sml_embed_loss_sgd_prepare", "incomplete exchange descriptor");
    if (batch > ctx->max_batch) return fail(SML_EINVAL, "sml_embed_loss_sgd_prepare", "batch exceeds ctx max_batch");
    if (n > 0x3fffffff) return fail(SML_EINVAL, "sml_embed_loss_sgd_prepare", "epoch too long");
    DevGuard g(ctx->device);
    return sort_epoch(&ctx->ix[slot], triples, n, batch, 0, n_user, n_item, true, (hipStream_t)stream, nullptr, xchg);
}

int64_t sml_index_lists_read(sml_ctx* ctx, int slot, int which, void* host, int64_t bytes) {
    if (!ctx || (slot != 0 && slot != 1) || !host || bytes < 0) return fail(SML_EINVAL, "sml_index_lists_read", "bad argument");
    DevGuard g(ctx->device);
    IndexSet* X = &ctx->ix[slot];
    const int64_t nb = X->batch > 0 ? (X->n + X->batch - 1) / X->batch : 0;
    const void* src = nullptr; int64_t have = 0;
    int two[2] = {0, X->hot_cap};
    switch (which) {
        case 0: src = X->runs_u.p; have = (int64_t)X->runs_u.cap * sizeof(SmlRun); break;
        case 1: src = X->runs_i.p; have = (int64_t)X->runs_i.cap * sizeof(SmlRun); break;
        case 2: src = X->off_u.p; have = (nb + 1) * 4; break;
        case 3: src = X->off_i.p; have = (nb + 1) * 4; break;
        case 4: case 5: {        // the batches' run counters sit a cache line apart on the device: gathered here
            if (!X->by_hand) return 0;
            std::vector<int> raw((size_t)nb * SML_PREP_CNT_STRIDE), cnt((size_t)nb);
            HIPCHK(hipMemcpy(raw.data(), which == 4 ? X->cnt_u.p : X->cnt_i.p, raw.size() * sizeof(int), hipMemcpyDeviceToHost));
            for (int64_t b = 0; b < nb; ++b) cnt[(size_t)b] = raw[(size_t)b * SML_PREP_CNT_STRIDE];
            const int64_t nbytes = nb * 4 < bytes ? nb * 4 : bytes;
            memcpy(host, cnt.data(), (size_t)nbytes);
            return nbytes;
        }
        case 6: src = X->val_u2.p; have = X->n * 4; break;
        case 7: src = X->val_i2.p; have = 2 * X->n * 4; break;
        case 8: src = X->uniq.p; have = 3 * nb * X->batch; break;
        case 9: src = X->hot_list.p; have = X->hot_cap ? nb * X->hot_cap * 12 : 0; break;
        case 10: src = X->hot_count.p; have = X->hot_cap ? nb * 4 : 0; break;
        case 11: HIPCHK(hipMemcpy(two, X->n_sel.p + 2, 4, hipMemcpyDeviceToHost)); memcpy(host, two, bytes < 8 ? bytes : 8); return bytes < 8 ? bytes : 8;
        default: return fail(SML_EINVAL, "sml_index_lists_read", "which");
    }
    const int64_t nbytes = have < bytes ? have : bytes;
    if (nbytes > 0 && src) HIPCHK(hipMemcpy(host, src, (size_t)nbytes, hipMemcpyDeviceToHost));
    return src ? nbytes : 0;
}

int sml_embed_loss_sgd_epoch(sml_ctx* ctx, void* w_user, void* w_item, int64_t n_user, int64_t n_item, int dtype_bytes,
                             const int64_t* triples, int64_t n, int batch, float lr, float lam_user, float lam_item,
                             int loss_kind, float* batch_loss, int prepared_slot, const sml_bare_exchange* xchg, void* stream) {
    if (!ctx || !w_user || !w_item || !triples || !batch_loss || n <= 0 || batch <= 0 || n_user <= 0 || n_item <= 0)
        return fail(SML_EINVAL, "sml_embed_loss_sgd_epoch", "bad argument");
    if (xchg && (xchg->world < 1 || !xchg->items_all || !xchg->dx_items_all || !xchg->dx_local))
        return fail(SML_EINVAL, "sml_embed_loss_sgd_epoch", "incomplete exchange descriptor");
    if (xchg && !xchg->hook && (!ctx->comm || ctx->comm_world != xchg->world))
        return fail(SML_ESTATE, "sml_embed_loss_sgd_epoch", "exchange without a hook needs sml_comm_init with the same world size");
    if (dtype_bytes != 4 && dtype_bytes != 2) return fail(SML_EINVAL, "sml_embed_loss_sgd_epoch", "dtype_bytes must be 4 or 2");
    if (loss_kind != SML_LOSS_BCE && loss_kind != SML_LOSS_BPR) return fail(SML_EINVAL, "sml_embed_loss_sgd_epoch", "loss_kind");
    if (batch > ctx->max_batch) return fail(SML_EINVAL, "sml_embed_loss_sgd_epoch", "batch exceeds ctx max_batch");
    if (n > 0x3fffffff) return fail(SML_EINVAL, "sml_embed_loss_sgd_epoch", "epoch too long");
    if (prepared_slot < -1 || prepared_slot > 1) return fail(SML_EINVAL, "sml_embed_loss_sgd_epoch", "prepared_slot");
    DevGuard g(ctx->device);
    hipStream_t st = (hipStream_t)stream;
    const int d = ctx->d;
    const int64_t nb = (n + batch - 1) / batch;
    int rc;
    HIPCHK(ctx->dx.ensure((size_t)3 * batch * d));
    float* const dxb = xchg ? xchg->dx_local : ctx->dx.p;       // per-occurrence gradient rows of the batch in flight
    const int lpr = d * dtype_bytes / 16;
    const int lstride = (int)(((int64_t)batch * lpr + 255) / 256);
    HIPCHK(ctx->loss_part.ensure((size_t)nb * lstride));
    IndexSet* X = &ctx->ix[prepared_slot < 0 ? 0 : prepared_slot];
    if (prepared_slot < 0) {
        ctx->prof.begin(PC_SORT, st); rc = sort_epoch(X, triples, n, batch, 0, n_user, n_item, true, st, nullptr, xchg); ctx->prof.end(st);
        if (rc) return rc;
    } else if (X->n != n || X->batch != batch || X->triples != triples || X->world != (xchg ? xchg->world : 1)) {
        return fail(SML_ESTATE, "sml_embed_loss_sgd_epoch", "index set was prepared for other triples");
    }
    HIPCHK(hipMemsetAsync(ctx->loss_part.p, 0, (size_t)nb * lstride * sizeof(float), st));
    // NO host wait: the event is only queried.  Lists prepared an epoch ahead on a side stream are long complete and
    // their longest run is known: an epoch without hot rows skips the hot-row kernels and runs the light-tailed run
    // kernel.  Lists still in flight (built inline just above, or prepared a moment ago): assume hot rows -- the hot
    // reducers then find, on the device, that their lists are empty.
    const int hot_cap = X->hot_cap;
    const bool known = hipEventQuery(X->ready) == hipSuccess;
    if ((rc = sort_order_check(X))) return rc;          // (the prepared lists' sorted-order invariant, once its read-back has landed)
    const bool hot = hot_cap > 0 && (!known || *X->max_len_host > SML_HOT);
    const int world = xchg ? xchg->world : 1;
    const int hot_chunks = (int)(((int64_t)batch + (int64_t)world * 2 * batch) / SML_HOT_CHUNK) + hot_cap;
    if (hot) {
        HIPCHK(ctx->hot_first.ensure((size_t)hot_cap));
        HIPCHK(ctx->hot_part.ensure((size_t)hot_chunks * d));
    }
    for (int64_t b = 0; b < nb; ++b) {
        const int B = (int)((n - b * batch) < batch ? (n - b * batch) : batch);
        SmlBareArgs a;
        memset(&a, 0, sizeof(a));
        a.w_user = w_user; a.w_item = w_item; a.tri = triples + b * batch * 3; a.B = B; a.dx = dxb;
        a.loss_part = ctx->loss_part.p + b * lstride; a.kind = loss_kind; a.lam_user = lam_user; a.lam_item = lam_item;
        a.uniq = X->uniq.p + (size_t)3 * b * batch; a.lr = lr; a.scale = xchg ? xchg->loss_scale : 1.0f;
        ctx->prof.begin(PC_BARE_GRAD, st); HIPCHK(sml_launch_bare_grad(d, dtype_bytes, a, nullptr, st)); ctx->prof.end(st);
        if (xchg) {     // every rank's item-gradient rows (slots [B, B + 2*batch) of dx; a ragged batch sends its tail along)
            if (xchg->hook) {
                if (xchg->hook(xchg->hook_user, b) != 0) return fail(SML_ESTATE, "sml_embed_loss_sgd_epoch", "exchange hook failed");
            } else {
                NCCLCHK(g_rccl.AllGather(dxb + (size_t)B * d, xchg->dx_items_all, (size_t)2 * batch * d, ncclFloat, ctx->comm, st));
            }
        }
        SmlRunArgs u;
        memset(&u, 0, sizeof(u));
        // compacted duplicated-run lists of the whole epoch; the kernels slice out batch b themselves and
        // stride over it (how many runs a batch has is only known on the device)
        u.run_u = X->runs_u.p; u.run_i = X->runs_i.p; u.off_u = X->off_u.p; u.off_i = X->off_i.p; u.batch_index = (int)b;
        u.val_u = X->val_u2.p; u.val_i = X->val_i2.p;
        if (X->by_hand) { u.cnt_u = X->cnt_u.p; u.cnt_i = X->cnt_i.p; }
        int64_t max_rec = xchg ? (int64_t)B / 2 + (int64_t)world * 2 * B : (int64_t)3 * B / 2;
        if (known && X->by_hand && !xchg && X->lists_nb == nb && env_int("SML_A3_KNOWN_LISTS", 1)) {
            // the prepared epoch's list slices are on the host: plain arguments, an exact grid
            const int* h = X->lists_host;
            const int ou = h[b], oi = h[(nb + 1) + b];
            const int cu = h[2 * (nb + 1) + b * SML_PREP_CNT_STRIDE], ci = h[2 * (nb + 1) + nb * SML_PREP_CNT_STRIDE + b * SML_PREP_CNT_STRIDE];
            u.run_u = X->runs_u.p + ou; u.n_u = cu; u.run_i = X->runs_i.p + oi; u.n_i = ci;
            u.off_u = u.off_i = nullptr; u.cnt_u = u.cnt_i = nullptr; u.known = 1;
            max_rec = (int64_t)cu + ci;
        }
        u.dx = dxb; u.dx_i = xchg ? xchg->dx_items_all : dxb; u.w_user = w_user; u.w_item = w_item; u.lr = lr;
        if (hot) {
            u.hot_list = X->hot_list.p + (size_t)b * hot_cap * 3; u.hot_count = X->hot_count.p + b; u.hot_first = ctx->hot_first.p;
            u.hot_part = ctx->hot_part.p; u.hot_cap = hot_cap; u.hot_blocks = hot_chunks < 512 ? hot_chunks : 512;
        }
        // (grid: one lane group per possible record -- duplicated runs are at most every second occurrence; on several
        // GPUs every item run of the job's list is a record)
        ctx->prof.begin(PC_SEG_SGD, st); HIPCHK(sml_launch_run_sgd(d, dtype_bytes, u, max_rec, st)); ctx->prof.end(st);
        if (hot) { ctx->prof.begin(PC_PAIR_LOSS, st); HIPCHK(sml_launch_hot_apply(d, dtype_bytes, u, st)); ctx->prof.end(st); }
    }
    ctx->prof.begin(PC_MISC, st); HIPCHK(sml_launch_loss_finalize(ctx->loss_part.p, (int)nb, lstride, nullptr, batch_loss, st)); ctx->prof.end(st);
    return SML_OK;
}

int sml_embed_loss_sgd_epoch_sharded(sml_ctx* ctx, void* w_user, int64_t n_user, int64_t n_item, int dtype_bytes,
                                     const int64_t* triples, int64_t n, int batch, float lr, float lam_user, float lam_item,
                                     int loss_kind, float* batch_loss, const sml_bare_shard* sh, void* stream) {
    if (!ctx || !w_user || !triples || !batch_loss || !sh || n <= 0 || batch <= 0 || n_user <= 0 || n_item <= 0)
        return fail(SML_EINVAL, "sml_embed_loss_sgd_epoch_sharded", "bad argument");
    if (dtype_bytes != 4 && dtype_bytes != 2) return fail(SML_EINVAL, "sml_embed_loss_sgd_epoch_sharded", "dtype_bytes must be 4 or 2");
    if (loss_kind != SML_LOSS_BCE && loss_kind != SML_LOSS_BPR) return fail(SML_EINVAL, "sml_embed_loss_sgd_epoch_sharded", "loss_kind");
    if (batch > ctx->max_batch) return fail(SML_EINVAL, "sml_embed_loss_sgd_epoch_sharded", "batch exceeds ctx max_batch");
    const int W = sh->world;
    if (ctx->peer.world <= 0 || ctx->peer.world != W || ctx->peer.rank != sh->rank)
        return fail(SML_ESTATE, "sml_embed_loss_sgd_epoch_sharded", "needs sml_peer_attach with the same world / rank");
    if (!sh->item_shard || !sh->items_all || sh->head_rows < 0 || sh->shard_rows <= 0 || (sh->head_rows > 0 && !sh->w_item_head))
        return fail(SML_EINVAL, "sml_embed_loss_sgd_epoch_sharded", "incomplete shard descriptor");
    for (int q = 0; q < W; ++q) if (!sh->item_shard[q]) return fail(SML_EINVAL, "sml_embed_loss_sgd_epoch_sharded", "every rank's shard pointer is needed");
    if (sh->head_rows + (int64_t)W * sh->shard_rows < n_item) return fail(SML_EINVAL, "sml_embed_loss_sgd_epoch_sharded", "head + shards do not cover the item table");
    if (ctx->peer.rows_cap < 2 * (int64_t)batch) return fail(SML_EINVAL, "sml_embed_loss_sgd_epoch_sharded", "the inboxes' rows_cap must hold a batch's 2*batch gradient rows");
    if (sh->head_rows * ctx->d > 2 * (int64_t)sml_net_size(ctx->d) || (sh->head_rows * ctx->d) % 4)
        return fail(SML_EINVAL, "sml_embed_loss_sgd_epoch_sharded", "the dense head partial must fit a theta slot (head_rows * d <= 2 * net size)");
    DevGuard g(ctx->device);
    hipStream_t st = (hipStream_t)stream;
    const int d = ctx->d, me = sh->rank;
    const int64_t nb = (n + batch - 1) / batch, H = sh->head_rows;
    int rc;
    HIPCHK(ctx->dx.ensure((size_t)3 * batch * d));
    const int lpr = d * dtype_bytes / 16;
    const int lstride = (int)(((int64_t)batch * lpr + 255) / 256);
    HIPCHK(ctx->loss_part.ensure((size_t)nb * lstride));
    if (H > 0) HIPCHK(ctx->head_part.ensure((size_t)H * d));
    HIPCHK(ctx->ptr_tab.ensure(16));
    ctx->prof.begin(PC_SORT, st);
    rc = sort_epoch_sharded(&ctx->ix[0], &ctx->ix[1], triples, n, batch, n_user, sh, ctx->peer.rows_cap, st);
    ctx->prof.end(st);
    if (rc) return rc;
    void* inboxes[8];
    for (int q = 0; q < 8; ++q) inboxes[q] = q < W ? ctx->peer.inbox[q] : nullptr;
    HIPCHK(sml_launch_set_ptr_tab(ctx->ptr_tab.p, sh->item_shard, W, st));
    HIPCHK(sml_launch_set_ptr_tab(ctx->ptr_tab.p + 8, inboxes, W, st));
    HIPCHK(hipMemsetAsync(ctx->loss_part.p, 0, (size_t)nb * lstride * sizeof(float), st));
    IndexSet* A = &ctx->ix[0];
    IndexSet* Bs = &ctx->ix[1];
    for (int64_t b = 0; b < nb; ++b) {
        const int B = (int)((n - b * batch) < batch ? (n - b * batch) : batch);
        // (1) every owner has applied the previous batch (its shard rows are final; this parity's inbox slots are free)
        if (ctx->peer.done_pending) { ctx->prof.begin(PC_MISC, st); HIPCHK(sml_launch_peer_wait(ctx->peer.done_poll, st)); ctx->prof.end(st); }
        // (2) gradient pass: tail rows read from their owners, tail gradient rows stored into their owners' inboxes
        SmlBareArgs a;
        memset(&a, 0, sizeof(a));
        SmlPeerPoll rows_poll;
        int n_blocks = (int)(((int64_t)B * lpr + 255) / 256);
        peer_step(ctx, 1, n_blocks, &a.peer, &rows_poll);
        a.w_user = w_user; a.w_item = sh->w_item_head; a.tri = triples + b * batch * 3; a.B = B; a.dx = ctx->dx.p;
        a.loss_part = ctx->loss_part.p + b * lstride; a.kind = loss_kind; a.lam_user = lam_user; a.lam_item = lam_item;
        a.uniq = A->uniq.p + (size_t)3 * b * batch; a.lr = lr; a.scale = sh->loss_scale;
        a.shard_tab = reinterpret_cast<const void* const*>(ctx->ptr_tab.p);
        a.inbox_tab = reinterpret_cast<float* const*>(ctx->ptr_tab.p + 8);
        a.head_rows = H; a.shard_rows = sh->shard_rows;
        a.push_off = (long long)(a.peer.dst[me] - reinterpret_cast<float*>(ctx->peer.inbox[me]));     // the same offset inside every inbox
        ctx->prof.begin(PC_BARE_GRAD, st); HIPCHK(sml_launch_bare_grad(d, dtype_bytes, a, nullptr, st)); ctx->prof.end(st);
        // (3) all ranks' gradient rows of this batch have landed here -> owner update of this rank's shard (+ its duplicated user rows)
        ctx->prof.begin(PC_MISC, st); HIPCHK(sml_launch_peer_wait(rows_poll, st)); ctx->prof.end(st);
        SmlRunArgs u;
        memset(&u, 0, sizeof(u));
        u.run_u = A->runs_u.p; u.run_i = A->runs_i.p; u.off_u = A->off_u.p; u.off_i = A->off_i.p; u.batch_index = (int)b;
        u.val_u = A->val_u2.p; u.val_i = A->val_i2.p;
        if (A->by_hand) { u.cnt_u = A->cnt_u.p; u.cnt_i = A->cnt_i.p; }
        u.dx = ctx->dx.p; u.dx_i = rows_poll.slot0; u.w_user = w_user; u.w_item = sh->item_shard[me]; u.lr = lr;
        ctx->prof.begin(PC_SEG_SGD, st); HIPCHK(sml_launch_run_sgd(d, dtype_bytes, u, (int64_t)B / 2 + (int64_t)W * 2 * B, st)); ctx->prof.end(st);
        // (4) tell every rank: this owner is done with batch b
        {
            SmlPeerPush done_push;
            peer_step(ctx, 2, 1, &done_push, &ctx->peer.done_poll);
            ctx->peer.done_pending = true;
            ctx->prof.begin(PC_MISC, st); HIPCHK(sml_launch_peer_signal(done_push, st)); ctx->prof.end(st);
        }
        // (5) the replicated head: this rank's occurrences -> dense partial -> one-shot all-reduce -> identical update everywhere
        if (H > 0) {
            HIPCHK(hipMemsetAsync(ctx->head_part.p, 0, (size_t)H * d * sizeof(float), st));
            SmlRunArgs hgt;
            memset(&hgt, 0, sizeof(hgt));
            hgt.run_u = Bs->runs_i.p; hgt.run_i = Bs->runs_i.p; hgt.off_u = Bs->off_u.p; hgt.off_i = Bs->off_i.p; hgt.batch_index = (int)b;
            hgt.val_u = Bs->val_i2.p; hgt.val_i = Bs->val_i2.p;
            if (Bs->by_hand) { hgt.cnt_u = Bs->cnt_u.p; hgt.cnt_i = Bs->cnt_i.p; }
            hgt.dx = ctx->dx.p; hgt.dx_i = ctx->dx.p; hgt.w_user = ctx->head_part.p; hgt.w_item = ctx->head_part.p;
            hgt.lr = -1.0f;           // 0 - (-1) * sum = the sum itself, exactly
            ctx->prof.begin(PC_SEG_SGD, st); HIPCHK(sml_launch_run_sgd(d, 4, hgt, (int64_t)2 * B, st)); ctx->prof.end(st);
            SmlPeerPush hp; SmlPeerPoll hq;
            peer_step(ctx, 0, sml_peer_push_blocks(H * d), &hp, &hq);
            ctx->prof.begin(PC_MISC, st);
            HIPCHK(sml_launch_peer_push(ctx->head_part.p, H * d, hp, st));
            HIPCHK(sml_launch_head_apply(d, dtype_bytes, sh->w_item_head, H, lr, hq, st));
            ctx->prof.end(st);
        }
    }
    ctx->prof.begin(PC_MISC, st); HIPCHK(sml_launch_loss_finalize(ctx->loss_part.p, (int)nb, lstride, nullptr, batch_loss, st)); ctx->prof.end(st);
    return SML_OK;
}

int sml_embed_loss_adam_epoch(sml_ctx* ctx, const sml_mf_tables* t, const int64_t* triples, int64_t n, int batch, float lr,
                              float lam_user, float lam_item, int loss_kind, int64_t* step, float* batch_loss, void* stream) {
    if (!ctx || !t || !t->w_user || !t->w_item || !t->m_user || !t->v_user || !t->m_item || !t->v_item || !t->step_user ||
        !t->step_item || !triples || !step || !batch_loss || n <= 0 || batch <= 0)
        return fail(SML_EINVAL, "sml_embed_loss_adam_epoch", "bad argument");
    if (loss_kind != SML_LOSS_BCE && loss_kind != SML_LOSS_BPR) return fail(SML_EINVAL, "sml_embed_loss_adam_epoch", "loss_kind");
    if (batch > ctx->max_batch) return fail(SML_EINVAL, "sml_embed_loss_adam_epoch", "batch exceeds ctx max_batch");
    if (n > 0x3fffffff) return fail(SML_EINVAL, "sml_embed_loss_adam_epoch", "epoch too long");
    DevGuard g(ctx->device);
    hipStream_t st = (hipStream_t)stream;
    const int d = ctx->d;
    const int64_t nb = (n + batch - 1) / batch;
    int rc;
    if ((rc = ensure_sched(ctx, lr, *step + nb + 1, st))) return rc;
    HIPCHK(ctx->dx.ensure((size_t)3 * batch * d));
    HIPCHK(ctx->xin.ensure((size_t)3 * batch * d)); HIPCHK(ctx->mrep.ensure((size_t)3 * batch * d)); HIPCHK(ctx->vrep.ensure((size_t)3 * batch * d));
    const int lstride = (int)(((int64_t)batch * (d / 4) + 255) / 256);
    HIPCHK(ctx->loss_part.ensure((size_t)nb * lstride));
    ctx->prof.begin(PC_SORT, st); rc = sort_epoch(&ctx->ix[0], triples, n, batch, 0, t->n_user, t->n_item, false, st); ctx->prof.end(st);
    if (rc) return rc;
    HIPCHK(hipMemsetAsync(ctx->loss_part.p, 0, (size_t)nb * lstride * sizeof(float), st));
    for (int64_t b = 0; b < nb; ++b) {
        const int B = (int)((n - b * batch) < batch ? (n - b * batch) : batch);
        const int cur = (int)(*step + 1 + b);
        SmlBareArgs a;
        memset(&a, 0, sizeof(a));
        a.w_user = t->w_user; a.w_item = t->w_item; a.tri = triples + b * batch * 3; a.B = B; a.dx = ctx->dx.p;
        a.loss_part = ctx->loss_part.p + b * lstride; a.kind = loss_kind; a.lam_user = lam_user; a.lam_item = lam_item;
        a.m_user = t->m_user; a.v_user = t->v_user; a.m_item = t->m_item; a.v_item = t->v_item;
        a.last_user = t->step_user; a.last_item = t->step_item; a.sched = ctx->sched.p; a.cur_step = cur;
        a.xrep = ctx->xin.p; a.mrep = ctx->mrep.p; a.vrep = ctx->vrep.p; a.scale = 1.0f;
        ctx->prof.begin(PC_BARE_GRAD, st); HIPCHK(sml_launch_bare_grad(d, 4, a, nullptr, st)); ctx->prof.end(st);
        SmlRunArgs u;
        memset(&u, 0, sizeof(u));
        u.run_u = ctx->ix[0].rec_u.p + b * batch; u.n_u = B; u.val_u = ctx->ix[0].val_u2.p;
        u.run_i = ctx->ix[0].rec_i.p + 2 * b * batch; u.n_i = 2 * B; u.val_i = ctx->ix[0].val_i2.p;
        u.dx = ctx->dx.p; u.dx_i = ctx->dx.p; u.w_user = t->w_user; u.w_item = t->w_item;
        u.m_user = t->m_user; u.v_user = t->v_user; u.m_item = t->m_item; u.v_item = t->v_item;
        u.last_user = t->step_user; u.last_item = t->step_item; u.sched = ctx->sched.p; u.cur_step = cur; u.lr = lr; u.sched_len = replay_len(ctx, cur - 1);
        u.rep_x = ctx->xin.p; u.rep_m = ctx->mrep.p; u.rep_v = ctx->vrep.p; u.rep_u = 1; u.rep_i = 1; u.rep_x_stride = d; u.rep_x_off = 0;
        ctx->prof.begin(PC_SEG_ADAM, st); HIPCHK(sml_launch_run_adam(d, u, (int64_t)3 * B, st)); ctx->prof.end(st);
    }
    ctx->prof.begin(PC_MISC, st); HIPCHK(sml_launch_loss_finalize(ctx->loss_part.p, (int)nb, lstride, nullptr, batch_loss, st)); ctx->prof.end(st);
    *step += nb;
    return SML_OK;
}

int sml_mf_forward(sml_ctx* ctx, const float* w_user, const float* w_item, const int64_t* user, const int64_t* item,
                   int64_t n, int norm, float* uemb, float* iemb, float* score, void* stream) {
    if (!ctx || !w_user || !w_item || !user || !item || !uemb || !iemb || !score || n < 0)
        return fail(SML_EINVAL, "sml_mf_forward", "bad argument");
    if (n == 0) return SML_OK;
    DevGuard g(ctx->device);
    HIPCHK(sml_launch_mf_forward(ctx->d, w_user, w_item, user, item, n, norm, uemb, iemb, score, (hipStream_t)stream));
    return SML_OK;
}

int sml_eval_ranks(sml_ctx* ctx, const float* w_user, const float* w_item, const int64_t* rows, int64_t n, int n_cols,
                   int32_t* rank, void* stream) {
    if (ctx && n == 0) return SML_OK;        // (a rank that owns no row of a test set passes empty -- null -- tensors)
    if (!ctx || !w_user || !w_item || !rows || !rank || n < 0 || n_cols < 2)
        return fail(SML_EINVAL, "sml_eval_ranks", "bad argument");
    DevGuard g(ctx->device);
    hipStream_t st = (hipStream_t)stream;
    ctx->prof.begin(PC_EVAL_RANKS, st); HIPCHK(sml_launch_eval_ranks(ctx->d, w_user, w_item, rows, n, n_cols, rank, st)); ctx->prof.end(st);
    return SML_OK;
}

int sml_eval_prepare(sml_ctx* ctx, const int64_t* rows, int64_t n, int n_cols, int64_t n_item, int32_t* rows_b,
                     int32_t* bucket_off, void* stream) {
    if (ctx && n == 0 && n_cols >= 2 && n_item > 0) return SML_OK;
    if (!ctx || !rows || !rows_b || !bucket_off || n < 0 || n_cols < 2 || n_item <= 0 || n_item > 0x7fffffff)
        return fail(SML_EINVAL, "sml_eval_prepare", "bad argument");
    // the blocked rank kernel addresses the item table with 32-bit BYTE offsets (row id * d * 4, + 16 per lane)
    if ((uint64_t)n_item * (uint64_t)ctx->d * 4ull > 0x100000000ull)
        return fail(SML_EINVAL, "sml_eval_prepare", "item table too large for the blocked evaluation (n_item * d * 4 > 2^32 bytes): use sml_eval_ranks");
    DevGuard g(ctx->device);
    hipStream_t st = (hipStream_t)stream;
    ctx->prof.begin(PC_MISC, st); HIPCHK(sml_launch_eval_bucketize(rows, n, n_cols, n_item, rows_b, bucket_off, st)); ctx->prof.end(st);
    return SML_OK;
}

int sml_eval_ranks_blocked(sml_ctx* ctx, const float* w_user, const float* w_item, const int32_t* rows_b,
                           const int32_t* bucket_off, int64_t n, int n_cols, int32_t* rank, int max_workgroups,
                           void* stream) {
    if (ctx && n == 0) return SML_OK;
    if (!ctx || !w_user || !w_item || !rows_b || !bucket_off || !rank || n < 0 || n_cols < 2)
        return fail(SML_EINVAL, "sml_eval_ranks_blocked", "bad argument");
    DevGuard g(ctx->device);
    hipStream_t st = (hipStream_t)stream;
    ctx->prof.begin(PC_EVAL_RANKS, st);
    HIPCHK(sml_launch_eval_ranks_bucketed(ctx->d, w_user, w_item, rows_b, bucket_off, n, n_cols, rank, max_workgroups, st));
    ctx->prof.end(st);
    return SML_OK;
}

// ---- LDS-sliced evaluation (d = 32): include/sml_hip.h ------------------------------------------------------------
namespace {
struct EvsGeom { int ns; int64_t n_mb, n_pad, cap; };
bool evs_geom(const sml_ctx* ctx, int64_t n, int n_cols, int64_t n_item, EvsGeom& g) {
    g.ns = sml_evs_slices(ctx->d, n_item);
    g.n_mb = (n + 63) / 64;
    g.n_pad = g.n_mb * 64;
    /* every (test row, slice) unit is padded to an even count: at most one extra entry per unit that holds a candidate */
    g.cap = n_cols >= 2 ? n * ((int64_t)(n_cols - 2) + (g.ns < n_cols - 2 ? g.ns : n_cols - 2)) : 0;
    return g.ns > 0 && n_cols >= 2 && n_cols - 2 <= 32767 && g.cap <= 0x7fffffffll && (int64_t)g.ns * g.n_mb < 0x7fffffffll;
}
}
int sml_eval_sliced_slices(sml_ctx* ctx, int64_t n, int n_cols, int64_t n_item) {
    if (!ctx || n < 0) return 0;
    EvsGeom g;
    return evs_geom(ctx, n, n_cols, n_item, g) ? g.ns : 0;
}
int64_t sml_eval_sliced_entries(sml_ctx* ctx, int64_t n, int n_cols, int64_t n_item) {
    EvsGeom g;
    if (!ctx || n < 0 || !evs_geom(ctx, n, n_cols, n_item, g)) return 0;
    return g.cap;
}
int64_t sml_eval_sliced_work_ints(sml_ctx* ctx, int64_t n, int n_cols, int64_t n_item) {
    EvsGeom g;
    if (!ctx || n < 0 || !evs_geom(ctx, n, n_cols, n_item, g)) return 0;
    return 5 * (int64_t)g.ns * g.n_mb;                  /* per-wavefront histograms [n_mb][4][ns] + segment counts [ns][n_mb] */
}
int64_t sml_eval_sliced_scratch_bytes(sml_ctx* ctx, int64_t n, int n_cols, int64_t n_item) {
    EvsGeom g;
    if (!ctx || n < 0 || !evs_geom(ctx, n, n_cols, n_item, g)) return 0;
    return n * ctx->d * 4 + g.n_pad * 4 + (int64_t)g.ns * g.n_pad * 2;
}
int sml_eval_prepare_sliced(sml_ctx* ctx, const int64_t* rows, int64_t n, int n_cols, int64_t n_item, uint32_t* entries,
                            int32_t* seg_off, int32_t* work, void* stream) {
    if (ctx && n == 0 && n_cols >= 2 && n_item > 0) return SML_OK;
    EvsGeom g;
    if (!ctx || !rows || (!entries && n_cols > 2) || !seg_off || !work || n < 0 || n_item <= 0)      /* (no negatives: no entries) */
        return fail(SML_EINVAL, "sml_eval_prepare_sliced", "bad argument");
    if (!evs_geom(ctx, n, n_cols, n_item, g))
        return fail(SML_EINVAL, "sml_eval_prepare_sliced", "the sliced evaluation needs d = 32, n_item <= 2^20, at most 32767 candidates per row and fewer than 2^31 entries: use sml_eval_prepare");
    DevGuard dg(ctx->device);
    hipStream_t st = (hipStream_t)stream;
    ctx->prof.begin(PC_MISC, st);
    HIPCHK(sml_launch_evs_prepare(rows, n, n_cols, g.ns, work, work + 4 * (int64_t)g.ns * g.n_mb, seg_off, entries, st));
    ctx->prof.end(st);
    return SML_OK;
}
int sml_eval_ranks_sliced(sml_ctx* ctx, const float* w_user, const float* w_item, const int64_t* rows, const uint32_t* entries,
                          const int32_t* seg_off, int64_t n, int n_cols, int64_t n_item, void* scratch, int32_t* rank,
                          int max_workgroups, void* stream) {
    if (ctx && n == 0) return SML_OK;
    EvsGeom g;
    if (!ctx || !w_user || !w_item || !rows || (!entries && n_cols > 2) || !seg_off || !scratch || !rank || n < 0 || n_item <= 0)
        return fail(SML_EINVAL, "sml_eval_ranks_sliced", "bad argument");
    if (!evs_geom(ctx, n, n_cols, n_item, g)) return fail(SML_EINVAL, "sml_eval_ranks_sliced", "shape outside the sliced evaluation's range");
    DevGuard dg(ctx->device);
    hipStream_t st = (hipStream_t)stream;
    float* ug = reinterpret_cast<float*>(scratch);
    float* s0 = ug + n * ctx->d;
    uint16_t* partial = reinterpret_cast<uint16_t*>(s0 + g.n_pad);
    ctx->prof.begin(PC_EVAL_RANKS, st);
    HIPCHK(sml_launch_evs_ranks(ctx->d, w_user, w_item, rows, entries, seg_off, n, n_cols, n_item, g.ns, ug, s0, partial, rank, max_workgroups, st));
    ctx->prof.end(st);
    return SML_OK;
}

int sml_eval_metrics(sml_ctx* ctx, const int32_t* rank, int64_t n, int topk, float* out, void* stream) {
    if (!ctx || !out || n < 0 || (n > 0 && !rank)) return fail(SML_EINVAL, "sml_eval_metrics", "bad argument");
    DevGuard g(ctx->device);
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) { HIPCHK(hipMemsetAsync(out, 0, 2 * sizeof(float), st)); return SML_OK; }    // no rows: (0 hits, 0 ndcg)
    ctx->prof.begin(PC_MISC, st); HIPCHK(sml_launch_eval_metrics(rank, n, topk, out, st)); ctx->prof.end(st);
    return SML_OK;
}

int sml_comm_load(const char* path) {
    if (g_rccl.ok()) return SML_OK;
    if (!path) return fail(SML_EINVAL, "sml_comm_load", "null path");
    void* h = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
    if (!h) return fail(SML_EINVAL, "sml_comm_load", dlerror());
    RcclApi a;
    a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))dlsym(h, "ncclCommInitRank");
    a.CommDestroy = (decltype(a.CommDestroy))dlsym(h, "ncclCommDestroy");
    a.AllReduce = (decltype(a.AllReduce))dlsym(h, "ncclAllReduce");
    a.AllGather = (decltype(a.AllGather))dlsym(h, "ncclAllGather");
    a.GetErrorString = (decltype(a.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllReduce || !a.AllGather)
        return fail(SML_EINVAL, "sml_comm_load", "librccl lacks a required symbol");
    a.handle = h;
    g_rccl = a;
    return SML_OK;
}
int sml_comm_unique_id(void* out128) {
    if (!g_rccl.ok() || !out128) return fail(SML_ESTATE, "sml_comm_unique_id", "call sml_comm_load first");
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    NCCLCHK(g_rccl.GetUniqueId(&id));
    memcpy(out128, &id, sizeof(id));
    return SML_OK;
}
int sml_comm_init(sml_ctx* ctx, int world, int rank, const void* id128) {
    if (!ctx || !id128 || world < 1 || rank < 0 || rank >= world) return fail(SML_EINVAL, "sml_comm_init", "bad argument");
    if (!g_rccl.ok()) return fail(SML_ESTATE, "sml_comm_init", "call sml_comm_load first");
    DevGuard g(ctx->device);
    if (ctx->comm) { (void)g_rccl.CommDestroy(ctx->comm); ctx->comm = nullptr; }
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    NCCLCHK(g_rccl.CommInitRank(&ctx->comm, world, id, rank));
    ctx->comm_world = world; ctx->comm_rank = rank;
    return SML_OK;
}
int sml_comm_destroy(sml_ctx* ctx) {
    if (!ctx) return SML_OK;
    if (ctx->comm && g_rccl.ok()) { DevGuard g(ctx->device); (void)g_rccl.CommDestroy(ctx->comm); }
    ctx->comm = nullptr; ctx->comm_world = 1; ctx->comm_rank = 0;
    return SML_OK;
}
int sml_comm_allreduce(sml_ctx* ctx, float* buf, int64_t n, void* stream) {
    if (!ctx || !buf || n <= 0) return fail(SML_EINVAL, "sml_comm_allreduce", "bad argument");
    if (!ctx->comm) return fail(SML_ESTATE, "sml_comm_allreduce", "no communicator");
    DevGuard g(ctx->device);
    NCCLCHK(g_rccl.AllReduce(buf, buf, (size_t)n, ncclFloat, ncclSum, ctx->comm, (hipStream_t)stream));
    return SML_OK;
}
int sml_comm_allgather(sml_ctx* ctx, const float* src, float* dst, int64_t n_per_rank, void* stream) {
    if (!ctx || !src || !dst || n_per_rank <= 0) return fail(SML_EINVAL, "sml_comm_allgather", "bad argument");
    if (!ctx->comm) return fail(SML_ESTATE, "sml_comm_allgather", "no communicator");
    DevGuard g(ctx->device);
    NCCLCHK(g_rccl.AllGather(src, dst, (size_t)n_per_rank, ncclFloat, ctx->comm, (hipStream_t)stream));
    return SML_OK;
}

// ---- one-shot exchange over peer mappings ------------------------------------------------------------------------
int sml_peer_region_bytes(sml_ctx* ctx, int world, int64_t rows_cap, int64_t* inbox_bytes, int64_t* flags_bytes) {
    if (!ctx || world < 1 || world > SML_MAX_PEERS || rows_cap < 0 || !inbox_bytes || !flags_bytes)
        return fail(SML_EINVAL, "sml_peer_region_bytes", "bad argument");
    *inbox_bytes = (int64_t)2 * world * (peer_theta_slot(ctx->d) + rows_cap * ctx->d) * (int64_t)sizeof(float);
    *flags_bytes = (int64_t)3 * 2 * world * (int64_t)sizeof(unsigned long long);     // kinds: theta / head, rows, "owner done"
    return SML_OK;
}
namespace {
std::mutex g_peer_kind_mu;
std::map<void*, int> g_peer_kind;          // allocation -> 0 uncached, 1 fine-grained, 2 plain device memory
}
int sml_peer_mem_kind(void* ptr) {
    std::lock_guard<std::mutex> l(g_peer_kind_mu);
    auto it = g_peer_kind.find(ptr);
    return it == g_peer_kind.end() ? -1 : it->second;
}
int sml_peer_alloc(int device, int64_t bytes, void** ptr) {
    if (!ptr || bytes <= 0) return fail(SML_EINVAL, "sml_peer_alloc", "bad argument");
    DevGuard g(device);
    void* p = nullptr;
    // uncached device memory: stores from peers land in HBM and the owner's loads never hit a stale L2 line;
    // fine-grained if the runtime refuses that flag; plain device memory as a last resort (one-device tests)
    // (SML_PEER_MEM = uncached | finegrained | plain picks the first kind tried: measurements)
    const char* kind = getenv("SML_PEER_MEM");
    const int first = kind && !strcmp(kind, "finegrained") ? 1 : kind && !strcmp(kind, "plain") ? 2 : 0;
    hipError_t e = hipErrorUnknown;
    int got = -1;
    if (first <= 0) { e = hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocUncached); got = 0; }
    if (e != hipSuccess && first <= 1) { (void)hipGetLastError(); e = hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocFinegrained); got = 1; }
    if (e != hipSuccess) { (void)hipGetLastError(); e = hipMalloc(&p, (size_t)bytes); got = 2; }
    if (getenv("SML_DEBUG_PEER")) fprintf(stderr, "[sml] sml_peer_alloc(%lld bytes): kind %d (0 uncached, 1 fine-grained, 2 plain), %s\n",
                                          (long long)bytes, got, hipGetErrorString(e));
    if (e != hipSuccess) return fail(SML_ENOMEM, "sml_peer_alloc", hipGetErrorString(e));
    e = hipMemset(p, 0, (size_t)bytes);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) { (void)hipFree(p); return fail(SML_EHIP, "sml_peer_alloc", hipGetErrorString(e)); }
    { std::lock_guard<std::mutex> l(g_peer_kind_mu); g_peer_kind[p] = got; }
    *ptr = p;
    return SML_OK;
}
int sml_peer_free(int device, void* ptr) {
    if (!ptr) return SML_OK;
    DevGuard g(device);
    { std::lock_guard<std::mutex> l(g_peer_kind_mu); g_peer_kind.erase(ptr); }
    HIPCHK(hipFree(ptr));
    return SML_OK;
}
int sml_peer_read(int device, const void* src, void* dst, int64_t bytes, void* stream) {
    if (!src || !dst || bytes <= 0 || bytes % 16 || (uintptr_t)src % 16 || (uintptr_t)dst % 16)
        return fail(SML_EINVAL, "sml_peer_read", "16-byte aligned pointers and size are needed");
    DevGuard g(device);
    HIPCHK(sml_launch_peer_read(src, dst, bytes / 16, (hipStream_t)stream));
    return SML_OK;
}
int sml_peer_export(void* ptr, void* handle64) {
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
    if (!ptr || !handle64) return fail(SML_EINVAL, "sml_peer_export", "null argument");
    hipIpcMemHandle_t h;
    HIPCHK(hipIpcGetMemHandle(&h, ptr));
    memcpy(handle64, &h, sizeof(h));
    return SML_OK;
}
int sml_peer_open(int device, const void* handle64, void** ptr) {
    if (!handle64 || !ptr) return fail(SML_EINVAL, "sml_peer_open", "null argument");
    DevGuard g(device);
    hipIpcMemHandle_t h;
    memcpy(&h, handle64, sizeof(h));
    HIPCHK(hipIpcOpenMemHandle(ptr, h, hipIpcMemLazyEnablePeerAccess));
    return SML_OK;
}
int sml_peer_close(int device, void* ptr) {
    if (!ptr) return SML_OK;
    DevGuard g(device);
    HIPCHK(hipIpcCloseMemHandle(ptr));
    return SML_OK;
}
int sml_peer_attach(sml_ctx* ctx, int world, int rank, void* const* inbox, void* const* flags, int64_t rows_cap, double timeout_s) {
    if (!ctx || world < 1 || world > SML_MAX_PEERS || rank < 0 || rank >= world || !inbox || !flags || rows_cap < 0 || !(timeout_s > 0))
        return fail(SML_EINVAL, "sml_peer_attach", "bad argument");
    for (int q = 0; q < world; ++q)
        if (!inbox[q] || !flags[q] || (uintptr_t)inbox[q] % 16 || (uintptr_t)flags[q] % 8)
            return fail(SML_EINVAL, "sml_peer_attach", "every rank's inbox (16-byte aligned) and flags (8-byte aligned) are needed");
    DevGuard g(ctx->device);
    if (!ctx->peer.err) {
        HIPCHK(hipMalloc(reinterpret_cast<void**>(&ctx->peer.err), sizeof(int)));
        HIPCHK(hipMemset(ctx->peer.err, 0, sizeof(int)));
    }
    if (ctx->peer.world == 0) g_graveyard.peers(+1);
    ctx->peer.world = world; ctx->peer.rank = rank;
    for (int q = 0; q < SML_MAX_PEERS; ++q) {
        ctx->peer.inbox[q] = q < world ? static_cast<char*>(inbox[q]) : nullptr;
        ctx->peer.flags[q] = q < world ? static_cast<unsigned long long*>(flags[q]) : nullptr;
    }
    ctx->peer.theta_slot = peer_theta_slot(ctx->d);
    ctx->peer.rows_cap = rows_cap;
    ctx->peer.tick[0] = ctx->peer.tick[1] = ctx->peer.tick[2] = 0;   // (the regions come zeroed from sml_peer_alloc: counters start at 0)
    ctx->peer.done_pending = false;
    memset(ctx->peer.expect, 0, sizeof(ctx->peer.expect));
    ctx->peer.timeout = (long long)(timeout_s * 1e8);   // wall_clock64: 100 MHz
    return SML_OK;
}
int sml_peer_detach(sml_ctx* ctx) {
    if (!ctx) return SML_OK;
    if (ctx->peer.world > 0) g_graveyard.peers(-1);
    ctx->peer.world = 0;
    return SML_OK;
}
int sml_peer_status(sml_ctx* ctx, int* timeouts) {
    if (!ctx || !timeouts) return fail(SML_EINVAL, "sml_peer_status", "null argument");
    *timeouts = 0;
    if (!ctx->peer.err) return SML_OK;
    DevGuard g(ctx->device);
    HIPCHK(hipMemcpy(timeouts, ctx->peer.err, sizeof(int), hipMemcpyDeviceToHost));
    return SML_OK;
}
int sml_peer_allreduce_check(sml_ctx* ctx, const float* src, float* dst, int64_t n, double timeout_s, void* stream) {
    if (!ctx || !src || !dst || n <= 0 || n % 4 || n > 2 * sml_net_size(ctx->d)) return fail(SML_EINVAL, "sml_peer_allreduce_check", "bad argument (n: a multiple of 4 within the theta slot)");
    if (ctx->peer.world <= 0) return fail(SML_ESTATE, "sml_peer_allreduce_check", "no peers attached");
    DevGuard g(ctx->device);
    hipStream_t st = (hipStream_t)stream;
    SmlPeerPush push; SmlPeerPoll poll;
    peer_step(ctx, 0, sml_peer_push_blocks(n), &push, &poll);
    if (timeout_s > 0) poll.timeout = (long long)(timeout_s * 1e8);
    HIPCHK(sml_launch_peer_push(src, n, push, st));
    HIPCHK(sml_launch_peer_sum(dst, n, poll, st));
    return SML_OK;
}

int sml_prof_enable(sml_ctx* ctx, int on) {
    if (!ctx) return fail(SML_EINVAL, "sml_prof_enable", "null ctx");
    DevGuard g(ctx->device);
    if (!on) ctx->prof.drain();
    ctx->prof.on = on != 0;
    return SML_OK;
}
int sml_prof_reset(sml_ctx* ctx) {
    if (!ctx) return fail(SML_EINVAL, "sml_prof_reset", "null ctx");
    DevGuard g(ctx->device);
    ctx->prof.drain();
    for (int i = 0; i < PC_COUNT; ++i) { ctx->prof.total_ms[i] = 0; ctx->prof.count[i] = 0; }
    return SML_OK;
}
int sml_prof_pair_overhead(sml_ctx* ctx, int n, void* stream, double* avg_us) {
    if (!ctx || n <= 0 || n > 4096 || !avg_us) return fail(SML_EINVAL, "sml_prof_pair_overhead", "bad argument");
    DevGuard g(ctx->device);
    hipStream_t st = (hipStream_t)stream;
    std::vector<hipEvent_t> ev((size_t)2 * n);
    for (auto& e : ev) HIPCHK(hipEventCreate(&e));
    // n EMPTY pairs: what a pair reads with nothing in between -- the part of a bracketed launch's reading that is not the kernel
    for (int i = 0; i < n; ++i) { HIPCHK(hipEventRecord(ev[2 * i], st)); HIPCHK(hipEventRecord(ev[2 * i + 1], st)); }
    HIPCHK(hipEventSynchronize(ev.back()));
    double tot = 0.0;
    for (int i = 0; i < n; ++i) { float ms = 0.f; HIPCHK(hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1])); tot += ms; }
    for (auto& e : ev) (void)hipEventDestroy(e);
    *avg_us = 1000.0 * tot / n;
    return SML_OK;
}
int sml_prof_classes(void) { return PC_COUNT; }
const char* sml_prof_name(int cls) { return (cls >= 0 && cls < PC_COUNT) ? kProfNames[cls] : ""; }
int sml_prof_get(sml_ctx* ctx, int cls, int64_t* count, double* total_ms) {
    if (!ctx || cls < 0 || cls >= PC_COUNT || !count || !total_ms) return fail(SML_EINVAL, "sml_prof_get", "bad argument");
    DevGuard g(ctx->device);
    ctx->prof.drain();
    *count = ctx->prof.count[cls];
    *total_ms = ctx->prof.total_ms[cls];
    return SML_OK;
}

int sml_host_resolve_negatives(const int64_t* users, int64_t n, const int64_t* cand, int64_t m,
                               const int64_t* pairs, int64_t n_pairs, int64_t stride, int64_t* negs,
                               int64_t* consumed, int64_t* resolved) {
    if (!users || !cand || !pairs || !negs || !consumed || !resolved || n < 0 || m < 0 || n_pairs < 0 || stride <= 0)
        return fail(SML_EINVAL, "sml_host_resolve_negatives", "bad argument");
    int64_t ptr = 0, e = 0;
    for (; e < n; ++e) {
        bool placed = false;
        while (ptr < m) {
            const int64_t c = cand[ptr++];
            const int64_t code = users[e] * stride + c;
            int64_t lo = 0, hi = n_pairs;                      // is (user, candidate) one of the user's own pairs?
            while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (pairs[mid] < code) lo = mid + 1; else hi = mid; }
            if (!(lo < n_pairs && pairs[lo] == code)) { negs[e] = c; placed = true; break; }
        }
        if (!placed) break;                                    // stream ran out inside element e
    }
    *consumed = ptr;
    *resolved = e;
    return SML_OK;
}

int sml_sample_negatives(sml_ctx* ctx, const int64_t* users, int64_t n, const int64_t* item_all, int64_t pop,
                         const int64_t* user_ptr, int64_t n_users, const int64_t* user_items, uint64_t seed, int64_t* negs,
                         int32_t* failed, void* stream) {
    if (!ctx || !users || !item_all || !user_ptr || !user_items || !negs || !failed || n < 0 || pop <= 0 || n_users < 0)
        return fail(SML_EINVAL, "sml_sample_negatives", "bad argument");
    if (n == 0) return SML_OK;
    DevGuard g(ctx->device);
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipMemsetAsync(failed, 0, sizeof(int32_t), st));
    ctx->prof.begin(PC_MISC, st);
    HIPCHK(sml_launch_sample_negatives(users, n, item_all, pop, user_ptr, n_users, user_items, seed, negs, failed, st));
    ctx->prof.end(st);
    return SML_OK;
}

int sml_device_epoch(sml_ctx* ctx, const int64_t* ui, const void* mat, int elem_bytes, int64_t row_stride, int64_t col, int64_t n,
                     uint64_t seed, int64_t* out3, void* stream) {
    if (!ctx || !ui || !out3 || n < 0 || (mat && (elem_bytes != 4 && elem_bytes != 8)) || (mat && (row_stride <= 0 || col < 0 || col >= row_stride)))
        return fail(SML_EINVAL, "sml_device_epoch", "bad argument");
    DevGuard g(ctx->device);
    HIPCHK(sml_launch_device_epoch(ui, mat, elem_bytes, row_stride, col, n, seed, out3, (hipStream_t)stream));
    return SML_OK;
}

int sml_host_resolve_negatives_csr(const int64_t* users, int64_t n, const int64_t* cand, int64_t m,
                                   const int64_t* user_ptr, int64_t n_users, const int64_t* user_items, int64_t* negs,
                                   int64_t* consumed, int64_t* resolved) {
    if (!users || !cand || !user_ptr || !user_items || !negs || !consumed || !resolved || n < 0 || m < 0 || n_users < 0)
        return fail(SML_EINVAL, "sml_host_resolve_negatives_csr", "bad argument");
    int64_t ptr = 0, e = 0;
    for (; e < n; ++e) {
        // (the walk is sequential in the candidate stream; the users are known ahead: their list heads are prefetched)
        if (e + 16 < n) { const int64_t u2 = users[e + 16]; if (u2 >= 0 && u2 < n_users) __builtin_prefetch(user_ptr + u2); }
        if (e + 8 < n) { const int64_t u1 = users[e + 8]; if (u1 >= 0 && u1 < n_users) __builtin_prefetch(user_items + user_ptr[u1]); }
        const int64_t u = users[e];
        int64_t b = 0, t = 0;
        if (u >= 0 && u < n_users) { b = user_ptr[u]; t = user_ptr[u + 1]; }     // the user's own items, ascending
        bool placed = false;
        while (ptr < m) {
            const int64_t c = cand[ptr++];
            bool own = false;
            if (t - b <= 8) { for (int64_t q = b; q < t; ++q) own |= (user_items[q] == c); }
            else {
                int64_t lo = b, hi = t;
                while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (user_items[mid] < c) lo = mid + 1; else hi = mid; }
                own = lo < t && user_items[lo] == c;
            }
            if (!own) { negs[e] = c; placed = true; break; }
        }
        if (!placed) break;                                    // stream ran out inside element e
    }
    *consumed = ptr;
    *resolved = e;
    return SML_OK;
}

// Host-side gathers of the reference-exact batch supply (data/dataset2.py:172-201, data/dataset.py:41-71): the epoch's
// (user, item) pairs in loader order straight into the [n,3] triple array, and one column of a row-major integer matrix
// into one of its columns.  Plain loops with software prefetch: numpy's fancy indexing builds a temporary and copies it
// through a strided view (1.9 ms per 75,000-row epoch against 0.2 ms here).
int sml_host_gather_pairs(const int64_t* ui, int64_t n_rows, const int64_t* order, int64_t n, int64_t* out3) {
    if (!ui || !order || !out3 || n < 0 || n_rows < 0) return fail(SML_EINVAL, "sml_host_gather_pairs", "bad argument");
    for (int64_t e = 0; e < n; ++e) {
        if (e + 16 < n) __builtin_prefetch(ui + 2 * order[e + 16]);
        const int64_t r = order[e];
        if (r < 0 || r >= n_rows) return fail(SML_EINVAL, "sml_host_gather_pairs", "index outside the data");
        out3[3 * e] = ui[2 * r]; out3[3 * e + 1] = ui[2 * r + 1];
    }
    return SML_OK;
}
int sml_host_gather_column(const void* mat, int64_t n_rows, int64_t row_stride_bytes, int elem_bytes, int64_t col,
                           const int64_t* order, int64_t n, int64_t* out3, int out_col) {
    if (!mat || !order || !out3 || n < 0 || (elem_bytes != 4 && elem_bytes != 8) || out_col < 0 || out_col > 2 || col < 0)
        return fail(SML_EINVAL, "sml_host_gather_column", "bad argument");
    const char* base = reinterpret_cast<const char*>(mat) + col * elem_bytes;
    for (int64_t e = 0; e < n; ++e) {
        if (e + 16 < n) __builtin_prefetch(base + order[e + 16] * row_stride_bytes);
        const int64_t r = order[e];
        if (r < 0 || r >= n_rows) return fail(SML_EINVAL, "sml_host_gather_column", "index outside the data");
        const char* p = base + r * row_stride_bytes;
        out3[3 * e + out_col] = elem_bytes == 8 ? *reinterpret_cast<const int64_t*>(p) : (int64_t)*reinterpret_cast<const int32_t*>(p);
    }
    return SML_OK;
}

int sml_stream_create_cu_range(void** stream, int device, int cu_lo, int cu_hi) {
    if (!stream) return fail(SML_EINVAL, "sml_stream_create_cu_range", "null stream pointer");
    DevGuard g(device);
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    const int n_cu = prop.multiProcessorCount;
    if (cu_lo < 0 || cu_hi > n_cu || cu_lo >= cu_hi) return fail(SML_EINVAL, "sml_stream_create_cu_range", "range outside the device's CUs");
    std::vector<uint32_t> mask((size_t)(n_cu + 31) / 32, 0u);
    for (int b = cu_lo; b < cu_hi; ++b) mask[(size_t)b / 32] |= 1u << (b % 32);
    hipStream_t st = nullptr;
    HIPCHK(hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()));
    *stream = st;
    return SML_OK;
}
int sml_stream_wait_stream(void* waiter, void* signaler) {
    // work queued on `waiter` after this call starts after everything queued on `signaler` so far.  The event is
    // device-scope (no system fence: nothing on the host reads what the signaler wrote) -- a default event makes the
    // signalling stream write back and invalidate the L2s for the host's benefit
    static const unsigned flags = getenv("SML_EVENT_FLAGS") ? (unsigned)strtoul(getenv("SML_EVENT_FLAGS"), nullptr, 0)
                                                            : (hipEventDisableTiming | hipEventDisableSystemFence);
    hipEvent_t ev;
    HIPCHK(hipEventCreateWithFlags(&ev, flags));
    hipError_t e = hipEventRecord(ev, (hipStream_t)signaler);
    if (e == hipSuccess) e = hipStreamWaitEvent((hipStream_t)waiter, ev, 0);
    (void)hipEventDestroy(ev);                   // released once it has completed
    if (e != hipSuccess) return fail(SML_EHIP, "sml_stream_wait_stream", hipGetErrorString(e));
    return SML_OK;
}
int sml_copy_tables(int n, void* const* dst, const void* const* src, const int64_t* bytes, void* stream) {
    if (n < 0 || n > 4 || (n && (!dst || !src || !bytes))) return fail(SML_EINVAL, "sml_copy_tables", "1..4 copies per call");
    long long b[4] = {0, 0, 0, 0};
    for (int q = 0; q < n; ++q) {
        if (bytes[q] == 0) continue;                 // an empty job: its pointers are not looked at
        if (!dst[q] || !src[q] || bytes[q] < 0 || bytes[q] % 16 || ((uintptr_t)dst[q] | (uintptr_t)src[q]) % 16)
            return fail(SML_EINVAL, "sml_copy_tables", "pointers and sizes must be multiples of 16 bytes");
        b[q] = bytes[q];
    }
    HIPCHK(sml_launch_copy_tables(n, dst, src, b, (hipStream_t)stream));
    return SML_OK;
}
int sml_debug_timeline(long long* buf) { return sml_debug_set_timeline(buf) == hipSuccess ? 0 : -1; }
int sml_flag_set(int32_t* flag, int value, void* stream) {
    if (!flag || value < 0) return fail(SML_EINVAL, "sml_flag_set", "bad argument");
    HIPCHK(sml_launch_flag_set(flag, value, (hipStream_t)stream));
    return SML_OK;
}
int sml_flag_wait(int32_t* flag, int value, double timeout_s, void* stream) {
    if (!flag || value < 0 || !(timeout_s > 0)) return fail(SML_EINVAL, "sml_flag_wait", "bad argument");
    HIPCHK(sml_launch_flag_wait(flag, value, (long long)(timeout_s * 1e8), (hipStream_t)stream));   // wall_clock64: 100 MHz
    return SML_OK;
}
int sml_stream_destroy(void* stream) {
    if (stream) HIPCHK(hipStreamDestroy((hipStream_t)stream));
    return SML_OK;
}

int sml_selftest(int device) {
    DevGuard g(device);
    const int M = 16, K = 32;
    std::vector<float> A(M * K), W(M * K), ref(M * M, 0.f), got(M * M, 0.f);
    for (int i = 0; i < M * K; ++i) {
        A[i] = (float)((i * 37 + 11) % 23) - 11.0f;          // asymmetric integer data: exact in fp32
        W[i] = (float)((i * 53 + 5) % 19) - 9.0f + (float)(i / K);
    }
    for (int r = 0; r < M; ++r)
        for (int c = 0; c < M; ++c) {
            float s = 0.f;
            for (int k = 0; k < K; ++k) s += A[r * K + k] * W[c * K + k];
            ref[r * M + c] = s;
        }
    float *dA = nullptr, *dW = nullptr, *dP = nullptr, *dO = nullptr;
    HIPCHK(hipMalloc((void**)&dA, A.size() * 4)); HIPCHK(hipMalloc((void**)&dW, W.size() * 4));
    HIPCHK(hipMalloc((void**)&dP, 512 * 4)); HIPCHK(hipMalloc((void**)&dO, got.size() * 4));
    HIPCHK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(sml_launch_selftest(dA, dW, dP, dO, nullptr));
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(got.data(), dO, got.size() * 4, hipMemcpyDeviceToHost));
    (void)hipFree(dA); (void)hipFree(dW); (void)hipFree(dP); (void)hipFree(dO);
    for (int i = 0; i < M * M; ++i)
        if (got[i] != ref[i]) {
            char buf[128];
            snprintf(buf, sizeof(buf), "element (%d,%d): got %g want %g", i / M, i % M, got[i], ref[i]);
            return fail(SML_ESTATE, "sml_selftest: MFMA lane map mismatch", buf);
        }
    return SML_OK;
}

}  // extern "C"
