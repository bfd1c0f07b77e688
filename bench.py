#!/usr/bin/env python3
"""bench.py -- SML per-period retraining throughput on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one full SML retrain period (reference meta_train.train_one_stage3,
model/transfer.py:753-792: multi_num x [MF epoch, updata, TR epoch, updata] with the
validation evaluations the reference performs) on a Yelp-shaped synthetic period
(BASELINE.json configs[1]: U=60,000, I=123,000, 75,000 interactions, 999 negatives,
d=32, MF batch 1024, TR batch 256, multi_num 10).  Inputs (every epoch's triples,
validation rows, tables) are resident in HBM before the timed region.

Prints ONE JSON line on rank 0.  `value` = training triples (MF + TR stage) per second,
whole job.  Extra objects: `roofline` for the kernel that dominates GPU time (measured
with HIP events inside this process) and `cpu_baseline` (the CPU oracle timed on a
bounded sample of the same workload, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3  # v_mfma_f32_32x32x2_f32 dense peak


class _StdoutToStderr(object):
    """fd-level redirect: RCCL prints a version banner to stdout when a communicator is created; the
    bench contract is ONE JSON line on stdout."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    @staticmethod
    def _flush_c():
        # RCCL printf()s into C stdio's buffer (fully buffered on a pipe): push it out while fd 1 still points at
        # stderr, or it surfaces on the real stdout when the process exits
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass

    def __exit__(self, *exc):
        sys.stdout.flush()
        self._flush_c()
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False

    def emit(self, text):
        """Write to the REAL stdout while the redirect stays in place."""
        sys.stdout.flush()
        self._flush_c()
        os.write(self._saved, (text + "\n").encode())


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--d", type=int, default=32)
    ap.add_argument("--users", type=int, default=60000)
    ap.add_argument("--items", type=int, default=123000)
    ap.add_argument("--inter", type=int, default=75000)
    ap.add_argument("--neg", type=int, default=999)
    ap.add_argument("--multi_num", type=int, default=10)
    ap.add_argument("--no-val", action="store_true", help="skip the validation evaluations (not the reference default)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-a3", action="store_true", help="skip the a3 (bare fused embed+loss+SGD) table-scale leg")
    ap.add_argument("--a3-child", action="store_true", help="(internal) run the a3 legs only and print their object: the default run measures "
                                                            "them in a fresh child process (see a3_in_child)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="evaluate in place on the training stream instead of on the side stream (single-queue runs for "
                         "rocprofv3 --pmc, which does not survive this workload's two queues)")
    ap.add_argument("--workload", default="yelp_period", choices=["yelp_period", "bare"],
                    help="yelp_period: the headline SML retrain period (default).  bare: the a3 fused embed+loss+SGD "
                         "step alone on large synthetic tables (HBM roofline study; not the headline metric)")
    ap.add_argument("--bare-batch", type=int, default=65536)
    ap.add_argument("--bare-triples", type=int, default=1 << 22)
    ap.add_argument("--bare-dtype", default="f32", choices=["f32", "f16"])
    ap.add_argument("--item-zipf", type=float, default=1.0)
    ap.add_argument("--bare-items", default="sharded", choices=["sharded", "replicated"],
                    help="several GPUs, --workload bare: item table sharded over the ranks (owner-computes over the peer exchange; "
                         "the form configs 4 / 5 need) or replicated with a per-batch all-gather of gradient rows (exchange-bound "
                         "by design: a labelled comparison)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="several GPUs, yelp_period: weak = every rank its own period over its own user shard (global batch = "
                         "N x the reference's); strong = ONE period with the reference's global batches (1024 / 256) split over "
                         "the ranks by user owner")
    ap.add_argument("--job-timeout", type=float, default=None,
                    help="--gpus N > 1: seconds before the launcher stops the rank processes and exits 124 (default: SML_JOB_TIMEOUT_S, "
                         "else 3600; 0: no limit) -- a hung rank must not hold the node's GPUs forever")
    ap.add_argument("--one-device", action="store_true",
                    help="test mode for a 1-GPU box: all N rank processes share device 0 (gloo carries torch.distributed)")
    return ap.parse_args()


def zipf_head_rows(n_item, zipf_a, occ_per_batch, d, net_size):
    """Rows of a Zipf(a) catalogue worth REPLICATING in the item-sharded step: those expected to occur at least 16 times
    in a global batch (their occurrences would otherwise all land on one owner), capped by what the dense head partial
    may hold (head_rows * d <= 2 * net size, a theta slot of the inbox).  0 for a uniform catalogue."""
    if zipf_a <= 0.0:
        return 0
    ranks = np.arange(1, min(n_item, 1 << 20) + 1, dtype=np.float64)
    h = float((1.0 / np.power(np.arange(1, n_item + 1, dtype=np.float64), zipf_a)).sum())
    want = int((occ_per_batch / np.power(ranks, zipf_a) / h >= 16.0).sum())
    cap = (2 * net_size // d) // 4 * 4
    return int(min(want, cap, n_item))


def bench_bare_sharded(a, device, dist):
    """a3 on several GPUs the way configs 4 / 5 need it (model/baseline.py:188-201 semantics, exact synchronous SGD of the
    GLOBAL batch): users row-sharded, the item table SHARDED too -- a replicated head of the popular rows, the tail
    owner-computes over the one-shot peer exchange (sml_embed_loss_sgd_epoch_sharded).  Weak scaling in the triples:
    every rank brings `bare_triples` triples per epoch and `bare_batch` per batch over a table of FIXED total size.
    Returns the result dict, or None when the peer exchange is not available (the caller then runs the replicated form)."""
    from sml_amd import dist as smldist
    from sml_amd import synth
    from sml_amd.engine import HipEngine
    world, rank = dist.get_world_size(), dist.get_rank()
    eng = HipEngine(device, a.d, a.bare_batch)
    ctx = smldist.attach(eng, None, dist, rows_cap=2 * a.bare_batch)
    ok = ctx.mode == "peer" and smldist.shard_visibility_check(eng, dist)
    if not ok:
        if rank == 0:
            print("[bench] item-sharded bare step unavailable (carrier %s, shard visibility %s): replicated form instead"
                  % (ctx.mode, "not checked" if ctx.mode != "peer" else "FAILED"), file=sys.stderr)
        eng.peer_detach()
        return None
    dt = torch.float32 if a.bare_dtype == "f32" else torch.float16
    users_local = -(-a.users // world)
    head = zipf_head_rows(a.items, a.item_zipf, 2 * a.bare_batch * world, a.d, eng.net_size)
    H, S = smldist.item_shard_layout(a.items, world, head)
    g = torch.Generator(device=device).manual_seed(4)
    w_head = (torch.randn(max(H, 1), a.d, device=device, generator=g) * 0.1).to(dt)[:H]      # the same replica on every rank
    g.manual_seed(400 + rank)
    shard = eng.peer_tensor((S, a.d), dt)
    shard.copy_((torch.randn(S, a.d, device=device, generator=g) * 0.1).to(dt))
    g.manual_seed(40 + rank)
    wu = (torch.randn(users_local, a.d, device=device, generator=g) * 0.1).to(dt)
    rng = np.random.RandomState(4 + 1000 * rank)
    u, i, j = synth.synth_triples(rng, a.bare_triples, users_local, a.items, a_user=0.0, a_item=a.item_zipf)
    tri = torch.from_numpy(np.stack([u, i, j], 1)).to(device)
    sh = ctx.bare_shard(eng, tri, a.items, head, w_head, shard, 0)

    def barrier():
        dist.barrier()
        torch.cuda.synchronize(device)

    for _ in range(a.warmup):
        eng.bare_epoch_sharded(wu, tri, a.bare_batch, 0.05, 1e-6, 1e-6, sh, bce=True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        eng.bare_epoch_sharded(wu, tri, a.bare_batch, 0.05, 1e-6, 1e-6, sh, bce=True)
    barrier()
    dtm = time.perf_counter() - t0
    tt = torch.tensor([dtm], device=device, dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dtm = float(tt.item())
    eng.profile(True)
    eng.bare_epoch_sharded(wu, tri, a.bare_batch, 0.05, 1e-6, 1e-6, sh, bce=True)
    torch.cuda.synchronize(device)
    prof = eng.profile_read()
    eng.profile(False)
    ctx.check_exchange(eng, "bench bare (item-sharded)", replicas=[w_head] if H > 0 else [])     # raises on a time-out / drifted head
    s = wu.element_size()
    a_sgd = 24 + 6 * a.d * s
    n = a.bare_triples
    t_k = sum(prof.get(k, (0, 0.0))[1] for k in ("k_bare_grad", "k_seg_update_sgd", "k_hot_rows")) / 1e3
    ach = n * a_sgd / t_k / 1e9 if t_k > 0 else None
    e2e = a.steps * n / dtm * a_sgd / 1e9
    tail_frac = float(((i >= H).mean() + (j >= H).mean()) / 2.0)
    remote = tail_frac * (world - 1) / world
    out = {"metric": "bare embed+loss+SGD step triples/s (a3), synthetic uniform users / Zipf(%g) items, d=%d %s" % (a.item_zipf, a.d, a.bare_dtype),
           "value": world * a.steps * n / dtm, "unit": "triples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
           "ms_per_step": 1000.0 * dtm / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": a.bare_dtype, "data": "synthetic",
           "config": {"workload": "bare: users=%d (row-sharded over %d GPU(s)) items=%d (SHARDED: %d replicated head rows + %d tail rows "
                                  "per rank) triples/epoch/GPU=%d batch/GPU=%d (global batch %d); index lists built inline every epoch"
                                  % (a.users, world, a.items, H, S, n, a.bare_batch, world * a.bare_batch),
                      "parallelism": "users row-sharded x%d, items sharded x%d with a replicated head: tail rows read from / gradient "
                                     "rows stored to their owner over peer mappings (owner-computes), dense one-shot all-reduce of the "
                                     "head" % (world, world),
                      "carrier": ctx.mode, "peer_timeouts": int(eng.peer_status()),
                      "remote_item_row_fraction": remote},
           "roofline": {"kernel": "k_bare_grad<SH> + k_run_update (owner update) per GPU", "bound": "hbm", "achieved": ach,
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS if ach else None, "traffic": None,
                        "end_to_end_achieved": e2e, "end_to_end_frac": e2e / HBM_PEAK_GBS,
                        "algorithmic_bytes_per_triple": a_sgd,
                        # what crosses xGMI per triple: the remote share of 2 item rows read + 2 fp32 gradient rows stored
                        "xgmi_bytes_per_triple": remote * (2 * a.d * s + 2 * a.d * 4),
                        "xgmi_GBps_per_gpu_end_to_end": remote * (2 * a.d * s + 2 * a.d * 4) * a.steps * n / dtm / 1e9},
           "kernels": {k: {"launches": c, "total_ms": round(m, 3), "avg_us": round(1000.0 * m / c, 2)} for k, (c, m) in prof.items()}}
    dist.barrier()                       # nobody frees a shard a peer may still read
    del eng, wu, tri, sh
    torch.cuda.empty_cache()
    return out


def bench_bare(a, device, dist=None):
    """a3 alone: synchronous minibatch SGD on (user, pos, neg) triples over large tables.  Returns the result dict.
    dist (several GPUs, weak scaling), the REPLICATED-items comparison form: the `users` rows are sharded over the ranks (each
    rank holds users/world rows and draws `bare_triples` triples over them), the item table is replicated, every global batch
    is the union of the ranks' batches: item-gradient rows are all-gathered per batch and every rank applies the identical item
    update -- exchange-bound by a factor of ten by design; bench_bare_sharded is the form that scales."""
    from sml_amd import synth
    from sml_amd.engine import HipEngine
    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0
    eng = HipEngine(device, a.d, a.bare_batch)
    dt = torch.float32 if a.bare_dtype == "f32" else torch.float16
    users_local = -(-a.users // world)
    g = torch.Generator(device=device).manual_seed(4)
    wi = (torch.randn(a.items, a.d, device=device, generator=g) * 0.1).to(dt)         # the same replica on every rank
    g.manual_seed(40 + rank)
    wu = (torch.randn(users_local, a.d, device=device, generator=g) * 0.1).to(dt)
    rng = np.random.RandomState(4 + 1000 * rank)
    u, i, j = synth.synth_triples(rng, a.bare_triples, users_local, a.items, a_user=0.0, a_item=a.item_zipf)
    tri = torch.from_numpy(np.stack([u, i, j], 1)).to(device)
    ex = None
    if dist is not None:
        from sml_amd import dist as smldist
        os.environ.setdefault("SML_COMM", "rccl")       # (the replicated form is not wired to the inboxes: RCCL, else the hooks)
        ctx = smldist.attach(eng, None, dist)
        ex = ctx.bare_exchange(tri, a.bare_batch, a.d, 0)         # item columns gathered once: the triples are reused every epoch

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(device)

    # the index lists of epoch e+1 (sort, unique marks, compaction) are built on a side stream while epoch e runs
    for _ in range(a.warmup):
        eng.bare_epoch(wu, wi, tri, a.bare_batch, 0.05, 1e-6, 1e-6, bce=True, exchange=ex)
    nxt = eng.bare_prepare(tri, a.bare_batch, users_local, a.items, exchange=ex)
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        cur, nxt = nxt, eng.bare_prepare(tri, a.bare_batch, users_local, a.items, exchange=ex)
        eng.bare_epoch(wu, wi, tri, a.bare_batch, 0.05, 1e-6, 1e-6, bce=True, prepared=cur)
    barrier()
    dtm = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dtm], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dtm = float(tt.item())
    # the index preparation alone (nothing else on the chip): its own time and rate
    barrier()
    tp0 = time.perf_counter()
    for _ in range(5):
        cur = eng.bare_prepare(tri, a.bare_batch, users_local, a.items, exchange=ex)
    torch.cuda.synchronize(device)
    prep_us = (time.perf_counter() - tp0) / 5 * 1e6
    cur = eng.bare_prepare(tri, a.bare_batch, users_local, a.items, exchange=ex)      # (prepared ahead, as in the timed loop)
    torch.cuda.synchronize(device)
    eng.profile(True)
    eng.bare_epoch(wu, wi, tri, a.bare_batch, 0.05, 1e-6, 1e-6, bce=True, prepared=cur)
    torch.cuda.synchronize(device)
    prof = eng.profile_read()
    eng.profile(False)
    # the same step WITHOUT a bracket per launch: one HIP-event pair around a whole prepared epoch (its index lists finished
    # and nothing else on the chip), divided by the epoch's batches.  A bracket per launch adds 1-3 us of its own to a 14-39 us
    # kernel (the period's roofline object measures and subtracts that; here both readings are given, neither corrected);
    # this one contains everything the epoch's stream does -- both step kernels of every batch, their boundaries, the
    # per-epoch loss reduction -- so it is an upper bound of the kernels' own time
    ev_us = []
    for _ in range(4):
        cur = eng.bare_prepare(tri, a.bare_batch, users_local, a.items, exchange=ex)
        torch.cuda.synchronize(device)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        eng.bare_epoch(wu, wi, tri, a.bare_batch, 0.05, 1e-6, 1e-6, bce=True, prepared=cur)
        e1.record()
        torch.cuda.synchronize(device)
        ev_us.append(1000.0 * e0.elapsed_time(e1))
    s = wu.element_size()
    a_sgd = 24 + 6 * a.d * s                       # int64 (u,i,j) + 3 rows read + 3 rows written
    n = a.bare_triples
    nb_epoch = -(-n // a.bare_batch)
    step_us_epoch = float(np.median(ev_us)) / nb_epoch
    ach_epoch = n * a_sgd / nb_epoch / (step_us_epoch * 1e-6) / 1e9
    # k_bare_grad = the fused pass (unique rows in place, duplicated rows by their last arriver); the other two classes
    # exist only in epochs with hot runs (chunk partial sums, per-row apply)
    t_grad, t_seg = prof["k_bare_grad"][1] / 1e3, prof.get("k_seg_update_sgd", (0, 0.0))[1] / 1e3
    t_hot = prof.get("k_hot_rows", (0, 0.0))[1] / 1e3
    ach = n * a_sgd / (t_grad + t_seg + t_hot) / 1e9
    e2e = a.steps * n / dtm * a_sgd / 1e9                 # whole step incl. the index preparation on its side stream (per GPU)
    out = {"metric": "bare embed+loss+SGD step triples/s (a3), synthetic uniform users / Zipf(%g) items, d=%d %s" % (a.item_zipf, a.d, a.bare_dtype),
           "value": world * a.steps * n / dtm, "unit": "triples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
           "ms_per_step": 1000.0 * dtm / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": a.bare_dtype, "data": "synthetic",
           "config": {"workload": "bare: users=%d (row-sharded over %d GPU(s)) items=%d (replicated) triples/epoch/GPU=%d batch/GPU=%d"
                                  % (a.users, world, a.items, n, a.bare_batch),
                      "parallelism": "single GPU" if world == 1 else
                      "COMPARISON FORM (exchange-bound by design): users row-sharded x%d, items replicated, per-batch all-gather of "
                      "item-gradient rows (%s)" % (world, {"rccl": "native RCCL", "torch": "torch.distributed hooks"}.get(ctx.mode, "torch.distributed hooks")),
                      "carrier": ctx.mode if dist is not None else None},
           # traffic (PMC bytes) cannot be read from inside the process: null here; the rocprofv3 --pmc passes of this
           # command are tools/profile_round.sh's, summarised under profiles/
           "roofline": {"kernel": "k_bare_grad + k_seg_update_sgd + k_hot_rows (one a3 step)", "bound": "hbm", "achieved": ach,
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                        "end_to_end_achieved": e2e, "end_to_end_frac": e2e / HBM_PEAK_GBS,
                        # one event pair around a prepared epoch / its batches (no bracket per launch; index lists ready before)
                        "step_us_epoch_events": step_us_epoch, "frac_epoch_events": ach_epoch / HBM_PEAK_GBS,
                        "algorithmic_bytes_per_step": a_sgd * min(a.bare_batch, n), "algorithmic_bytes_per_triple": a_sgd},
           "kernels": {k: {"launches": c, "total_ms": round(m, 3), "avg_us": round(1000.0 * m / c, 2)} for k, (c, m) in prof.items()},
           # index lists of an epoch (index_prep.hip), alone on the chip: wall per epoch; algorithmic rate = the triples once
           # + one mark per occurrence (27 B per triple); fabric bytes per epoch from the committed counter passes
           "index_prep": dict({"us_per_epoch": prep_us, "algorithmic_GBps": n * 27 / (prep_us * 1e-6) / 1e9}, **prep_fabric())}
    del eng, wu, wi, tri
    torch.cuda.empty_cache()
    return out


A3_CONFIGS = (   # (tag, users, items, d, dtype, item zipf): BASELINE.json configs 4 / 5 shapes and the d=32 target of north_star
    ("d32_f32_uniform", 10000000, 1000000, 32, "f32", 0.0),
    ("d32_f32_zipf", 10000000, 1000000, 32, "f32", 1.0),
    ("d64_f32_zipf", 10000000, 1000000, 64, "f32", 1.0),
    ("d128_f16_uniform", 50000000, 5000000, 128, "f16", 0.0),
)


def prep_fabric():
    """Fabric bytes per prepared epoch (FETCH x 2 + WRITE of the k_prep_* kernels) from the latest committed
    profiles/r*_index_prep.json (tools/profile_round.sh; 10M x 1M, d = 32, uniform items)."""
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_index_prep.json")))
    if not files:
        return {}
    try:
        z = json.load(open(files[-1])).get("bare_z0", {})
        return {"fabric_bytes_per_epoch_from_profiles": z.get("fabric_bytes_per_epoch"), "fabric_profile": os.path.basename(files[-1])}
    except Exception:
        return {}


def gather_ceiling(d, dtype):
    """Measured ceiling of the a3 access pattern for this row shape: three random row reads + three in-place row writes per
    triple with nothing else in the kernel (tools/micro_gather.hip), from the latest committed profiles/r*_micro_gather.json.
    -> {"ceiling_GBps", "ceiling_profile"} or {}."""
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_micro_gather.json")))
    if not files:
        return {}
    try:
        z = json.load(open(files[-1])).get("d%d_%s" % (d, "fp32" if dtype == "f32" else "fp16"))
        return {"ceiling_GBps": 1000.0 * z["reads_writes"]["TBps"], "ceiling_profile": os.path.basename(files[-1])}
    except Exception:
        return {}


def a3_in_child(a, device):
    """The a3 legs in a FRESH process of their own (a child of this one; its JSON is merged into the line).  The period workload
    partitions the chip with CU-masked streams, and the HIP runtime hands streams to a small pool of hardware queues: in a process
    that has had masked queues, the a3 legs' preparation stream can end up on a 64-CU queue -- seen again in round 5 although the
    stream is created before any masked one: the first a3 leg behind the period ran end to end at HALF speed (0.20 instead of
    0.38; its kernels' own times unchanged) in one of three default runs.  A process that never creates a masked stream cannot
    inherit one.  (This process keeps running and waits: a child process, not an exec.)  SML_A3_INPROC=1: in this process, as before."""
    import subprocess
    if os.environ.get("SML_A3_INPROC") == "1":
        return a3_object(a, device)
    torch.cuda.synchronize(device)
    torch.cuda.empty_cache()
    cmd = [sys.executable, os.path.abspath(__file__), "--a3-child"]
    try:
        p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=1500)
        lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
        if p.returncode == 0 and lines:
            return json.loads(lines[-1])
        sys.stderr.write("[bench] the a3 child failed (rc %d): %s\n" % (p.returncode, p.stderr[-2000:]))
    except Exception as e:      # noqa: BLE001
        sys.stderr.write("[bench] the a3 child could not run: %s\n" % e)
    return a3_object(a, device)


def a3_rocprof(tag, bytes_per_batch):
    """The a3 step's kernels in the COMMITTED rocprofv3 --kernel-trace --stats summary of `bench.py --workload bare` at the same
    shape (profiles/r*_bare_z0 / _z1_kernel_stats.csv, tools/profile_round.sh): steady-state µs per batch = k_bare_grad + every
    step-kernel instance launched at least half as often (the first epoch's variants with the hot-row early-out are left out),
    and the fraction of the 8 TB/s roofline that gives.  A HIP-event pair reads a kernel plus 1-3 us of its own: both are quoted."""
    import csv
    import glob
    leg = {"d32_f32_uniform": "bare_z0", "d32_f32_zipf": "bare_z1"}.get(tag)
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_%s_kernel_stats.csv" % leg))) if leg else []
    if not files:
        return {}
    try:
        rows = [(r["kernel"], int(r["calls"]), float(r["avg_us"])) for r in csv.DictReader(open(files[-1]))
                if r["kernel"].startswith(("k_bare_grad", "k_run_update", "k_hot_"))]
        base = max((c for k, c, _ in rows if k.startswith("k_bare_grad")), default=0)
        us = sum(avg for _, c, avg in rows if base and 2 * c >= base)
        if not us:
            return {}
        return {"kernel_us_rocprof": us, "kernel_frac_rocprof": bytes_per_batch / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                "rocprof_profile": os.path.basename(files[-1])}
    except Exception:
        return {}


def a3_object(a, device):
    """The fused embed+loss+SGD kernel pair (north_star's HBM-roofline target) at table scale,
    measured in this same run: per configuration the kernel-only fraction of the 8 TB/s roofline (HIP events over
    k_bare_grad + k_run_update + k_hot_apply), the end-to-end fraction (wall clock of whole epochs, index
    preparation included) and the algorithmic bytes per triple (24 + 6*d*s)."""
    import copy
    res = {}
    order = os.environ.get("SML_A3_ORDER")          # (measurement aid: the legs in another order / a subset)
    cfgs = A3_CONFIGS if not order else [c for t in order.split(",") for c in A3_CONFIGS if c[0] == t]
    for tag, users, items, d, dt, zipf in cfgs:
        while tag in res:
            tag += "'"
        b = copy.copy(a)
        b.users, b.items, b.d, b.bare_dtype, b.item_zipf = users, items, d, dt, zipf
        # (an epoch is about a millisecond: 4 untimed + 16 timed epochs per configuration -- the first epochs after the
        # tables are allocated run up to 25 % slower than the steady state)
        b.bare_batch, b.bare_triples, b.steps, b.warmup = 262144, 1 << 22, 16, 4
        try:
            r = bench_bare(b, device)
            res[tag] = {"users": users, "items": items, "d": d, "dtype": dt, "item_zipf": zipf, "batch": b.bare_batch,
                        "epochs_timed": b.steps, "epochs_warmup": b.warmup, "ms_per_epoch": r["ms_per_step"],
                        "triples_per_s": r["value"], "bytes_per_triple": r["roofline"]["algorithmic_bytes_per_triple"],
                        "kernel_frac": r["roofline"]["frac"], "end_to_end_frac": r["roofline"]["end_to_end_frac"],
                        "step_us_epoch_events": r["roofline"]["step_us_epoch_events"], "kernel_frac_epoch_events": r["roofline"]["frac_epoch_events"],
                        "kernel_GBps": r["roofline"]["achieved"],
                        "kernels_avg_us": {k: v["avg_us"] for k, v in r["kernels"].items()},
                        "index_prep": r.get("index_prep")}
            # the kernels' algorithmic rate against the MEASURED rate of the bare access pattern (random rows read and
            # rewritten in place, tools/micro_gather.hip) -- what "of the 8 TB/s roofline" cannot say for 128-byte random rows
            res[tag].update(a3_rocprof(tag, r["roofline"]["algorithmic_bytes_per_triple"] * b.bare_batch))
            c = gather_ceiling(d, dt)
            if c:
                res[tag].update(c, ceiling_frac=r["roofline"]["achieved"] / c["ceiling_GBps"])
        except Exception as e:      # noqa: BLE001 -- e.g. a smaller-memory part: report, do not lose the headline line
            res[tag] = {"error": "%s: %s" % (type(e).__name__, e)}
            torch.cuda.empty_cache()
    return res


def build_state(engine, U, I, d, device, seed):
    from sml_amd.conv_transfer import ConvTransfer_com
    from sml_amd.mf import MFbasemode
    from sml_amd.period import PeriodState
    import contextlib, io
    torch.manual_seed(seed)
    mf = MFbasemode(U, I, d)
    with torch.no_grad():   # pretrained-MF-like scale (scores not saturated)
        mf.user_laten.weight.mul_(0.3)
        mf.item_laten.weight.mul_(0.3)
    with contextlib.redirect_stdout(io.StringIO()):
        net = ConvTransfer_com(d, d)
    mf, net = mf.to(device), net.to(device)
    mf._sml_engine = engine
    net._sml_engine = engine
    engine.adopt(net)
    return PeriodState(mf, net)


def kernel_work(name, a, hp, U_local, side_forwards=0):
    """Algorithmic work of one period per kernel class: (bound, unit_work_total, launches_expected).
    Bytes for HBM-bound kernels, FLOP for the MFMA-bound ones.
    side_forwards: table-sized forward LAUNCHES that ran on the evaluation stream under their own class
    (k_side_transfer_fwd: two per evaluation-only updata) -- their rows are that class's, not k_transfer_fwd's."""
    d, n = a.d, a.inter
    nb_mf = -(-n // hp.MF_batch_size)
    nb_tr = -(-n // hp.TR_batch_size)
    evals = 0 if a.no_val else hp.multi_num * (2 + hp.MF_epochs + hp.TR_epochs)
    n_updata = hp.multi_num * (1 + (hp.TR_epochs if not a.no_val else 0)) + 1
    if name == "k_side_transfer_fwd":
        return "mfma", (side_forwards / 2.0) * (U_local + a.items) * 6304.0 * d, side_forwards
    n_updata -= side_forwards / 2.0
    rows_fwd = hp.multi_num * (hp.MF_epochs + hp.TR_epochs) * 3 * n + n_updata * (U_local + a.items)
    f_row = 6304.0 * d                      # conv prologue + fc1 + fc2 forward, FLOP per row
    b_row = (2.0 * d * 512 + 2.0 * 512 * 5 * d) + 160.0 * d   # dA2 + dA1 GEMMs + per-coordinate tail
    if name == "k_eval_ranks":
        per = n * ((2 + a.neg) * d * 4 + (2 + a.neg) * 4 + 4)     # candidate rows + int32 candidate ids + the rank
        return "hbm", per * evals, evals
    if name == "k_transfer_fwd":
        return "mfma", rows_fwd * f_row, hp.multi_num * (hp.MF_epochs * nb_mf + hp.TR_epochs * nb_tr) + 2 * n_updata
    if name == "k_transfer_bwd":
        return "mfma", hp.multi_num * (hp.MF_epochs + hp.TR_epochs) * 3 * n * b_row, hp.multi_num * (hp.MF_epochs * nb_mf + hp.TR_epochs * nb_tr)
    if name == "k_transfer_wgrad":
        return "mfma", hp.multi_num * hp.TR_epochs * 3 * n * (2.0 * 512 * 5 * d + 2.0 * 512 * d), hp.multi_num * hp.TR_epochs * nb_tr
    if name == "k_seg_update_adam":
        # per occurrence: gradient row read + (p, m, v) read and write
        return "hbm", hp.multi_num * hp.MF_epochs * 3 * n * (d * 4 * 7 + 12), hp.multi_num * hp.MF_epochs * nb_mf
    if name == "k_adam_flush":
        return "hbm", hp.multi_num * hp.MF_epochs * (U_local + a.items) * (d * 4 * 6 + 8), 2 * hp.multi_num * hp.MF_epochs
    return None, 0.0, 0


# kernel names a timing class covers in the rocprofv3 summaries (the restructured TR step's kernels carry their own names)
CLASS_KERNELS = {"k_transfer_wgrad": ("k_transfer_wgrad", "k_tr_wgrad2"), "k_transfer_bwd": ("k_transfer_bwd", "k_tr_bwd_head"),
                 # round 5: the MF stage's forward and the table-sized forwards at d = 32 run on bf16x3 products under their own kernel names
                 # (the evaluation stream's table-sized forwards are the <32,true> instantiation)
                 "k_transfer_fwd": ("k_transfer_fwd<", "k_mf_fwd_bx3", "k_transfer_fwd_bx3<32,false>"),
                 "k_side_transfer_fwd": ("k_side_transfer_fwd", "k_transfer_fwd_bx3<32,true>")}


def _in_class(name, kernel):
    return any(name.startswith(p) for p in CLASS_KERNELS.get(kernel, (kernel,)))


def pmc_traffic(kernel):
    """Fabric bytes per launch of kernel class `kernel` from the COMMITTED rocprofv3 --pmc passes of this same
    command (profiles/*_pmc_per_launch.json, made by tools/profile_round.sh: FETCH_SIZE and WRITE_SIZE in
    separate passes; units KB; FETCH doubled per the gfx950 correction in MI355X_MICROARCH.md), averaged
    over the class's template instances by launch count.  PMC cannot be read from inside the process: the bench
    line reports this under `traffic_from_profiles` (with the file's name) and leaves `traffic` null."""
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_pmc_per_launch.json")))
    if not files:
        return None, None
    path = files[-1]
    try:
        runs = json.load(open(path))["period"]
        tot, n = 0.0, 0
        for name, c in runs.items():
            if not _in_class(name, kernel):
                continue
            k = c["FETCH_SIZE"]["launches"]
            tot += k * (2.0 * c["FETCH_SIZE"]["avg_counter_per_launch"] + c["WRITE_SIZE"]["avg_counter_per_launch"]) * 1024.0
            n += k
        return (tot / n if n else None), os.path.basename(path)
    except Exception:
        return None, None


def rocprof_avg_us(kernel):
    """Average duration of kernel class `kernel` in the COMMITTED rocprofv3 --kernel-trace --stats summary of this same
    command (profiles/r*_yelp_period_kernel_stats.csv): a HIP-event pair around one short kernel reads the kernel
    plus a few microseconds of the pair's own cost, rocprofv3 reads the kernel alone -- both are quoted."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_yelp_period_kernel_stats.csv")))
    if not files:
        return None, None
    try:
        tot_ms, calls = 0.0, 0        # tools/summarize_prof.py's columns: kernel, calls, total_ms, avg_us
        for row in csv.DictReader(open(files[-1])):
            if _in_class(row["kernel"], kernel):
                tot_ms += float(row["total_ms"])
                calls += int(row["calls"])
        return (1000.0 * tot_ms / calls if calls else None), os.path.basename(files[-1])
    except Exception:
        return None, None


def pmc_mfma(kernel):
    """MFMA-pipe busy fraction of kernel class `kernel` from the COMMITTED rocprofv3 --pmc pass of this same command
    (profiles/r*_pmc_mfma.json, tools/profile_round.sh: SQ_VALU_MFMA_BUSY_CYCLES, SQ_INSTS_VALU_MFMA_MOPS_F32,
    GRBM_GUI_ACTIVE), averaged over the class's template instances by launch count."""
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_pmc_mfma.json")))
    if not files:
        return None, None
    try:
        tot, n = 0.0, 0
        for name, e in json.load(open(files[-1])).items():
            if _in_class(name, kernel) and "mfma_busy_frac_of_all_simds" in e:
                tot += e["launches"] * e["mfma_busy_frac_of_all_simds"]
                n += e["launches"]
        return (tot / n if n else None), os.path.basename(files[-1])
    except Exception:
        return None, None


def host_cpu():
    model = "unknown"
    try:
        for l in open("/proc/cpuinfo"):
            if l.startswith("model name"):
                model = l.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"os_cpu_count": os.cpu_count(), "model": model}


def cpu_baseline(a, hp):
    """The CPU oracle (oracle/sml_oracle.py, kind 'port') on a bounded sample of the same
    period: full-size tables (dense Adam cost scales with the table), 8 MF batches, 16 TR
    batches, updata over 1/16 of the rows, evaluation of 2048 rows; scaled to one period.
    `value` is the full bounded sample at min(16, os.cpu_count()) threads; ONE batch per stage is also timed at
    os.cpu_count() threads and reported beside it (`by_threads`): intra-op threading of the oracle's small tensors stops
    scaling well before a 256-core host's core count, which is why the cap exists."""
    from oracle import sml_oracle as O
    from sml_amd.conv_transfer import ConvTransfer_com
    from sml_amd.mf import MFbasemode
    import contextlib, io
    U, I, d, n = a.users, a.items, a.d, a.inter
    n_mf, n_tr = 8 * hp.MF_batch_size, 16 * hp.TR_batch_size
    evals = 0 if a.no_val else hp.multi_num * (2 + hp.MF_epochs + hp.TR_epochs)
    n_updata = hp.multi_num * (1 + (hp.TR_epochs if not a.no_val else 0)) + 1
    triples = hp.multi_num * (hp.MF_epochs + hp.TR_epochs) * n

    def sample(cores, full):
        """full: the whole bounded sample.  Otherwise ONE MF batch and ONE TR batch only -- the arm that shows what the
        host's full core count does to these small-tensor loops must not cost minutes (on a 256-core host one batch
        takes over twenty seconds with 256 intra-op threads against tens of milliseconds with 16)."""
        torch.set_num_threads(cores)
        torch.manual_seed(1)
        mf = MFbasemode(U, I, d)
        with torch.no_grad():
            mf.user_laten.weight.mul_(0.3)
            mf.item_laten.weight.mul_(0.3)
        with contextlib.redirect_stdout(io.StringIO()):
            net = ConvTransfer_com(d, d)
        eng = O.OracleEngine(d)
        lu, li = mf.user_laten.weight.detach() * 0.9, mf.item_laten.weight.detach() * 0.9
        rng = np.random.RandomState(0)
        tri = lambda k: torch.from_numpy(np.stack([rng.randint(0, U, k), rng.randint(0, I, k), rng.randint(0, I, k)], 1))
        k_mf, k_tr = (n_mf, n_tr) if full else (hp.MF_batch_size, hp.TR_batch_size)
        hu, hi = mf.user_laten.weight.detach().clone(), mf.item_laten.weight.detach().clone()
        t0 = time.time(); eng.tr_stage_epoch(net, lu, li, hu, hi, tri(k_tr), hp.TR_batch_size, hp.TR_lr, hp.TR_l2); t_tr = time.time() - t0
        res = {"cores": cores, "tr_batch_s": t_tr / (k_tr // hp.TR_batch_size)}
        if not full and t_tr > 5.0:         # already conclusive: the MF batch would cost as much again
            return res
        t0 = time.time(); eng.mf_stage_epoch(mf, net, lu, li, tri(k_mf), hp.MF_batch_size, hp.MF_lr, hp.l2); t_mf = time.time() - t0
        res["mf_batch_s"] = t_mf / (k_mf // hp.MF_batch_size)
        if not full:
            return res
        ru, ri = U // 16, I // 16
        ou, oi = torch.empty(ru, d), torch.empty(ri, d)
        t0 = time.time(); eng.updata(net, lu[:ru], hu[:ru], li[:ri], hi[:ri], ou, oi); t_up = (time.time() - t0) * 16
        rows = torch.from_numpy(np.concatenate([rng.randint(0, U, (2048, 1)), rng.randint(0, I, (2048, 1 + a.neg))], 1))
        t0 = time.time(); O.eval_ranks(hu, hi, rows); t_ev = (time.time() - t0) * n / 2048.0
        parts = {"mf_s": hp.multi_num * hp.MF_epochs * t_mf * n / n_mf, "tr_s": hp.multi_num * hp.TR_epochs * t_tr * n / n_tr,
                 "updata_s": n_updata * t_up, "eval_s": evals * t_ev}
        period_s = sum(parts.values())
        res.update({"value": triples / period_s, "period_s_est": period_s, **{k: round(v, 2) for k, v in parts.items()}})
        return res

    # the full bounded sample at <= 16 threads (`value`); one batch of each stage at os.cpu_count() threads beside it,
    # so that the cap is a measurement in this line, not a comment (BASELINE.md section 3 promised cpu_count threads)
    c16, call = min(os.cpu_count() or 1, 16), os.cpu_count() or 1
    best = sample(c16, True)
    runs = [best]
    if call != c16:
        runs.append(sample(call, False))
        torch.set_num_threads(c16)
    return {"value": best["value"], "unit": "triples/s", "cores": best["cores"], "kind": "port", "host": host_cpu(),
            "extrapolated": True, "by_threads": runs,
            "sample": "oracle on full-size tables: %d MF triples, %d TR triples, updata on 1/16 of rows, eval of 2048 rows; "
                      "scaled to one period (est. %.1f s/period at %d threads: MF %.1f s, TR %.1f s, updata %.1f s, eval %.1f s); "
                      "one batch per stage also timed at os.cpu_count() threads (by_threads: seconds per batch)"
                      % (n_mf, n_tr, best["period_s_est"], best["cores"], best["mf_s"], best["tr_s"], best["updata_s"], best["eval_s"])}


def _init_ranks(device):
    """torch.distributed for a rank process: RCCL one process per GPU; gloo when the ranks share a device (test mode)."""
    import torch.distributed as dist
    from sml_amd import launch
    if launch.backend() == "nccl":
        dist.init_process_group("nccl", device_id=device)
    else:
        dist.init_process_group("gloo")
    return dist


def main():
    a = parse()
    from sml_amd import launch
    if a.gpus > 1 and not launch.is_rank_process():
        # `python bench.py --gpus N` as the driver runs it: this process has made NO GPU call.  It starts N fresh rank
        # processes (rendezvous and HSA_ENABLE_IPC_MODE_LEGACY=0 in their environment), relays rank 0's ONE JSON line
        # and leaves with the ranks' exit code.
        argv = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
        one = a.one_device or os.environ.get("SML_ONE_DEVICE") == "1"
        have = launch.visible_gpus()              # (sysfs, no HIP / HSA call; None: unknown -- a rank's set_device then fails cleanly)
        if not one and have is not None and have < a.gpus:
            raise SystemExit("bench.py --gpus %d: this node shows %d GPU(s) (--one-device maps every rank to device 0: a test mode)" % (a.gpus, have))
        code, _ = launch.spawn_ranks(argv, a.gpus, one_device=one, timeout=launch.job_timeout(a.job_timeout, 3600.0))
        raise SystemExit(code)
    if a.a3_child:
        torch.cuda.set_device(0)
        quiet = _StdoutToStderr()
        quiet.__enter__()
        res = a3_object(a, torch.device("cuda", 0))
        quiet.emit(json.dumps(res))
        quiet.__exit__()
        return
    launch.prepare_rank_env()           # (ranks started by torchrun: the IPC mode, before the first HIP call)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if launch.one_device():             # (test mode on a 1-GPU box, also under torchrun: SML_ONE_DEVICE=1 -- every rank on device 0)
        local = 0
    if world != a.gpus:
        raise SystemExit("bench.py --gpus %d inside a job of WORLD_SIZE %d" % (a.gpus, world))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if a.workload == "bare":
        quiet = _StdoutToStderr()
        quiet.__enter__()
        dist = None
        force_dist = os.environ.get("SML_FORCE_DIST") == "1" and "MASTER_ADDR" in os.environ
        if world > 1 or force_dist:
            dist = _init_ranks(device)
        res = None
        if dist is not None and a.bare_items == "sharded":
            res = bench_bare_sharded(a, device, dist)
        if res is None:
            res = bench_bare(a, device, dist)
        if rank == 0:
            quiet.emit(json.dumps(res))
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        quiet.__exit__()
        return
    from sml_amd.engine import HipEngine
    from sml_amd.period import Hyper, route_plan, run_period, synth_plan
    hp = Hyper(multi_num=a.multi_num)
    quiet = _StdoutToStderr()
    quiet.__enter__()          # for the whole run: RCCL prints its banner whenever a communicator is first used --
    #                            the one JSON line goes to the real stdout through quiet.emit()
    dist = None
    force_dist = os.environ.get("SML_FORCE_DIST") == "1" and "MASTER_ADDR" in os.environ   # exercise the exchange path at N=1
    if world > 1 or force_dist:
        dist = _init_ranks(device)
        from sml_amd import dist as smldist
    engine = HipEngine(device, a.d, max(hp.MF_batch_size, hp.TR_batch_size))
    strong = dist is not None and a.scaling == "strong"
    # weak scaling: every rank owns a shard of `users` users and processes its own period of `inter` interactions over
    # them (global batch = world x the reference's).  strong scaling: ONE period over `users` users, the reference's
    # global batches split over the ranks by user owner.  Items and theta are replicated either way.
    if strong:
        lo, hi = smldist.user_range(a.users, world, rank)
        U_local = max(hi - lo, 1)
    else:
        U_local = a.users
    st = build_state(engine, U_local, a.items, a.d, device, seed=2000 + (0 if strong else rank))
    dctx = None
    if dist is not None:
        dctx = smldist.attach(engine, st, dist, hp)
    n_plans = max(min(a.steps + a.warmup, 2), 1)
    exchanges = None
    if strong:
        # every rank draws the SAME periods (same seeds) and keeps its users' share of every global batch
        glob = [synth_plan(100 + p, a.inter, a.users, a.items, a.neg, hp, device, with_val=not a.no_val) for p in range(n_plans)]
        plans = [route_plan(dctx, g, hp, a.users, a.d, device) for g in glob]
        triples_per_period = plans[0].n_global_triples
        del glob
    else:
        plans = [synth_plan(100 + 17 * rank + p, a.inter, U_local, a.items, a.neg, hp, device, with_val=not a.no_val)
                 for p in range(n_plans)]
        triples_per_period = plans[0].n_train_triples() * world
        if dctx is not None:
            # the job-wide item-occurrence lists of every MF epoch, from the resident inputs (one all-gather of the item
            # columns per epoch, here, not in the timed loop)
            exchanges = {id(t): dctx.mf_exchange(t, hp.MF_batch_size, a.d, 0) for pl in plans for ph in pl.mf_triples for t in ph}

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(device)

    def replicas():
        return [st.MFbase.item_laten.weight.data, engine.adopt(st.transfer)]

    # the chip is partitioned: training kernels on 192 CUs, the side-stream evaluations on the other 64 (engine.partition)
    with engine.partition():
        for w in range(a.warmup):
            run_period(engine, st, plans[w % len(plans)], hp, overlap=not a.no_overlap, exchanges=exchanges)
    # A full (generation-2) collection of the interpreter's garbage collector walks every object torch and this program
    # have alive: 40-60 ms of host time during which nothing is queued (tools/bench_steps.py STEPS_NOGC, round 5).  What
    # exists now is set aside (gc.freeze): collections inside the timed region look at the periods' own garbage only.
    import gc
    gc.collect()
    gc.freeze()
    barrier()
    t0 = time.perf_counter()
    with engine.partition():
        for s in range(a.steps):
            run_period(engine, st, plans[(a.warmup + s) % len(plans)], hp, overlap=not a.no_overlap, exchanges=exchanges)
    barrier()
    dt = time.perf_counter() - t0
    engine.side_sync_check()          # the periods dropped their evaluation results: an unordered evaluation still fails the run
    peer_timeouts = None
    if dist is not None:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        # the exchange was whole: no consumer timed out, theta and the item table are bit-identical on every rank (raises)
        dctx.check_exchange(engine, "bench (%s scaling)" % a.scaling, replicas=replicas())
        peer_timeouts = int(engine.peer_status()) if dctx.mode == "peer" else 0
    value = a.steps * triples_per_period / dt

    carriers = {"peer": "one-shot exchange over peer mappings (hipIpc, no collective library on the data path)",
                "rccl": "native RCCL exchange", "torch": "torch.distributed hooks"}
    out = {"metric": "SML retrain-period training triples/s (MF + transfer stages), Yelp-shaped synthetic, d=%d" % a.d,
           "value": value, "unit": "triples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
           "ms_per_step": 1000.0 * dt / a.steps, "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
           "dtype": "f32", "data": "synthetic",
           "config": {"workload": "yelp_period: users=%d items=%d interactions/period=%d neg=%d d=%d multi_num=%d "
                                  "MF_batch=%d TR_batch=%d val_eval=%s (the reference's 40 validation evaluations per period: "
                                  "31 computed on the GPU, 9 memoised because the tables did not change in between -- identical "
                                  "numbers); MF stage: the transfer net runs once per DISTINCT row of a batch (the reference gathers one "
                                  "row per occurrence: same outputs, the duplicates' gradients summed before the backward instead of after); "
                                  "fc1 / fc2 of the MF forward and of updata on bf16 matrix products of exactly split fp32 operands (fp32-grade "
                                  "results, dtype f32); inputs resident in HBM: host batch supply and H2D are outside the timed region"
                                  % (a.users, a.items, a.inter, a.neg, a.d, hp.multi_num, hp.MF_batch_size, hp.TR_batch_size,
                                     not a.no_val),
                      "parallelism": (("users row-sharded x%d, items and theta replicated, %s, %s" %
                                       (world, "ONE period: the reference's global batches (%d / %d) split over the ranks by user owner"
                                        % (hp.MF_batch_size, hp.TR_batch_size) if strong else
                                        "independent shards: every rank its own period (global batch = %d x the reference's)" % world,
                                        carriers[dctx.mode]))
                                      if dist is not None else "single GPU")}}
    if dist is not None:
        out["config"].update({"carrier": dctx.mode, "carrier_wanted": getattr(dctx, "wanted", None), "peer_timeouts": peer_timeouts,
                              "replicas_bit_identical": True, "ranks_share_one_device": launch.one_device()})
        if not strong and not a.no_roofline and os.environ.get("SML_BENCH_STRONG_LEG", "1") != "0":
            # the strong-scaling leg beside the weak one: ONE period (the same on every rank), reference batches split by owner
            out["strong_scaling"] = strong_leg(a, hp, engine, dctx, dist, device, world, rank)

    if not a.no_roofline:
        engine.profile(True)
        with engine.partition():
            run_period(engine, st, plans[0], hp, overlap=not a.no_overlap, exchanges=exchanges)
        torch.cuda.synchronize(device)
        prof = engine.profile_read()                 # every launch of the period, on the stream it was launched on
        prof_main = engine.profile_read("main")       # ... the training stream's alone (the evaluation-only forwards run beside it)
        engine.profile(False)
        empty_pair_us = engine.profile_pair_overhead()
        if prof and rank == 0:
            # the evaluations run throttled (256 workgroups) on a low-priority side stream underneath the
            # training kernels: their span is not on the critical path, so the roofline object describes
            # the kernel class that dominates the TRAINING stream
            # (k_side_transfer_fwd: the evaluation-only table-sized forwards, on the evaluation stream's 64 CUs beside the
            # training stream -- a class and a kernel name of their own, in this run and in the rocprofv3 summaries)
            side_fwd = prof.get("k_side_transfer_fwd", (0, 0.0))[0]
            dom = max(((k, v) for k, v in prof.items() if k not in ("k_eval_ranks", "k_side_transfer_fwd")), key=lambda kv: kv[1][1])
            name, (cnt, ms) = dom
            bound, work, _ = kernel_work(name, a, hp, U_local, side_forwards=side_fwd)
            kern = {k: {"launches": c, "total_ms": round(m, 3), "avg_us": round(1000.0 * m / c, 2)} for k, (c, m) in prof.items()}
            # What a HIP-event bracket adds to a short kernel's reading, measured in THIS run: the training stream is
            # gap-free (kernel time = step time), so the bracketed kernel totals of the training stream exceed the
            # un-bracketed step by (launches x overhead).  An EMPTY pair's own reading (also measured) is an upper bound:
            # it reads more than a bracket adds around a kernel, so it is quoted, not subtracted.
            train = {k: v for k, v in prof_main.items() if k != "k_eval_ranks"}
            n_launch = sum(c for c, _ in train.values())
            derived = (sum(m for _, m in train.values()) - 1000.0 * dt / a.steps) * 1000.0 / max(n_launch, 1)
            overhead_us = min(max(derived, 0.0), empty_pair_us)
            if bound is not None and cnt and not strong:
                per_launch = work / cnt
                avg_raw_s = ms / 1000.0 / cnt
                avg_s = max(avg_raw_s - overhead_us * 1e-6, 0.25 * avg_raw_s)
                if bound == "hbm":
                    ach, peak, unit = per_launch / avg_s / 1e9, HBM_PEAK_GBS, "GB/s"
                else:
                    ach, peak, unit = per_launch / avg_s / 1e12, MFMA_F32_PEAK_TFLOPS, "TFLOP/s"
                tp, tfile = pmc_traffic(name)
                rp_us, rp_file = rocprof_avg_us(name)
                mb, mfile = pmc_mfma(name)
                out["roofline"] = {"kernel": name, "bound": bound, "achieved": ach, "peak": peak, "unit": unit,
                                   "frac": ach / peak, "traffic": None, "traffic_from_profiles": tp, "traffic_profile": tfile,
                                   "launches": cnt, "avg_launch_us": 1e6 * avg_s, "algorithmic_per_launch": per_launch,
                                   # avg_launch_us = the bracketed reading minus the bracket's own cost, both measured in this run
                                   "avg_launch_us_bracketed": 1e6 * avg_raw_s, "event_bracket_overhead_us": overhead_us,
                                   "event_pair_empty_us": empty_pair_us,
                                   "mfma_busy_frac_from_profiles": mb, "mfma_profile": mfile,
                                   # in-run HIP events (above) vs the committed rocprofv3 kernel-trace of the same command
                                   "avg_launch_us_rocprof": rp_us, "rocprof_profile": rp_file,
                                   "frac_rocprof": ((per_launch / (rp_us * 1e-6) / (1e9 if bound == "hbm" else 1e12)) / peak) if rp_us else None}
            out["kernels"] = kern
    if not a.no_a3:
        del plans, st, exchanges
        torch.cuda.empty_cache()
        if world == 1 and dist is None:
            out["a3"] = a3_in_child(a, device)
        elif dist is not None and not launch.one_device():
            # the fused embed+loss+SGD step at configs 4 / 5's shapes with the item table SHARDED over the job's ranks
            # (every rank takes part: the legs hold collectives)
            if dctx.mode == "peer":
                engine.peer_detach()
            out["a3"] = a3_object_sharded(a, device, dist)
    if rank == 0 and world == 1 and not a.no_cpu:
        out["cpu_baseline"] = cpu_baseline(a, hp)
    if rank == 0:
        quiet.emit(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    quiet.__exit__()


def strong_leg(a, hp, engine, dctx, dist, device, world, rank):
    """Strong scaling beside the weak headline: ONE Yelp-shaped period, the same on every rank (same seed), with the
    reference's global batches (1024 / 256 triples) split over the ranks by user owner (sml_amd.dist.EpochRoute): what
    `main_yelp.py --gpus N` runs.  A fresh state over this rank's user range; 1 untimed + 2 timed periods."""
    from sml_amd import dist as smldist
    from sml_amd.period import route_plan, run_period, synth_plan
    lo, hi = smldist.user_range(a.users, world, rank)
    st = build_state(engine, max(hi - lo, 1), a.items, a.d, device, seed=2000)
    dctx.sync_replicas([st.MFbase.item_laten.weight.data, st.last_item, st.hat_item, st.prev_hat_item, engine.adopt(st.transfer)])
    plan = route_plan(dctx, synth_plan(100, a.inter, a.users, a.items, a.neg, hp, device, with_val=not a.no_val), hp, a.users, a.d, device)
    with engine.partition():
        run_period(engine, st, plan, hp, overlap=not a.no_overlap)
    dist.barrier()
    torch.cuda.synchronize(device)
    k = 2
    t0 = time.perf_counter()
    with engine.partition():
        for _ in range(k):
            run_period(engine, st, plan, hp, overlap=not a.no_overlap)
    dist.barrier()
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt], device=device, dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
    engine.side_sync_check()
    dctx.check_exchange(engine, "bench (strong-scaling leg)", replicas=[st.MFbase.item_laten.weight.data, engine.adopt(st.transfer)])
    return {"scaling": "strong", "periods_timed": k, "ms_per_period": 1000.0 * dt / k, "triples_per_s": k * plan.n_global_triples / dt,
            "global_batches": [hp.MF_batch_size, hp.TR_batch_size], "carrier": dctx.mode,
            "peer_timeouts": int(engine.peer_status()) if dctx.mode == "peer" else 0}


A3_SHARDED_CONFIGS = (   # BASELINE.json configs 4 and 5: the shapes north_star's >= 6x at 8 GPUs is quoted on
    ("c4_d64_f32_zipf", 10000000, 1000000, 64, "f32", 1.0),
    ("c5_d128_f16_uniform", 50000000, 5000000, 128, "f16", 0.0),
)


def a3_object_sharded(a, device, dist):
    import copy
    res = {}
    for tag, users, items, d, dt, zipf in A3_SHARDED_CONFIGS:
        b = copy.copy(a)
        b.users, b.items, b.d, b.bare_dtype, b.item_zipf = users, items, d, dt, zipf
        b.bare_batch, b.bare_triples, b.steps, b.warmup = 262144, 1 << 22, 8, 2
        err = None
        try:
            r = bench_bare_sharded(b, device, dist)
        except Exception as e:      # noqa: BLE001 -- reported below; the vote keeps the ranks together
            r, err = None, "%s: %s" % (type(e).__name__, e)
        ok = torch.tensor([0.0 if r is None else 1.0], device=device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if float(ok.item()) < 0.5:
            res[tag] = {"error": err or "unavailable on some rank (peer exchange / shard visibility check)"}
            torch.cuda.empty_cache()
            if err is not None:     # a rank that raised mid-epoch cannot rejoin the collectives of the next leg
                break
            continue
        res[tag] = {"users": users, "items": items, "d": d, "dtype": dt, "item_zipf": zipf, "batch_per_gpu": b.bare_batch,
                    "n_gpus": r["n_gpus"], "ms_per_epoch": r["ms_per_step"], "triples_per_s": r["value"],
                    "bytes_per_triple": r["roofline"]["algorithmic_bytes_per_triple"], "kernel_frac_per_gpu": r["roofline"]["frac"],
                    "end_to_end_frac_per_gpu": r["roofline"]["end_to_end_frac"], "xgmi_GBps_per_gpu": r["roofline"]["xgmi_GBps_per_gpu_end_to_end"],
                    "carrier": r["config"]["carrier"], "peer_timeouts": r["config"]["peer_timeouts"],
                    "kernels_avg_us": {k: v["avg_us"] for k, v in r["kernels"].items()}}
    return res


if __name__ == "__main__":
    main()
