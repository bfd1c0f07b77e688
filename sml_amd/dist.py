"""Multi-GPU layout of the SML period: one process per GPU, RCCL over xGMI via
torch.distributed (backend "nccl"); CPU tests drive the same code over gloo.

  users   row-sharded by owner: rank r owns the contiguous range [lo_r, hi_r) and holds ONLY those rows (local row =
          user - lo_r).  A period's triples and test rows are routed to the owner of their user, so user gathers,
          the user-row Adam, updata over users and evaluation need no communication at all;
  items   replicated (the item table is the small one: 16 MB at Yelp scale, 0.26 GB for 1M x 64 -- nothing next to
          288 GB of HBM; see DESIGN.md for where that stops being the right call): every MF batch the ranks
          all-gather their per-occurrence item-gradient rows and each applies the update over the global,
          row-sorted occurrence list -> bit-identical replicas and exact global-batch semantics (the reference's
          dense gradient sums duplicates the same way);
  theta   replicated: every TR batch the flat theta-gradient (0.79 MB at d=32) is all-reduced, then every rank
          takes the identical Adam step.

Two ways to get a rank its share of an epoch:

  EpochRoute (the real driver, `torchrun ... main_yelp.py`): every rank runs the SAME host program with the same
          seeds, so every rank holds the same GLOBAL epoch (the reference's batches, in the reference's order) and
          simply keeps the triples whose user it owns.  Nothing about indices is ever communicated: each rank
          derives every rank's per-batch counts and the job's global item-occurrence list from data it already has.
          The local batch sizes differ from rank to rank and batch to batch (a rank may own no triple of a batch):
          the C ABI's `sml_batch_plan` carries the offsets and the per-batch loss scales.  Results equal the
          single-GPU run of the same command up to floating-point summation order.
  independent shards (bench.py weak scaling; period.py): every rank draws its own period over its own users, all
          with the same n and batch size; the item indices of the other ranks are all-gathered once per epoch.

Loss scaling: BCE is a MEAN over the global batch -> local terms are scaled by B_local / B_global; the BPR kinds are
SUMS -> scale 1.
"""
import numpy as np
import torch

LOSS_BCE = 0       # sml_hip.h: the only mean-type loss kind


def user_range(n_user, world, rank):
    """Contiguous shard [lo, hi) of rank `rank`."""
    per = -(-n_user // world)
    lo = min(n_user, rank * per)
    return lo, min(n_user, lo + per)


def owner_of(users, n_user, world):
    per = -(-n_user // world)
    return users // per


def item_shard_layout(n_item, world, head_rows=0):
    """Layout of an item table sharded over `world` ranks with a replicated head: rows [0, head_rows) live on every rank,
    row r >= head_rows on rank (r - head_rows) // shard_rows (local row (r - head_rows) % shard_rows).
    Returns (head_rows, shard_rows)."""
    head_rows = int(min(max(head_rows, 0), n_item))
    tail = n_item - head_rows
    return head_rows, max(1, -(-tail // world))


def item_owner(rows, world, head_rows, shard_rows):
    """(owner rank or -1 for the replicated head, local row) of global item rows (numpy / torch integer arrays)."""
    tail = rows - head_rows
    owner = tail // shard_rows
    local = tail - owner * shard_rows
    is_head = rows < head_rows
    return (owner * (~is_head) - 1 * is_head), (local * (~is_head) + rows * is_head)


def _sort_occurrences(keys, vals):
    """(keys, vals) stably sorted by key, on the keys' device."""
    order = torch.sort(keys, stable=True).indices
    return keys[order].contiguous(), vals[order].to(torch.int32).contiguous()


def global_item_lists(all_triples, batch, stride=None):
    """Independent-shards mode.  all_triples int64 [world, n, 3] (same n on every rank) -> (keys uint64-as-int64
    [nb-major], vals int32): for every batch b the world*2*B_b item occurrences sorted (stably) by
    (b << 32 | item row); value = slot in the gathered gradient buffer [world][stride][d] (stride = 2*batch unless
    the buffer is an inbox with wider slots): rank q's positives at q*stride + t, negatives at q*stride + B_b + t."""
    world, n, _ = all_triples.shape
    dev = all_triples.device
    e = torch.arange(n, device=dev)
    b = e // batch
    Bb = torch.clamp(n - b * batch, max=batch)
    t = e - b * batch
    stride = 2 * batch if stride is None else int(stride)
    keys, vals = [], []
    for q in range(world):
        base = q * stride
        keys += [(b << 32) | all_triples[q, :, 1], (b << 32) | all_triples[q, :, 2]]
        vals += [base + t, base + Bb + t]
    return _sort_occurrences(torch.cat(keys), torch.cat(vals))


class EpochRoute(object):
    """One epoch of GLOBAL batches split over the ranks by user owner (see the module docstring).

    local_tri   int64 [n_local, 3]: this rank's triples in global order, users re-indexed into the rank's shard
    plan        {batch_off, loss_scale}: the sml_batch_plan of the local triples (empty batches included)
    cap         the longest local batch of ANY rank (>= 1): the `batch` the scratch is sized for, and half the rows
                every rank contributes per batch to the gathered item-gradient buffer
    counts      int64 [world, nb]: every rank's local batch sizes"""

    def __init__(self, ctx, global_tri, batch, n_user, mean_loss):
        tri = np.ascontiguousarray(global_tri, dtype=np.int64)
        n = tri.shape[0]
        world, rank = ctx.world, ctx.rank
        nb = max(1, -(-n // batch))
        owner = owner_of(tri[:, 0], n_user, world)
        b_of = np.arange(n, dtype=np.int64) // batch
        group = owner * nb + b_of
        self.counts = np.bincount(group, minlength=world * nb).reshape(world, nb).astype(np.int64)
        Bg = self.counts.sum(0)
        lo, _hi = user_range(n_user, world, rank)
        mine = owner == rank
        self.local_tri = tri[mine].copy()
        self.local_tri[:, 0] -= lo
        off = np.zeros(nb + 1, dtype=np.int64)
        np.cumsum(self.counts[rank], out=off[1:])
        scale = (self.counts[rank] / np.maximum(Bg, 1)).astype(np.float32) if mean_loss else np.ones(nb, dtype=np.float32)
        self.plan = dict(batch_off=off, loss_scale=scale)
        self.cap = int(max(1, self.counts.max()))
        self.nb, self.n, self.batch, self.ctx = nb, n, batch, ctx
        self.global_batch_sizes = Bg
        # position of every element inside its (owner, batch) group, in global order
        order = np.argsort(group, kind="stable")
        start = np.zeros(world * nb + 1, dtype=np.int64)
        np.cumsum(self.counts.reshape(-1), out=start[1:])
        t_in = np.empty(n, dtype=np.int64)
        t_in[order] = np.arange(n, dtype=np.int64) - start[group[order]]
        self._owner, self._b_of, self._t_in, self._items = owner, b_of, t_in, tri[:, 1:3]

    def exchange(self, d):
        """Exchange descriptor of the MF stage (engine.mf_stage_epoch(exchange=...)): the job's global item-occurrence
        list -- for batch b every rank's 2*B_{q,b} occurrences sorted (stably) by (b << 32 | item row), value = slot
        in the gathered buffer [world][2*cap][d]: rank q's positives at q*2*cap + t, negatives at q*2*cap + B_{q,b}
        + t -- derived locally, plus the scratch and the gather hook."""
        ctx = self.ctx
        push_rows = 2 * self.cap
        # (peer path: the gathered buffer is one parity of this rank's inbox row slots, rows_cap rows per source rank)
        stride = ctx.peer_stride(push_rows)
        cnt = self.counts[self._owner, self._b_of]
        base = self._owner * stride + self._t_in
        dev = ctx.device
        b = torch.from_numpy(self._b_of).to(dev)
        keys = torch.cat([(b << 32) | torch.from_numpy(np.ascontiguousarray(self._items[:, 0])).to(dev),
                          (b << 32) | torch.from_numpy(np.ascontiguousarray(self._items[:, 1])).to(dev)])
        vals = torch.cat([torch.from_numpy(base).to(dev), torch.from_numpy(base + cnt).to(dev)])
        keys, vals = _sort_occurrences(keys, vals)
        item_off = np.zeros(self.nb + 1, dtype=np.int64)
        np.cumsum(2 * self.global_batch_sizes, out=item_off[1:])
        dx_local, dx_all = ctx.scratch(self.cap, stride, d)
        mine = self.counts[ctx.rank]
        tile = 32

        def hook(bi):
            ioff = -(-int(mine[bi]) // tile) * tile
            src = dx_local[ioff * d:(ioff + stride) * d]
            ctx.all_gather_rows(dx_all[:ctx.world * stride * d], src)

        return dict(world=ctx.world, keys=keys, vals=vals, dx_local=dx_local, dx_all=dx_all,
                    hook=None if ctx.native else hook, loss_scale=1.0, slot_stride=stride, item_off=item_off, push_rows=push_rows)


class DistContext(object):
    def __init__(self, dist, device, group=None):
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.device = torch.device(device)
        self._buf = {}
        self.native_gather = dist.get_backend(group) == "nccl"
        self.native = False          # True: the library does the per-batch exchange itself (no host callback per batch)
        self.mode = "torch"          # "peer": one-shot push / poll over peer mappings; "rccl": the library's own RCCL
        #                              communicator; "torch": torch.distributed hooks
        self.peer_rows_cap = 0

    def peer_stride(self, rows):
        """Slot stride of the gathered item-gradient buffer for `rows` rows per rank: the inbox geometry on the peer path."""
        if self.mode != "peer":
            return rows
        if rows > self.peer_rows_cap:
            raise ValueError("a batch contributes %d item rows per rank; the peer inboxes were sized for %d" % (rows, self.peer_rows_cap))
        return self.peer_rows_cap

    # ---- loss scaling
    def loss_scale(self, loss_kind):
        """Equal local batches on every rank: B_local / B_global = 1 / world for the mean-type BCE, 1 for the BPR sums."""
        return 1.0 / self.world if loss_kind == LOSS_BCE else 1.0

    def tr_loss_scale(self, loss_kind=LOSS_BCE):
        return self.loss_scale(loss_kind)

    # ---- TR stage: all-reduce of the flat theta gradient
    def tr_grad_hook(self, grad, batch_index):
        self.dist.all_reduce(grad, op=self.dist.ReduceOp.SUM, group=self.group)

    # ---- MF stage: all-gather of item-gradient rows
    def scratch(self, cap, stride, d):
        """(dx_local, dx_all) grown on demand: the backward's gradient rows of one local batch and the gathered
        item-gradient rows of all ranks."""
        need_l, need_a = (3 * cap + 64 + stride) * d, self.world * stride * d
        cur = self._buf.get(d)
        if cur is None or cur[0].numel() < need_l or cur[1].numel() < need_a:
            cur = (torch.zeros(max(need_l, cur[0].numel() if cur else 0), device=self.device, dtype=torch.float32),
                   torch.zeros(max(need_a, cur[1].numel() if cur else 0), device=self.device, dtype=torch.float32))
            self._buf[d] = cur
        return cur

    def all_gather_rows(self, dst, src):
        if self.native_gather:
            self.dist.all_gather_into_tensor(dst, src, group=self.group)
        else:
            self.dist.all_gather(list(dst.view(self.world, -1).unbind(0)), src, group=self.group)

    def same_on_all_ranks(self, value, what):
        """Independent-shards mode needs the same epoch length on every rank: check it, do not hang.  Every call joins
        the (tiny) all-reduce, so the ranks' collective sequences stay aligned whatever they see; but the host WAITS for
        the answer only the first time it sees a value -- the property does not change from epoch to epoch, and a drain
        of the device per MF epoch would cost the run-ahead the single-GPU path lives on.  Later answers are read back
        asynchronously and verified by the next call."""
        t = torch.tensor([float(value), -float(value)], device=self.device, dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)

        def verify(hi, lo):
            if hi != -lo:
                raise ValueError("%s differs between ranks (max %d, min %d): independent-shard epochs need equal sizes; "
                                 "route a shared epoch with DistContext.route_epoch instead" % (what, int(hi), int(-lo)))
        seen = self.__dict__.setdefault("_same_seen", set())
        pend = self.__dict__.get("_same_pending")
        if pend is not None and (pend[1] is None or pend[1].query()):
            verify(float(pend[0][0]), float(pend[0][1]))
            self._same_pending = pend = None
        if (what, value) not in seen:
            h = t.cpu()
            verify(float(h[0]), float(h[1]))
            seen.add((what, value))
        elif pend is None:
            if t.is_cuda:
                host = torch.empty(2, dtype=torch.float64).pin_memory()
                host.copy_(t, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(self.device))
                self._same_pending = (host, ev)
            else:
                self._same_pending = (t, None)

    def mf_exchange(self, triples, batch, d, loss_kind=LOSS_BCE):
        """Independent-shards mode: exchange descriptor of one MF epoch over this rank's `triples` [n,3] (every rank
        brings the same n).  The other ranks' item indices are all-gathered once per epoch."""
        n = triples.shape[0]
        self.same_on_all_ranks(n, "the epoch length n")
        items = triples[:, 1:3].contiguous()
        alli = torch.empty((self.world,) + tuple(items.shape), dtype=items.dtype, device=items.device)
        self.dist.all_gather(list(alli.unbind(0)), items, group=self.group)
        allt = torch.cat([torch.zeros((self.world, n, 1), dtype=items.dtype, device=items.device), alli], dim=2)
        stride = self.peer_stride(2 * batch)
        keys, vals = global_item_lists(allt, batch, stride)
        dx_local, dx_all = self.scratch(batch, stride, d)
        tile = 32

        def hook(b):
            Bb = min(batch, n - b * batch)
            ioff = -(-Bb // tile) * tile
            self.all_gather_rows(dx_all[:self.world * stride * d], dx_local[ioff * d:(ioff + stride) * d])

        return dict(world=self.world, keys=keys, vals=vals, dx_local=dx_local, dx_all=dx_all,
                    hook=None if self.native else hook, loss_scale=self.loss_scale(loss_kind), slot_stride=stride,
                    item_off=None, push_rows=2 * batch)

    # ---- bare a3 step (SGD): users sharded, items replicated, per-occurrence item-gradient rows all-gathered
    def bare_exchange(self, triples, batch, d, loss_kind=LOSS_BCE):
        """Exchange descriptor of the bare step over this rank's `triples` [n,3] (local user index; every rank brings
        the same n and batch): the item columns of every rank are gathered ONCE here (the library builds the job's
        item-occurrence list from them on the device), and per batch the ranks' item-gradient rows are gathered (by
        the library's own communicator, or by the hook)."""
        n = triples.shape[0]
        self.same_on_all_ranks(n, "the epoch length n")
        items = triples[:, 1:3].contiguous()
        items_all = torch.empty((self.world,) + tuple(items.shape), dtype=items.dtype, device=items.device)
        self.dist.all_gather(list(items_all.unbind(0)), items, group=self.group)
        key = ("bare", d, batch)
        if key not in self._buf:
            self._buf[key] = (torch.zeros(3 * batch * d, device=self.device, dtype=torch.float32),
                              torch.zeros(self.world * 2 * batch * d, device=self.device, dtype=torch.float32))
        dx_local, dx_all = self._buf[key]

        def hook(b):
            Bb = min(batch, n - b * batch)
            self.all_gather_rows(dx_all, dx_local[Bb * d:(Bb + 2 * batch) * d])

        # (the replicated-items bare step is exchange-bound by design -- DESIGN.md section 6 -- and is not wired to the
        # peer inboxes: RCCL when the library has a communicator, else the torch.distributed hook)
        return dict(world=self.world, items_all=items_all.contiguous(), dx_local=dx_local, dx_all=dx_all,
                    hook=None if self.mode == "rccl" else hook, loss_scale=self.loss_scale(loss_kind))

    # ---- bare a3 step with the ITEM TABLE SHARDED (configs 4 / 5): head replicated, tail owner-computes over the peer exchange
    def bare_shard(self, engine, triples, n_item, head_rows, w_item_head, w_item_shard, loss_kind=LOSS_BCE):
        """Shard descriptor of one bare epoch over this rank's `triples` [n,3] (local user index, GLOBAL item index;
        every rank brings the same n and batch).  w_item_head: this rank's replica of rows [0, head_rows);
        w_item_shard: this rank's tail shard [shard_rows, d] (item_shard_layout).  Every rank's shard must be readable
        from every device: ranks that are threads of one process hand over the addresses; one process per GPU exports
        the shard with hipIpc -- it must then be a whole allocation (HipEngine.peer_tensor).  The item columns of every
        rank are gathered once here."""
        if self.mode != "peer":
            raise RuntimeError("the item-sharded bare step runs on the one-shot peer exchange (SML_COMM=peer)")
        n = triples.shape[0]
        self.same_on_all_ranks(n, "the epoch length n")
        head_rows, shard_rows = item_shard_layout(n_item, self.world, head_rows)
        if tuple(w_item_shard.shape[:1]) != (shard_rows,):
            raise ValueError("this rank's tail shard must have %d rows" % shard_rows)
        items = triples[:, 1:3].contiguous()
        items_all = torch.empty((self.world,) + tuple(items.shape), dtype=items.dtype, device=items.device)
        self.dist.all_gather(list(items_all.unbind(0)), items, group=self.group)
        key = ("shard_ptrs", w_item_shard.data_ptr())
        ptrs = self._buf.get(key)
        if ptrs is None:
            if self.dist.get_backend(self.group) == "threads":
                ptrs = _gather_objects(self.dist, w_item_shard.data_ptr(), self.group)
            else:
                raw = getattr(w_item_shard, "_sml_peer_ptr", None)
                if raw is None or raw != w_item_shard.data_ptr():
                    raise ValueError("one process per GPU: allocate the shard with HipEngine.peer_tensor (a whole, exportable allocation)")
                handles = _gather_objects(self.dist, engine.peer_export(raw), self.group)
                ptrs = [raw if q == self.rank else engine.peer_open(handles[q]) for q in range(self.world)]
            self._buf[key] = ptrs
        return dict(world=self.world, rank=self.rank, head_rows=head_rows, shard_rows=shard_rows, item_shard=ptrs,
                    w_item_head=w_item_head, items_all=items_all.contiguous(), loss_scale=self.loss_scale(loss_kind), n_item=int(n_item))

    # ---- the real driver: a shared global epoch, split by user owner
    def route_epoch(self, global_tri, batch, n_user, mean_loss):
        return EpochRoute(self, global_tri, batch, n_user, mean_loss)

    def route_rows(self, rows, n_user):
        """Test rows [n, 2+neg] (numpy) of the users this rank owns, user column re-indexed into the shard."""
        rows = np.asarray(rows)
        lo, hi = user_range(n_user, self.world, self.rank)
        mine = (rows[:, 0] >= lo) & (rows[:, 0] < hi)
        out = rows[mine].copy()
        out[:, 0] -= lo
        return out

    def sum_scalars(self, values):
        """Sum a few host numbers over the ranks (evaluation hit counts, ...)."""
        t = torch.tensor([float(v) for v in values], device=self.device, dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return [float(v) for v in t.cpu()]

    def sum_tensor_(self, t):
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t

    # ---- replicas start identical
    def sync_replicas(self, tensors):
        for t in tensors:
            self.dist.broadcast(t, src=0, group=self.group)


def _gather_objects(dist, obj, group):
    out = [None] * dist.get_world_size(group)
    dist.all_gather_object(out, obj, group=group)
    return out


def peer_setup(engine, dist, group, rows_cap, timeout_s=None):
    """Allocate this rank's inbox / flags regions, exchange them with every rank of `group` and attach (sml_peer_attach).
    One process per GPU: the regions travel as hipIpc handles (dmabuf IPC: HSA_ENABLE_IPC_MODE_LEGACY=0 must be in the
    environment) and are opened with lazy peer access.  Ranks that are threads of ONE process (tests on one GPU):
    the raw addresses travel.  Ends with a start-up self-check: every rank pushes a pattern through the theta slots and
    reads back the rank-order sum.  Returns True when this rank's check passed."""
    import os
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if timeout_s is None:
        timeout_s = float(os.environ.get("SML_PEER_TIMEOUT_S", "120"))
    ib, fb = engine.peer_region_bytes(world, rows_cap)
    inbox, flags = engine.peer_alloc(ib), engine.peer_alloc(fb)
    same_process = dist.get_backend(group) == "threads"
    if same_process:
        got = _gather_objects(dist, (inbox, flags), group)
        inboxes, flagses = [g[0] for g in got], [g[1] for g in got]
    else:
        got = _gather_objects(dist, (engine.peer_export(inbox), engine.peer_export(flags)), group)
        inboxes = [inbox if q == rank else engine.peer_open(got[q][0]) for q in range(world)]
        flagses = [flags if q == rank else engine.peer_open(got[q][1]) for q in range(world)]
    engine.peer_attach(world, rank, inboxes, flagses, rows_cap, timeout_s=timeout_s)
    dist.barrier(group=group)                 # every rank is attached before anybody pushes
    n = 1024
    src = torch.arange(n, device=engine.device, dtype=torch.float32) * 0.25 + float(rank + 1)
    want = torch.arange(n, device=engine.device, dtype=torch.float32) * 0.25 * world + world * (world + 1) / 2.0
    ok = True
    for _ in range(2):                        # both parities; a short hang guard of its own
        ok = bool(torch.equal(engine.peer_allreduce_check(src, timeout_s=min(timeout_s, 20.0)), want)) and ok
    return ok and engine.peer_status() == 0


def attach(engine, state, dist, hp=None, group=None, rows_cap=None):
    """Wire an engine (and a PeriodState) into the process group: theta and the item tables
    are broadcast from rank 0 so that replicas start identical.

    Who carries the per-batch exchange (SML_COMM = peer | rccl | torch; default peer):
      peer    one-shot push / poll over peer mappings (sml_peer_*): no collective library on the data path;
      rccl    the library's own RCCL communicator issues ncclAllReduce / ncclAllGather on the compute stream;
      torch   torch.distributed hooks called from the library per batch.
    Every rank runs the chosen path's start-up self-check; unless ALL ranks pass, the job falls back to the next one."""
    ctx = DistContext(dist, engine.device if hasattr(engine, "device") else "cpu", group)
    engine.dist = ctx
    engine.grad_hook = ctx.tr_grad_hook
    import os
    import sys
    want = os.environ.get("SML_COMM", "peer")

    def all_ok(ok):
        flag = torch.tensor([1.0 if ok else 0.0], device=ctx.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        return bool(flag.item() > 0.5)

    if want == "peer" and hasattr(engine, "peer_attach"):
        if rows_cap is None:
            rows_cap = 2 * int(hp.MF_batch_size) if hp is not None and hasattr(hp, "MF_batch_size") else 2 * int(engine.max_batch)
        try:
            ok = peer_setup(engine, dist, group, int(rows_cap))
        except Exception as e:   # noqa: BLE001 -- any failure means: use the next path
            print("[sml_amd.dist] peer-mapping exchange unavailable on rank %d: %s" % (ctx.rank, e), file=sys.stderr)
            ok = False
        if all_ok(ok):
            ctx.mode, ctx.native, ctx.peer_rows_cap = "peer", True, int(rows_cap)
            engine.grad_hook = None
        else:
            engine.peer_detach()
            want = "rccl"
    # the native RCCL exchange (issued by the library on the compute stream, no host callback per batch)
    if ctx.mode == "torch" and want == "rccl" and hasattr(engine, "comm_init") and dist.get_backend(group) == "nccl":
        try:
            ok = engine.comm_init(dist, group)
        except Exception as e:   # noqa: BLE001 -- any failure means: use the hook path
            print("[sml_amd.dist] native RCCL exchange unavailable on rank %d: %s" % (ctx.rank, e), file=sys.stderr)
            ok = False
        if all_ok(ok):
            ctx.mode, ctx.native = "rccl", True
            engine.grad_hook = None
        elif ok:
            engine.comm_destroy()
    if state is not None:
        theta = engine.adopt(state.transfer) if hasattr(engine, "adopt") else None
        reps = [state.MFbase.item_laten.weight.data, state.last_item, state.hat_item, state.prev_hat_item]
        reps += [theta] if theta is not None else [p.data for p in state.transfer.parameters()]
        ctx.sync_replicas(reps)
    return ctx
