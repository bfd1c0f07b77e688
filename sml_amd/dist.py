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


def _place(keys, vals, dest, total):
    """(keys, vals) laid out BATCH-MAJOR by the destination index `dest` (a permutation of range(total)): no sort -- the
    library builds the per-batch run lists from the unsorted occurrences itself (index_prep.hip; `lists_unsorted`)."""
    k = torch.empty(total, dtype=torch.int64, device=keys.device)
    v = torch.empty(total, dtype=torch.int32, device=keys.device)
    k[dest] = keys
    v[dest] = vals.to(torch.int32)
    return k, v


def global_item_lists(all_triples, batch, stride=None):
    """Independent-shards mode.  all_triples int64 [world, n, 3] (same n on every rank) -> (keys uint64-as-int64, vals int32),
    BATCH-MAJOR and unsorted inside a batch: batch b's world*2*B_b item occurrences as (b << 32 | item row) in the order
    (rank, positive / negative, element) -- the order the run kernel sums a row's occurrences in is then a function of the
    input alone, the same on every rank; value = slot in the gathered gradient buffer [world][stride][d] (stride = 2*batch
    unless the buffer is an inbox with wider slots): rank q's positives at q*stride + t, negatives at q*stride + B_b + t."""
    world, n, _ = all_triples.shape
    dev = all_triples.device
    e = torch.arange(n, device=dev)
    b = e // batch
    Bb = torch.clamp(n - b * batch, max=batch)
    t = e - b * batch
    stride = 2 * batch if stride is None else int(stride)
    keys, vals, dest = [], [], []
    for q in range(world):
        base = q * stride
        keys += [(b << 32) | all_triples[q, :, 1], (b << 32) | all_triples[q, :, 2]]
        vals += [base + t, base + Bb + t]
        dest += [world * 2 * batch * b + (2 * q) * Bb + t, world * 2 * batch * b + (2 * q + 1) * Bb + t]
    return _place(torch.cat(keys), torch.cat(vals), torch.cat(dest), world * 2 * n)


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
        list -- for batch b every rank's 2*B_{q,b} occurrences, keyed (b << 32 | item row), batch-major and UNSORTED inside a
        batch (index_prep.hip builds every batch's run list on the device: `unsorted=True` -> sml_mf_exchange.lists_unsorted), value = slot
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
        item_off = np.zeros(self.nb + 1, dtype=np.int64)
        np.cumsum(2 * self.global_batch_sizes, out=item_off[1:])
        # batch-major, unsorted: global batch b holds its positives (in global order) and then its negatives
        tg = np.arange(self.n, dtype=np.int64) - self._b_of * self.batch
        dpos = item_off[self._b_of] + tg
        dest = torch.cat([torch.from_numpy(dpos).to(dev), torch.from_numpy(dpos + self.global_batch_sizes[self._b_of]).to(dev)])
        keys, vals = _place(keys, vals, dest, 2 * self.n)
        dx_local, dx_all = ctx.scratch(self.cap, stride, d)
        mine = self.counts[ctx.rank]
        tile = 32

        def hook(bi):
            ioff = -(-int(mine[bi]) // tile) * tile
            src = dx_local[ioff * d:(ioff + stride) * d]
            ctx.all_gather_rows(dx_all[:ctx.world * stride * d], src)

        return dict(world=ctx.world, keys=keys, vals=vals, dx_local=dx_local, dx_all=dx_all, unsorted=True,
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
        elif src.is_cuda:                        # gloo has no device all-gather (two ranks sharing one device in the tests)
            dst.view(self.world, -1).copy_(self.all_gather_dev(src.reshape(-1)))
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
            self._same_pending = None
            pend[2](float(pend[0][0]), float(pend[0][1]))       # (verified under the name it was queued with)
            pend = None
        if (what, value) not in seen or self.mode == "peer":
            # first sight of this value -- or the peer path, where a rank that runs ahead on a mismatched epoch leaves
            # its peers polling until the hang guard: there every answer is waited for (a host wait per EPOCH, not per
            # batch)
            h = t.cpu()
            verify(float(h[0]), float(h[1]))
            seen.add((what, value))
        elif pend is None:
            if t.is_cuda:
                host = torch.empty(2, dtype=torch.float64).pin_memory()
                host.copy_(t, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(self.device))
                self._same_pending = (host, ev, verify)
            else:
                self._same_pending = (t, None, verify)

    def flush_checks(self):
        """Wait for (and verify) the last deferred same_on_all_ranks answer: the driver calls this when a stage ends."""
        pend = self.__dict__.get("_same_pending")
        if pend is not None:
            self._same_pending = None
            if pend[1] is not None:
                pend[1].synchronize()
            pend[2](float(pend[0][0]), float(pend[0][1]))

    # ---- what the control plane can carry: RCCL moves device tensors; gloo (CPU tests, two ranks sharing one device)
    # all-reduces / broadcasts them but has no device all-gather
    def all_gather_dev(self, t):
        """[world, *t.shape] on t's device: every rank's `t` (same shape everywhere)."""
        out = torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        if t.is_cuda and not self.native_gather:
            host = [torch.empty(t.shape, dtype=t.dtype) for _ in range(self.world)]
            self.dist.all_gather(host, t.cpu(), group=self.group)
            for q in range(self.world):
                out[q].copy_(host[q])
        else:
            self.dist.all_gather(list(out.unbind(0)), t.contiguous(), group=self.group)
        return out

    # ---- safety nets of the exchange: a consumer that gave up waiting, replicas that drifted apart
    def check_exchange(self, engine, where, replicas=()):
        """Raise when the one-shot peer exchange lost a step since the last call (a consumer that was not released
        within SML_PEER_TIMEOUT_S went on with a partial sum: the replicas are then wrong) or when `replicas` --
        tensors that must be bit-identical on every rank (theta, the item tables) -- are not.  Synchronises the
        device; called once per stage / at the end of a benchmark, never per batch."""
        bad = 0
        if self.mode == "peer" and hasattr(engine, "peer_status"):
            torch.cuda.synchronize(self.device) if self.device.type == "cuda" else None
            bad = int(engine.peer_status()) - int(self.__dict__.get("_peer_seen", 0))
            self._peer_seen = int(self.__dict__.get("_peer_seen", 0)) + max(bad, 0)
        sums = []
        for r in replicas:
            x = r.detach().reshape(-1)
            # an order-independent fingerprint of the BITS (a float sum would hide a sign flip behind rounding)
            bits = x.view(torch.int32).to(torch.int64) if x.dtype == torch.float32 else x.view(torch.int16).to(torch.int64)
            sums.append(torch.stack([bits.sum(), (bits * bits % 1000003).sum()]))
        # (int64 end to end: a float64 cast is inexact above 2^53, which a 64M-element table's bit sum exceeds -- ADVICE r4)
        t = torch.cat([torch.tensor([int(bad)], dtype=torch.int64, device=self.device)] + sums)
        hi, lo = t.clone(), -t
        self.dist.all_reduce(hi, op=self.dist.ReduceOp.MAX, group=self.group)
        self.dist.all_reduce(lo, op=self.dist.ReduceOp.MAX, group=self.group)
        hi, lo = hi.cpu(), (-lo).cpu()
        if float(hi[0]) > 0:
            raise RuntimeError("%s: %d consumer(s) of the one-shot peer exchange gave up waiting for a peer's push on some rank "
                               "(SML_PEER_TIMEOUT_S): sums were formed from partial data; the replicas are not trustworthy"
                               % (where, int(hi[0])))
        if not torch.equal(hi[1:], lo[1:]):
            which = [k for k in range(len(replicas)) if not torch.equal(hi[1 + 2 * k:3 + 2 * k], lo[1 + 2 * k:3 + 2 * k])]
            raise RuntimeError("%s: replicated tensors %s differ between the ranks (exchange carrier: %s)" % (where, which, self.mode))

    def mf_exchange(self, triples, batch, d, loss_kind=LOSS_BCE):
        """Independent-shards mode: exchange descriptor of one MF epoch over this rank's `triples` [n,3] (every rank
        brings the same n).  The other ranks' item indices are all-gathered once per epoch."""
        n = triples.shape[0]
        self.same_on_all_ranks(n, "the epoch length n")
        items = triples[:, 1:3].contiguous()
        alli = self.all_gather_dev(items)
        allt = torch.cat([torch.zeros((self.world, n, 1), dtype=items.dtype, device=items.device), alli], dim=2)
        stride = self.peer_stride(2 * batch)
        keys, vals = global_item_lists(allt, batch, stride)
        dx_local, dx_all = self.scratch(batch, stride, d)
        tile = 32

        def hook(b):
            Bb = min(batch, n - b * batch)
            ioff = -(-Bb // tile) * tile
            self.all_gather_rows(dx_all[:self.world * stride * d], dx_local[ioff * d:(ioff + stride) * d])

        return dict(world=self.world, keys=keys, vals=vals, dx_local=dx_local, dx_all=dx_all, unsorted=True,
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
        items_all = self.all_gather_dev(items)
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
        items_all = self.all_gather_dev(items)
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


def _vote(dist, group, device, ok):
    """True when EVERY rank says ok.  Every rank calls this at the same points of the set-up, whatever happened to it
    locally: the ranks' collective sequences stay aligned, and they leave a failing path together."""
    flag = torch.tensor([1.0 if ok else 0.0], device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return bool(flag.item() > 0.5)


class _PeerSetup(object):
    """The one-shot exchange's set-up as LOCAL steps with an all-ranks vote after each (a rank that raises
    inside a step that also holds collectives leaves the others in a mismatched collective).  Owned regions and opened
    mappings are remembered so that a fall-back can release them."""

    def __init__(self, engine, dist, group, rows_cap, timeout_s):
        import os
        self.e, self.dist, self.group = engine, dist, group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.rows_cap = int(rows_cap)
        self.timeout_s = float(os.environ.get("SML_PEER_TIMEOUT_S", "120")) if timeout_s is None else float(timeout_s)
        self.same_process = dist.get_backend(group) == "threads"
        self.inbox = self.flags = None
        self.opened = []
        self.why = None

    def _try(self, fn):
        try:
            fn()
            return True
        except Exception as e:      # noqa: BLE001 -- any local failure is a "no" in the next vote
            self.why = "%s: %s" % (type(e).__name__, e)
            return False

    def run(self):
        import os
        import sys
        e, dist, group = self.e, self.dist, self.group
        dev = e.device

        def vote(ok, step):
            good = _vote(dist, group, dev, ok)
            if not good and self.why and os.environ.get("SML_DEBUG_PEER"):
                print("[sml_amd.dist] rank %d, peer set-up step %r: %s" % (self.rank, step, self.why), file=sys.stderr)
            return good

        # 1. this rank's regions
        def alloc():
            ib, fb = e.peer_region_bytes(self.world, self.rows_cap)
            self.inbox, self.flags = e.peer_alloc(ib), e.peer_alloc(fb)
        if not vote(self._try(alloc), "alloc"):
            return False
        # 2. which device every rank sits on, and what kind of memory it got: other DEVICES may only write into
        #    uncached / fine-grained regions (a plain allocation leaves the owner's L2 free to serve stale lines)
        kinds = (e.peer_mem_kind(self.inbox), e.peer_mem_kind(self.flags))
        where = _gather_objects(dist, (_device_identity(e), kinds), group)
        several_devices = len({w[0] for w in where}) > 1
        plain = any(k == 2 for w in where for k in w[1])
        if several_devices and plain:
            self.why = "an inbox / flags region is plain device memory and the ranks sit on different devices"
            if self.rank == 0:
                print("[sml_amd.dist] peer exchange refused: %s" % self.why, file=sys.stderr)
            return False
        # 3. handles out, mappings in
        handles = {}

        def export():
            handles["mine"] = (self.inbox, self.flags) if self.same_process else (e.peer_export(self.inbox), e.peer_export(self.flags))
        if not vote(self._try(export), "export"):
            return False
        got = _gather_objects(dist, handles["mine"], group)
        maps = {}

        def open_all():
            if self.same_process:
                maps["inbox"], maps["flags"] = [g[0] for g in got], [g[1] for g in got]
                return
            ib, fl = [], []
            for q in range(self.world):
                if q == self.rank:
                    ib.append(self.inbox); fl.append(self.flags)
                else:
                    a = e.peer_open(got[q][0]); self.opened.append(a)
                    b = e.peer_open(got[q][1]); self.opened.append(b)
                    ib.append(a); fl.append(b)
            maps["inbox"], maps["flags"] = ib, fl
        if not vote(self._try(open_all), "open"):
            return False
        if not vote(self._try(lambda: e.peer_attach(self.world, self.rank, maps["inbox"], maps["flags"], self.rows_cap,
                                                    timeout_s=self.timeout_s)), "attach"):
            return False
        dist.barrier(group=group)                 # every rank is attached before anybody pushes
        # 4. self-check: SML_PEER_CHECK_ROUNDS (default 6 = three per parity) one-shot all-reduces with a pattern that
        #    changes every round -- a slot served from a stale cache line, or a counter that lags, shows as the previous
        #    round's values; a short hang guard of its own
        rounds = max(2, int(os.environ.get("SML_PEER_CHECK_ROUNDS", "6")))
        n = 4096
        base = torch.arange(n, device=dev, dtype=torch.float32)
        state = {"ok": True}

        def check():
            for k in range(rounds):
                src = base * (0.25 + k) + float((self.rank + 1) * (k + 1))
                want = base * (0.25 + k) * self.world + (k + 1) * self.world * (self.world + 1) / 2.0
                got_ = e.peer_allreduce_check(src, timeout_s=min(self.timeout_s, 20.0))
                state["ok"] = bool(torch.equal(got_, want)) and state["ok"]
            state["ok"] = state["ok"] and e.peer_status() == 0
            if not state["ok"]:
                self.why = "self-check mismatch or time-out"
        ok = self._try(check) and state["ok"]
        return vote(ok, "self-check")

    def release(self):
        """After a failed set-up: nobody will use the regions.  Detach, wait for every rank, then unmap and free."""
        e = self.e
        try:
            e.peer_detach()
        except Exception:      # noqa: BLE001
            pass
        self.dist.barrier(group=self.group)       # no peer has a kernel in flight that targets a region freed below
        if not self.same_process:
            for a in self.opened:
                try:
                    e.peer_close(a)
                except Exception:      # noqa: BLE001
                    pass
        self.dist.barrier(group=self.group)       # every mapping is closed before its owner frees the memory
        for a in (self.inbox, self.flags):
            if a:
                try:
                    e.peer_free(a)
                except Exception:      # noqa: BLE001
                    pass
        self.inbox = self.flags = None
        self.opened = []


def _device_identity(engine):
    """Something two ranks share exactly when they sit on the same physical device: (host, bus id)."""
    import socket
    try:
        bus = torch.cuda.get_device_properties(engine.device).pci_bus_id
        dom = getattr(torch.cuda.get_device_properties(engine.device), "pci_domain_id", 0)
        dvc = getattr(torch.cuda.get_device_properties(engine.device), "pci_device_id", 0)
        return (socket.gethostname(), int(dom), int(bus), int(dvc))
    except Exception:      # noqa: BLE001
        return (socket.gethostname(), "index", int(engine.device.index or 0))


def peer_setup(engine, dist, group, rows_cap, timeout_s=None):
    """Allocate this rank's inbox / flags regions, exchange them with every rank of `group` and attach (sml_peer_attach).
    One process per GPU: the regions travel as hipIpc handles (dmabuf IPC: HSA_ENABLE_IPC_MODE_LEGACY=0 must be in the
    environment) and are opened with lazy peer access.  Ranks that are threads of ONE process (tests on one GPU):
    the raw addresses travel.  Every local step is followed by an all-ranks vote (the ranks leave a failing set-up
    together, collectives aligned); it ends with the start-up self-check.  Returns True when ALL ranks passed; on False
    everything the set-up allocated or mapped has been released."""
    ps = _PeerSetup(engine, dist, group, rows_cap, timeout_s)
    ok = ps.run()
    if not ok:
        ps.release()
    else:
        engine._peer_setup = ps
    return ok


def shard_visibility_check(engine, dist, group=None, rounds=3, n=4096):
    """Start-up check of the item-sharded bare step's one cross-device assumption: a row an OWNER rewrote
    with ordinary stores in one kernel is seen by a READER on another device in a later kernel.  A probe of the same
    memory kind as the shards (HipEngine.peer_tensor) is rewritten `rounds` times by its owner (a plain torch kernel),
    and after each round every rank reads every other rank's probe through its peer mapping with the loads the gradient
    pass uses (sml_peer_read: system scope, cache-bypassing) -- a stale line shows as the previous round's value.
    Returns True when all ranks saw all rounds right."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = engine.device
    state = {"ok": True}
    probe = None
    opened = []

    def body():
        nonlocal probe
        try:
            probe = engine.peer_tensor((n,), torch.float32)
            mine = probe.data_ptr() if dist.get_backend(group) == "threads" else engine.peer_export(probe.data_ptr())
        except Exception:      # noqa: BLE001
            state["ok"], mine = False, None
        if not _vote(dist, group, dev, state["ok"]):
            return False
        got = _gather_objects(dist, mine, group)
        ptrs = []
        try:
            for q in range(world):
                if q == rank or dist.get_backend(group) == "threads":
                    ptrs.append(got[q] if q != rank else probe.data_ptr())
                else:
                    a = engine.peer_open(got[q]); opened.append(a); ptrs.append(a)
        except Exception:      # noqa: BLE001
            state["ok"] = False
        if not _vote(dist, group, dev, state["ok"]):
            return False
        base = torch.arange(n, device=dev, dtype=torch.float32)
        for k in range(rounds):
            probe.copy_(base * (k + 1) + float(rank * 1000 + k))       # the owner's ordinary stores
            torch.cuda.synchronize(dev)
            dist.barrier(group=group)                                  # every owner has written round k
            for q in range(world):
                seen = engine.peer_read(ptrs[q], n)
                state["ok"] = state["ok"] and bool(torch.equal(seen, base * (k + 1) + float(q * 1000 + k)))
            torch.cuda.synchronize(dev)
            dist.barrier(group=group)                                  # everybody has read round k before it is overwritten
        return _vote(dist, group, dev, state["ok"])

    # Every path out of body() that RETURNS ends with a vote all ranks took part in (collectives stay aligned); what this rank
    # mapped or allocated is then released -- mappings closed, then a barrier (no rank frees a probe somebody still has
    # mapped), then the probe (a dedicated peer allocation, not a pooled torch tensor) -- as _PeerSetup.release does (ADVICE
    # r4: the early returns leaked both).  An EXCEPTION out of body() (peer_read / copy_ failing on this rank only) is not a
    # collective exit: this rank closes its own mappings, takes part in NO barrier (it would pair with the peers' in-loop
    # barrier, or block where nobody else arrives) and does not free the probe its peers may still read; it re-raises, the
    # process ends with an error and the launcher tears the job down.
    def close_mappings():
        for a in opened:
            try:
                engine.peer_close(a)
            except Exception:      # noqa: BLE001
                pass

    try:
        ok = body()
    except BaseException:
        close_mappings()
        raise
    close_mappings()
    dist.barrier(group=group)
    if probe is not None:
        try:
            engine.peer_tensor_free(probe)
        except Exception:      # noqa: BLE001
            pass
    return ok


def attach(engine, state, dist, hp=None, group=None, rows_cap=None):
    """Wire an engine (and a PeriodState) into the process group: theta and the item tables
    are broadcast from rank 0 so that replicas start identical.

    Who carries the per-batch exchange (SML_COMM = peer | rccl | torch; default peer):
      peer    one-shot push / poll over peer mappings (sml_peer_*): no collective library on the data path.  It is the
              default BECAUSE its start-up self-check runs the protocol itself across the job's devices (six rounds of
              changing patterns through both slot parities, every counter and every mapping) and every rank must pass;
              plain-memory regions are refused when the ranks sit on different devices; and the driver / bench verify
              after every stage that no consumer timed out and that the replicas are still bit-identical
              (DistContext.check_exchange).  Anything short of that falls through to
      rccl    the library's own RCCL communicator issues ncclAllReduce / ncclAllGather on the compute stream;
      torch   torch.distributed hooks called from the library per batch.
    Every rank runs the chosen path's start-up self-check; unless ALL ranks pass, the job falls back to the next one."""
    ctx = DistContext(dist, engine.device if hasattr(engine, "device") else "cpu", group)
    engine.dist = ctx
    engine.grad_hook = ctx.tr_grad_hook
    import os
    import sys
    want = os.environ.get("SML_COMM", "peer")
    ctx.wanted = want

    if want == "peer" and hasattr(engine, "peer_attach"):
        if rows_cap is None:
            rows_cap = 2 * int(hp.MF_batch_size) if hp is not None and hasattr(hp, "MF_batch_size") else 2 * int(engine.max_batch)
        if peer_setup(engine, dist, group, int(rows_cap)):
            ctx.mode, ctx.native, ctx.peer_rows_cap = "peer", True, int(rows_cap)
            engine.grad_hook = None
        else:
            if ctx.rank == 0:
                print("[sml_amd.dist] one-shot peer exchange unavailable (set SML_DEBUG_PEER=1 for the step that failed): "
                      "falling back to the RCCL exchange", file=sys.stderr)
            want = "rccl"
    # the native RCCL exchange (issued by the library on the compute stream, no host callback per batch)
    if ctx.mode == "torch" and want == "rccl" and hasattr(engine, "comm_init") and dist.get_backend(group) == "nccl":
        try:
            ok = engine.comm_init(dist, group)
        except Exception as e:   # noqa: BLE001 -- any failure means: use the hook path
            print("[sml_amd.dist] native RCCL exchange unavailable on rank %d: %s" % (ctx.rank, e), file=sys.stderr)
            ok = False
        if _vote(dist, group, ctx.device, ok):
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
