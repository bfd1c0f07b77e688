"""Multi-GPU layout of the SML period: one process per GPU, RCCL over xGMI via
torch.distributed (backend "nccl"); CPU tests drive the same code over gloo.

  users   row-sharded by owner: rank r owns users [lo_r, hi_r); a period's triples and
          test rows are routed to the owner of their user, so user gathers, the user-row
          Adam, updata over users and evaluation need no communication at all;
  items   replicated (the item table is the small one; a replica is 16 MB at Yelp scale,
          0.26 GB for 1M x 64 -- nothing next to 288 GB of HBM): every MF batch the ranks
          all-gather their per-occurrence item-gradient rows and each applies the update
          over the global, row-sorted occurrence list -> bit-identical replicas and exact
          global-batch semantics (the reference's dense gradient sums duplicates the same way);
  theta   replicated: every TR batch the flat theta-gradient (0.79 MB at d=32) is
          all-reduced, then every rank takes the identical Adam step.

BCE is a MEAN over the global batch: local terms are scaled by B_local / B_global.
"""
import torch


def user_range(n_user, world, rank):
    """Contiguous shard [lo, hi) of rank `rank`."""
    per = -(-n_user // world)
    lo = min(n_user, rank * per)
    return lo, min(n_user, lo + per)


def owner_of(users, n_user, world):
    per = -(-n_user // world)
    return users // per


def global_item_lists(all_triples, batch):
    """all_triples int64 [world, n, 3] (same n on every rank) -> (keys uint64-as-int64 [nb-major],
    vals int32): for every batch b the world*2*B_b item occurrences sorted (stably) by
    (b << 32 | item row); value = slot in the gathered gradient buffer
    [world][2*batch][d]: rank q's positives at q*2*batch + t, negatives at q*2*batch + B_b + t."""
    world, n, _ = all_triples.shape
    dev = all_triples.device
    e = torch.arange(n, device=dev)
    b = e // batch
    Bb = torch.clamp(n - b * batch, max=batch)
    t = e - b * batch
    keys, vals = [], []
    for q in range(world):
        base = q * 2 * batch
        keys += [(b << 32) | all_triples[q, :, 1], (b << 32) | all_triples[q, :, 2]]
        vals += [base + t, base + Bb + t]
    keys = torch.cat(keys)
    vals = torch.cat(vals)
    order = torch.sort(keys, stable=True).indices
    return keys[order].contiguous(), vals[order].to(torch.int32).contiguous()


class DistContext(object):
    def __init__(self, dist, device, group=None):
        self.dist = dist
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.device = torch.device(device)
        self._buf = {}
        self.native_gather = dist.get_backend(group) == "nccl"
        self.native = False          # True: libsml_hip's own RCCL communicator does the per-batch exchange

    # ---- TR stage: all-reduce of the flat theta gradient
    def tr_loss_scale(self):
        return 1.0 / self.world

    def tr_grad_hook(self, grad, batch_index):
        self.dist.all_reduce(grad, op=self.dist.ReduceOp.SUM, group=self.group)

    # ---- MF stage: all-gather of item-gradient rows
    def mf_exchange(self, triples, batch, d):
        """Exchange descriptor of one MF epoch over this rank's `triples` [n,3]."""
        n = triples.shape[0]
        allt = torch.empty((self.world,) + tuple(triples.shape), dtype=triples.dtype, device=triples.device)
        self.dist.all_gather(list(allt.unbind(0)), triples.contiguous(), group=self.group)
        keys, vals = global_item_lists(allt, batch)
        key = (batch, d)
        if key not in self._buf:
            self._buf[key] = (torch.zeros((3 * batch + 64) * d, device=self.device, dtype=torch.float32),
                              torch.zeros(self.world * 2 * batch * d, device=self.device, dtype=torch.float32))
        dx_local, dx_all = self._buf[key]
        tile = 32

        def hook(b):
            Bb = min(batch, n - b * batch)
            ioff = -(-Bb // tile) * tile
            src = dx_local[ioff * d:(ioff + 2 * batch) * d]
            if self.native_gather:
                self.dist.all_gather_into_tensor(dx_all, src, group=self.group)
            else:
                self.dist.all_gather(list(dx_all.view(self.world, -1).unbind(0)), src, group=self.group)

        return dict(world=self.world, keys=keys, vals=vals, dx_local=dx_local, dx_all=dx_all,
                    hook=None if self.native else hook, loss_scale=1.0 / self.world)

    # ---- replicas start identical
    def sync_replicas(self, tensors):
        for t in tensors:
            self.dist.broadcast(t, src=0, group=self.group)


def attach(engine, state, dist, hp=None, group=None):
    """Wire an engine (and a PeriodState) into the process group: theta and the item tables
    are broadcast from rank 0 so that replicas start identical."""
    ctx = DistContext(dist, engine.device if hasattr(engine, "device") else "cpu", group)
    engine.dist = ctx
    engine.grad_hook = ctx.tr_grad_hook
    # Prefer the native exchange (RCCL issued by the library on the compute stream, no host callback per
    # batch); fall back to the torch.distributed hooks unless EVERY rank's communicator passed its check.
    import os
    if hasattr(engine, "comm_init") and os.environ.get("SML_COMM", "rccl") == "rccl":
        try:
            ok = engine.comm_init(dist, group)
        except Exception as e:   # noqa: BLE001 -- any failure means: use the hook path
            import sys
            print("[sml_amd.dist] native RCCL exchange unavailable on rank %d: %s" % (ctx.rank, e), file=sys.stderr)
            ok = False
        flag = torch.tensor([1.0 if ok else 0.0], device=ctx.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        ctx.native = bool(flag.item() > 0.5)
        if ctx.native:
            engine.grad_hook = None
        elif ok:
            engine.comm_destroy()
    if state is not None:
        theta = engine.adopt(state.transfer) if hasattr(engine, "adopt") else None
        reps = [state.MFbase.item_laten.weight.data, state.last_item, state.hat_item, state.prev_hat_item]
        reps += [theta] if theta is not None else [p.data for p in state.transfer.parameters()]
        ctx.sync_replicas(reps)
    return ctx
