"""One SML retrain period as a sequence of engine calls on HBM-resident inputs.

This is the compute of meta_train.train_one_stage3 (reference model/transfer.py:753-792,
the online-training branch with the validation prints the reference always makes)
with the batch supply hoisted out: every epoch's (user, item, neg) triples and the
validation rows are device tensors before the period starts.  bench.py times this.
"""
import numpy as np
import torch

from . import synth


class Hyper(object):
    """The reference's hot-path defaults (main_yelp.py:24-73)."""

    def __init__(self, **kw):
        self.multi_num = 10
        self.MF_epochs = 1
        self.TR_epochs = 1
        self.MF_batch_size = 1024
        self.TR_batch_size = 256
        self.MF_lr = 0.01
        self.l2 = 1e-6
        self.TR_lr = 0.001
        self.TR_l2 = 1e-4
        self.topK = 20
        self.__dict__.update(kw)


class PeriodState(object):
    """W (inside MFbase), W_{t-1}, W_hat, previous W_hat, theta -- the driver's state
    (reference model/transfer.py:347-364, 381)."""

    def __init__(self, mfbase, transfer):
        self.MFbase = mfbase
        self.transfer = transfer
        wu, wi = mfbase.user_laten.weight.data, mfbase.item_laten.weight.data
        self.last_user, self.last_item = torch.zeros_like(wu), torch.zeros_like(wi)
        self.hat_user, self.hat_item = wu.clone(), wi.clone()
        self.prev_hat_user, self.prev_hat_item = wu.clone(), wi.clone()


class PeriodPlan(object):
    """HBM-resident inputs of one period."""

    def __init__(self, val_rows, mf_triples, tr_triples):
        self.val_rows = val_rows          # int64 [n, 2+neg] or None
        self.mf_triples = mf_triples      # [phase][epoch] -> int64 [n,3]
        self.tr_triples = tr_triples

    def n_train_triples(self):
        return sum(t.shape[0] for ph in self.mf_triples for t in ph) + \
               sum(t.shape[0] for ph in self.tr_triples for t in ph)


def synth_plan(seed, n_inter, n_user, n_item, neg, hp, device, user_lo=0, user_hi=None, with_val=True):
    """Synthetic period in the reference's data semantics: D_t rows carry pre-sampled
    negatives (MF_sample 'all': one negative column per epoch), D_{t+1} pairs get a uniform
    negative from the period's items (TR_sample_type 'alone'), validation rows carry `neg`
    negatives.  Users are drawn from [user_lo, user_hi) (a rank's shard)."""
    rng = np.random.RandomState(seed)
    user_hi = n_user if user_hi is None else user_hi
    span = user_hi - user_lo

    def period(n):
        train, test = synth.sample_period(rng, n, span, n_item, neg=neg)
        train[:, 0] += user_lo
        test[:, 0] += user_lo
        return train, test

    _, set_t = period(n_inter)
    set_tt, val = period(n_inter)
    mf, tr = [], []
    for ph in range(hp.multi_num):
        eps = []
        for ep in range(hp.MF_epochs):
            order = rng.permutation(n_inter)
            col = 2 + rng.randint(0, neg)
            eps.append(torch.from_numpy(np.ascontiguousarray(set_t[order][:, [0, 1, col]])).to(device))
        mf.append(eps)
        eps = []
        items = np.unique(set_tt[:, 1])
        for ep in range(hp.TR_epochs):
            order = rng.permutation(n_inter)
            negs = items[rng.randint(0, items.shape[0], size=n_inter)]
            tri = np.concatenate([set_tt[order], negs[:, None]], axis=1)
            eps.append(torch.from_numpy(np.ascontiguousarray(tri)).to(device))
        tr.append(eps)
    val_rows = torch.from_numpy(val).to(device) if with_val else None
    return PeriodPlan(val_rows, mf, tr)


class RoutedEpoch(object):
    """One epoch of the reference's GLOBAL batches as this rank runs it (strong scaling): its share of every batch
    (sml_amd.dist.EpochRoute: split by user owner, unequal and empty local batches), resident on the device."""

    def __init__(self, route, d, device, with_exchange):
        self.local_tri = torch.from_numpy(route.local_tri).to(device)
        self.cap, self.plan = route.cap, route.plan
        self.exchange = route.exchange(d) if with_exchange else None
        self.n_global = route.n


def route_plan(dctx, plan, hp, n_user_global, d, device, mean_loss=True):
    """Strong scaling: the same period every rank holds (same seed), with every epoch's GLOBAL batches (hp.MF_batch_size /
    hp.TR_batch_size, the reference's) split over the ranks by user owner.  Nothing about indices is communicated: every
    rank derives all ranks' counts from the global epoch.  Returns a PeriodPlan whose epochs are RoutedEpoch objects and
    whose validation rows are this rank's users' rows (user column re-indexed into the shard)."""
    mf = [[RoutedEpoch(dctx.route_epoch(t.cpu().numpy(), hp.MF_batch_size, n_user_global, mean_loss), d, device, True) for t in ph]
          for ph in plan.mf_triples]
    tr = [[RoutedEpoch(dctx.route_epoch(t.cpu().numpy(), hp.TR_batch_size, n_user_global, mean_loss), d, device, False) for t in ph]
          for ph in plan.tr_triples]
    val = None
    if plan.val_rows is not None:
        val = torch.from_numpy(dctx.route_rows(plan.val_rows.cpu().numpy(), n_user_global)).to(device)
    out = PeriodPlan(val, mf, tr)
    out.n_global_triples = sum(e.n_global for ph in mf for e in ph) + sum(e.n_global for ph in tr for e in ph)
    return out


def run_period(engine, st, plan, hp, record=None, overlap=True, exchanges=None):
    """Execute one period.  Returns (last MF batch losses, last TR batch losses) device tensors.

    Validation scheduling (results identical to evaluating in place):
      * an evaluation of tables that were not modified since the previous evaluation of the same rows
        returns the previous result (the reference's "before train MF" numbers always repeat the
        preceding "val result" line);
      * every other evaluation is queued on a snapshot of the tables (engine.eval_submit) and runs on the
        engine's low-priority side stream underneath the training kernels that follow; the numbers are
        collected when the period ends.  overlap=False evaluates in place on the training stream."""
    mf, net = st.MFbase, st.transfer
    wu, wi = mf.user_laten.weight.data, mf.item_laten.weight.data
    state = {"version": 0, "cached": None}
    notes = []          # (tag, resolver)

    def evaluate(tag):
        if plan.val_rows is None:
            return
        if state["cached"] is None or state["cached"][0] != state["version"]:
            if overlap and hasattr(engine, "eval_submit"):
                pending = engine.eval_metrics_submit(engine.eval_submit(wu, wi, plan.val_rows), hp.topK)
                box = {}

                def res(pending=pending, box=box):
                    if "v" not in box:
                        box["v"] = engine.eval_result(pending)
                    return box["v"]
            else:
                val = engine.eval_metrics(engine.eval_ranks(wu, wi, plan.val_rows), hp.topK)

                def res(val=val):
                    return val
            state["cached"] = (state["version"], res)
        notes.append((tag, state["cached"][1]))

    def updata():
        engine.updata(net, st.last_user, st.hat_user, st.last_item, st.hat_item, wu, wi)
        state["version"] += 1

    # An updata whose output only an evaluation reads (the next updata overwrites it before any training kernel looks at
    # the tables) is queued WITH that evaluation on the evaluation stream (engine.eval_submit_transferred): the tables
    # this period holds are not written, the numbers are the same.
    transferred_ok = (overlap and plan.val_rows is not None and hasattr(engine, "eval_submit_transferred")
                      and engine.can_submit_transferred(wu, wi))

    def evaluate_transferred(tag):
        pending = engine.eval_metrics_submit(
            engine.eval_submit_transferred(net, st.last_user, st.hat_user, st.last_item, st.hat_item, plan.val_rows), hp.topK)
        box = {}

        def res(pending=pending, box=box):
            if "v" not in box:
                box["v"] = engine.eval_result(pending)
            return box["v"]
        notes.append((tag, res))

    # save_MF_weight('last')
    copy = engine.copy_tables if hasattr(engine, "copy_tables") else (lambda pairs: [d.copy_(s) for d, s in pairs])
    copy([(st.last_user, wu), (st.last_item, wi)])
    mf_loss = tr_loss = None
    for ph in range(hp.multi_num):
        evaluate("before MF")
        for tri in plan.mf_triples[ph]:
            if isinstance(tri, RoutedEpoch):      # this rank's share of the reference's global batches
                mf_loss = engine.mf_stage_epoch(mf, net, st.last_user, st.last_item, tri.local_tri, tri.cap,
                                                hp.MF_lr, hp.l2, norm=False, bce=True, plan=tri.plan, exchange=tri.exchange)
            else:
                # (exchanges: several GPUs, independent shards -- the epoch's job-wide item lists, built ahead from the
                # resident inputs; without them engine.mf_stage_epoch gathers the item columns itself, every epoch)
                ex = exchanges.get(id(tri)) if exchanges else None
                mf_loss = engine.mf_stage_epoch(mf, net, st.last_user, st.last_item, tri, hp.MF_batch_size,
                                                hp.MF_lr, hp.l2, norm=False, bce=True, exchange=ex)
            engine.mf_flush(mf)
            state["version"] += 1
            evaluate("MF epoch")
        # save_MF_weight('hat')
        copy([(st.prev_hat_user, st.hat_user), (st.prev_hat_item, st.hat_item)])
        copy([(st.hat_user, wu), (st.hat_item, wi)])
        n_tr = len(plan.tr_triples[ph])
        if transferred_ok and n_tr > 0:
            evaluate_transferred("before TR")         # (the updata after the first TR epoch overwrites what this one would write)
        else:
            updata()
            evaluate("before TR")
        for k_tr, tri in enumerate(plan.tr_triples[ph]):
            if isinstance(tri, RoutedEpoch):
                tr_loss = engine.tr_stage_epoch(net, st.last_user, st.last_item, st.hat_user, st.hat_item, tri.local_tri,
                                                tri.cap, hp.TR_lr, hp.TR_l2, bce=True, plan=tri.plan)
            else:
                tr_loss = engine.tr_stage_epoch(net, st.last_user, st.last_item, st.hat_user, st.hat_item, tri,
                                                hp.TR_batch_size, hp.TR_lr, hp.TR_l2, bce=True)
            if plan.val_rows is not None:
                if transferred_ok and k_tr + 1 < n_tr:
                    evaluate_transferred("TR epoch")
                else:
                    updata()
                    evaluate("TR epoch")
    updata()
    if record is not None and plan.val_rows is not None:
        n = plan.val_rows.shape[0]
        for tag, res in notes:
            hits, ndcg = res()
            record.append((tag, hits / n, ndcg / n))
    elif plan.val_rows is not None and overlap and hasattr(engine, "side_sync_check"):
        # results dropped: an evaluation that ran unordered (flag time-out) must still surface.  No host wait here --
        # this looks at the previous period's read-back; the caller ends its run with engine.side_sync_check()
        engine.side_sync_check(block=False)
    return mf_loss, tr_loss
