"""SML sequential-retraining driver: the module surface of the reference's
model/transfer.py `meta_train` with its inner loops on the HIP engine.

Control flow, state names, hyper-parameter handling, random-number consumption and
printed lines follow the reference (model/transfer.py:302-1029) so the class drops
into main_yelp.py / main_news.py; the three hot loops
  MF_train_onestage       (model/transfer.py:417-534)
  transfer_train_onestage (model/transfer.py:644-749)
  updata                  (model/transfer.py:884-902)
are single calls into libsml_hip.so per epoch / per table, with whole-epoch triples
built by sml_amd.datasets instead of a per-item DataLoader.
"""
import contextlib
import copy
import io
import sys
import time

import numpy as np
import torch

from . import datasets as D
from .conv_transfer import ConvTransfer, ConvTransfer_com
from .evaluation import DeviceRows, test_model
from .mf import MFbasemode

SampleDaset = D.offlineDataset_withsample
PreSampleDatast = D.trainDataset_withPreSample


def _default_device():
    if not torch.cuda.is_available():
        raise RuntimeError("SML on this build needs a GPU: torch.cuda.is_available() is False and "
                           "there is no CPU path")
    return torch.device("cuda", torch.cuda.current_device())


def _make_engine(device, d, max_batch):
    from .engine import HipEngine
    return HipEngine(device, d, max_batch)


def _np(x):
    return x.detach().cpu().numpy() if isinstance(x, torch.Tensor) else np.asarray(x)


class _Lazy(object):
    """A number that is still being computed on the device; resolved when a line is printed."""

    def __init__(self, fn):
        self._fn, self._done, self._v = fn, False, None

    def get(self):
        if not self._done:
            self._v, self._done, self._fn = self._fn(), True, None
        return self._v


def _val(x):
    return x.get() if isinstance(x, _Lazy) else x


class _OptimizerInfo(object):
    """What the reference exposes as MF_optimizer / transfer_optimizer, reduced to the
    hyper-parameters; the Adam state itself lives in the engine."""

    def __init__(self, lr, weight_decay):
        self.param_groups = [dict(lr=lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=weight_decay)]


class meta_train(object):
    def __init__(self, args, datasets, user_num, item_num, laten_dim, dist=None):
        """dist: an initialised torch.distributed module (one process per GPU, `torchrun ... main_yelp.py`), or None.
        Under dist every rank runs this same program on the same seeds and data; users are row-sharded by owner
        (this rank keeps rows [lo, hi) of every user-side table), items and theta are replicated, and every global
        batch is split by user owner (sml_amd.dist.EpochRoute): the printed numbers are the single-GPU run's."""
        self.device = _default_device()
        self.n_user_global = int(user_num)
        if args.data_name != 'yelp':
            # the reference builds (and then discards) a fresh MF model here, which advances
            # the global RNG before the transfer net is initialised (model/transfer.py:314-317)
            MFbasemode(num_user=user_num, num_item=item_num, laten_factor=laten_dim)
        # whole-module pickle, class path model.MF.MFbasemode (model/transfer.py:322-325)
        self.MFbase = torch.load(args.pre_model, map_location='cpu', weights_only=False).to(self.device)
        table_dim = self.MFbase.user_laten.weight.shape[1]
        if table_dim != laten_dim:
            raise ValueError("--laten %d does not match the checkpoint's embedding width %d" % (laten_dim, table_dim))
        self.dist = None
        import os
        use_dist = dist is not None and (dist.get_world_size() > 1 or os.environ.get("SML_FORCE_DIST") == "1")   # (forced: tests drive the exchange path on one GPU)
        if use_dist:
            # keep this rank's user rows only (the checkpoint holds the whole table)
            from . import dist as smldist
            lo, hi = smldist.user_range(self.n_user_global, dist.get_world_size(), dist.get_rank())
            full = self.MFbase
            rng = torch.get_rng_state()        # (the shard's constructor draws initial values: keep the run's random tape)
            self.MFbase = MFbasemode(num_user=max(hi - lo, 1), num_item=full.item_num, laten_factor=table_dim).to(self.device)
            torch.set_rng_state(rng)
            with torch.no_grad():
                self.MFbase.user_laten.weight[:hi - lo].copy_(full.user_laten.weight.data[lo:hi])
                self.MFbase.user_bais.weight[:hi - lo].copy_(full.user_bais.weight.data[lo:hi])
                self.MFbase.item_laten.weight.copy_(full.item_laten.weight.data)
                self.MFbase.item_bais.weight.copy_(full.item_bais.weight.data)
            del full

        self.transfer_type = args.transfer_type
        self.with_MF_bias = args.TR_with_MF_bias
        self.test_in_TR_train = args.test_in_TR_Train
        self.TR_train_sampleTYpe = args.TR_sample_type
        print("with MF bias:", self.with_MF_bias)
        print("transfer type:", self.transfer_type)
        self.need_writer = args.need_writer
        self.MF_TrainDataset = {'alone': SampleDaset, 'all': PreSampleDatast}.get(args.MF_sample)
        self.writer = None
        if args.need_writer:
            from torch.utils.tensorboard import SummaryWriter   # optional dependency, imported lazily
            tag = "m-num" + str(args.multi_num) + "-MF-lr" + str(args.MF_lr) + "-l2-" + str(args.l2) + "e-" + \
                  str(args.MF_epochs) + "--TR-lr" + str(args.TR_lr) + "-l2-" + str(args.TR_l2) + "-e-" + \
                  str(args.TR_epochs) + str(args.TR_sample_type) + "user-norm" + str(args.norm)
            self.writer = SummaryWriter(comment=tag)
        if self.with_MF_bias:
            # The reference cannot run this flag with the convolutional transfers either (the only ones in scope): W_{t-1} gets d + 1
            # columns (model/transfer.py:347-354) while MFbase.user_laten(user) keeps d, and ConvTransfer_com.forward multiplies the two
            # (model/conv_transfer.py:93) -- measured on the reference here: "RuntimeError: The size of tensor a (33) must match the size
            # of tensor b (32) at non-singleton dimension 1", raised from MF_train_onestage (transfer.py:476) in the first stage.  Same
            # error type, raised up front.
            raise RuntimeError("--TR_with_MF_bias: the size of tensor a (%d) must match the size of tensor b (%d) at non-singleton dimension 1 "
                               "(W_{t-1} carries the bias column, the MF rows do not: the reference's conv / conv_com transfers fail the same way "
                               "in their first MF batch, model/conv_transfer.py:93; the flag only works with its MLP transfers, which are out of scope)"
                               % (laten_dim + 1, laten_dim))
        # W_{t-1}, W_hat_t, previous W_hat (model/transfer.py:358-364)
        wu, wi = self.MFbase.user_laten.weight.data, self.MFbase.item_laten.weight.data
        self.last_user_weight = torch.zeros_like(wu)
        self.last_item_weight = torch.zeros_like(wi)
        self.user_weight_hat = wu.clone()
        self.item_weight_hat = wi.clone()
        self.last_user_weight_hat = self.user_weight_hat.clone()
        self.last_item_weight_hat = self.item_weight_hat.clone()

        if self.transfer_type == "conv_com":
            self.transfer = ConvTransfer_com(laten_dim, laten_dim).to(self.device)
            self.transfer_type = "transfer2"
        elif self.transfer_type == "conv":
            self.transfer = ConvTransfer(laten_dim, laten_dim).to(self.device)
            self.transfer_type = "transfer2"
        elif self.transfer_type in ("transfer", "transfer2", "GRU", "transfer3", "conv_com2"):
            raise NotImplementedError("transfer type %r is one of the reference's unused variants "
                                      "(model/transfer.py:1-4); only conv_com and conv are built" % self.transfer_type)
        else:
            raise TypeError("No such type transfer!!!")

        self.dataset = datasets
        self.engine = _make_engine(self.device, laten_dim, max(args.MF_batch_size, args.TR_batch_size))
        self.MFbase._sml_engine = self.engine
        self.transfer._sml_engine = self.engine
        if use_dist:
            from . import dist as smldist
            self.dist = smldist.attach(self.engine, None, dist, hp=args)
            theta = self.engine.adopt(self.transfer) if hasattr(self.engine, "adopt") else None
            # (every rank built theta and loaded the item table from the same seeds / file: the broadcast is a guard)
            self.dist.sync_replicas([self.MFbase.item_laten.weight.data] +
                                    ([theta] if theta is not None else [p.data for p in self.transfer.parameters()]))
        self.MF_optimizer = _OptimizerInfo(args.MF_lr, 0)
        self.transfer_optimizer = _OptimizerInfo(args.TR_lr, args.TR_l2)

        self.recall, self.ndcg, self.test_num = [], [], []
        self.recall_5, self.ndcg_5 = [], []
        self.recall_10, self.ndcg_10 = [], []
        self.MF_itr = 0
        self.TR_itr = 0
        self.timing = {"mf": 0.0, "tr": 0.0, "updata": 0.0, "eval": 0.0, "mf_triples": 0, "tr_triples": 0}
        self._rows_cache = {}

    # ------------------------------------------------------------------ helpers
    def get_next_data(self, stage_id):
        pre, self._next_data = getattr(self, "_next_data", None), None
        if pre is not None and pre[0] == stage_id:
            sys.stdout.write(pre[1])          # the loader's lines, where the reference prints them
            return pre[2]
        return self.dataset.next_train(stage_id)

    def _prefetch_next(self, stage_id):
        """Load period `stage_id` and upload its test rows NOW -- called when a stage's kernels are all queued and
        the host would otherwise just wait for the device: the next stage then starts with its rows resident
        (the upload of 0.6 GB of validation rows is 13 ms during which the device used to have nothing queued).
        The period loader draws no random numbers, so loading early does not move the run's random streams; its
        prints are captured and replayed by get_next_data at the point the reference prints them."""
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            data = self.dataset.next_train(stage_id)
        self._next_data = (stage_id, buf.getvalue(), data)
        # what the next stage would otherwise build on the host while the device has nothing queued: the TR sampler of
        # D_{t+1} (np.unique, CSR lists: 1.5 ms; no random numbers) and the (user, item) columns of D_t
        try:
            if data[1] is not None and self.TR_train_sampleTYpe == "alone":
                self._sample_cache_next = self._build_sample(data[1])
            if data[0] is not None and self.MF_TrainDataset is PreSampleDatast:
                prev = getattr(self, "_pairs_cache", None)
                self._pairs_cache_next = (data[0], np.ascontiguousarray(data[0][:, :2], dtype=np.int64))
                if prev is None:
                    self._pairs_cache = self._pairs_cache_next
        except Exception:          # (a prefetch may never fail the stage that triggered it: the next stage builds them itself)
            self._sample_cache_next = None
        self._uploading = True                # (on a stream of its own: the queued kernels keep running meanwhile)
        try:
            for arr in (data[3], data[2]):
                if arr is not None:
                    self._rows(arr)
        finally:
            self._uploading = False

    def _rows(self, arr):
        """Device-resident copy of a test array (cached per array object)."""
        key = id(arr)
        hit = self._rows_cache.get(key)
        if hit is None or hit[0] is not arr:
            if len(self._rows_cache) > 4:
                self._rows_cache.clear()
                self._rank_cache = None        # its rows object went with the entries above
            local = arr if self.dist is None else self.dist.route_rows(arr, self.n_user_global)
            up = None
            if getattr(self, "_uploading", False) and torch.device(self.device).type == "cuda":
                up = self.__dict__.get("_upload_stream") or self.__dict__.setdefault("_upload_stream", torch.cuda.Stream(device=self.device))
            rows = DeviceRows(local, self.device, stream=up)
            rows.n_global = int(np.asarray(arr).shape[0])      # recall / ndcg are over ALL rows of the set
            hit = (arr, rows)
            self._rows_cache[key] = hit
        return hit[1]

    def _touch_tables(self):
        """The MF tables changed: cached validation ranks are stale."""
        self._version = getattr(self, "_version", 0) + 1

    def _ranks(self, rows):
        """Rank of every row's positive under the current tables; reused while the tables are unchanged
        (the reference re-evaluates identical tables several times per phase, and @20/@10/@5 share ranks).
        On the GPU engine this is a handle of an evaluation queued on a snapshot of the tables
        (engine.eval_submit): the training stream does not wait for it."""
        key = (rows, getattr(self, "_version", 0))          # holds the rows object itself: an id() can be recycled
        hit = getattr(self, "_rank_cache", None)
        if hit is None or hit[0][0] is not rows or hit[0][1] != key[1]:
            wu, wi = self.MFbase.user_laten.weight.data, self.MFbase.item_laten.weight.data
            if hasattr(rows, "wait_ready"):
                rows.wait_ready()               # rows uploaded ahead of their stage (_prefetch_next)
            if getattr(self, "_pending_transfer", False):
                # the tables this evaluation is about are the ones a deferred updata would write (_updata_for_tests):
                # forward and ranks are queued together on the evaluation stream, the MF tables are not touched
                ranks = self.engine.eval_submit_transferred(self.transfer, self.last_user_weight, self.user_weight_hat,
                                                            self.last_item_weight, self.item_weight_hat, rows.rows)
            elif hasattr(self.engine, "eval_submit"):
                ranks = self.engine.eval_submit(wu, wi, rows.rows)
            else:
                ranks = self.engine.eval_ranks(wu, wi, rows.rows)
            self._rank_cache = hit = (key, ranks)
        return hit[1]

    # -- deferred output.  Inside train_one_stage3 nothing the host does (random draws, batch building,
    # launches) depends on device results, so lines are queued with lazy numbers and printed, in the
    # reference's order, when the stage ends: the host never waits for the device mid-stage.
    def _emit(self, fn, *args):
        if getattr(self, "_defer", False):
            self._queue.append((fn, args))
        else:
            fn(*[_val(a) for a in args])

    def _flush_output(self):
        q, self._queue = getattr(self, "_queue", []), []
        for fn, args in q:
            fn(*[_val(a) for a in args])
        if hasattr(self.engine, "side_sync_check"):
            self.engine.side_sync_check()      # one check for all the evaluations collected above (not one read-back each)
        if not getattr(self, "_stage_failed", False):     # (a stage that raised must not enter collectives its peers may never reach)
            self._check_exchange("stage")

    def _check_exchange(self, where):
        """Several GPUs: a stage ends with proof that its exchange was whole -- no consumer of the one-shot peer exchange
        timed out (it would have gone on with a partial sum) and theta / the item table are still bit-identical on
        every rank (sml_amd.dist.DistContext.check_exchange raises otherwise).  The stage's results have just been
        waited for, so the synchronisation this needs costs nothing more."""
        if self.dist is None:
            return
        self.dist.flush_checks()
        reps = [self.MFbase.item_laten.weight.data]
        if hasattr(self.engine, "adopt"):
            reps.append(self.engine.adopt(self.transfer))
        else:
            reps += [p.data for p in self.transfer.parameters()]
        self.dist.check_exchange(self.engine, where, replicas=reps)

    def _pair(self, resolve, n):
        """(recall, ndcg) from a (hits, ndcg_sum) resolver; lazy while output is deferred."""
        if getattr(self, "_defer", False):
            box = _Lazy(resolve)
            return (_Lazy(lambda: box.get()[0] / n), _Lazy(lambda: torch.tensor(np.float32(box.get()[1] / n))))
        hits, ndcg = resolve()
        return hits / n, torch.tensor(np.float32(ndcg / n))

    def _metrics(self, ranks, n, topK):
        if isinstance(ranks, dict):               # a submitted evaluation (engine.eval_submit)
            pending = self.engine.eval_metrics_submit(ranks, topK)

            def local():
                if getattr(self, "_defer", False):
                    return self.engine.eval_result(pending, check=False)      # (_flush_output checks once for the whole stage)
                return self.engine.eval_result(pending)
        elif hasattr(self.engine, "eval_metrics_device"):
            out = self.engine.eval_metrics_device(ranks, topK)

            def local():
                h = out.cpu()
                return float(h[0]), float(h[1])
        else:
            res = self.engine.eval_metrics(ranks, topK)

            def local():
                return res
        resolve = local
        if self.dist is not None:
            # every rank evaluated the rows of ITS users: hits and ndcg sums add up.  (Resolved when the stage's lines
            # are printed -- in the same order on every rank, which run the same program.)
            def resolve():
                return tuple(self.dist.sum_scalars(local()))
        return self._pair(resolve, n)

    def _test(self, rows, topK):
        """recall@K, ndcg@K of the current tables on `rows` (reference evalution/evaluation2.py:8-26)."""
        t0 = time.time()
        D.loader_base_seed_draw()          # the draw the reference's DataLoader iteration makes
        out = self._metrics(self._ranks(rows), getattr(rows, "n_global", rows.rows.shape[0]), topK)
        self.timing["eval"] += time.time() - t0
        return out

    def _sample_dataset(self, arr):
        """SampleDaset(arr), built once per array (it holds no state that changes between epochs; the
        reference rebuilds it every phase).  Its constructor's "user max:" lines are replayed on every
        call, in order with the rest of the (possibly deferred) output."""
        hit = getattr(self, "_sample_cache", None)
        if hit is None or hit[0] is not arr:
            nxt = getattr(self, "_sample_cache_next", None)      # built ahead by _prefetch_next
            if nxt is not None and nxt[0] is arr:
                hit, self._sample_cache_next = nxt, None
            else:
                hit = self._build_sample(arr)
            self._sample_cache = hit
        text = hit[2]
        self._emit(lambda: print(text, end=""))
        return hit[1]

    @staticmethod
    def _build_sample(arr):
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            obj = SampleDaset(arr)
        return (arr, obj, buf.getvalue())

    def _pairs_of(self, arr):
        """Contiguous int64 (user, item) columns of a pre-sampled training array, built once per array (the dataset object
        itself is rebuilt every phase -- its constructor draws from numpy's generator -- and would re-read the two columns
        through the 8 KB rows each time)."""
        hit = getattr(self, "_pairs_cache", None)
        if hit is None or hit[0] is not arr:
            nxt = getattr(self, "_pairs_cache_next", None)
            if nxt is not None and nxt[0] is arr:
                hit = nxt
            else:
                hit = (arr, np.ascontiguousarray(arr[:, :2], dtype=np.int64))
            self._pairs_cache, self._pairs_cache_next = hit, None
        return hit[1]

    def transfer_variant_is_bce(self):
        """ConvTransfer_com.run_MF is called with its BCE default (a MEAN over the batch); ConvTransfer.run_MF has only
        the BPR SUM (model/conv_transfer.py:71-85, 113-126): decides how a split batch's loss terms are scaled."""
        return not isinstance(self.transfer, ConvTransfer)

    def _sum_losses(self, losses):
        """Per-batch losses of a split global batch: the ranks' parts add up (device tensor or numpy array)."""
        if isinstance(losses, torch.Tensor):
            return self.dist.sum_tensor_(losses)
        t = torch.as_tensor(np.asarray(losses, dtype=np.float64)).to(self.dist.device)
        return self.dist.sum_tensor_(t).cpu().numpy()

    @staticmethod
    def _epoch_loss(losses):
        """mean of the batch losses, accumulated in fp32 like the reference's `loss_all += loss.data`."""
        acc = np.float32(0)
        arr = _np(losses).astype(np.float32)
        for l in arr:
            acc = np.float32(acc + l)
        return float(np.float32(acc / np.float32(len(arr))))

    # ------------------------------------------------------------------ hot loop 1
    def MF_train_onestage(self, args, set_t, stage_id, val=None):
        """Train W_hat on D_t with theta frozen (reference model/transfer.py:417-534)."""
        # --need_adaptive (model/transfer.py:427, 490-499): beta = 0.1 there
        adaptive = 0.1 if args.need_adaptive else None
        self.transfer.eval()
        if val is not None:
            val = self._rows(val)
        self._emit(lambda: print("******MF (inner) training ******"))
        # (SampleDaset's constructor prints: built once per array, its lines replayed in order with the deferred
        # output; PreSampleDatast draws np.random.shuffle when constructed, so it is built here every time)
        train_set = self._sample_dataset(set_t) if self.MF_TrainDataset is SampleDaset else self.MF_TrainDataset(set_t)
        if isinstance(train_set, PreSampleDatast) and getattr(train_set, "_ui", 0) is None:
            train_set._ui = self._pairs_of(set_t)
        if val is not None:
            recall, ndcg = self._test(val, args.topK)
            self._emit(lambda r, n: print("before train MF test:recall:{:.4f} ndcg:{:.4f}".format(r, n)), recall, ndcg)
            self._log_mf(args, recall, ndcg, None)
        for epoch in range(args.MF_epochs):
            self.MFbase.train()
            self.transfer.eval()
            if getattr(args, "device_batches", 0) and hasattr(train_set, "epoch_triples_device") \
                    and hasattr(self.engine, "device_epoch"):
                # fast mode (an extension): the pass is shuffled and assembled on the device -- same distribution, not the
                # reference's torch / numpy streams; ONE draw from the shared torch generator seeds it on every rank alike
                triples = train_set.epoch_triples_device(self.engine, int(torch.randint(0, 2 ** 62, (1,))))
                failed = getattr(train_set, "_last_failed", None)
                if failed is not None and self.MF_TrainDataset is SampleDaset:
                    self._emit(lambda failed=failed: self._raise_if_failed(failed))
                if self.dist is not None:
                    triples = triples.cpu().numpy()        # (split by user owner on the host, as for the exact path)
            else:
                order = D.loader_order(len(train_set), shuffle=True)
                triples = train_set.epoch_triples(order)
            t0 = time.time()
            if self.dist is None:
                losses = self.engine.mf_stage_epoch(self.MFbase, self.transfer, self.last_user_weight,
                                                    self.last_item_weight, triples, args.MF_batch_size,
                                                    args.MF_lr, args.l2, norm=args.norm, bce=True, adaptive_beta=adaptive)
            else:
                # the same global batches, this rank's share of each (its users' triples); batch losses add up
                route = self.dist.route_epoch(triples, args.MF_batch_size, self.n_user_global,
                                              mean_loss=self.transfer_variant_is_bce())
                losses = self.engine.mf_stage_epoch(self.MFbase, self.transfer, self.last_user_weight,
                                                    self.last_item_weight, route.local_tri, route.cap,
                                                    args.MF_lr, args.l2, norm=args.norm, bce=True,
                                                    plan=route.plan, exchange=route.exchange(self.MFbase.user_laten.weight.shape[1]),
                                                    adaptive_beta=adaptive)
                losses = self._sum_losses(losses)
            self.engine.mf_flush(self.MFbase)
            self._touch_tables()
            self.timing["mf"] += time.time() - t0
            self.timing["mf_triples"] += triples.shape[0]
            bs = args.MF_batch_size
            loss_all = _Lazy(lambda losses=losses, bs=bs: self._epoch_loss(losses) / bs)
            if val is not None:
                recall, ndcg = self._test(val, args.topK)
                self._emit(lambda l, r, n, epoch=epoch: print("MF-stage:", stage_id, "epoch:", epoch, "loss:{:.5f}".format(l),
                                                              "recall:{:.4f}".format(r), "ndcg:{:.4f}".format(n)),
                           loss_all, recall, ndcg)
                self._log_mf(args, recall, ndcg, loss_all)
            else:
                self._emit(lambda l, epoch=epoch: print("MF-stage:", stage_id, "epoch:", epoch, "loss:", l), loss_all)

    @staticmethod
    def _raise_if_failed(failed):
        if int(failed.item()) != 0:
            raise RuntimeError("negative sampling does not terminate: a user owns (almost) every item")

    def _log_mf(self, args, recall, ndcg, loss):
        if self.writer is None:
            return
        recall, ndcg, loss = _val(recall), _val(ndcg), _val(loss)
        self.writer.add_scalar("Acc/MF-recall" + str(args.topK), recall, self.MF_itr)
        self.writer.add_scalar("Acc/MF-ndcg" + str(args.topK), float(ndcg), self.MF_itr)
        if loss is not None:
            self.writer.add_scalar("Loss/MF-loss", loss, self.MF_itr)
        self.writer.add_scalar("norm/user-norm", float((self.MFbase.user_laten.weight.data ** 2).sum(-1).mean()),
                               self.MF_itr)
        self.MF_itr += 1

    def _speculate_next_tr_pass(self, args, train_set, epoch):
        """Start drawing the NEXT transfer-stage pass on a helper thread (datasets.EpochSpeculation: adopted only if its
        predicted inputs turn out right, so the reference's random streams are untouched).  Next = the following epoch of
        this call, or the first epoch of the next phase of this stage -- before which the driver draws two torch seeds per
        MF epoch and the pre-sampled MF dataset's constructor shuffles its columns.  SML_TR_SPECULATE=0: off."""
        import os
        if os.environ.get("SML_TR_SPECULATE", "1") == "0" or not isinstance(train_set, D.offlineDataset_withsample):
            return
        ph = getattr(self, "_phase", None)               # (phase, phases of the stage, columns the MF dataset shuffles)
        if ph is None:
            return
        gaps = self.__dict__.get("_tr_gaps", {})
        if epoch + 1 < int(args.TR_epochs):
            nxt, shuffles = (ph[0], epoch + 1), ()
        else:
            if ph[0] + 1 >= ph[1] or self.MF_TrainDataset is not PreSampleDatast:
                return
            nxt, shuffles = (ph[0] + 1, 0), (ph[2],)
        # torch draws expected before that pass: what the same transition took in the previous stage, else (first stage) what
        # the transition into THIS pass took -- phases repeat; a wrong guess costs one discarded pass
        draws = gaps.get(nxt, gaps.get((ph[0], epoch)))
        if draws is None:
            return
        self._tr_spec = D.EpochSpeculation(train_set, draws, shuffles)

    # ------------------------------------------------------------------ hot loop 2
    def transfer_train_onestage(self, args, set_tt, stage_id, compute_performance=False, val=None):
        """Train theta on D_{t+1} with the tables frozen (reference model/transfer.py:644-749)."""
        # --clip_grad / --maxnorm_grad (model/transfer.py:656, 724-727): the norm of theta's gradient is clipped between
        # backward and the optimiser step; the engine then runs the TR step un-fused
        clip = float(args.maxnorm_grad) if args.clip_grad else None
        self._emit(lambda: print("********* this is Transfer model training stage ***********"))
        self.MFbase.eval()
        now_test = None
        if self.TR_train_sampleTYpe == "alone":
            train_set = self._sample_dataset(set_tt)
            compute_performance = False
            if val is not None:
                now_test = self._rows(val)
                compute_performance = True
        elif self.TR_train_sampleTYpe == 'all':
            now_test = self._rows(set_tt)
            train_set = PreSampleDatast(set_tt)
            compute_performance = True
        else:
            raise TypeError("no such TR sample type")
        def report_before(recall, ndcg):
            self._emit(lambda r, n: print("before train transfer test:recall:{:.4f} ndcg:{:.4f}".format(r, n)), recall, ndcg)
            if self.writer is not None:
                self.writer.add_scalar("Acc/tr-TR-recall@" + str(args.topK), _val(recall), self.TR_itr)
                self.writer.add_scalar("Acc/tr-TR-ndcg@" + str(args.topK), float(_val(ndcg)), self.TR_itr)
                self.TR_itr += 1

        if compute_performance:
            # (queued on a snapshot of the tables: it runs underneath the TR epoch that follows)
            report_before(*self._test(now_test, args.topK))
        s_time = time.time()
        for epoch in range(args.TR_epochs):
            self.transfer.train()
            if getattr(args, "device_batches", 0) and hasattr(train_set, "epoch_triples_device") \
                    and hasattr(self.engine, "sample_negatives"):
                # fast mode: permutation and rejection-sampled negatives on the device (same distribution, other streams)
                triples = train_set.epoch_triples_device(self.engine, int(torch.randint(0, 2 ** 62, (1,))))
                failed = train_set._last_failed

                def check_failed(failed=failed):
                    if int(failed.item()) != 0:
                        raise RuntimeError("negative sampling does not terminate: a user owns (almost) every item")
                self._emit(check_failed)       # read when the stage's output is flushed: no mid-stage host wait
            else:
                # (torch draws since the previous transfer pass's shuffle: what the next speculation has to predict)
                key = (getattr(self, "_phase", None) or (None,))[0], epoch
                seen = self.__dict__.setdefault("_tr_gaps", {})
                if getattr(self, "_tr_draw_mark", None) is not None:
                    seen[key] = D.torch_draws() - self._tr_draw_mark
                elif getattr(self, "_tr_first_mark", None) is not None:
                    seen["first"] = D.torch_draws() - self._tr_first_mark        # (from the end of the previous stage's queuing)
                self._tr_first_mark = None
                order = D.loader_order(len(train_set), shuffle=True)
                self._tr_draw_mark = D.torch_draws()
                spec = self.__dict__.pop("_tr_spec", None)
                triples = spec.adopt(train_set, order) if spec is not None else None
                self.timing["tr_spec_adopted"] = self.timing.get("tr_spec_adopted", 0) + (triples is not None)
                if triples is None:
                    triples = train_set.epoch_triples(order)
                self._speculate_next_tr_pass(args, train_set, epoch)
            t0 = time.time()
            if self.dist is None:
                losses = self.engine.tr_stage_epoch(self.transfer, self.last_user_weight, self.last_item_weight,
                                                    self.user_weight_hat, self.item_weight_hat, triples,
                                                    args.TR_batch_size, args.TR_lr, args.TR_l2, bce=True, clip_max_norm=clip)
            else:
                if isinstance(triples, torch.Tensor):
                    # --device_batches on several GPUs: every rank drew the SAME epoch (the seed came from the shared
                    # torch generator, the device generator is counter-based) -- nothing is communicated; the split by
                    # user owner runs on the host as for the exact path (one read-back per epoch)
                    triples = triples.cpu().numpy()
                route = self.dist.route_epoch(triples, args.TR_batch_size, self.n_user_global,
                                              mean_loss=self.transfer_variant_is_bce())
                losses = self.engine.tr_stage_epoch(self.transfer, self.last_user_weight, self.last_item_weight,
                                                    self.user_weight_hat, self.item_weight_hat, route.local_tri,
                                                    route.cap, args.TR_lr, args.TR_l2, bce=True, plan=route.plan,
                                                    clip_max_norm=clip)
                losses = self._sum_losses(losses)
            self.timing["tr"] += time.time() - t0
            self.timing["tr_triples"] += triples.shape[0]
            bs = args.TR_batch_size
            loss_all = _Lazy(lambda losses=losses: self._epoch_loss(losses))
            self._emit(lambda dt: print("one epcohs TR time cost:", dt), time.time() - s_time)
            if self.writer is not None:
                self.writer.add_scalar("Loss/TR-loss", _val(loss_all) / bs, self.TR_itr)
            if compute_performance:
                self.updata()
                recall, ndcg = self._test(now_test, args.topK)
                # (the full-width punctuation is the reference's, model/transfer.py:741)
                self._emit(lambda l, r, n, epoch=epoch: print(
                    "stage:{}, epcoh：{}，loss:{:.4f},*****val result  reacll:{:.4f}  ndcg:{:.4f}".format(
                        stage_id, epoch, l / bs, r, n)), loss_all, recall, ndcg)
                if self.writer is not None:
                    self.writer.add_scalar("Acc/tr-TR-recall@" + str(args.topK), _val(recall), self.TR_itr)
                    self.writer.add_scalar("Acc/tr-TR-ndcg@" + str(args.topK), float(_val(ndcg)), self.TR_itr)
            else:
                self._emit(lambda l, epoch=epoch: print("stage:", stage_id, "epoch:", epoch, "transfer train loss:", l / bs),
                           loss_all)
        self._emit(lambda: print("stage ", stage_id, " transfer trained finished!!!!"))

    # ------------------------------------------------------------------ one period
    def _real_test(self, now_test):
        """recall/ndcg @20, @10, @5 on the coming period (model/transfer.py:846-868)."""
        self.test_num.append(now_test.shape[0])
        rows = self._rows(now_test)
        for k, tag, rl, nl in ((20, "", self.recall, self.ndcg), (10, "@10 ", self.recall_10, self.ndcg_10),
                               (5, "@5 ", self.recall_5, self.ndcg_5)):
            recall, ndcg = self._test(rows, k)

            def show(r, n, tag=tag, rl=rl, nl=nl):
                print("test result --------- {}reacll:{:.4f}  ndcg:{:.4f}".format(tag, r, n))
                rl.append(r)
                nl.append(n.cpu().numpy())
            self._emit(show, recall, ndcg)

    def train_one_stage3(self, args, stage_id):
        """One period of SML (reference model/transfer.py:753-881).  False when no data is left."""
        self.save_MF_weight(save_as='last')
        set_t, set_tt, now_test, val = self.get_next_data(stage_id)
        if set_t is None:
            return False
        self._defer, self._queue = self.writer is None, []
        self._stage_failed = True
        try:
            # the training kernels of the stage run on the training partition of the chip, the queued evaluations on
            # the side stream's own CUs (HipEngine.partition; the CPU test double has no such thing)
            scope = self.engine.partition() if hasattr(self.engine, "partition") else contextlib.nullcontext()
            with scope:
                more = self._stage_body(args, stage_id, set_t, set_tt, now_test, val)
            if more and getattr(self, "_prefetch", False):
                self._prefetch_next(stage_id + 1)
                self._speculate_first_tr_pass()
            self._stage_failed = False
            return more
        finally:
            self._defer = False
            self._pending_transfer = False       # (a stage that raised half-way leaves no deferred forward behind)
            self._flush_output()

    def _speculate_first_tr_pass(self):
        """The NEXT stage's first transfer pass, drawn while this stage's kernels run and the host only waits for them
        (its sampler was built by _prefetch_next).  Between here and that pass the driver makes the torch draws the same
        stretch took one stage earlier (counted) and the MF dataset's constructor shuffles its columns once; a wrong guess
        -- a test stage that trains no transfer, the first stages -- is discarded (datasets.EpochSpeculation)."""
        import os
        self._tr_first_mark = D.torch_draws()
        nxt, data = getattr(self, "_sample_cache_next", None), getattr(self, "_next_data", (None, None, (None,)))[2]
        draws = self.__dict__.get("_tr_gaps", {}).get("first")
        if (os.environ.get("SML_TR_SPECULATE", "1") == "0" or nxt is None or draws is None or data[0] is None
                or self.MF_TrainDataset is not PreSampleDatast or not isinstance(nxt[1], D.offlineDataset_withsample)):
            return
        self._tr_spec = D.EpochSpeculation(nxt[1], draws, (int(data[0].shape[1]) - 1,))
        self._tr_spec.first_of_stage = True

    def _stage_body(self, args, stage_id, set_t, set_tt, now_test, val):
        self._tr_draw_mark = None                        # (transfer-pass speculation inside a stage: from its second pass on)
        if not getattr(self.__dict__.get("_tr_spec"), "first_of_stage", False):
            self.__dict__.pop("_tr_spec", None)
        if now_test is not None and set_tt is None:
            # --TR_stop_: theta frozen during the test periods
            s_time = time.time()
            self._emit(lambda: print("stop train transfer while test###!!!!!"))
            args.MF_epochs = 2
            self.MF_train_onestage(args, set_t, stage_id, val=val)
            self.MFbase.eval()
            self.save_MF_weight(save_as='hat')
            self.updata()
            self._emit(lambda dt: print("only traning time cost:", dt), time.time() - s_time)
            self._real_test(now_test)
            self._emit(lambda dt: print("include test time cost:", dt), time.time() - s_time)
            return True
        for phase in range(args.multi_num):
            self.MF_train_onestage(args, set_t, stage_id, val=val)
            self.MFbase.eval()
            self.save_MF_weight(save_as='hat')
            if self.writer is not None:
                self.writer.add_scalars("Scale/user_weight", {"weight_hat": torch.norm(self.user_weight_hat).item(),
                                                              "weight_last": torch.norm(self.last_user_weight).item()},
                                        stage_id)
                self.writer.add_scalars("Scale/item_weight", {"weight_hat": torch.norm(self.item_weight_hat).item(),
                                                              "weight_last": torch.norm(self.last_item_weight).item()},
                                        stage_id)
            self._updata_for_tests(args, val)
            if now_test is not None and phase == 0:
                # test D_{t+1} with the first outer loop's model, before it trains theta
                self._real_test(now_test)
            self._phase = (phase, int(args.multi_num), int(set_t.shape[1]) - 1 if hasattr(set_t, "shape") and len(set_t.shape) == 2 else 0)
            try:
                self.transfer_train_onestage(args, set_tt, stage_id, val=val)
            finally:
                self._phase = None
            self._materialise_tables()
            if args.Load_W_hat:
                self.load_MFbase_weight(self.user_weight_hat, self.item_weight_hat)
        self.updata()
        return True

    # ------------------------------------------------------------------ hot loop 3
    def _updata_for_tests(self, args, val):
        """The updata between save_MF_weight('hat') and the transfer stage (model/transfer.py:829).  Only TESTS read what it
        writes -- the phase-0 test of D_{t+1} and "before train transfer" -- before the updata behind the first transfer
        epoch overwrites it (the transfer epochs read the last / hat tables, never the MF tables).  When that later updata
        is certain (validation on, at least one transfer epoch) and the engine can queue a forward with an evaluation
        (HipEngine.eval_submit_transferred), the forward is NOT run here: the tables count as changed, and every test
        until the next updata queues forward + ranks on the evaluation stream.  Otherwise: updata() as the reference."""
        eng = self.engine
        later = (self.TR_train_sampleTYpe == 'all' or (self.TR_train_sampleTYpe == 'alone' and val is not None)) \
            and int(args.TR_epochs) > 0
        wu, wi = self.MFbase.user_laten.weight.data, self.MFbase.item_laten.weight.data
        if later and self.transfer_type == 'transfer2' and hasattr(eng, "eval_submit_transferred") \
                and eng.can_submit_transferred(wu, wi):
            self.MFbase.eval()
            self.transfer.eval()
            self._pending_transfer = True
            self._touch_tables()
            return
        self.updata()

    def _materialise_tables(self):
        """A deferred updata nothing overwrote (no transfer epoch ran after all): run it now."""
        if getattr(self, "_pending_transfer", False):
            self.updata()

    def updata(self):
        """W <- transfer(W_{t-1}, W_hat) over every user and item row (model/transfer.py:884-902)."""
        self.MFbase.eval()
        self.transfer.eval()
        if self.transfer_type != 'transfer2':
            raise TypeError("No such type transfer!!!")
        self._pending_transfer = False
        t0 = time.time()
        self.engine.updata(self.transfer, self.last_user_weight, self.user_weight_hat, self.last_item_weight,
                           self.item_weight_hat, self.MFbase.user_laten.weight.data, self.MFbase.item_laten.weight.data)
        self._touch_tables()
        self.timing["updata"] += time.time() - t0

    def save_MF_weight(self, save_as="last"):
        """'last': W_{t-1} <- W.  'hat': previous W_hat <- W_hat; W_hat <- W.  (model/transfer.py:911-943)"""
        wu, wi = self.MFbase.user_laten.weight.data, self.MFbase.item_laten.weight.data
        copy = getattr(self.engine, "copy_tables", None) or (lambda pairs: [d.copy_(s) for d, s in pairs])
        if save_as == "last":
            copy([(self.last_user_weight, wu), (self.last_item_weight, wi)])
        elif save_as == "hat":
            copy([(self.last_user_weight_hat, self.user_weight_hat), (self.last_item_weight_hat, self.item_weight_hat)])
            copy([(self.user_weight_hat, wu), (self.item_weight_hat, wi)])       # (own launch: reads what the first wrote from)
        else:
            raise TypeError("save MFbase weight type is wrong")

    def load_MFbase_weight(self, user_weight, item_weight):
        """Overwrite the MF tables in place; the Adam moments are left as they are, as in
        the reference (model/transfer.py:945-959, note at :764)."""
        self.MFbase.user_laten.weight.data.copy_(user_weight)
        self.MFbase.item_laten.weight.data.copy_(item_weight)
        self._touch_tables()

    # ------------------------------------------------------------------ the sequence
    def run(self, args):
        """All periods, then the weighted averages (reference model/transfer.py:965-1029)."""
        pass_num = args.pass_num
        self._prefetch = True                 # consecutive stages: each one loads its successor's data while it drains
        for pass_id in range(pass_num):
            stage_id = 0
            self._next_data = None
            self.dataset.reinit()
            while True:
                if self.train_one_stage3(args, stage_id):
                    stage_id += 1
                    if pass_id < (pass_num - 1) and stage_id >= 19:
                        break
                    continue
                print(str(pass_id) + "--trained over!!!!!")
                self._report()
                break

    def _report(self):
        test_num = np.array(self.test_num)
        n3 = round(test_num.shape[0] * 1 / 3)
        val_num = test_num[0:n3]
        test_num = test_num[n3:-1]        # the last period is dropped, as in the reference
        recall = np.array(self.recall)
        ndcg = np.array(self.ndcg)
        print(test_num)
        print(recall)
        print(ndcg)
        print("include stage 0 of test:")
        val_w = val_num * 1.0 / val_num.sum()
        test_w = test_num * 1.0 / test_num.sum()
        for k, rl, nl in ((20, self.recall, self.ndcg), (10, self.recall_10, self.ndcg_10),
                          (5, self.recall_5, self.ndcg_5)):
            recall = np.array(rl)
            ndcg = np.array(nl)
            if k != 20:
                print("\n")
            print("val average recall@%d:" % k, (recall[0:n3] * val_w).sum())
            print("val average ndcg@%d:" % k, (ndcg[0:n3] * val_w).sum())
            print("test average recall@%d:" % k, (recall[n3:-1] * test_w).sum())
            print("test average ndcg@%d:" % k, (ndcg[n3:-1] * test_w).sum())
