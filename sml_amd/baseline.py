"""The baselines' bare-MF retraining step as one more caller of the a3 kernels.

The reference's model/baseline.py is a separate program (own CLI, hard-coded checkpoint paths, the SPMF
reservoir): out of scope here.  What IS on this build's path is the inner loop its fine-tune / full-retrain
baselines run every period -- `SPMF.run_one_stage2` (model/baseline.py:306-386): rejection-sampled negatives,
BCE + L2, torch.optim.Adam over the dense tables, recall/ndcg of the period's test rows -- because that loop is the
only live caller of the bare embed+loss step (model/baseline.py:343-361).  This module hosts exactly that
method on the HIP engine (HipEngine.bare_adam_epoch = sml_embed_loss_adam_epoch, evaluation = the rank kernel),
with the reference's class / method names and printed lines so its recorded run (fixture G10) can be replayed
against it.  The reservoir (`Reservious`), `SPMF.run_one_stage` / `base_train` / `run`, `StreamingData` and the
command line are not provided.
"""
import time

import numpy as np
import torch

from . import datasets as D
from .datasets import offlineDataset_withsample
from .engine import get_engine
from .mf import MFbasemode


class SPMF(object):
    """Fine-tune / full-retrain MF baseline, one period at a time (`run_one_stage2` only).

    `datasets` supplies `get_next(stage_id, types=...) -> (train [n,2], test [n,2+neg])` and the arrays
    `test_new_user` / `test_new_item` (the surface of the reference's StreamingData that this method touches)."""

    def __init__(self, args, datasets, user_num, item_num, laten_dim, device=None, engine=None):
        if engine is not None:                      # an injected engine (tests drive the control flow with a double)
            device = engine.device if device is None else device
        elif device is None:
            if not torch.cuda.is_available():
                raise RuntimeError("the MF baseline step runs on the HIP engine only; there is no CPU path")
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = torch.device(device)
        self.MFbase = MFbasemode(num_user=user_num, num_item=item_num, laten_factor=laten_dim).to(self.device)
        self.lr = args.lr
        self.dataset = datasets
        self.new_user = np.asarray(datasets.test_new_user).astype(np.int64)
        self.new_item = np.asarray(datasets.test_new_item).astype(np.int64)
        self.batch_size = args.batch_size
        self.lambda_u, self.lambda_i = args.l2_u, args.l2_i
        self.epochs = args.epochs
        self.early_stop = getattr(args, "pool_init_type", 0) == 1     # the Adressa setting of the reference
        self.recall, self.ndcg, self.hit_new_user, self.hit_new_item, self.test_num = [], [], [], [], []
        self.engine = engine if engine is not None else get_engine(self.device, laten_dim, max(int(args.batch_size), 4096))
        self.MFbase._sml_engine = self.engine

    def get_next_data(self, stage_id, types="only_new"):
        return self.dataset.get_next(stage_id, types=types)

    def _epoch(self, train):
        """One shuffled pass: batches drawn with the reference's random-number consumption, trained in ONE engine
        call; returns the mean batch loss accumulated in fp32 as `loss_all += loss.data` does."""
        tri = train.epoch_triples(D.loader_order(len(train), shuffle=True))
        self.MFbase.train()
        losses = self.engine.bare_adam_epoch(self.MFbase, tri, self.batch_size, self.lr, self.lambda_u, self.lambda_i, bce=True)
        self.engine.mf_flush(self.MFbase)          # dense-Adam state of every row is current again
        arr = losses.detach().cpu().numpy() if isinstance(losses, torch.Tensor) else np.asarray(losses)
        acc = np.float32(0)
        for l in arr.astype(np.float32):
            acc = np.float32(acc + l)
        return float(acc / np.float32(len(arr)))

    def run_one_stage2(self, stage_id, read_data_type='only_new'):
        """One period of the fine-tune ('only_new') / full-retrain ('not_only_new') baseline."""
        set_t, now_test = self.get_next_data(stage_id, types=read_data_type)
        if set_t is None:
            return False
        self.test_num.append(now_test.shape[0])
        train = offlineDataset_withsample(set_t)
        print("start train...")
        show = lambda tag, r, n, *more: print(tag, "recall(5,10,20):", r, "ndcg (5,10,20):", n, *more)
        rec, nd, _, _ = self.test(now_test)
        show("before train test---", rec, nd)
        best = (0, None, None)
        stale = 0
        for epoch in range(self.epochs):
            t0 = time.time()
            loss = self._epoch(train)
            print("epoch: {} ,time:{:.1f}, loss:{:.4f}".format(epoch, time.time() - t0, loss))
            stale += 1
            if epoch % 5:
                continue
            rec, nd, _, _ = self.test(now_test)
            show("        epoch test---", rec, nd)
            if rec[-1] > best[0]:
                best, stale = (rec[-1], rec, nd), 0
            if self.early_stop and stale > 5:
                break
        rec, nd, hit_u, hit_i = self.test(now_test)
        print("max result ", best[1], best[2])
        show("FInal test---", rec, nd, "hit user:", hit_u, "hit item:", hit_i)
        self.recall.append(rec)
        self.ndcg.append(nd)
        self.hit_new_user.append(hit_u)
        self.hit_new_item.append(hit_i)
        return True

    def test(self, test_data, topk=(5, 10, 20)):
        """(recall@k, ndcg@k for every k, share of @topk[-1] hits on new users, ... on new items) over the period's
        test rows (model/baseline.py:388-443)."""
        self.MFbase.eval()
        n = test_data.shape[0]
        rows = torch.from_numpy(np.ascontiguousarray(test_data)).to(self.device).long()
        ranks = self.engine.eval_ranks(self.MFbase.user_laten.weight.data, self.MFbase.item_laten.weight.data, rows)
        pairs = [self.engine.eval_metrics(ranks, k) for k in topk]
        hit_rows = np.asarray(test_data)[(ranks < topk[-1]).nonzero()[:, 0].cpu().numpy()]
        on_new_user = int(np.isin(hit_rows[:, 0], self.new_user).sum())
        on_new_item = int(np.isin(hit_rows[:, 1], self.new_item).sum())
        return (np.array([p[0] for p in pairs]) / n, np.array([p[1] for p in pairs]) / n, on_new_user / n, on_new_item / n)
