"""MF baselines of the reference (model/baseline.py): full-retrain MF, fine-tune MF and SPMF, on the
HIP engine.  Same class / method names, arguments, prints and control flow as the reference's SPMF
class; the inner loops -- gather, BCE + L2 loss, gradients, torch.optim.Adam over the dense tables --
are HipEngine.bare_adam_epoch (sml_embed_loss_adam_epoch: the lazy dense-Adam form of the a3 step) and
the evaluation is the engine's rank kernel.

What differs from the reference on purpose:
  * batches are built in-process (the reference uses DataLoader workers, whose numpy streams are
    process-local and unseeded: its batch composition is not reproducible; here it is the
    `num_workers=0` sequence, bit for bit -- tests/test_host_logic.py);
  * an epoch's batches are drawn first and trained in ONE engine call (nothing drawn depends on the model
    inside an epoch: SPMF's sampling weights are computed once per stage, model/baseline.py:247);
  * base_train's hard-coded checkpoint paths (model/baseline.py:215, 221) are replaced by `save_dir`
    (None: nothing is written).
"""
import argparse
import os
import time

import numpy as np
import torch

from . import datasets as D
from .datasets import offlineDataset_withsample
from .engine import get_engine
from .mf import MFbasemode


def test_hit_new(data, have_idx, new_user, new_item):
    """How many of the hit rows belong to a new user / a new item (model/baseline.py:18-30)."""
    hit = np.asarray(data)[np.asarray(have_idx)][:, 0:2]
    return int(np.isin(hit[:, 0], np.asarray(new_user)).sum()), int(np.isin(hit[:, 1], np.asarray(new_item)).sum())


class Reservious(object):
    """SPMF's reservoir (model/baseline.py:68-100)."""

    def __init__(self, length):
        self.t = 0
        self.len = length
        self.pool = np.zeros((length, 2), dtype=np.int64)
        print("pool size:", self.pool.shape)
        self.pool_have = 0

    def updata(self, new_data):
        if self.t <= self.len:
            new_num = new_data.shape[0]
            max_id = min(self.len, self.pool_have + new_num)
            self.pool[self.pool_have:max_id] = new_data[:max_id - self.pool_have]
            if max_id != self.len:
                new_data = new_data[max_id - self.pool_have:]
            self.pool_have = self.pool_have + max_id          # (sic: model/baseline.py:85)
            self.t = max_id
        new_num = new_data.shape[0]
        p = self.len * 1.0 / (self.t + np.arange(new_num) + 1)
        m = np.random.rand(new_num)
        select_data = new_data[np.where(m < p)]
        for i in range(select_data.shape[0]):
            idx = np.random.randint(0, self.len, 1)
            self.pool[idx] = select_data[i]
        self.t += new_num

    def init_pool(self, new_data):
        num = new_data.shape[0]
        np.random.randint(0, num, self.len)                   # drawn and unused by the reference (:96)
        self.pool[:] = new_data[-self.len:]
        self.pool_have = self.len
        self.t = num


class SPMF(object):
    """MF-base baselines: SPMF, full-retrain MF, fine-tune MF (model/baseline.py:102-556)."""

    def __init__(self, args, datasets, user_num, item_num, laten_dim, device=None, save_dir=None, engine=None):
        if engine is not None:                      # an injected engine (tests drive the control flow with a double)
            device = engine.device if device is None else device
        elif device is None:
            if not torch.cuda.is_available():
                raise RuntimeError("the MF baselines run on the HIP engine only; there is no CPU path")
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = torch.device(device)
        self.MFbase = MFbasemode(num_user=user_num, num_item=item_num, laten_factor=laten_dim).to(self.device)
        print("args lr:", args.lr)
        self.lr = args.lr
        # kept for its printed state and for callers that inspect it; the steps are taken by the engine
        self.optimizer = torch.optim.Adam(self.MFbase.parameters(), lr=args.lr, weight_decay=0)
        print("optimizer state:", self.optimizer.state_dict())
        self.pool_size = args.pool_size
        self.Reservious = Reservious(self.pool_size)
        print("self.pool_size:", self.pool_size)
        self.all_item = np.ones(0, dtype=np.int64)
        self.dataset = datasets
        self.new_user = np.asarray(datasets.test_new_user).astype(np.int64)
        self.new_item = np.asarray(datasets.test_new_item).astype(np.int64)
        self.neg_num = args.neg_num
        self.batch_size = args.batch_size
        self.lambda_u = args.l2_u
        self.lambda_i = args.l2_i
        self.recall, self.ndcg, self.hit_new_user, self.hit_new_item = [], [], [], []
        self.epochs = args.epochs
        self.run_stage = 0
        self.test_num = []
        self.user_hit = None
        self.pool_init_type = args.pool_init_type
        self.save_dir = save_dir
        self.engine = engine if engine is not None else get_engine(self.device, laten_dim, max(int(args.batch_size), 4096))
        self.MFbase._sml_engine = self.engine

    # ------------------------------------------------------------------ data
    def get_next_data(self, stage_id, types="only_new"):
        return self.dataset.get_next(stage_id, types=types)

    def base_train_not_train(self, stage_id):
        set_t, now_test = self.get_next_data(stage_id, types="not_only_new")
        if self.pool_init_type == 1:
            self.Reservious.init_pool(set_t)
        F_recall, F_ndcg, _, _ = self.test(now_test)
        print("before train test---", "recall(5,10,20):", F_recall, "ndcg (5,10,20):", F_ndcg)
        if self.pool_init_type == 0:
            self.updata_reservious(set_t)

    def init_pool(self):
        pass

    # ------------------------------------------------------------------ the bare step, an epoch at a time
    def _train_epoch(self, triples, l2_u, l2_i):
        """All batches of one epoch: BCE + L2, dense Adam (model/baseline.py:188-201 / 343-361).
        Returns the batch losses (device tensor)."""
        self.MFbase.train()
        losses = self.engine.bare_adam_epoch(self.MFbase, triples, self.batch_size, self.lr, l2_u, l2_i, bce=True)
        self.engine.mf_flush(self.MFbase)          # dense-Adam state of every row is current again
        return losses

    @staticmethod
    def _mean_loss(losses):
        """`loss_all += loss.data` over the batches in fp32, as the reference accumulates it."""
        acc = np.float32(0)
        arr = losses.detach().cpu().numpy() if isinstance(losses, torch.Tensor) else np.asarray(losses)
        for l in arr.astype(np.float32):
            acc = np.float32(acc + l)
        return acc, len(losses)

    def base_train(self, stage_id, epochs, l2_u, l2_i):
        """Pretrain the MF base model for the baselines (model/baseline.py:161-224)."""
        print("********base train: (l2_u,l2_i): ({},{})*****".format(l2_u, l2_i))
        set_t, now_test = self.get_next_data(stage_id, types="not_only_new")
        train = offlineDataset_withsample(set_t)
        max_recall20, max_REC, max_Ndcg, max_epoch, not_change_num = 0, None, None, 0, 0
        for epoch in range(epochs):
            s_time = time.time()
            tri = train.epoch_triples(D.loader_order(len(train), shuffle=True))
            acc, nb = self._mean_loss(self._train_epoch(tri, l2_u, l2_i))
            loss_all = acc / np.float32(nb * self.batch_size)
            print("epoch:{}, time:{:.1f}, loss:{:.4f}".format(epoch, time.time() - s_time, float(loss_all)))
            if (epoch % 2) == 0:
                F_recall, F_ndcg, _, _ = self.test(now_test)
                not_change_num += 1
                if F_recall[-1] > max_recall20:
                    max_recall20, max_REC, max_Ndcg, max_epoch, not_change_num = F_recall[-1], F_recall, F_ndcg, epoch, 0
                    self._save("best-mean-start29-spmf--%s-%slr.pt" % (l2_u, self.lr))
                print("test---", "recall(5,10,20):", F_recall, "ndcg (5,10,20):", F_ndcg, "max reccall", max_REC,
                      "max epoch:", max_epoch)
                if not_change_num > 50:
                    print("max not change up to 20 epochs, stop ......")
                    break
            if epoch % 50 == 0:
                self._save("mean-start29-spmf-%d-%s-%slr.pt" % (epoch, l2_u, self.lr))
        F_recall, F_ndcg, _, _ = self.test(now_test)
        print("FInal test---", "recall(5,10,20):", F_recall, "ndcg (5,10,20):", F_ndcg)
        print("max: epoch", max_epoch, "max_recall:", max_REC, "max_Ndcg", max_Ndcg)

    def _save(self, name):
        if self.save_dir is not None:
            torch.save(self.MFbase.state_dict(), os.path.join(self.save_dir, name))

    def _stage_head(self, set_t):
        self.all_item = np.union1d(self.all_item, set_t[:, 1])
        if self.Reservious.pool_have > 0:
            return np.concatenate([self.Reservious.pool[0:self.Reservious.pool_have], set_t], axis=0), True
        return set_t, False

    def run_one_stage(self, stage_id):
        """SPMF: one period (model/baseline.py:227-304)."""
        set_t, now_test = self.get_next_data(stage_id)
        if set_t is None:
            return False
        self.test_num.append(now_test.shape[0])
        train_data, _ = self._stage_head(set_t)
        self.user_hit_num_in_W_R(train_data)
        itr = round(train_data.shape[0] / self.batch_size)
        p = self.compute_R_W_P(train_data)
        print("start train...")
        F_recall, F_ndcg, _, _ = self.test(now_test)
        print("before train test---", "recall(5,10,20):", F_recall, "ndcg (5,10,20):", F_ndcg)
        max_recall20, max_recall, max_Ndcg, not_chang = 0, None, None, 0
        for epoch in range(self.epochs):
            s_time = time.time()
            batches = [np.concatenate(self.sample_batch(train_data, self.batch_size, p, self.neg_num), axis=1)
                       for _ in range(itr)]
            acc, nb = self._mean_loss(self._train_epoch(np.concatenate(batches, 0), self.lambda_u, self.lambda_i))
            loss_all = acc / np.float32(nb)
            print("epoch: {} ,time:{:.1f}, loss:{:.4f}".format(epoch, time.time() - s_time, float(loss_all)))
            not_chang += 1
            F_recall, F_ndcg, _, _ = self.test(now_test)
            print("        epoch test---", "recall(5,10,20):", F_recall, "ndcg (5,10,20):", F_ndcg)
            if max_recall20 < F_recall[-1]:
                max_recall20, max_recall, max_Ndcg, not_chang = F_recall[-1], F_recall, F_ndcg, 0
            if not_chang >= 5 and self.pool_init_type == 1:
                break
        self.updata_reservious(set_t)
        F_recall, F_ndcg, hit_new_user, hit_new_item = self.test(now_test)
        print("FInal test---", "recall(5,10,20):", F_recall, "ndcg (5,10,20):", F_ndcg, "hit new user:", hit_new_user,
              "hit new item:", hit_new_item)
        self.recall.append(F_recall)
        self.ndcg.append(F_ndcg)
        return True

    def run_one_stage2(self, stage_id, read_data_type='only_new'):
        """Full-retrain ('not_only_new') / fine-tune ('only_new') MF: one period (model/baseline.py:306-386)."""
        set_t, now_test = self.get_next_data(stage_id, types=read_data_type)
        if set_t is None:
            return False
        self.test_num.append(now_test.shape[0])
        train_data, pooled = self._stage_head(set_t)
        if pooled:
            print("pool having.....")
        self.user_hit_num_in_W_R(train_data)
        train = offlineDataset_withsample(train_data)
        print("start train...")
        F_recall, F_ndcg, _, _ = self.test(now_test)
        print("before train test---", "recall(5,10,20):", F_recall, "ndcg (5,10,20):", F_ndcg)
        max_recall20, max_recall, max_Ndcg, not_chang = 0, None, None, 0
        for epoch in range(self.epochs):
            s_time = time.time()
            tri = train.epoch_triples(D.loader_order(len(train), shuffle=True))
            acc, nb = self._mean_loss(self._train_epoch(tri, self.lambda_u, self.lambda_i))
            loss_all = acc / np.float32(nb)
            print("epoch: {} ,time:{:.1f}, loss:{:.4f}".format(epoch, time.time() - s_time, float(loss_all)))
            not_chang += 1
            if epoch % 5 == 0:
                F_recall, F_ndcg, _, _ = self.test(now_test)
                print("        epoch test---", "recall(5,10,20):", F_recall, "ndcg (5,10,20):", F_ndcg)
                if max_recall20 < F_recall[-1]:
                    max_recall20, max_recall, max_Ndcg, not_chang = F_recall[-1], F_recall, F_ndcg, 0
                if not_chang > 5 and self.pool_init_type == 1:
                    break
        F_recall, F_ndcg, hit_newu, hit_newi = self.test(now_test, stage_idx=stage_id)
        print("max result ", max_recall, max_Ndcg)
        print("FInal test---", "recall(5,10,20):", F_recall, "ndcg (5,10,20):", F_ndcg, "hit user:", hit_newu,
              "hit item:", hit_newi)
        self.recall.append(F_recall)
        self.ndcg.append(F_ndcg)
        self.hit_new_user.append(hit_newu)
        self.hit_new_item.append(hit_newi)
        return True

    # ------------------------------------------------------------------ evaluation
    def test(self, test_data, topk=[5, 10, 20], stage_idx=None):
        """recall / ndcg at every k plus the share of @topk[-1] hits on new users / new items
        (model/baseline.py:388-443: summed over 1024-row batches, divided by the number of rows)."""
        self.MFbase.eval()
        test_num = test_data.shape[0]
        rows = torch.from_numpy(np.ascontiguousarray(test_data)).to(self.device).long()
        wu, wi = self.MFbase.user_laten.weight.data, self.MFbase.item_laten.weight.data
        ranks = self.engine.eval_ranks(wu, wi, rows)
        recall, ndcg = [], []
        for k in topk:
            hits, nd = self.engine.eval_metrics(ranks, k)
            recall.append(hits)
            ndcg.append(nd)
        hit_idx = (ranks < topk[-1]).nonzero()[:, 0].cpu().numpy()
        hit_u, hit_i = test_hit_new(test_data, hit_idx, self.new_user, self.new_item)
        return (np.array(recall) / test_num, np.array(ndcg) / test_num, hit_u * 1.0 / test_num, hit_i * 1.0 / test_num)

    # ------------------------------------------------------------------ SPMF sampling
    def updata_reservious(self, train_data):
        self.Reservious.updata(train_data)

    def compute_R_W_P(self, R_TR_data):
        """Sampling probability from the score rank of every (user, item) in reservoir + new data
        (model/baseline.py:448-476)."""
        self.MFbase.eval()
        user = torch.from_numpy(np.ascontiguousarray(R_TR_data[:, 0])).to(self.device)
        item = torch.from_numpy(np.ascontiguousarray(R_TR_data[:, 1])).to(self.device)
        _, _, score = self.MFbase(user, item)
        rank_idx = torch.argsort(score, descending=True)
        num = rank_idx.shape[0]
        p = torch.zeros_like(score)
        p[rank_idx] = (torch.arange(num, device=score.device) + 1).float()
        w = torch.exp(p * 1.0 / num)
        return (w / w.sum()).cpu().numpy()

    def user_hit_num_in_W_R(self, data):
        if self.user_hit is None:
            self.user_hit = {}
        for u, i in zip(data[:, 0].tolist(), data[:, 1].tolist()):
            self.user_hit.setdefault(u, set()).add(i)

    def sample_batch(self, data, batch_size, p, neg_num):
        """One weighted batch with rejection-sampled negatives (model/baseline.py:489-503)."""
        bat_data = data[np.random.choice(np.arange(data.shape[0]), batch_size, p=p)]
        bat_user, bat_item = bat_data[:, 0], bat_data[:, 1]
        bat_neg = []
        for i in range(bat_data.shape[0]):
            m = np.random.choice(self.all_item, neg_num)
            while m[0] in self.user_hit[bat_user[i]]:
                m = np.random.choice(self.all_item, neg_num)
            bat_neg.append(m)
        return bat_user.reshape(-1, 1), bat_item.reshape(-1, 1), np.array(bat_neg)

    # ------------------------------------------------------------------ the stream of periods
    def run(self, start_stage, method='full'):
        """model/baseline.py:505-556."""
        self.run_stage = 0
        stage_id = start_stage
        while True:
            print("#################################runing stage:{}########################".format(stage_id))
            if method == 'spmf':
                run_flag = self.run_one_stage(stage_id)
            elif method == 'full':
                run_flag = self.run_one_stage2(stage_id, read_data_type='not_only_new')
            else:
                run_flag = self.run_one_stage2(stage_id, read_data_type='only_new')
            if run_flag:
                stage_id += 1
                self.run_stage += 1
                continue
            test_num = np.array(self.test_num).reshape(-1, 1)
            recall, ndcg = np.array(self.recall), np.array(self.ndcg)
            print("average recall:", recall.mean(axis=0))
            print("average recall:", ndcg.mean(axis=0))        # (sic)
            print(test_num)
            print(recall)
            print(ndcg)
            print("hit new user:", self.hit_new_user)
            print("hit new item:", self.hit_new_item)
            N = test_num.shape[0]
            N3 = round(N * 1.0 / 3)
            rate3 = test_num[0:N3] / test_num[0:N3].sum()
            print("pre 3 (val) reslut,recall,ndcg:", (recall[0:N3] * rate3).sum(axis=0), (ndcg[0:N3] * rate3).sum(axis=0))
            rate_7 = test_num[N3:] / test_num[N3:].sum()
            print("last 7 (test) results,recall ,ndcg:", (recall[N3:] * rate_7).sum(axis=0), (ndcg[N3:] * rate_7).sum(axis=0))
            rate = test_num / test_num.sum()
            print("weight average recall@20:", (recall * rate).sum(axis=0))
            print("weight average ndcg@20:", (ndcg * rate).sum(axis=0))
            break


class StreamingData(object):
    """Period files of the baselines (model/baseline.py:558-588): information.npy, test_new_user.npy,
    test_new_item.npy, train/{p}.npy, test/{p}.npy."""

    def __init__(self, file_pathe):
        information = np.load(file_pathe + "information.npy")
        self.user_num = information[1]
        self.item_num = information[2]
        self.itr_num = information[0]
        self.path = file_pathe
        self.test_new_user = np.load(file_pathe + "test_new_user.npy").astype(np.int64)
        self.test_new_item = np.load(file_pathe + "test_new_item.npy").astype(np.int64)

    def get_next(self, stage_id, types="not_only_new"):
        try:
            if types == "not_only_new":
                train_data = np.concatenate([np.load(self.path + "train/" + str(i) + ".npy").astype(np.int64)
                                             for i in range(0, stage_id)], axis=0)
            else:
                train_data = np.load(self.path + "train/" + str(stage_id - 1) + ".npy").astype(np.int64)
        except Exception:
            print("read train data roung , may be there is no new data,finished")
            return None, None
        try:
            test_data = np.load(self.path + "test/" + str(stage_id) + ".npy").astype(np.int64)
        except Exception:
            print("read test data roung , may be there is no new data,finished")
            return None, None
        print("NOTICED: will train: {} , will test:{} ".format(stage_id - 1, stage_id))
        return train_data, test_data


def get_parse():
    """Flags and defaults of model/baseline.py:592-627."""
    parser = argparse.ArgumentParser(description='MF and TR parameters.')
    parser.add_argument('--lr', type=float, default=0.01, help='Learning rate.')
    parser.add_argument('--l2_u', type=float, default=1e-5, help='user l2. should be same to l2_i')
    parser.add_argument('--l2_i', type=float, default=1e-5, help='item l2.should be same to l2_u ')
    parser.add_argument('--epochs', type=int, default=20, help='Number of epochs to train of each stage.')
    parser.add_argument('--batch_size', type=int, default=256, help='batch size of train.')
    parser.add_argument('--laten_dim', type=int, default=64, help='dim of embedding.')
    parser.add_argument('--neg_num', type=int, default=1, help='neg num.')
    parser.add_argument('--pool_size', type=int, default=0, help='reservoir size (SPMF)')
    parser.add_argument('--laten', type=int, default=64, help='dim of embedding.')
    parser.add_argument('--cuda', type=int, default=1, help='which GPU be used?.default 1')
    parser.add_argument('--method', default='full', help='full, fine, spmf')
    parser.add_argument('--pool_init_type', type=int, default=0,
                        help='Reservious of SPMF init methods, 0: update , 1: init, yelp=0, news (adressa) =1 ')
    parser.add_argument('--data_path', default='/home/sml/dataset/', help='data path')
    parser.add_argument('--data_name', default='yelp', help='dataset name')
    parser.add_argument('--pre_model', default=None, help='pretrained MF state_dict')
    parser.add_argument('--start_idx', type=int, default=30, help='retraining from which period: yelp 30, news(adressa) 48')
    return parser


def main(argv=None):
    """`python -m sml_amd.baseline` = the reference's `python model/baseline.py` (model/baseline.py:629-680)."""
    print("start")
    args = get_parse().parse_args(argv)
    print("parameters:", args)
    data_path = args.data_path + args.data_name + "/"
    args.pool_init_type = 1 if args.data_name == 'news' else 0
    dataset = StreamingData(data_path)
    args.l2_i = args.l2_u
    print("*******************(l2_u,pool size):({},{})********".format(0, 0))
    print("*##**##*")
    print(args)
    torch.manual_seed(2000)
    torch.cuda.manual_seed(2001)
    np.random.seed(2002)
    model = SPMF(args, dataset, int(dataset.user_num), int(dataset.item_num), args.laten_dim)
    if args.pre_model:
        model.MFbase.load_state_dict(torch.load(args.pre_model, map_location=model.device))
    if args.method == 'spmf':
        model.base_train_not_train(args.start_idx - 1)
    model.run(args.start_idx, method=args.method)
    print("\n *##**##* \n")


if __name__ == "__main__":
    main()
