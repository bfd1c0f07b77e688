#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by IMPORTING the reference.

Runs only in the build container, where the reference checkout is mounted at
/root/reference (read-only).  Nothing from the reference is copied: this script
drives the reference's own classes on seeded inputs and records inputs and
outputs as .npz data.  The GPU box never sees /root/reference; tests there read
only the committed fixtures.

Harness-side shims (the reference is untouched):
  1. a stub torch.utils.tensorboard.SummaryWriter (tensorboard is not installed),
  2. Tensor.cuda / Module.cuda -> identity (there is no GPU here),
  3. torch.load(..., weights_only=False) for the whole-module checkpoint pickle,
  4. test_model's NDCG wrapped back into a tensor (numpy.float32 has no .cpu()).

Fixtures:  G1 transfer forward, G2 run_MF loss and
gradients, G14 MF2.forward (training and test branch), G3 MF-stage steps, G4 TR-stage steps, G5 updata, G6 evaluation, G13 evaluation at 999 negatives,
G7 end-to-end main_yelp.py log on a tiny 40-period dataset, G8 batch supply,
G9 parameter initialisation, G7-news the main_news.py (Adressa) path end to end, G12 a mid-size period
sequence (10,000 test rows per period: Recall@20 resolves 1e-4) with full-precision per-batch losses and
teacher-forcing state snapshots, G11 the ConvTransfer variant (--transfer_type conv), G10 the baselines' bare-MF fine-tune loop (model/baseline.py
SPMF.run_one_stage2: BCE + L2 + dense Adam on recorded batches).

usage: python tests/golden/make_golden.py [--ref /root/reference]
"""
import argparse
import contextlib
import io
import os
import runpy
import shutil
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))


# --------------------------------------------------------------------------- shims
def install_shims(ref):
    if "torch.utils.tensorboard" not in sys.modules:
        tb = types.ModuleType("torch.utils.tensorboard")

        class SummaryWriter(object):
            def __init__(self, *a, **k):
                pass

            def add_scalar(self, *a, **k):
                pass

            def add_scalars(self, *a, **k):
                pass

        tb.SummaryWriter = SummaryWriter
        sys.modules["torch.utils.tensorboard"] = tb
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    torch.cuda.manual_seed = lambda *a, **k: None
    _orig_load = torch.load

    def _load(f, *a, **k):
        k["weights_only"] = False
        return _orig_load(f, *a, **k)

    torch.load = _load
    # The build ships drop-in packages with the SAME names (model/, data/, evalution/ at the repo root, with
    # __init__.py -- a regular package beats the reference's __init__-less directories wherever both are on
    # sys.path).  So: only the reference is on the path while its modules are imported, every module's
    # origin is checked, and the repo root is appended afterwards, for sml_amd.synth alone.
    for name in [m for m in sys.modules if m.split(".")[0] in ("model", "data", "evalution")]:
        del sys.modules[name]
    sys.path[:] = [p for p in sys.path if os.path.realpath(p or os.getcwd()) != os.path.realpath(REPO)]
    sys.path.insert(0, ref)
    if not hasattr(np, "long"):
        np.long = np.int64              # model/baseline.py predates numpy 2
    import model.transfer as T
    import model.MF, model.conv_transfer, model.baseline            # noqa: F401,E401
    import data.dataset2, data.dataset, evalution.evaluation2       # noqa: F401,E401
    root = os.path.realpath(ref) + os.sep
    for name in ("model.transfer", "model.MF", "model.conv_transfer", "model.baseline", "data.dataset2", "data.dataset",
                 "evalution.evaluation2"):
        origin = os.path.realpath(sys.modules[name].__file__)
        assert origin.startswith(root), "%s was imported from %s, not from the reference" % (name, origin)
    sys.path.append(REPO)  # for sml_amd.synth (the build's own generator)

    _orig_tm = T.test_model

    def _tm(*a, **k):
        r, n = _orig_tm(*a, **k)
        return r, torch.as_tensor(np.float32(n))

    T.test_model = _tm
    return T


def sd_np(module, prefix=""):
    return {prefix + k: v.detach().cpu().numpy().copy() for k, v in module.state_dict().items()}


OUT_DIR = [HERE]


def save(name, **arrs):
    path = os.path.join(OUT_DIR[0], name)
    np.savez_compressed(path, **arrs)
    print("wrote", name, "%.1f KB" % (os.path.getsize(path) / 1024.0))


@contextlib.contextmanager
def record_backward(sink):
    """Every scalar the reference calls .backward() on (the per-batch training loss, l2 term included) is
    appended to `sink` at full precision.  A harness-side wrapper: the reference is untouched."""
    orig = torch.Tensor.backward

    def rec(self, *a, **k):
        if self.dim() == 0:
            sink.append(float(self.detach()))
        return orig(self, *a, **k)

    torch.Tensor.backward = rec
    try:
        yield
    finally:
        torch.Tensor.backward = orig


class Args(object):
    """Minimal args namespace with the reference's defaults (main_yelp.py:10-120)."""

    def __init__(self, **kw):
        d = dict(data_name="yelp", data_path="", multi_num=2, MF_lr=0.01, MF_epochs=1, l2=1e-6,
                 MF_batch_size=32, laten=32, pre_model="", MF_sample="all", Load_W_hat=False,
                 clip_grad=False, need_adaptive=False, maxnorm_grad=3.0, TR_lr=0.001, TR_l2=1e-4,
                 TR_epochs=1, TR_batch_size=16, TR_sample_type="alone", TR_with_MF_bias=False,
                 TR_stop_=False, transfer_type="conv_com", seed=2000, numworkers=0, cuda=0, topK=20,
                 pass_num=1, norm=False, Lambda_lr=0.01, min_l2=1e-4, set_t_as_tt=False, tqdm=False,
                 need_writer=False, test_in_TR_Train=False)
        d.update(kw)
        self.__dict__.update(d)


# --------------------------------------------------------------------------- G1 / G2 / G9
def gen_g1_g2_g9(T):
    from model.conv_transfer import ConvTransfer_com
    from model.MF import MFbasemode

    for d in (32, 64):
        torch.manual_seed(100 + d)
        with contextlib.redirect_stdout(io.StringIO()):
            net = ConvTransfer_com(d, d)
        g = torch.Generator().manual_seed(7 + d)
        x_t = torch.randn(64, d, generator=g)
        x_hat = torch.randn(64, d, generator=g)
        with torch.no_grad():
            yu = net(x_t, x_hat, "user")
            yi = net(x_t, x_hat, "item")
        out = sd_np(net, "theta.")
        out.update(x_t=x_t.numpy(), x_hat=x_hat.numpy(), y_user=yu.numpy(), y_item=yi.numpy())
        save("g1_transfer_forward_d%d.npz" % d, **out)

        # G2: run_MF loss + gradients (w.r.t. x_hat inputs and theta)
        B = 48
        ten = [torch.randn(B, d, generator=g) * 0.7 for _ in range(6)]
        res = {}
        for tag, kw in (("bce", dict()), ("bpr", dict(BCE=False)), ("bprnorm", dict(BCE=False, norm=True))):
            ins = [t.clone() for t in ten]
            for k in (1, 3, 5):
                ins[k].requires_grad_(True)
            net.zero_grad()
            loss = net.run_MF(*ins, **kw)
            loss.backward()
            res["loss_" + tag] = loss.detach().numpy()
            res["gu_" + tag] = ins[1].grad.numpy().copy()
            res["gi_" + tag] = ins[3].grad.numpy().copy()
            res["gn_" + tag] = ins[5].grad.numpy().copy()
            for k, p in net.named_parameters():
                res["gtheta_%s.%s" % (tag, k)] = p.grad.detach().numpy().copy()
        out = sd_np(net, "theta.")
        out.update(res)
        for k, nm in enumerate(("ul", "uh", "il", "ih", "nl", "nh")):
            out[nm] = ten[k].numpy()
        save("g2_run_mf_d%d.npz" % d, **out)

    # G9: creation order / RNG draw order of the parameter initialisers
    torch.manual_seed(2000)
    mf = MFbasemode(37, 23, 32)
    with contextlib.redirect_stdout(io.StringIO()):
        net = ConvTransfer_com(32, 32)
    out = sd_np(mf, "mf.")
    out.update(sd_np(net, "theta."))
    out["mf_keys"] = np.array(list(mf.state_dict().keys()))
    out["theta_keys"] = np.array(list(net.state_dict().keys()))
    save("g9_init.npz", **out)


# --------------------------------------------------------------------------- G3 / G4 / G5
class _FakeData(object):
    user_number = 0
    item_number = 0


def build_meta(T, tmp, U, I, d, args, seed):
    """Construct the reference meta_train on a seeded random checkpoint."""
    from model.MF import MFbasemode

    torch.manual_seed(seed)
    mf = MFbasemode(U, I, d)
    ck = os.path.join(tmp, "init_%d.pkl" % seed)
    torch.save(mf, ck)
    args.pre_model = ck
    args.laten = d
    with contextlib.redirect_stdout(io.StringIO()):
        meta = T.meta_train(args, _FakeData(), U, I, d)
    return meta


def gen_g3_g4_g5(T, tmp, ttype="conv_com", suffix=""):
    from data.dataset2 import trainDataset_withPreSample
    from data.dataset import offlineDataset_withsample

    U, I, d = 50, 40, 32
    args = Args(MF_batch_size=32, TR_batch_size=16, MF_epochs=2, TR_epochs=2, transfer_type=ttype)
    meta = build_meta(T, tmp, U, I, d, args, seed=31)
    rng = np.random.RandomState(5)
    # the transfer net only sees non-trivial x_t after a 'last' save; make W_{t-1} != W_hat
    meta.save_MF_weight("last")
    with torch.no_grad():
        meta.MFbase.user_laten.weight.add_(0.05 * torch.randn(U, d))
        meta.MFbase.item_laten.weight.add_(0.05 * torch.randn(I, d))

    out = {}
    out.update(sd_np(meta.transfer, "theta0."))
    out["W_user0"] = meta.MFbase.user_laten.weight.detach().numpy().copy()
    out["W_item0"] = meta.MFbase.item_laten.weight.detach().numpy().copy()
    out["Wlast_user"] = meta.last_user_weight.numpy().copy()
    out["Wlast_item"] = meta.last_item_weight.numpy().copy()

    # period data: set_t in the 'test' layout [n, 2+neg] (MF_sample == 'all')
    n_t, neg = 200, 9
    users = rng.randint(0, U, size=n_t)
    users[:40] = 3  # heavy duplicates
    items = rng.randint(0, I, size=n_t)
    negs = rng.randint(0, I - 1, size=(n_t, neg))
    negs += negs >= items[:, None]
    set_t = np.concatenate([users[:, None], items[:, None], negs], axis=1).astype(np.int64)
    out["set_t"] = set_t

    # record batches in DataLoader order through a recording dataset wrapper
    log = []

    class RecMF(trainDataset_withPreSample):
        def __getitem__(self, idx):
            r = trainDataset_withPreSample.__getitem__(self, idx)
            log.append((int(r[0]), int(r[1]), int(r[2])))
            return r

    meta.MF_TrainDataset = RecMF
    losses = []
    orig_run = meta.transfer.run_MF

    def rec_run(*a, **k):
        l = orig_run(*a, **k)
        losses.append(float(l.detach()))
        return l

    meta.transfer.run_MF = rec_run
    torch.manual_seed(77)
    np.random.seed(78)
    buf = io.StringIO()
    full = []          # loss_batch as the reference backpropagates it: run_MF + l2 * l2loss (model/transfer.py:488, 502)
    with contextlib.redirect_stdout(buf), record_backward(full):
        meta.MF_train_onestage(args, set_t, 0, val=None)
    out["mf_batch_loss"] = np.array(full, dtype=np.float64)
    out["mf_log"] = np.array(buf.getvalue())
    out["mf_triples"] = np.array(log, dtype=np.int64)  # [epochs*n_t, 3] in consumption order
    out["mf_runmf_loss"] = np.array(losses, dtype=np.float64)
    out["W_user1"] = meta.MFbase.user_laten.weight.detach().numpy().copy()
    out["W_item1"] = meta.MFbase.item_laten.weight.detach().numpy().copy()
    st = meta.MF_optimizer.state
    pu, pi = meta.MFbase.user_laten.weight, meta.MFbase.item_laten.weight
    out["adam_m_user"] = st[pu]["exp_avg"].numpy().copy()
    out["adam_v_user"] = st[pu]["exp_avg_sq"].numpy().copy()
    out["adam_m_item"] = st[pi]["exp_avg"].numpy().copy()
    out["adam_v_item"] = st[pi]["exp_avg_sq"].numpy().copy()
    out["adam_step"] = np.array(float(st[pu]["step"]))
    out["hp_mf"] = np.array([args.MF_lr, args.l2, args.MF_batch_size, args.MF_epochs], dtype=np.float64)
    save("g3_mf_stage%s.npz" % suffix, **out)

    # ---- G5 updata (after save 'hat'), then G4 TR-stage steps on the same state
    meta.MFbase.eval()
    meta.save_MF_weight("hat")
    g5 = {}
    g5.update(sd_np(meta.transfer, "theta."))
    g5["Wlast_user"] = meta.last_user_weight.numpy().copy()
    g5["Wlast_item"] = meta.last_item_weight.numpy().copy()
    g5["What_user"] = meta.user_weight_hat.numpy().copy()
    g5["What_item"] = meta.item_weight_hat.numpy().copy()
    meta.updata()
    g5["Wnew_user"] = meta.MFbase.user_laten.weight.detach().numpy().copy()
    g5["Wnew_item"] = meta.MFbase.item_laten.weight.detach().numpy().copy()
    save("g5_updata%s.npz" % suffix, **g5)

    g4 = {}
    g4.update(sd_np(meta.transfer, "theta0."))
    g4["Wlast_user"] = g5["Wlast_user"]
    g4["Wlast_item"] = g5["Wlast_item"]
    g4["What_user"] = g5["What_user"]
    g4["What_item"] = g5["What_item"]
    n_tt = 90
    set_tt = np.stack([rng.randint(0, U, size=n_tt), rng.randint(0, I // 2, size=n_tt)], axis=1).astype(np.int64)
    g4["set_tt"] = set_tt
    log2 = []

    class RecTR(offlineDataset_withsample):
        def __getitem__(self, idx):
            r = offlineDataset_withsample.__getitem__(self, idx)
            log2.append((int(r[0]), int(r[1]), int(r[2])))
            return r

    T.SampleDaset = RecTR
    losses.clear()
    torch.manual_seed(177)
    np.random.seed(178)
    buf = io.StringIO()
    full = []
    with contextlib.redirect_stdout(buf), record_backward(full):
        meta.transfer_train_onestage(args, set_tt, 0, val=None)
    g4["tr_batch_loss"] = np.array(full, dtype=np.float64)
    T.SampleDaset = offlineDataset_withsample
    g4["tr_log"] = np.array(buf.getvalue())
    g4["tr_triples"] = np.array(log2, dtype=np.int64)
    g4["tr_runmf_loss"] = np.array(losses, dtype=np.float64)
    g4.update(sd_np(meta.transfer, "theta1."))
    ost = meta.transfer_optimizer.state
    for k, p in meta.transfer.named_parameters():
        g4["adam_m." + k] = ost[p]["exp_avg"].numpy().copy()
        g4["adam_v." + k] = ost[p]["exp_avg_sq"].numpy().copy()
    g4["adam_step"] = np.array(float(ost[next(meta.transfer.parameters())]["step"]))
    g4["hp_tr"] = np.array([args.TR_lr, args.TR_l2, args.TR_batch_size, args.TR_epochs], dtype=np.float64)
    save("g4_tr_stage%s.npz" % suffix, **g4)


# --------------------------------------------------------------------------- G11 ConvTransfer (--transfer_type conv)
def gen_g11(T, tmp):
    """The reference's other convolutional transfer (model/conv_transfer.py:52-85: kernel (2,1), no x_com row,
    user output divided by its detached norm, BPR sum loss): forward, run_MF loss and gradients, then the
    same recorded MF-stage / updata / TR-stage runs as G3-G5 through meta_train(transfer_type='conv')."""
    from model.conv_transfer import ConvTransfer
    d = 32
    torch.manual_seed(300 + d)
    with contextlib.redirect_stdout(io.StringIO()):
        net = ConvTransfer(d, d)
    g = torch.Generator().manual_seed(9 + d)
    x_t = torch.randn(64, d, generator=g)
    x_hat = torch.randn(64, d, generator=g)
    with torch.no_grad():
        yu = net(x_t, x_hat, "user")
        yi = net(x_t, x_hat, "item")
    out = sd_np(net, "theta.")
    out.update(x_t=x_t.numpy(), x_hat=x_hat.numpy(), y_user=yu.numpy(), y_item=yi.numpy())
    B = 48
    ten = [torch.randn(B, d, generator=g) * 0.7 for _ in range(6)]
    ins = [t.clone() for t in ten]
    for k in (1, 3, 5):
        ins[k].requires_grad_(True)
    net.zero_grad()
    loss = net.run_MF(*ins, norm=False)
    loss.backward()
    out["loss_bpr"] = loss.detach().numpy()
    out["gu_bpr"], out["gi_bpr"], out["gn_bpr"] = ins[1].grad.numpy().copy(), ins[3].grad.numpy().copy(), ins[5].grad.numpy().copy()
    for k, p in net.named_parameters():
        out["gtheta_bpr.%s" % k] = p.grad.detach().numpy().copy()
    for k, nm in enumerate(("ul", "uh", "il", "ih", "nl", "nh")):
        out[nm] = ten[k].numpy()
    # round 5: run_MF(norm=True) -- model/conv_transfer.py:79-81: the score divided by the norm of the (already unit-norm)
    # user output, which is NOT detached there
    ins = [t.clone() for t in ten]
    for k in (1, 3, 5):
        ins[k].requires_grad_(True)
    net.zero_grad()
    loss = net.run_MF(*ins, norm=True)
    loss.backward()
    out["loss_bprn"] = loss.detach().numpy()
    out["gu_bprn"], out["gi_bprn"], out["gn_bprn"] = ins[1].grad.numpy().copy(), ins[3].grad.numpy().copy(), ins[5].grad.numpy().copy()
    for k, p in net.named_parameters():
        out["gtheta_bprn.%s" % k] = p.grad.detach().numpy().copy()
    save("g11_convtransfer_d32.npz", **out)
    gen_g3_g4_g5(T, tmp, ttype="conv", suffix="_conv")


# --------------------------------------------------------------------------- G6 eval
def gen_g6(T):
    from model.MF import MFbasemode
    from data.dataset2 import testDataset

    torch.manual_seed(11)
    U, I, d, n, neg = 60, 150, 32, 97, 99
    mf = MFbasemode(U, I, d)
    rng = np.random.RandomState(12)
    users = rng.randint(0, U, size=n)
    pos = rng.randint(0, I, size=n)
    rows = np.zeros((n, 2 + neg), dtype=np.int64)
    for r in range(n):
        cand = np.setdiff1d(np.arange(I), [pos[r]])
        rows[r, 0] = users[r]
        rows[r, 1] = pos[r]
        rows[r, 2:] = rng.choice(cand, size=neg, replace=False)  # distinct items: tie-free
    out = sd_np(mf, "mf.")
    out["rows"] = rows
    for K in (5, 10, 20):
        with torch.no_grad():
            hit, ndcg, idx = mf.test(torch.from_numpy(rows), topK=K)
        out["hit_%d" % K] = np.array(float(hit))
        out["ndcg_%d" % K] = np.array(float(ndcg))
        out["hitidx_%d" % K] = idx.numpy().copy()
        loader = torch.utils.data.DataLoader(testDataset(rows), batch_size=32, num_workers=0)
        import evalution.evaluation2 as E
        r, nd = E.test_model(mf, loader, topK=K)
        out["recall_%d" % K] = np.array(float(r))
        out["ndcgavg_%d" % K] = np.array(float(nd))
    save("g6_eval.npz", **out)


# --------------------------------------------------------------------------- G14 MF2 (model/MF.py:118-156)
def gen_g14(T):
    """MF2.forward: the training branch (BPR with the item-bias difference in the score; "l2" as a sum of row NORMS, the
    negatives' as ONE Frobenius norm -- the reference's expression, model/MF.py:139-146) with its gradients, and the test
    branch (scores with both biases)."""
    from model.MF import MF2

    torch.manual_seed(14)
    U, I, d, B = 40, 30, 32, 24
    mf = MF2(U, I, d)
    rng = np.random.RandomState(15)
    user, item, neg = rng.randint(0, U, B), rng.randint(0, I, B), rng.randint(0, I, B)
    user[:4] = 3                        # duplicates inside the batch
    item[5] = neg[5]
    out = sd_np(mf, "mf.")
    out["user"], out["item"], out["neg"] = user, item, neg
    u, i, j = (torch.from_numpy(x) for x in (user, item, neg))
    bpr, l2 = mf(u, i, j)
    (bpr + 0.01 * l2).backward()
    out["bpr_loss"] = np.array(float(bpr.detach()), dtype=np.float64)
    out["l2loss"] = np.array(float(l2.detach()), dtype=np.float64)
    for name, prm in mf.named_parameters():
        out["grad." + name] = prm.grad.numpy().copy()
    with torch.no_grad():
        ue, ie, res = mf(u, i)
    out["test_uemb"], out["test_iemb"], out["test_result"] = ue.numpy().copy(), ie.numpy().copy(), res.numpy().copy()
    save("g14_mf2.npz", **out)


# --------------------------------------------------------------------------- G13 eval at the shipped format's width
def gen_g13(T):
    """MFbasemode.test / test_model on rows of the shipped test format: 1 positive + 999 negatives (data/dataset2.py:356
    neg_num=999; model/MF.py:50).  Candidates are distinct and every negative's score is at least 1e-3 away from the
    positive's (checked in float64, offending candidates redrawn), so the ranks are exact whatever the order an
    implementation sums the 32 products in.  Recorded from the reference: hits / ndcg / hit indices at K = 5, 10, 20, the
    full rank of every row (topK over all 1000 candidates) and test_model's averages."""
    from model.MF import MFbasemode
    from data.dataset2 import testDataset
    import evalution.evaluation2 as E

    torch.manual_seed(13)
    U, I, d, n, neg = 100, 1300, 32, 96, 999
    mf = MFbasemode(U, I, d)
    wu = mf.user_laten.weight.detach().double().numpy()
    wi = mf.item_laten.weight.detach().double().numpy()
    rng = np.random.RandomState(14)
    rows = np.zeros((n, 2 + neg), dtype=np.int64)
    for r in range(n):
        u, pos = rng.randint(0, U), rng.randint(0, I)
        s = wi @ wu[u]
        ok = np.abs(s - s[pos]) >= 1e-3
        ok[pos] = False
        cand = np.nonzero(ok)[0]
        assert cand.shape[0] >= neg
        rows[r, 0], rows[r, 1] = u, pos
        rows[r, 2:] = rng.choice(cand, size=neg, replace=False)
    out = sd_np(mf, "mf.")
    out["rows"] = rows.astype(np.int32)
    with torch.no_grad():
        _, _, _ = mf.test(torch.from_numpy(rows), topK=20)
        sc = (mf.user_laten(torch.from_numpy(rows[:, 0])).unsqueeze(1) * mf.item_laten(torch.from_numpy(rows[:, 1:]))).sum(-1)
        _, order = torch.topk(sc, 1 + neg)                  # the reference's ranking primitive over ALL candidates
        full = (order < 1).nonzero()
        assert full.shape[0] == n and torch.equal(full[:, 0], torch.arange(n))
        out["rank_full"] = full[:, 1].numpy().astype(np.int32)
    for K in (5, 10, 20):
        with torch.no_grad():
            hit, ndcg, idx = mf.test(torch.from_numpy(rows), topK=K)
        out["hit_%d" % K] = np.array(float(hit))
        out["ndcg_%d" % K] = np.array(float(ndcg))
        out["hitidx_%d" % K] = idx.numpy().copy()
        loader = torch.utils.data.DataLoader(testDataset(rows), batch_size=32, num_workers=0)
        r, nd = E.test_model(mf, loader, topK=K)
        out["recall_%d" % K] = np.array(float(r))
        out["ndcgavg_%d" % K] = np.array(float(nd))
    save("g13_eval_999.npz", **out)


# --------------------------------------------------------------------------- G8 batch supply
def gen_g8(T):
    from data.dataset2 import trainDataset_withPreSample
    from data.dataset import offlineDataset_withsample

    rng = np.random.RandomState(21)
    n, neg, U, I = 70, 6, 20, 30
    items = rng.randint(0, I, size=n)
    negs = rng.randint(0, I - 1, size=(n, neg))
    negs += negs >= items[:, None]
    table = np.concatenate([rng.randint(0, U, size=(n, 1)), items[:, None], negs], axis=1).astype(np.int64)
    out = {"table": table}
    torch.manual_seed(501)
    np.random.seed(502)
    ds = trainDataset_withPreSample(table)
    dl = torch.utils.data.DataLoader(ds, batch_size=16, shuffle=True, num_workers=0)
    seq = []
    for ep in range(8):  # > neg epochs: exercises the column switch and the reshuffle of neg_flag
        for (u, i, j) in dl:
            seq.append(torch.stack([u, i, j], 1).numpy())
    out["presample_seq"] = np.concatenate(seq, 0)
    pairs = np.stack([rng.randint(0, U, size=n), rng.randint(0, 12, size=n)], axis=1).astype(np.int64)
    out["pairs"] = pairs
    torch.manual_seed(503)
    np.random.seed(504)
    with contextlib.redirect_stdout(io.StringIO()):
        ds2 = offlineDataset_withsample(pairs)
    dl2 = torch.utils.data.DataLoader(ds2, batch_size=16, shuffle=True, num_workers=0)
    seq = []
    for ep in range(3):
        for (u, i, j) in dl2:
            seq.append(torch.stack([torch.as_tensor(u), torch.as_tensor(i), torch.as_tensor(j)], 1).numpy())
    out["withsample_seq"] = np.concatenate(seq, 0)
    save("g8_batches.npz", **out)


# --------------------------------------------------------------------------- G7 end to end
def gen_g7(T, ref, tmp, extra=(), suffix="", which="yelp"):
    """which='yelp': main_yelp.py (40 periods).  which='news': main_news.py (63 periods, train from 21, test from
    48, multi_num 7, 2 + 2 epochs: main_news.py:22,34,68,221-227) -- the Adressa path, which also makes the
    discarded-MFbasemode RNG draw of model/transfer.py:314-317 before the transfer net is initialised."""
    from sml_amd import synth
    from model.MF import MFbasemode

    root = os.path.join(tmp, "data_" + which) + "/"
    U, I, d = 300, 120, 32
    n_periods = 40 if which == "yelp" else 63
    synth.write_dataset(root, which, n_periods=n_periods, n_inter=160, n_user=U, n_item=I, neg=49,
                        a_user=0.8, a_item=0.8, seed=2000)
    torch.manual_seed(4242)
    mf = MFbasemode(U, I, d)
    with torch.no_grad():  # a small-norm init behaves like a pretrained MF (scores not saturated)
        mf.user_laten.weight.mul_(0.3)
        mf.item_laten.weight.mul_(0.3)
    ck = os.path.join(tmp, "BCE_init.pkl")
    torch.save(mf, ck)
    script = "main_%s.py" % which
    argv = [script, "--data_path", root, "--pre_model", ck, "--laten", str(d)] + \
           (["--multi_num", "2"] if which == "yelp" else []) + \
           ["--numworkers", "0", "--MF_batch_size", "64", "--TR_batch_size", "32"] + list(extra)
    buf = io.StringIO()
    old = sys.argv
    sys.argv = argv
    try:
        with contextlib.redirect_stdout(buf):
            runpy.run_path(os.path.join(ref, script), run_name="__main__")
    finally:
        sys.argv = old
    log = buf.getvalue()
    out = sd_np(mf, "mf.")
    out["log"] = np.array(log)
    out["argv"] = np.array(argv[5:])
    out["dataset"] = np.array([n_periods, 160, U, I, 49, 2000], dtype=np.int64)
    out["dataset_zipf"] = np.array([0.8, 0.8])
    save("g7_end_to_end%s.npz" % suffix, **out)
    if not suffix and which == "yelp":
        # a reference-pickled whole-module checkpoint: the on-disk contract for --pre_model
        shutil.copy(ck, os.path.join(OUT_DIR[0], "ref_BCE_init_tiny.pkl"))
    print("log lines:", len(log.splitlines()))


# --------------------------------------------------------------------------- G12 mid-size period sequence
G12 = dict(U=1500, I=2500, d=32, n_inter=10000, neg=99, n_periods=6, train_from=1, test_from=3, multi_num=3,
           seed=2000, data_seed=4100, a_user=0.9, a_item=0.9, ck_seed=4343, snap_stages=(2, 3))


# G15 (round 5): the SAME kind of sequence at the reference's DEFAULT depth -- multi_num 10 (main_yelp.py's default; G12 runs 3) --
# over 7 stages, 4 of them test stages, with the complete numerical state recorded at the START OF EVERY stage: the transfer
# stage's trajectory is the chaotic part of the loop, and ten phases per stage is where a re-implementation drifts furthest.
G15 = dict(U=1500, I=2500, d=32, n_inter=10000, neg=99, n_periods=8, train_from=1, test_from=4, multi_num=10,
           seed=2000, data_seed=4100, a_user=0.9, a_item=0.9, ck_seed=4343, snap_stages=(1, 2, 3, 4, 5, 6))      # (stage 0 starts from the seeds)


def gen_g15(T, ref, tmp):
    gen_g12(T, ref, tmp, c=G15, name="g15_fulldepth")


def gen_g12(T, ref, tmp, c=None, name="g12_midsize"):
    """The reference's period loop at a size where Recall@20 resolves 1e-4: 10,000 validation / test rows per
    period.  main_yelp.py's __main__ body (seeding order main_yelp.py:137-139, transfer_data, meta_train,
    run: :159-168) is driven here with SIX periods instead of forty (train from period 1, test from period 3
    -> one pure-training stage and three test stages), main_yelp.py's defaults otherwise (MF batch 1024, TR
    batch 256, lr / l2) except multi_num 3.  Recorded: the whole printed log; every per-batch training loss at
    full precision (the scalar the reference backpropagates); and -- for the teacher-forced test -- the complete
    numerical state (tables, theta, both Adam states) at the START of stages 2 and 3, so a re-implementation can
    be set onto the reference's trajectory there and must then reproduce that stage's printed numbers."""
    from sml_amd import synth
    from model.MF import MFbasemode
    import data.dataset2 as dataset2

    c = G12 if c is None else c
    root = os.path.join(tmp, "data_" + name) + "/"
    synth.write_dataset(root, "yelp", n_periods=c["n_periods"], n_inter=c["n_inter"], n_user=c["U"], n_item=c["I"],
                        neg=c["neg"], a_user=c["a_user"], a_item=c["a_item"], seed=c["data_seed"])
    torch.manual_seed(c["ck_seed"])
    mf = MFbasemode(c["U"], c["I"], c["d"])
    with torch.no_grad():
        mf.user_laten.weight.mul_(0.3)
        mf.item_laten.weight.mul_(0.3)
    ck = os.path.join(tmp, name + "_init.pkl")
    torch.save(mf, ck)
    args = Args(data_path=root, pre_model=ck, laten=c["d"], multi_num=c["multi_num"], MF_batch_size=1024,
                TR_batch_size=256, numworkers=0, seed=c["seed"])
    # main_yelp.py:137-139
    torch.manual_seed(args.seed)
    torch.cuda.manual_seed(args.seed + 1)
    np.random.seed(args.seed + 2)
    file_list = [str(i) for i in range(c["n_periods"])]
    test_list = [str(j) for j in range(c["test_from"], c["n_periods"])]
    buf = io.StringIO()
    losses, tags, snaps = [], [], {}
    with contextlib.redirect_stdout(buf):
        sets = dataset2.transfer_data(args, path=root, datasetname="yelp", file_path_list=file_list, test_list=test_list,
                                      validation_list=None, online_train_time=c["train_from"], online_test_time=c["test_from"])
        meta = T.meta_train(args, sets, sets.user_number, sets.item_number, args.laten)
        theta0 = sd_np(meta.transfer, "theta0.")
        stage_now = [0]
        for meth, tag in (("MF_train_onestage", 0), ("transfer_train_onestage", 1)):
            def wrap(fn, tag=tag):
                def inner(*a, **k):
                    n0 = len(losses)
                    r = fn(*a, **k)
                    tags.extend([(stage_now[0], tag)] * (len(losses) - n0))
                    return r
                return inner
            setattr(meta, meth, wrap(getattr(meta, meth)))
        real_stage = meta.train_one_stage3

        def stage(a, stage_id):
            stage_now[0] = stage_id
            if stage_id in c["snap_stages"]:
                sn = {}
                sn["W_user"] = meta.MFbase.user_laten.weight.detach().numpy().copy()
                sn["W_item"] = meta.MFbase.item_laten.weight.detach().numpy().copy()
                sn.update(sd_np(meta.transfer, "theta."))
                st = meta.MF_optimizer.state
                for nm, p in (("user", meta.MFbase.user_laten.weight), ("item", meta.MFbase.item_laten.weight)):
                    sn["mf_m_" + nm] = st[p]["exp_avg"].numpy().copy()
                    sn["mf_v_" + nm] = st[p]["exp_avg_sq"].numpy().copy()
                    sn["mf_step"] = np.array(int(st[p]["step"]))
                ost = meta.transfer_optimizer.state
                for k, p in meta.transfer.named_parameters():
                    sn["tr_m." + k] = ost[p]["exp_avg"].numpy().copy()
                    sn["tr_v." + k] = ost[p]["exp_avg_sq"].numpy().copy()
                    sn["tr_step"] = np.array(int(ost[p]["step"]))
                snaps[stage_id] = sn
            return real_stage(a, stage_id)

        meta.train_one_stage3 = stage
        with record_backward(losses):
            meta.run(args)
    log = buf.getvalue()
    out = {"log": np.array(log), "batch_loss": np.array(losses, dtype=np.float64),
           "batch_tag": np.array(tags, dtype=np.int64),           # [n, 2]: (stage, 0 = MF batch / 1 = TR batch)
           "config": np.array([c[k] for k in ("U", "I", "d", "n_inter", "neg", "n_periods", "train_from", "test_from",
                                              "multi_num", "seed", "data_seed", "ck_seed")], dtype=np.int64),
           "zipf": np.array([c["a_user"], c["a_item"]]),
           "final_sum": np.array([float(meta.MFbase.user_laten.weight.double().sum()), float(meta.MFbase.item_laten.weight.double().sum()),
                                  float(meta.MFbase.user_laten.weight.double().abs().sum()), float(meta.MFbase.item_laten.weight.double().abs().sum())]),
           "recall": np.array(meta.recall, dtype=np.float64), "ndcg": np.array([float(x) for x in meta.ndcg], dtype=np.float64),
           "test_num": np.array(meta.test_num, dtype=np.int64)}
    out.update(theta0)
    save(name + ".npz", **out)
    for sid, sn in snaps.items():
        save(name + "_state_s%d.npz" % sid, **sn)
    print(name, "log lines:", len(log.splitlines()), "batches:", len(losses), "recall:", meta.recall)


# --------------------------------------------------------------------------- G10 baseline bare-MF loop
def gen_g10(ref):
    """Drive the reference's SPMF.run_one_stage2 (model/baseline.py:306-386: the fine-tune / full-retrain
    baseline = bare MF with BCE + L2 and torch.optim.Adam) on a tiny stream.  Harness shims only: np.long
    (removed from numpy 2), and the DataLoader the method builds is forced to num_workers=0 and its batches
    are recorded -- those recorded (user, item, neg) batches are the fixture's inputs."""
    import model.baseline as B
    assert os.path.realpath(B.__file__).startswith(os.path.realpath(ref))

    U, I, d, n_train, n_test, neg = 80, 60, 32, 700, 90, 49
    rng = np.random.RandomState(31)
    train = np.stack([rng.randint(0, U, n_train), rng.randint(0, I, n_train)], 1).astype(np.int64)
    test = np.zeros((n_test, 2 + neg), dtype=np.int64)
    for r in range(n_test):
        test[r, 0] = rng.randint(0, U)
        test[r, 1] = rng.randint(0, I)
        test[r, 2:] = rng.choice(np.setdiff1d(np.arange(I), [test[r, 1]]), size=neg, replace=False)

    class Stream(object):                      # the surface of baseline.StreamingData that SPMF touches
        test_new_user = np.zeros(0, dtype=np.int64)
        test_new_item = np.zeros(0, dtype=np.int64)

        def get_next(self, stage_id, types="not_only_new"):
            return train, test

    args = types.SimpleNamespace(lr=0.01, pool_size=0, neg_num=1, batch_size=128, l2_u=1e-5, l2_i=1e-5,
                                 epochs=3, pool_init_type=0)
    recorded = []
    real_loader = torch.utils.data.DataLoader

    class RecLoader(object):
        def __init__(self, ds, batch_size=1, shuffle=False, num_workers=0, **k):
            self.inner = real_loader(ds, batch_size=batch_size, shuffle=shuffle, num_workers=0)

        def __iter__(self):
            for (u, i, j) in self.inner:
                recorded.append(np.stack([np.asarray(u), np.asarray(i), np.asarray(j)], 1).astype(np.int64))
                yield (u, i, j)

    torch.manual_seed(41)
    np.random.seed(42)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        sp = B.SPMF(args, Stream(), U, I, d)
        with torch.no_grad():                  # pretrained-like scale (scores not saturated)
            sp.MFbase.user_laten.weight.mul_(0.3)
            sp.MFbase.item_laten.weight.mul_(0.3)
        init = {k: v.detach().numpy().copy() for k, v in sp.MFbase.state_dict().items()}
        torch.utils.data.DataLoader = RecLoader
        try:
            ok = sp.run_one_stage2(1, read_data_type="only_new")
        finally:
            torch.utils.data.DataLoader = real_loader
    assert ok
    log = buf.getvalue()
    losses = [float(l.split("loss:")[1]) for l in log.splitlines() if l.startswith("epoch:")]
    nb = -(-n_train // args.batch_size)
    assert len(recorded) == args.epochs * nb and len(losses) == args.epochs, (len(recorded), losses)
    out = {"init." + k: v for k, v in init.items()}
    for e in range(args.epochs):
        out["triples_%d" % e] = np.concatenate(recorded[e * nb:(e + 1) * nb], 0)
    out["epoch_loss"] = np.array(losses, dtype=np.float64)          # as printed: 4 decimals
    st = sp.optimizer.state
    Wu, Wi = sp.MFbase.user_laten.weight, sp.MFbase.item_laten.weight
    out["final.user"], out["final.item"] = Wu.detach().numpy().copy(), Wi.detach().numpy().copy()
    out["adam.m_user"], out["adam.v_user"] = st[Wu]["exp_avg"].numpy().copy(), st[Wu]["exp_avg_sq"].numpy().copy()
    out["adam.m_item"], out["adam.v_item"] = st[Wi]["exp_avg"].numpy().copy(), st[Wi]["exp_avg_sq"].numpy().copy()
    out["adam.step"] = np.array(int(st[Wu]["step"]))
    out["test_rows"] = test
    out["recall_5_10_20"] = np.asarray(sp.recall[-1], dtype=np.float64)
    out["ndcg_5_10_20"] = np.asarray(sp.ndcg[-1], dtype=np.float64)
    out["hyper"] = np.array([args.lr, args.l2_u, args.l2_i, args.batch_size, args.epochs], dtype=np.float64)
    save("g10_baseline_adam.npz", **out)
    print("G10 losses:", losses, "recall:", out["recall_5_10_20"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--only", default="")
    ap.add_argument("--out", default=HERE, help="where to write the fixtures (default: next to this script)")
    a = ap.parse_args()
    OUT_DIR[0] = a.out
    os.makedirs(a.out, exist_ok=True)
    T = install_shims(a.ref)
    tmp = tempfile.mkdtemp(prefix="sml_golden_")
    try:
        only = set(a.only.split(",")) if a.only else None
        if not only or "g1" in only:
            gen_g1_g2_g9(T)
        if not only or "g3" in only:
            gen_g3_g4_g5(T, tmp)
        if not only or "g6" in only:
            gen_g6(T)
        if not only or "g13" in only:
            gen_g13(T)
        if not only or "g14" in only:
            gen_g14(T)
        if not only or "g8" in only:
            gen_g8(T)
        if not only or "g7" in only:
            gen_g7(T, a.ref, tmp)
        if not only or "g7news" in only:
            gen_g7(T, a.ref, tmp, suffix="_news", which="news")
        if not only or "g12" in only:
            gen_g12(T, a.ref, tmp)
        if not only or "g15" in only:
            gen_g15(T, a.ref, tmp)
        if not only or "g10" in only:
            gen_g10(a.ref)
        if not only or "g11" in only:
            gen_g11(T, tmp)
            gen_g7(T, a.ref, tmp, extra=("--transfer_type", "conv"), suffix="_conv")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
