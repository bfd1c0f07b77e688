"""The CPU oracle against the golden vectors recorded from the reference (G1-G6).
These pin oracle/sml_oracle.py; the -m gpu tests then compare the HIP path to it."""
import numpy as np
import pytest
import torch

from conftest import golden, make_mf, make_transfer, T
from oracle import sml_oracle as O


def close(a, b, tol):
    """max-norm relative agreement: fp32 sums cancel, so tiny elements are compared
    against the tensor's scale, not their own."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    scale = max(np.abs(b).max(), 1e-30)
    err = np.abs(a - b).max() / scale
    assert err <= tol, "max-norm relative error %.3e > %.1e" % (err, tol)


@pytest.mark.parametrize("d", [32, 64])
def test_g1_transfer_forward(d):
    z = golden("g1_transfer_forward_d%d.npz" % d)
    theta = O.split_theta(z, "theta.")
    x_t, x_hat = T(z["x_t"]), T(z["x_hat"])
    for net, key in (("user", "y_user"), ("item", "y_item")):
        y = O.transfer_forward(theta[net], x_t, x_hat)
        np.testing.assert_allclose(y.numpy(), z[key], rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("d", [32, 64])
@pytest.mark.parametrize("tag,kw", [("bce", {}), ("bpr", dict(bce=False)), ("bprnorm", dict(bce=False, norm=True))])
def test_g2_run_mf_loss_and_grads(d, tag, kw):
    z = golden("g2_run_mf_d%d.npz" % d)
    net = make_transfer(d, z)
    theta = O.OracleEngine.theta_of(net)
    ins = [T(z[k]).clone() for k in ("ul", "uh", "il", "ih", "nl", "nh")]
    for k in (1, 3, 5):
        ins[k].requires_grad_(True)
    loss = O.run_mf(theta, *ins, **kw)
    loss.backward()
    np.testing.assert_allclose(float(loss.detach()), float(z["loss_" + tag]), rtol=2e-6)
    for k, name in ((1, "gu_"), (3, "gi_"), (5, "gn_")):
        ref = z[name + tag]
        close(ins[k].grad.numpy(), ref, 2e-5)
    for name, p in net.named_parameters():
        ref = z["gtheta_%s.%s" % (tag, name)]
        # fc2.bias is a heavily cancelling sum (the fp32 reference itself sits 7e-5 from fp64)
        close(p.grad.numpy(), ref, 3e-4 if name.endswith("bias") else 3e-5)


def adam_close(a, b, lr, steps, frac=0.999):
    """Adam divides by sqrt(v): an element whose gradient is at rounding-noise level moves by
    up to +-lr per step in either implementation.  So: nearly all elements agree tightly, and
    none is further apart than a small fraction of the distance Adam can move it."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    tight = np.abs(a - b) <= 2e-4 * np.abs(b) + 2e-5 * lr * steps
    # (small tensors: a couple of noise-level elements must not fail the fraction test)
    assert tight.mean() >= frac or (~tight).sum() <= 2, "only %.5f of elements agree tightly" % tight.mean()
    assert np.abs(a - b).max() <= 0.05 * lr * steps, "max abs diff %.3e" % np.abs(a - b).max()


@pytest.mark.parametrize("variant", ["", "_conv"])
def test_g3_mf_stage_steps(variant):
    """K batches of MF_train_onestage incl. duplicate rows: per-batch loss, touched and
    untouched rows (dense-Adam drift), Adam moments and step count."""
    z = golden("g3_mf_stage%s.npz" % variant)      # "_conv": the same run through --transfer_type conv
    lr, l2, B, epochs = z["hp_mf"]
    B, epochs = int(B), int(epochs)
    U, d = z["W_user0"].shape
    I = z["W_item0"].shape[0]
    mf = make_mf(U, I, d, z["W_user0"], z["W_item0"])
    net = make_transfer(d, z, prefix="theta0.")
    eng = O.OracleEngine(d)
    tri = T(z["mf_triples"])
    n = z["set_t"].shape[0]
    losses = []
    for ep in range(epochs):
        losses.append(eng.mf_stage_epoch(mf, net, T(z["Wlast_user"]), T(z["Wlast_item"]), tri[ep * n:(ep + 1) * n],
                                         B, lr, l2))
    losses = np.concatenate(losses)
    # mf_batch_loss: the scalar the reference backpropagates per batch (run_MF + l2 * l2loss, model/transfer.py:488);
    # mf_runmf_loss: run_MF's own value before the l2 term
    np.testing.assert_allclose(losses, z["mf_batch_loss"], rtol=2e-5)
    assert np.all(losses >= z["mf_runmf_loss"] - 1e-6)
    assert eng.mf_step == int(z["adam_step"])
    adam_close(mf.user_laten.weight.detach().numpy(), z["W_user1"], lr, eng.mf_step)
    adam_close(mf.item_laten.weight.detach().numpy(), z["W_item1"], lr, eng.mf_step)
    su, si = eng.mf_state
    np.testing.assert_allclose(su.m.numpy(), z["adam_m_user"], rtol=1e-3, atol=1e-9)
    np.testing.assert_allclose(si.v.numpy(), z["adam_v_item"], rtol=1e-3, atol=1e-14)
    # epoch-level printed loss of the reference
    log = str(z["mf_log"])
    printed = [float(l.split("loss:")[1]) for l in log.splitlines() if "loss:" in l]
    nb = (n + B - 1) // B
    mine = [losses[e * nb:(e + 1) * nb].mean() / B for e in range(epochs)]
    np.testing.assert_allclose(mine, printed, rtol=1e-5)


@pytest.mark.parametrize("variant", ["", "_conv"])
def test_g4_tr_stage_steps(variant):
    z = golden("g4_tr_stage%s.npz" % variant)
    lr, wd, B, epochs = z["hp_tr"]
    B, epochs = int(B), int(epochs)
    d = z["Wlast_user"].shape[1]
    net = make_transfer(d, z, prefix="theta0.")
    eng = O.OracleEngine(d)
    tri = T(z["tr_triples"])
    n = z["set_tt"].shape[0]
    losses = []
    for ep in range(epochs):
        losses.append(eng.tr_stage_epoch(net, T(z["Wlast_user"]), T(z["Wlast_item"]), T(z["What_user"]),
                                         T(z["What_item"]), tri[ep * n:(ep + 1) * n], B, lr, wd))
    losses = np.concatenate(losses)
    np.testing.assert_allclose(losses, z["tr_runmf_loss"], rtol=2e-5)
    np.testing.assert_allclose(losses, z["tr_batch_loss"], rtol=2e-5)      # the backpropagated scalar (no l2 term here)
    assert eng.tr_step == int(z["adam_step"])
    for name, p in net.named_parameters():
        ref = z["theta1." + name]
        adam_close(p.detach().numpy(), ref, lr, eng.tr_step)


@pytest.mark.parametrize("variant", ["", "_conv"])
def test_g5_updata(variant):
    z = golden("g5_updata%s.npz" % variant)
    d = z["Wlast_user"].shape[1]
    net = make_transfer(d, z)
    eng = O.OracleEngine(d)
    out_u = torch.empty_like(T(z["What_user"]))
    out_i = torch.empty_like(T(z["What_item"]))
    eng.updata(net, T(z["Wlast_user"]), T(z["What_user"]), T(z["Wlast_item"]), T(z["What_item"]), out_u, out_i)
    np.testing.assert_allclose(out_u.numpy(), z["Wnew_user"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(out_i.numpy(), z["Wnew_item"], rtol=2e-5, atol=2e-6)


def test_g6_eval():
    z = golden("g6_eval.npz")
    wu, wi = T(z["mf.user_laten.weight"]), T(z["mf.item_laten.weight"])
    ranks = O.eval_ranks(wu, wi, z["rows"])
    n = z["rows"].shape[0]
    for K in (5, 10, 20):
        hits, ndcg = O.eval_metrics(ranks, K)
        assert hits == float(z["hit_%d" % K])
        np.testing.assert_allclose(ndcg, float(z["ndcg_%d" % K]), rtol=1e-6)
        np.testing.assert_array_equal(np.nonzero((ranks < K).numpy())[0], z["hitidx_%d" % K])
        np.testing.assert_allclose(hits / n, float(z["recall_%d" % K]), rtol=1e-7)
        np.testing.assert_allclose(ndcg / n, float(z["ndcgavg_%d" % K]), rtol=1e-6)


def test_g13_eval_999_negatives():
    """The shipped test format's width (1 + 999 candidates): the oracle's rank-by-count equals the reference's topk
    position for every row, and hits / ndcg / hit indices at K = 5, 10, 20."""
    z = golden("g13_eval_999.npz")
    wu, wi = T(z["mf.user_laten.weight"]), T(z["mf.item_laten.weight"])
    rows = torch.from_numpy(z["rows"].astype(np.int64))
    n = rows.shape[0]
    assert rows.shape[1] == 1001
    ranks = O.eval_ranks(wu, wi, rows)
    np.testing.assert_array_equal(ranks.numpy(), z["rank_full"])
    for K in (5, 10, 20):
        hits, ndcg = O.eval_metrics(ranks, K)
        assert hits == float(z["hit_%d" % K])
        np.testing.assert_allclose(ndcg, float(z["ndcg_%d" % K]), rtol=1e-6)
        np.testing.assert_array_equal(np.nonzero((ranks < K).numpy())[0], z["hitidx_%d" % K])
        np.testing.assert_allclose(hits / n, float(z["recall_%d" % K]), rtol=1e-7)
        np.testing.assert_allclose(ndcg / n, float(z["ndcgavg_%d" % K]), rtol=1e-6)


def test_bare_step_matches_autograd_sgd():
    """a3: the oracle's bare step is synchronous minibatch SGD on the baseline.py loss."""
    g = torch.Generator().manual_seed(3)
    U, I, d, B = 30, 20, 32, 64
    Wu, Wi = torch.randn(U, d, generator=g) * 0.3, torch.randn(I, d, generator=g) * 0.3
    u = torch.randint(0, U, (B,), generator=g); u[:10] = 2
    i = torch.randint(0, I, (B,), generator=g)
    j = torch.randint(0, I, (B,), generator=g)
    for bce in (True, False):
        a, b = Wu.clone(), Wi.clone()
        loss = O.bare_step(a, b, u, i, j, 0.05, 1e-3, 2e-3, bce=bce)
        pu, pi = torch.nn.Parameter(Wu.clone()), torch.nn.Parameter(Wi.clone())
        opt = torch.optim.SGD([pu, pi], lr=0.05)
        sp, sn = (pu[u] * pi[i]).sum(-1), (pu[u] * pi[j]).sum(-1)
        l = O.pair_loss(sp, sn, bce) + 1e-3 * 0.5 * (pu[u] ** 2).sum() + 2e-3 * 0.5 * ((pi[i] ** 2).sum() + (pi[j] ** 2).sum())
        l.backward(); opt.step()
        np.testing.assert_allclose(loss, float(l), rtol=1e-6)
        np.testing.assert_allclose(a.numpy(), pu.detach().numpy(), rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(b.numpy(), pi.detach().numpy(), rtol=1e-6, atol=1e-7)


def test_g10_baseline_bare_adam_loop():
    """G10: the baselines' bare-MF loop (reference model/baseline.py SPMF.run_one_stage2 on its own recorded
    batches): per-epoch losses as printed, final tables, Adam state, and the final test's recall/ndcg."""
    g = golden("g10_baseline_adam.npz")
    lr, l2u, l2i, B, epochs = (float(x) for x in g["hyper"])
    B, epochs = int(B), int(epochs)
    from sml_amd.mf import MFbasemode
    U, d = g["init.user_laten.weight"].shape
    I = g["init.item_laten.weight"].shape[0]
    mf = MFbasemode(U, I, d)
    mf.load_state_dict({k[5:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("init.")})
    eng = O.OracleEngine(d)
    for e in range(epochs):
        losses = eng.bare_adam_epoch(mf, torch.from_numpy(g["triples_%d" % e]), B, lr, l2u, l2i, bce=True)
        # reference prints loss_all / n_batches with 4 decimals (model/baseline.py:363-364)
        assert abs(float(np.mean(losses)) - float(g["epoch_loss"][e])) < 6e-5
    assert eng.mf_step == int(g["adam.step"])
    np.testing.assert_allclose(mf.user_laten.weight.detach().numpy(), g["final.user"], rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(mf.item_laten.weight.detach().numpy(), g["final.item"], rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(eng.mf_state[0].m.numpy(), g["adam.m_user"], rtol=2e-4, atol=1e-7)
    np.testing.assert_allclose(eng.mf_state[1].v.numpy(), g["adam.v_item"], rtol=2e-4, atol=1e-9)
    rows = torch.from_numpy(g["test_rows"])
    ranks = O.eval_ranks(mf.user_laten.weight.detach(), mf.item_laten.weight.detach(), rows)
    n = rows.shape[0]
    for k, topk in enumerate((5, 10, 20)):
        hits, ndcg = O.eval_metrics(ranks, topk)
        assert abs(hits / n - g["recall_5_10_20"][k]) < 1e-9
        assert abs(ndcg / n - g["ndcg_5_10_20"][k]) < 1e-5


def test_g11_convtransfer_forward_loss_and_gradients():
    """ConvTransfer (--transfer_type conv; reference model/conv_transfer.py:52-85): forward of both nets (user
    output unit-norm), run_MF's BPR loss, gradients w.r.t. the x_hat inputs and theta."""
    z = golden("g11_convtransfer_d32.npz")
    theta = O.split_theta(z, "theta.")
    assert O.net_kernel(theta["user"]) == 2
    x_t, x_hat = T(z["x_t"]), T(z["x_hat"])
    np.testing.assert_allclose(O.transfer_forward(theta["user"], x_t, x_hat, unit_norm=True).numpy(), z["y_user"],
                               rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(O.transfer_forward(theta["item"], x_t, x_hat).numpy(), z["y_item"], rtol=2e-5, atol=2e-6)
    theta = {n: {k: v.clone().requires_grad_(True) for k, v in t.items()} for n, t in theta.items()}
    ins = [T(z[k]).clone() for k in ("ul", "uh", "il", "ih", "nl", "nh")]
    for k in (1, 3, 5):
        ins[k].requires_grad_(True)
    loss = O.run_mf(theta, *ins, norm=False, bce=True)      # bce is ignored for kernel-2 nets: BPR only
    loss.backward()
    np.testing.assert_allclose(float(loss), float(z["loss_bpr"]), rtol=1e-5)
    close(ins[1].grad.numpy(), z["gu_bpr"], 3e-5)
    close(ins[3].grad.numpy(), z["gi_bpr"], 3e-5)
    close(ins[5].grad.numpy(), z["gn_bpr"], 3e-5)
    for net, mod in (("user", "user_transfer"), ("item", "item_transfer")):
        for k, p in theta[net].items():
            close(p.grad.numpy(), z["gtheta_bpr.%s.%s" % (mod, k)], 3e-4 if k.endswith("bias") else 3e-5)
    # run_MF(norm=True), model/conv_transfer.py:79-81 (round 5): the score over the norm of the unit-norm user output, not detached
    theta = {n: {k: v.detach().clone().requires_grad_(True) for k, v in t.items()} for n, t in theta.items()}
    ins = [T(z[k]).clone() for k in ("ul", "uh", "il", "ih", "nl", "nh")]
    for k in (1, 3, 5):
        ins[k].requires_grad_(True)
    loss = O.run_mf(theta, *ins, norm=True, bce=False)
    loss.backward()
    np.testing.assert_allclose(float(loss), float(z["loss_bprn"]), rtol=1e-5)
    close(ins[1].grad.numpy(), z["gu_bprn"], 3e-5)
    close(ins[3].grad.numpy(), z["gi_bprn"], 3e-5)
    close(ins[5].grad.numpy(), z["gn_bprn"], 3e-5)
    assert np.abs(z["gu_bprn"] - z["gu_bpr"]).max() > 1e-4          # (the flag changes the user rows' gradient: the fixture says so)
    for net, mod in (("user", "user_transfer"), ("item", "item_transfer")):
        for k, p in theta[net].items():
            close(p.grad.numpy(), z["gtheta_bprn.%s.%s" % (mod, k)], 3e-4 if k.endswith("bias") else 3e-5)
