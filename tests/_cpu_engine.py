"""Test doubles with HipEngine's surface, built on the CPU oracle (tests only).

CpuDistEngine consumes the SAME exchange descriptor sml_amd.dist builds for the HIP
library (sorted global item-occurrence keys/values, gather hook, loss scale), so the
gloo tests exercise the product's distributed logic end to end on CPU."""
import numpy as np
import torch

from oracle import sml_oracle as O


class CpuEngine(O.OracleEngine):
    def __init__(self, device="cpu", d=32, max_batch=0):
        O.OracleEngine.__init__(self, d)
        self.dist = None
        self.grad_hook = None
        self.device = torch.device("cpu")

    def mf_stage_epoch(self, mfbase, transfer, last_user, last_item, triples, *a, **k):
        return O.OracleEngine.mf_stage_epoch(self, mfbase, transfer, last_user, last_item,
                                             torch.as_tensor(triples), *a, **k)

    def tr_stage_epoch(self, transfer, lu, li, hu, hi, triples, *a, **k):
        return O.OracleEngine.tr_stage_epoch(self, transfer, lu, li, hu, hi, torch.as_tensor(triples), *a, **k)


class CpuDistEngine(CpuEngine):
    def mf_stage_epoch(self, mfbase, transfer, last_user, last_item, triples, batch_size, lr, l2, norm=False, bce=True):
        tri = torch.as_tensor(triples)
        ex = self.dist.mf_exchange(tri, batch_size, self.d)
        d, B, n = self.d, batch_size, tri.shape[0]
        theta = {nn: {k: p.detach() for k, p in t.items()} for nn, t in self.theta_of(transfer).items()}
        Wu, Wi = mfbase.user_laten.weight, mfbase.item_laten.weight
        if self.mf_state is None:
            self.mf_state = (O.AdamState(Wu.data), O.AdamState(Wi.data))
        su, si = self.mf_state
        keys, vals, dx_local, dx_all = ex["keys"], ex["vals"].long(), ex["dx_local"], ex["dx_all"]
        losses = []
        nb = (n + B - 1) // B
        for b in range(nb):
            t = tri[b * B:(b + 1) * B]
            Bb = t.shape[0]
            u, i, j = t[:, 0], t[:, 1], t[:, 2]
            uh = Wu.data[u].clone().requires_grad_(True)
            ih = Wi.data[i].clone().requires_grad_(True)
            nh = Wi.data[j].clone().requires_grad_(True)
            core = O.run_mf(theta, last_user[u], uh, last_item[i], ih, last_item[j], nh, norm=norm, bce=bce)
            loss = ex["loss_scale"] * core + l2 * 0.5 * torch.sum(uh ** 2 + ih ** 2 + nh ** 2)
            loss.backward()
            ioff = -(-Bb // 32) * 32
            dxl = dx_local.view(-1, d)
            dxl[ioff:ioff + Bb] = ih.grad
            dxl[ioff + Bb:ioff + 2 * Bb] = nh.grad
            ex["hook"](b)
            off = self.dist.world * 2 * B * b
            cnt = self.dist.world * 2 * Bb
            rows = (keys[off:off + cnt] & 0xffffffff).long()
            assert torch.all((keys[off:off + cnt] >> 32) == b)
            gi = torch.zeros_like(Wi.data).index_add_(0, rows, dx_all.view(-1, d)[vals[off:off + cnt]])
            gu = torch.zeros_like(Wu.data).index_add_(0, u, uh.grad)
            self.mf_step += 1
            O.adam_dense_step(Wu.data, gu, su.m, su.v, self.mf_step, lr)
            O.adam_dense_step(Wi.data, gi, si.m, si.v, self.mf_step, lr)
            losses.append(float(loss.detach()))
        return np.array(losses, dtype=np.float64)

    def tr_stage_epoch(self, transfer, last_user, last_item, hat_user, hat_item, triples, batch_size, lr,
                       weight_decay, bce=True, loss_scale=None):
        tri = torch.as_tensor(triples)
        scale = self.dist.tr_loss_scale() if loss_scale is None else loss_scale
        params = list(transfer.parameters())
        if self.tr_state is None:
            self.tr_state = [O.AdamState(p.data) for p in params]
        losses = []
        for b0 in range(0, tri.shape[0], batch_size):
            t = tri[b0:b0 + batch_size]
            u, i, j = t[:, 0], t[:, 1], t[:, 2]
            for p in params:
                p.grad = None
            loss = scale * O.run_mf(self.theta_of(transfer), last_user[u], hat_user[u], last_item[i], hat_item[i],
                                    last_item[j], hat_item[j], norm=False, bce=bce)
            loss.backward()
            flat = torch.cat([p.grad.reshape(-1) for p in params])
            self.grad_hook(flat, b0 // batch_size)
            self.tr_step += 1
            o = 0
            with torch.no_grad():
                for p, s in zip(params, self.tr_state):
                    g = flat[o:o + p.numel()].view_as(p)
                    o += p.numel()
                    O.adam_dense_step(p.data, g, s.m, s.v, self.tr_step, lr, weight_decay=weight_decay)
            losses.append(float(loss.detach()))
        return np.array(losses, dtype=np.float64)
