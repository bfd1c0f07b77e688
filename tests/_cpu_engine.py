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
    """Consumes what the HIP library consumes: the exchange descriptor (sorted global item occurrences, slot stride,
    per-batch list offsets, gather hook) and the batch plan (unequal local batches, per-batch loss scales)."""

    @staticmethod
    def _batches(n, batch_size, plan):
        if plan is None:
            return [(b0, min(b0 + batch_size, n)) for b0 in range(0, n, batch_size)], None
        off = [int(v) for v in plan["batch_off"]]
        return [(off[b], off[b + 1]) for b in range(len(off) - 1)], plan.get("loss_scale")

    def mf_stage_epoch(self, mfbase, transfer, last_user, last_item, triples, batch_size, lr, l2, norm=False, bce=True,
                       plan=None, exchange=None, adaptive_beta=None):
        assert not adaptive_beta, "the CPU stand-in of the distributed tests has no --need_adaptive term"
        tri = torch.as_tensor(triples).reshape(-1, 3)
        ex = exchange if exchange is not None else self.dist.mf_exchange(tri, batch_size, self.d, 0 if bce else 1)
        d, n = self.d, tri.shape[0]
        theta = {nn: {k: p.detach() for k, p in t.items()} for nn, t in self.theta_of(transfer).items()}
        Wu, Wi = mfbase.user_laten.weight, mfbase.item_laten.weight
        if self.mf_state is None:
            self.mf_state = (O.AdamState(Wu.data), O.AdamState(Wi.data))
        su, si = self.mf_state
        keys, vals, dx_local, dx_all = ex["keys"], ex["vals"].long(), ex["dx_local"], ex["dx_all"]
        stride = ex["slot_stride"] or 2 * batch_size
        spans, scales = self._batches(n, batch_size, plan)
        losses = []
        for b, (b0, b1) in enumerate(spans):
            t = tri[b0:b1]
            Bb = t.shape[0]
            scale = float(scales[b]) if scales is not None else ex["loss_scale"]
            u, i, j = t[:, 0], t[:, 1], t[:, 2]
            gu = torch.zeros_like(Wu.data)
            loss_v = 0.0
            if Bb:
                uh = Wu.data[u].clone().requires_grad_(True)
                ih = Wi.data[i].clone().requires_grad_(True)
                nh = Wi.data[j].clone().requires_grad_(True)
                core = O.run_mf(theta, last_user[u], uh, last_item[i], ih, last_item[j], nh, norm=norm, bce=bce)
                loss = scale * core + l2 * 0.5 * torch.sum(uh ** 2 + ih ** 2 + nh ** 2)
                loss.backward()
                loss_v = float(loss.detach())
                ioff = -(-Bb // 32) * 32
                dxl = dx_local.view(-1, d)
                dxl[ioff:ioff + Bb] = ih.grad
                dxl[ioff + Bb:ioff + 2 * Bb] = nh.grad
                gu.index_add_(0, u, uh.grad)
            ex["hook"](b)
            if ex.get("item_off") is not None:
                off, cnt = int(ex["item_off"][b]), int(ex["item_off"][b + 1] - ex["item_off"][b])
            else:
                off, cnt = self.dist.world * 2 * batch_size * b, self.dist.world * 2 * Bb
            rows = (keys[off:off + cnt] & 0xffffffff).long()
            assert torch.all((keys[off:off + cnt] >> 32) == b)
            gi = torch.zeros_like(Wi.data).index_add_(0, rows, dx_all.view(-1, d)[vals[off:off + cnt]])
            self.mf_step += 1
            O.adam_dense_step(Wu.data, gu, su.m, su.v, self.mf_step, lr)
            O.adam_dense_step(Wi.data, gi, si.m, si.v, self.mf_step, lr)
            losses.append(loss_v)
        return np.array(losses, dtype=np.float64)

    def tr_stage_epoch(self, transfer, last_user, last_item, hat_user, hat_item, triples, batch_size, lr,
                       weight_decay, bce=True, loss_scale=None, plan=None, clip_max_norm=None):
        assert not clip_max_norm, "the CPU stand-in of the distributed tests has no --clip_grad"
        tri = torch.as_tensor(triples).reshape(-1, 3)
        base_scale = self.dist.tr_loss_scale(0 if bce else 1) if loss_scale is None else loss_scale
        params = list(transfer.parameters())
        if self.tr_state is None:
            self.tr_state = [O.AdamState(p.data) for p in params]
        spans, scales = self._batches(tri.shape[0], batch_size, plan)
        losses = []
        for b, (b0, b1) in enumerate(spans):
            t = tri[b0:b1]
            u, i, j = t[:, 0], t[:, 1], t[:, 2]
            for p in params:
                p.grad = None
            scale = float(scales[b]) if scales is not None else base_scale
            if t.shape[0]:
                loss = scale * O.run_mf(self.theta_of(transfer), last_user[u], hat_user[u], last_item[i], hat_item[i],
                                        last_item[j], hat_item[j], norm=False, bce=bce)
                loss.backward()
                flat = torch.cat([p.grad.reshape(-1) for p in params])
                loss_v = float(loss.detach())
            else:                                   # a batch none of whose users this rank owns: zero gradient, still exchanged
                flat = torch.zeros(sum(p.numel() for p in params))
                loss_v = 0.0
            self.grad_hook(flat, b)
            self.tr_step += 1
            o = 0
            with torch.no_grad():
                for p, s in zip(params, self.tr_state):
                    g = flat[o:o + p.numel()].view_as(p)
                    o += p.numel()
                    O.adam_dense_step(p.data, g, s.m, s.v, self.tr_step, lr, weight_decay=weight_decay)
            losses.append(loss_v)
        return np.array(losses, dtype=np.float64)
