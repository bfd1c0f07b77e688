"""HipEngine: the host-side handle on libsml_hip.so for one GPU.

Everything numerical on the SML hot path goes through this object; it owns the
sml_ctx (scratch), the flat theta buffer the transfer module's parameters are
re-pointed into, and the Adam state of both optimisers.  torch is used for device
memory, streams and (in sml_amd.dist) RCCL -- not for arithmetic on the path.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import check

NET_PARAM_ORDER = ("conv1.weight", "conv1.bias", "conv2.weight", "conv2.bias",
                   "fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias")


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


class HipEngine(object):
    def __init__(self, device, d, max_batch=4096, lib=None):
        self.lib = _lib.load() if lib is None else lib           # (lib: a test hook -- _lib.load_other)
        if not torch.cuda.is_available():
            raise RuntimeError("HipEngine needs a GPU (torch.cuda.is_available() is False); there is no CPU path")
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("HipEngine device must be a cuda (HIP) device, got %s" % device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.d = int(d)
        self.max_batch = int(max_batch)
        h = ctypes.c_void_p()
        check(self.lib.sml_ctx_create(ctypes.byref(h), self.device.index, self.d, self.max_batch), "sml_ctx_create")
        self._ctx = h
        self.net_size = int(self.lib.sml_theta_net_size(self.d))
        self.offsets = [int(self.lib.sml_theta_offset(self.d, w)) for w in range(8)]
        self._flat = {}       # id(transfer) -> (flat theta, [(param, off, numel)], transfer): the entry keeps the module alive
        self.mf_state = None  # lazy-Adam state of the MF tables
        self.mf_step = 0      # torch.optim.Adam's step counter; never resets (reference model/transfer.py:392)
        self.tr_state = None  # (m, v) flat, theta layout
        self.tr_step = 0
        self.grad_hook = None  # set by sml_amd.dist for the multi-GPU TR stage
        self.dist = None       # sml_amd.dist.DistContext when the job spans several GPUs
        self._plain_stream()   # (created before any CU-masked stream of the process: see there)

    _PLAIN = {}      # device index -> the process's one side stream over all CUs

    def _plain_stream(self):
        """The side stream for work that may use the whole chip (the bare step's index preparation).  ONE per process and
        device, created with the first engine -- before any CU-masked stream exists: the HIP runtime keeps a small pool
        of hardware queues and hands a new stream the least-used one, and a queue made for hipExtStreamCreateWithCUMask
        joins that pool WITH its mask.  A stream created after the masked ones can land on the 64-CU evaluation queue
        and run four times slower (measured: every third stream of torch's pool, in a process that had partitioned the
        chip).  A stream that exists already keeps the unmasked queue it was given."""
        key = self.device.index or 0
        st = HipEngine._PLAIN.get(key)
        if st is None:
            st = HipEngine._PLAIN[key] = torch.cuda.Stream(device=self.device)
            with torch.cuda.stream(st):                  # (the runtime binds the queue at the stream's first use)
                torch.empty(1, device=self.device).zero_()
            st.synchronize()
        return st

    def close(self):
        if getattr(self, "_side_ctx_h", None):
            self.lib.sml_ctx_destroy(self._side_ctx_h)
            self._side_ctx_h = None
        if getattr(self, "_ctx", None):
            self.lib.sml_ctx_destroy(self._ctx)
            self._ctx = None
            for p in self.__dict__.pop("_peer_opened", []):
                self.lib.sml_peer_close(self.device.index, ctypes.c_void_p(p))
            # (own regions are NOT freed here: a peer may still be writing into them; they go with the process,
            # or explicitly through sml_peer_free once every rank has detached)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ helpers
    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _dev(self, t, dtype):
        if isinstance(t, np.ndarray):
            # pinned staging (torch's caching host allocator) + an asynchronous copy: a pageable upload
            # would make the host wait for everything already queued on the stream
            t = torch.from_numpy(np.ascontiguousarray(t))
            if t.dtype != dtype:
                t = t.to(dtype)
            return t.pin_memory().to(device=self.device, non_blocking=True)
        t = t.to(device=self.device, dtype=dtype)
        return t if t.is_contiguous() else t.contiguous()

    def _table(self, t):
        if t.device != self.device or t.dtype != torch.float32 or not t.is_contiguous() or t.shape[-1] != self.d:
            raise ValueError("expected a contiguous fp32 [rows,%d] tensor on %s, got %s %s on %s"
                             % (self.d, self.device, tuple(t.shape), t.dtype, t.device))
        return t

    def adopt(self, transfer):
        """Re-point the transfer module's parameters into one flat device buffer
        (user net then item net, layout sml_theta_offset) and return that buffer."""
        key = id(transfer)
        ent = self._flat.get(key)
        if ent is not None and ent[2] is transfer:
            flat, views = ent[0], ent[1]
            if all(p.data_ptr() == flat.data_ptr() + 4 * off for (p, off, _) in views):
                return flat
        flat = torch.zeros(2 * self.net_size, device=self.device, dtype=torch.float32)
        views = []
        for ni, net in enumerate((transfer.user_transfer, transfer.item_transfer)):
            named = dict(net.named_parameters())
            for w, name in enumerate(NET_PARAM_ORDER):
                p = named[name]
                off = ni * self.net_size + self.offsets[w]
                n = p.numel()
                if name == "conv1.weight" and p.shape[2] == 2:
                    # ConvTransfer's (2,1) kernel lives in the first two columns of the [10][3] block; the third
                    # column multiplies the zero x_com row and stays zero (its gradient is exactly zero)
                    block = flat[off:off + 30].view(10, 3)
                    block[:, :2].copy_(p.data.reshape(10, 2).to(self.device, torch.float32))
                    p.data = block[:, :2].view(10, 1, 2, 1)
                    views.append((p, off, n))
                    continue
                flat[off:off + n].copy_(p.data.reshape(-1).to(self.device, torch.float32))
                p.data = flat[off:off + n].view(p.shape)
                views.append((p, off, n))
        if len(self._flat) >= 8:                 # a long-lived engine does not collect every module it ever saw
            self._flat.pop(next(iter(self._flat)))
        self._flat[key] = (flat, views, transfer)
        return flat

    def load_optimizer_state(self, mfbase=None, mf_state=None, transfer=None, tr_state=None):
        """Resume from saved torch.optim.Adam states (the reference keeps both optimisers alive for the whole run,
        model/transfer.py:392-393; this is how a run is continued from a checkpoint of them).
        mf_state: dict(m_user, v_user, m_item, v_item, step) -- exp_avg / exp_avg_sq of the two tables, all rows
        current as of `step`.  tr_state: dict(m={param name: exp_avg}, v={param name: exp_avg_sq}, step) over
        transfer.named_parameters()."""
        if mf_state is not None:
            self._mf_tables(mfbase, None, None)
            s = self.mf_state
            for key, src in (("m_u", "m_user"), ("v_u", "v_user"), ("m_i", "m_item"), ("v_i", "v_item")):
                s[key].copy_(torch.as_tensor(mf_state[src]).to(self.device, torch.float32))
            self.mf_step = int(mf_state["step"])
            s["s_u"].fill_(self.mf_step)
            s["s_i"].fill_(self.mf_step)
        if tr_state is not None:
            theta = self._select(transfer)
            if self.tr_state is None or self.tr_state[0].shape != theta.shape:
                self.tr_state = (torch.zeros_like(theta), torch.zeros_like(theta), torch.zeros_like(theta))
            names = {id(p): n for n, p in transfer.named_parameters()}
            for flat, src in ((self.tr_state[0], tr_state["m"]), (self.tr_state[1], tr_state["v"])):
                flat.zero_()
                for p, off, n in self._flat[id(transfer)][1]:
                    t = torch.as_tensor(src[names[id(p)]]).to(self.device, torch.float32)
                    if p.dim() == 4 and p.shape[2] == 2 and p.shape[0] == 10:   # ConvTransfer's (2,1) conv1 kernel: see adopt()
                        flat[off:off + 30].view(10, 3)[:, :2].copy_(t.reshape(10, 2))
                    else:
                        flat[off:off + n].copy_(t.reshape(-1))
            self.tr_step = int(tr_state["step"])

    def _select(self, transfer):
        """Tell the library which transfer architecture the next calls are for (0: ConvTransfer_com,
        1: ConvTransfer -- kernel-2 nets, unit-norm user output, BPR) and return the flat theta."""
        variant = 1 if int(getattr(transfer.user_transfer, "kernel", 3)) == 2 else 0
        if variant != getattr(self, "_variant", 0):
            check(self.lib.sml_ctx_set_variant(self._ctx, variant), "sml_ctx_set_variant")
            self._variant = variant
        return self.adopt(transfer)

    # ------------------------------------------------------------------ a5/a6/a10
    def transfer_forward(self, transfer, x_t, x_hat, which):
        net = {"user": 0, "item": 1}.get(which)
        if net is None:
            raise TypeError("convtransfer has not this type")
        theta = self._select(transfer)
        x_t = self._table(self._dev(x_t, torch.float32))
        x_hat = self._table(self._dev(x_hat.detach(), torch.float32))
        out = torch.empty_like(x_t)
        check(self.lib.sml_transfer_forward(self._ctx, _ptr(theta), net, _ptr(x_t), _ptr(x_hat), _ptr(out),
                                            x_t.shape[0], self._stream()), "sml_transfer_forward")
        return out

    def updata(self, transfer, last_user, hat_user, last_item, hat_item, out_user, out_item):
        theta = self._select(transfer)
        for net, (xt, xh, out) in enumerate(((last_user, hat_user, out_user), (last_item, hat_item, out_item))):
            xt, xh, out = self._table(xt), self._table(xh), self._table(out)
            if out.data_ptr() in (xt.data_ptr(), xh.data_ptr()):
                raise ValueError("updata output may not alias its inputs")
            check(self.lib.sml_transfer_forward(self._ctx, _ptr(theta), net, _ptr(xt), _ptr(xh), _ptr(out),
                                                xt.shape[0], self._stream()), "sml_transfer_forward")

    # ------------------------------------------------------------------ a8
    def _mf_tables(self, mfbase, last_user, last_item):
        wu, wi = self._table(mfbase.user_laten.weight.data), self._table(mfbase.item_laten.weight.data)
        if self.mf_state is None or self.mf_state["m_u"].shape != wu.shape or self.mf_state["m_i"].shape != wi.shape:
            z = lambda t: torch.zeros_like(t)
            self.mf_state = dict(m_u=z(wu), v_u=z(wu), m_i=z(wi), v_i=z(wi),
                                 # -1: never touched (m = v = 0: nothing to replay, nothing to flush)
                                 s_u=torch.full((wu.shape[0],), -1, device=self.device, dtype=torch.int32),
                                 s_i=torch.full((wi.shape[0],), -1, device=self.device, dtype=torch.int32))
        s = self.mf_state
        t = _lib.MFTables()
        t.w_user, t.w_item = wu.data_ptr(), wi.data_ptr()
        t.last_user = self._table(last_user).data_ptr() if last_user is not None else 0
        t.last_item = self._table(last_item).data_ptr() if last_item is not None else 0
        t.m_user, t.v_user, t.m_item, t.v_item = (s["m_u"].data_ptr(), s["v_u"].data_ptr(),
                                                  s["m_i"].data_ptr(), s["v_i"].data_ptr())
        t.step_user, t.step_item = s["s_u"].data_ptr(), s["s_i"].data_ptr()
        t.n_user, t.n_item = wu.shape[0], wi.shape[0]
        return t

    def _loss_kind(self, bce, norm):
        if getattr(self, "_variant", 0) == 1:
            # ConvTransfer.run_MF has one loss: BPR over the unit-norm user output (model/conv_transfer.py:71-85).  norm=True
            # (:79-81) divides the score by the norm of that unit-norm output WITHOUT detaching it: with u = x / stop(|x|) the
            # value is x.(i - n) / |x| and d/dx = (i - n) / |x| - (x.(i - n)) x / |x|^3 -- exactly the differentiable-norm BPR
            # (SML_LOSS_BPR_NORM) on the net's raw user rows
            return _lib.LOSS_BPR_NORM if norm else _lib.LOSS_BPR_UNIT
        if bce:
            return _lib.LOSS_BCE
        return _lib.LOSS_BPR_NORM if norm else _lib.LOSS_BPR

    def _plan(self, plan, n):
        """ctypes view of a batch plan {batch_off: int64 [nb+1] (numpy), loss_scale: float32 [nb] or None}: batches
        of unequal size (a global batch split over ranks by user owner).  Returns (struct pointer, keep-alive)."""
        if plan is None:
            return None, None
        off = np.ascontiguousarray(plan["batch_off"], dtype=np.int64)
        if off[0] != 0 or off[-1] != n:
            raise ValueError("batch plan does not cover the %d triples" % n)
        off_dev = torch.from_numpy(off.astype(np.int32)).pin_memory().to(self.device, non_blocking=True)
        sc = plan.get("loss_scale")
        sc = None if sc is None else np.ascontiguousarray(sc, dtype=np.float32)
        p = _lib.BatchPlan()
        p.n_batches = off.shape[0] - 1
        p.batch_off = off.ctypes.data
        p.batch_off_dev = off_dev.data_ptr()
        p.loss_scale = sc.ctypes.data if sc is not None else None
        return ctypes.byref(p), (p, off, off_dev, sc)

    def mf_stage_epoch(self, mfbase, transfer, last_user, last_item, triples, batch_size, lr, l2, norm=False, bce=True,
                       plan=None, exchange=None, adaptive_beta=None):
        """plan / exchange: the multi-GPU driver's per-epoch descriptors (sml_amd.dist.EpochRoute): batches of unequal
        local size, and the global item-occurrence list of the job.  adaptive_beta: the reference's --need_adaptive
        term (model/transfer.py:490-499; 0.1 there)."""
        check(self.lib.sml_ctx_set_adaptive(self._ctx, float(adaptive_beta or 0.0)), "sml_ctx_set_adaptive")
        theta = self._select(transfer)
        tri = self._dev(triples, torch.int64)
        n = tri.shape[0]
        nb = (n + batch_size - 1) // batch_size if plan is None else len(plan["batch_off"]) - 1
        t = self._mf_tables(mfbase, last_user, last_item)
        self._mf_lr = float(lr)
        losses = torch.empty(nb, device=self.device, dtype=torch.float32)
        step = ctypes.c_int64(self.mf_step)
        xp = None
        pp, keep = self._plan(plan, n)
        if self.dist is not None:
            ex = exchange if exchange is not None else self.dist.mf_exchange(tri, int(batch_size), self.d, self._loss_kind(bce, norm))
            hook_fn = ex["hook"]

            def _cb(_user, b):
                try:
                    hook_fn(int(b))
                    return 0
                except Exception:
                    import traceback
                    traceback.print_exc()
                    return 1
            x = _lib.MFExchange()
            x.world = ex["world"]
            x.key_items, x.val_items = ex["keys"].data_ptr(), ex["vals"].data_ptr()
            x.dx_local, x.dx_items_all = ex["dx_local"].data_ptr(), ex["dx_all"].data_ptr()
            x.hook = _lib.MF_HOOK(_cb) if hook_fn is not None else ctypes.cast(None, _lib.MF_HOOK)
            x.hook_user = None
            x.loss_scale = ex["loss_scale"]
            x.slot_stride = int(ex.get("slot_stride", 0))
            ioff = ex.get("item_off")
            if ioff is not None:
                ioff = np.ascontiguousarray(ioff, dtype=np.int64)
                keep = (keep, ioff)
            x.item_off = ioff.ctypes.data if ioff is not None else None
            x.push_rows = int(ex.get("push_rows", 0))
            x.lists_unsorted = 1 if ex.get("unsorted") else 0
            xp = ctypes.byref(x)
        check(self.lib.sml_mf_stage_epoch(self._ctx, _ptr(theta), ctypes.byref(t), _ptr(tri), n, int(batch_size),
                                          float(lr), float(l2), self._loss_kind(bce, norm), ctypes.byref(step),
                                          _ptr(losses), xp, pp, self._stream()), "sml_mf_stage_epoch")
        self.mf_step = step.value
        return losses

    def mf_flush(self, mfbase):
        """Replay pending zero-gradient Adam steps so the tables can be read."""
        if self.mf_state is None or self.mf_step == 0:
            return
        t = self._mf_tables(mfbase, None, None)
        check(self.lib.sml_mf_adam_flush(self._ctx, ctypes.byref(t), float(self._mf_lr), self.mf_step, self._stream()),
              "sml_mf_adam_flush")

    # ------------------------------------------------------------------ a9
    def tr_stage_epoch(self, transfer, last_user, last_item, hat_user, hat_item, triples, batch_size, lr,
                       weight_decay, bce=True, loss_scale=None, plan=None, clip_max_norm=None):
        """clip_max_norm: the reference's --clip_grad / --maxnorm_grad (model/transfer.py:725-727)."""
        check(self.lib.sml_ctx_set_grad_clip(self._ctx, float(clip_max_norm or 0.0)), "sml_ctx_set_grad_clip")
        theta = self._select(transfer)
        if loss_scale is None:
            loss_scale = self.dist.tr_loss_scale(self._loss_kind(bce, False)) if self.dist is not None else 1.0
        if self.tr_state is None or self.tr_state[0].shape != theta.shape:
            self.tr_state = (torch.zeros_like(theta), torch.zeros_like(theta), torch.zeros_like(theta))
        m, v, grad = self.tr_state
        tri = self._dev(triples, torch.int64)
        n = tri.shape[0]
        nb = (n + batch_size - 1) // batch_size if plan is None else len(plan["batch_off"]) - 1
        pp, keep = self._plan(plan, n)
        t = _lib.TRTables()
        t.last_user, t.last_item = self._table(last_user).data_ptr(), self._table(last_item).data_ptr()
        t.hat_user, t.hat_item = self._table(hat_user).data_ptr(), self._table(hat_item).data_ptr()
        t.n_user, t.n_item = last_user.shape[0], last_item.shape[0]
        losses = torch.empty(nb, device=self.device, dtype=torch.float32)
        step = ctypes.c_int64(self.tr_step)
        if self.grad_hook is not None:
            hook_fn = self.grad_hook

            def _cb(_user, _grad, _n, b):
                try:
                    hook_fn(grad, int(b))
                    return 0
                except Exception:  # surfaced as SML_ESTATE by the library
                    import traceback
                    traceback.print_exc()
                    return 1
            cb = _lib.GRAD_HOOK(_cb)
        else:
            cb = ctypes.cast(None, _lib.GRAD_HOOK)
        # the flat theta-gradient is an output only where somebody reads it: the exchange on several GPUs, or a caller
        # that set keep_theta_grad (the G2 parity test); the fused single-GPU step does not write it otherwise
        want_grad = self.grad_hook is not None or self.dist is not None or getattr(self, "keep_theta_grad", False)
        check(self.lib.sml_tr_stage_epoch(self._ctx, _ptr(theta), _ptr(m), _ptr(v), _ptr(grad) if want_grad else None, ctypes.byref(t),
                                          _ptr(tri), n, int(batch_size), float(lr), float(weight_decay),
                                          self._loss_kind(bce, False), float(loss_scale), ctypes.byref(step),
                                          _ptr(losses), cb, None, pp, self._stream()), "sml_tr_stage_epoch")
        self.tr_step = step.value
        return losses

    # ------------------------------------------------------------------ a7 with gradients (the autograd surface)
    def run_mf_grad(self, transfer, user_last, user_hat, item_last, item_hat, bce=True, norm=False,
                    want_rows=True, want_theta=True):
        """run_MF of one batch given as row blocks (user_* [B,d]; item_* [2B,d]: positives then negatives) with its
        gradients: (loss 0-dim, d loss / d user_hat or None, d loss / d item_hat or None, flat theta gradient or None).
        No optimiser step -- this is what sml_amd.conv_transfer's autograd Function calls."""
        theta = self._select(transfer)
        ul, uh = self._table(user_last), self._table(user_hat)
        il, ih = self._table(item_last), self._table(item_hat)
        B = ul.shape[0]
        if uh.shape[0] != B or il.shape[0] != 2 * B or ih.shape[0] != 2 * B:
            raise ValueError("run_mf_grad: user blocks [B,d], item blocks [2B,d]")
        loss = torch.empty((), device=self.device, dtype=torch.float32)
        du = torch.empty_like(uh) if want_rows else None
        di = torch.empty_like(ih) if want_rows else None
        gt = torch.empty_like(theta) if want_theta else None
        check(self.lib.sml_run_mf_grad(self._ctx, _ptr(theta), _ptr(ul), _ptr(uh), _ptr(il), _ptr(ih), B, self._loss_kind(bce, norm),
                                       _ptr(loss), _ptr(du), _ptr(di), _ptr(gt), self._stream()), "sml_run_mf_grad")
        return loss, du, di, gt

    def theta_views(self, transfer):
        """[(parameter, offset, numel)] of the adopted transfer module inside its flat theta (adopt() first)."""
        self.adopt(transfer)
        return self._flat[id(transfer)][1]

    # ------------------------------------------------------------------ a3
    def _bare_exchange(self, ex):
        """ctypes view of sml_amd.dist.DistContext.bare_exchange(): (struct pointer or None, keep-alive)."""
        if ex is None:
            return None, None
        hook_fn = ex["hook"]

        def _cb(_user, b):
            try:
                hook_fn(int(b))
                return 0
            except Exception:
                import traceback
                traceback.print_exc()
                return 1
        x = _lib.BareExchange()
        x.world = ex["world"]
        x.items_all, x.dx_local, x.dx_items_all = ex["items_all"].data_ptr(), ex["dx_local"].data_ptr(), ex["dx_all"].data_ptr()
        cb = _lib.MF_HOOK(_cb) if hook_fn is not None else ctypes.cast(None, _lib.MF_HOOK)
        x.hook = cb
        x.hook_user = None
        x.loss_scale = ex["loss_scale"]
        return ctypes.byref(x), (x, cb, ex)

    def bare_prepare(self, triples, batch_size, n_user=0, n_item=0, exchange=None):
        """Build the index lists of a coming bare epoch on the engine's side stream (sort by (batch,row),
        unique marks, duplicates-only compaction) while earlier work keeps the main stream busy.
        n_user / n_item (the table heights; 0 = unknown) let the sort use 32-bit keys.
        exchange: several GPUs (DistContext.bare_exchange): the item lists are then the job's.
        Returns a handle for bare_epoch(prepared=...)."""
        if getattr(self, "_prep", None) is None:
            import os
            k = int(os.environ.get("SML_PREP_CUS", "0"))
            if k > 0:                                    # the preparation confined to the last k compute units
                self._prep = self._masked_stream(self._n_cus() - k, self._n_cus())
            elif os.environ.get("SML_PREP_PRIO", "") in ("low", "high"):
                lo, hi = torch.cuda.Stream.priority_range()
                self._prep = torch.cuda.Stream(device=self.device, priority=int(lo if os.environ["SML_PREP_PRIO"] == "low" else hi))
            else:
                self._prep = self._plain_stream()
        tri = self._dev(triples, torch.int64)
        slot = self._prep_slot = 1 - getattr(self, "_prep_slot", 1)
        cur = torch.cuda.current_stream(self.device)
        self._prep.wait_stream(cur)          # the slot's previous user (two epochs back) has been queued before this
        xp, keep = self._bare_exchange(exchange)
        with torch.cuda.stream(self._prep):
            check(self.lib.sml_embed_loss_sgd_prepare(self._ctx, _ptr(tri), tri.shape[0], int(batch_size), int(n_user),
                                                      int(n_item), slot, xp, self._stream()), "sml_embed_loss_sgd_prepare")
            ev = torch.cuda.Event()
            ev.record(self._prep)
        tri.record_stream(self._prep)
        return dict(slot=slot, event=ev, tri=tri, batch=int(batch_size), exchange=exchange)

    def index_lists(self, prepared):
        """Test hook: the lists a bare_prepare handle holds, as numpy arrays (see sml_index_lists_read)."""
        import numpy as np
        prepared["event"].synchronize()
        n, batch = int(prepared["tri"].shape[0]), prepared["batch"]
        nb = (n + batch - 1) // batch

        def rd(which, dtype, count):
            buf = np.zeros(max(int(count), 1), dtype)
            got = self.lib.sml_index_lists_read(self._ctx, prepared["slot"], which, buf.ctypes.data, buf.nbytes)
            if got < 0:
                check(int(got), "sml_index_lists_read")
            return buf[:got // buf.itemsize]

        two = rd(11, np.int32, 2)
        out = dict(runs_u=rd(0, np.uint32, 8 * (n // 2 + 8)).reshape(-1, 8), runs_i=rd(1, np.uint32, 8 * (n + 8)).reshape(-1, 8),
                   off_u=rd(2, np.int32, nb + 1), off_i=rd(3, np.int32, nb + 1), cnt_u=rd(4, np.int32, nb), cnt_i=rd(5, np.int32, nb),
                   val_u=rd(6, np.uint32, n), val_i=rd(7, np.uint32, 2 * n), uniq=rd(8, np.uint8, 3 * nb * batch),
                   max_len=int(two[0]), hot_cap=int(two[1]))
        out["hot_list"] = rd(9, np.uint32, nb * out["hot_cap"] * 3).reshape(nb, -1, 3) if out["hot_cap"] else None
        out["hot_count"] = rd(10, np.int32, nb) if out["hot_cap"] else None
        return out

    def bare_epoch(self, w_user, w_item, triples, batch_size, lr, lam_user, lam_item, bce=True, prepared=None, exchange=None):
        if w_user.dtype not in (torch.float32, torch.float16) or w_item.dtype != w_user.dtype:
            raise ValueError("tables must both be fp32 or both fp16")
        for w in (w_user, w_item):
            if w.device != self.device or not w.is_contiguous() or w.shape[-1] != self.d:
                raise ValueError("tables must be contiguous [rows,%d] on %s" % (self.d, self.device))
        slot = -1
        if prepared is not None:
            if prepared["batch"] != int(batch_size):
                raise ValueError("prepared for another batch size")
            tri, slot = prepared["tri"], prepared["slot"]
            exchange = prepared.get("exchange") if exchange is None else exchange
            torch.cuda.current_stream(self.device).wait_event(prepared["event"])
        else:
            tri = self._dev(triples, torch.int64)
        n = tri.shape[0]
        nb = (n + batch_size - 1) // batch_size
        losses = torch.empty(nb, device=self.device, dtype=torch.float32)
        xp, keep = self._bare_exchange(exchange)
        check(self.lib.sml_embed_loss_sgd_epoch(self._ctx, _ptr(w_user), _ptr(w_item), w_user.shape[0], w_item.shape[0],
                                                w_user.element_size(), _ptr(tri), n, int(batch_size), float(lr),
                                                float(lam_user), float(lam_item),
                                                _lib.LOSS_BCE if bce else _lib.LOSS_BPR, _ptr(losses), slot, xp,
                                                self._stream()),
              "sml_embed_loss_sgd_epoch")
        return losses

    def bare_epoch_sharded(self, w_user, triples, batch_size, lr, lam_user, lam_item, shard, bce=True):
        """The bare step with the item table SHARDED over the ranks (sml_embed_loss_sgd_epoch_sharded; peers attached):
        `shard` is sml_amd.dist.DistContext.bare_shard()'s descriptor -- head_rows / shard_rows, every rank's tail-shard
        address, this rank's head replica, the gathered item columns, the loss scale.  n_item = the GLOBAL table height."""
        if w_user.dtype not in (torch.float32, torch.float16) or not w_user.is_contiguous() or w_user.shape[-1] != self.d:
            raise ValueError("user table: contiguous fp32 / fp16 [rows,%d]" % self.d)
        tri = self._dev(triples, torch.int64)
        n = tri.shape[0]
        nb = (n + batch_size - 1) // batch_size
        losses = torch.empty(nb, device=self.device, dtype=torch.float32)
        world = int(shard["world"])
        ptrs = (ctypes.c_void_p * world)(*[int(p) for p in shard["item_shard"]])
        x = _lib.BareShard()
        x.world, x.rank = world, int(shard["rank"])
        x.head_rows, x.shard_rows = int(shard["head_rows"]), int(shard["shard_rows"])
        x.item_shard = ptrs
        head = shard.get("w_item_head")
        x.w_item_head = head.data_ptr() if head is not None and head.numel() else None
        x.items_all = shard["items_all"].data_ptr()
        x.loss_scale = float(shard["loss_scale"])
        check(self.lib.sml_embed_loss_sgd_epoch_sharded(self._ctx, _ptr(w_user), w_user.shape[0], int(shard["n_item"]), w_user.element_size(),
                                                        _ptr(tri), n, int(batch_size), float(lr), float(lam_user), float(lam_item),
                                                        _lib.LOSS_BCE if bce else _lib.LOSS_BPR, _ptr(losses), ctypes.byref(x), self._stream()),
              "sml_embed_loss_sgd_epoch_sharded")
        return losses

    def bare_adam_epoch(self, mfbase, triples, batch_size, lr, lam_user, lam_item, bce=True):
        """One epoch of the baselines' bare-MF step (reference model/baseline.py:343-361): BCE (or BPR) + L2 with
        the dense-Adam semantics of torch.optim.Adam over MFbase, replayed lazily per row.  Shares the MF
        optimiser state (m, v, step) with mf_stage_epoch; call mf_flush before reading the tables."""
        tri = self._dev(triples, torch.int64)
        n = tri.shape[0]
        nb = (n + batch_size - 1) // batch_size
        t = self._mf_tables(mfbase, None, None)
        self._mf_lr = float(lr)
        losses = torch.empty(nb, device=self.device, dtype=torch.float32)
        step = ctypes.c_int64(self.mf_step)
        check(self.lib.sml_embed_loss_adam_epoch(self._ctx, ctypes.byref(t), _ptr(tri), n, int(batch_size), float(lr),
                                                 float(lam_user), float(lam_item),
                                                 _lib.LOSS_BCE if bce else _lib.LOSS_BPR, ctypes.byref(step),
                                                 _ptr(losses), self._stream()), "sml_embed_loss_adam_epoch")
        self.mf_step = step.value
        return losses

    # ------------------------------------------------------------------ device batch supply (fast mode)
    def sample_negatives(self, users, item_all, user_ptr, user_items, seed):
        """A negative per element of `users` (device int64), uniform over item_all, never one of the user's own items
        (CSR user_ptr / user_items, device int64).  Distribution of the reference's sampler, not its numpy stream."""
        users = self._dev(users, torch.int64)
        n = users.shape[0]
        negs = torch.empty(n, device=self.device, dtype=torch.int64)
        failed = torch.zeros(1, device=self.device, dtype=torch.int32)
        check(self.lib.sml_sample_negatives(self._ctx, _ptr(users), n, _ptr(item_all), item_all.shape[0], _ptr(user_ptr),
                                            user_ptr.shape[0] - 1, _ptr(user_items), ctypes.c_uint64(int(seed) & (2 ** 64 - 1)),
                                            _ptr(negs), _ptr(failed), self._stream()), "sml_sample_negatives")
        return negs, failed

    def device_epoch(self, ui, n, seed, mat=None, row_stride=1, col=0):
        """int64 [n,3] device triples of one shuffled pass (sml_device_epoch): columns 0, 1 = ui[perm(e)] (ui: device int64
        [n,2]); column 2 = mat[perm(e) * row_stride + col] when a device integer matrix / column vector is given, else
        left for sample_negatives.  perm: the counter-based permutation keyed by `seed`."""
        out = torch.empty((int(n), 3), device=self.device, dtype=torch.int64)
        eb = 8
        if mat is not None:
            if mat.dtype not in (torch.int32, torch.int64) or not mat.is_contiguous() or mat.device != self.device:
                raise ValueError("mat: a contiguous int32 / int64 device tensor")
            eb = mat.element_size()
        check(self.lib.sml_device_epoch(self._ctx, _ptr(ui), _ptr(mat), eb, int(row_stride), int(col), int(n),
                                        ctypes.c_uint64(int(seed) & (2 ** 64 - 1)), _ptr(out), self._stream()), "sml_device_epoch")
        return out

    # ------------------------------------------------------------------ a2
    def mf_forward(self, w_user, w_item, user, item, norm=False):
        wu, wi = self._table(w_user), self._table(w_item)
        user, item = self._dev(user, torch.int64), self._dev(item, torch.int64)
        n = user.shape[0]
        ue = torch.empty(n, self.d, device=self.device, dtype=torch.float32)
        ie = torch.empty_like(ue)
        sc = torch.empty(n, device=self.device, dtype=torch.float32)
        check(self.lib.sml_mf_forward(self._ctx, _ptr(wu), _ptr(wi), _ptr(user), _ptr(item), n, int(bool(norm)),
                                      _ptr(ue), _ptr(ie), _ptr(sc), self._stream()), "sml_mf_forward")
        return ue, ie, sc

    # ------------------------------------------------------------------ a13
    BLOCKED_EVAL_MIN_ITEM_BYTES = 6 << 20   # L2-block the evaluation once the item table outgrows an XCD's 4 MB L2

    def _blocked_rows(self, rows, n_item):
        """Candidates of `rows` grouped by item range (built once per test set, kept while the tensor lives)."""
        key = (rows.data_ptr(), tuple(rows.shape), int(n_item))
        cache = self.__dict__.setdefault("_eval_blocked", {})
        hit = cache.get(key)
        if hit is None:
            if len(cache) >= 3:
                cache.pop(next(iter(cache)))
            rows_b = torch.empty(rows.shape, device=self.device, dtype=torch.int32)
            off = torch.empty((rows.shape[0], 9), device=self.device, dtype=torch.int32)
            check(self.lib.sml_eval_prepare(self._ctx, _ptr(rows), rows.shape[0], rows.shape[1], int(n_item), _ptr(rows_b),
                                            _ptr(off), self._stream()), "sml_eval_prepare")
            rows.record_stream(torch.cuda.current_stream(self.device))      # (it may have been uploaded on another stream)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            hit = cache[key] = (rows_b, off, rows, ev)     # keeps `rows` alive: the key is its address
        else:
            torch.cuda.current_stream(self.device).wait_event(hit[3])   # built on another stream, perhaps
        return hit[0], hit[1]

    def _sliced_rows(self, rows, n_item):
        """Candidates of `rows` re-ordered slice-major for the LDS-sliced rank pass (built once per test set, kept while
        the tensor lives); None when the shape is outside that pass's range."""
        n, c = rows.shape
        ns = int(self.lib.sml_eval_sliced_slices(self._ctx, n, c, int(n_item)))
        if ns <= 0:
            return None
        key = (rows.data_ptr(), tuple(rows.shape), int(n_item))
        cache = self.__dict__.setdefault("_eval_sliced", {})
        hit = cache.get(key)
        if hit is None:
            if len(cache) >= 3:
                cache.pop(next(iter(cache)))
            entries = torch.empty(int(self.lib.sml_eval_sliced_entries(self._ctx, n, c, int(n_item))), device=self.device, dtype=torch.int32)
            seg_off = torch.empty(ns * ((n + 63) // 64) + 1, device=self.device, dtype=torch.int32)
            work = torch.empty(int(self.lib.sml_eval_sliced_work_ints(self._ctx, n, c, int(n_item))), device=self.device, dtype=torch.int32)
            check(self.lib.sml_eval_prepare_sliced(self._ctx, _ptr(rows), n, c, int(n_item), _ptr(entries), _ptr(seg_off), _ptr(work),
                                                   self._stream()), "sml_eval_prepare_sliced")
            cur = torch.cuda.current_stream(self.device)
            rows.record_stream(cur)
            ev = torch.cuda.Event()
            ev.record(cur)
            hit = cache[key] = (entries, seg_off, rows, ev)     # keeps `rows` alive: the key is its address
        else:
            torch.cuda.current_stream(self.device).wait_event(hit[3])
        return hit[0], hit[1]

    # The form large test sets get: blocked (default) | sliced | plain.  The LDS-sliced pass is 1.8x faster on the evaluation
    # partition (1.38 against 2.55 ms per Yelp-shaped evaluation) and leaves that stream 45 % busy instead of 86 %, but the period
    # is 0.5-1 ms LONGER with it in a same-box A/B (the training kernels beside it run 1 % slower: profiles/r05s_*): the
    # evaluation stream is not the critical path, so the cheaper-to-sit-beside kernel stays the default.
    EVAL_MODE = __import__("os").environ.get("SML_EVAL", "blocked")

    def eval_ranks(self, user_tab, item_tab, rows, blocked=None, max_workgroups=0, sliced=None):
        wu, wi = self._table(user_tab), self._table(item_tab)
        rows = self._dev(rows, torch.int64)
        n, c = rows.shape
        rank = torch.empty(n, device=self.device, dtype=torch.int32)
        if n == 0:                         # a rank that owns no row of this test set
            return rank
        if wi.numel() * 4 > (1 << 32):     # past the blocked kernel's 32-bit byte offsets: the plain kernel ranks it
            blocked = False
        large = wi.numel() * 4 >= self.BLOCKED_EVAL_MIN_ITEM_BYTES and n * c >= (1 << 22)
        if sliced is None:
            sliced = blocked is None and large and self.EVAL_MODE == "sliced"
        if sliced:
            prep = self._sliced_rows(rows, wi.shape[0])
            if prep is not None:
                entries, seg_off = prep
                # one scratch block per (stream, size), kept: passes queued on one stream use it one after the other (stream
                # order), and nothing of it outlives a pass
                nbytes = int(self.lib.sml_eval_sliced_scratch_bytes(self._ctx, n, c, wi.shape[0]))
                skey = (torch.cuda.current_stream(self.device).cuda_stream, nbytes)
                pool = self.__dict__.setdefault("_eval_sliced_scratch", {})
                scratch = pool.get(skey)
                if scratch is None:
                    if len(pool) >= 4:
                        pool.pop(next(iter(pool)))
                    scratch = pool[skey] = torch.empty(nbytes, device=self.device, dtype=torch.uint8)
                check(self.lib.sml_eval_ranks_sliced(self._ctx, _ptr(wu), _ptr(wi), _ptr(rows), _ptr(entries), _ptr(seg_off), n, c,
                                                     wi.shape[0], _ptr(scratch), _ptr(rank), int(max_workgroups), self._stream()),
                      "sml_eval_ranks_sliced")
                for t in (entries, seg_off):
                    t.record_stream(torch.cuda.current_stream(self.device))
                return rank
        if blocked is None:
            blocked = large and self.EVAL_MODE != "plain"
        if blocked:
            rows_b, off = self._blocked_rows(rows, wi.shape[0])
            check(self.lib.sml_eval_ranks_blocked(self._ctx, _ptr(wu), _ptr(wi), _ptr(rows_b), _ptr(off), n, c,
                                                  _ptr(rank), int(max_workgroups), self._stream()), "sml_eval_ranks_blocked")
            for t in (rows_b, off):            # built on one stream, read here on another: not to be recycled under this launch
                t.record_stream(torch.cuda.current_stream(self.device))
        else:
            check(self.lib.sml_eval_ranks(self._ctx, _ptr(wu), _ptr(wi), _ptr(rows), n, c, _ptr(rank), self._stream()),
                  "sml_eval_ranks")
        return rank

    # Evaluations never sit on the training stream.  eval_submit copies the two tables into a snapshot (one copy
    # kernel, a few tens of MB) on the CURRENT stream and queues the rank pass over the snapshot on the engine's side
    # stream; the training kernels that follow on the current stream -- which may overwrite the tables at once -- do
    # not wait for it.
    SNAPSHOTS = 3
    # The chip is PARTITIONED between the two streams: the side stream owns the last SIDE_EVAL_CUS compute units of
    # the CU mask (mask bit i is a CU of XCD i % 8: a contiguous range takes the same share of every XCD), and the
    # training loop may run on the rest (`training_stream()`); the evaluation then neither shares SIMD issue slots
    # nor a CU's L2 port with the latency-bound training kernels.  SML_SIDE_EVAL_CUS=0 falls back to a low-priority
    # stream over all CUs with a capped grid.
    SIDE_EVAL_CUS = int(__import__("os").environ.get("SML_SIDE_EVAL_CUS", "-1"))      # -1: a quarter of the chip (64 of 256 CUs)
    SIDE_EVAL_WORKGROUPS = int(__import__("os").environ.get("SML_SIDE_EVAL_WGS", "0"))   # grid cap of a side-stream evaluation

    def _n_cus(self):
        return int(torch.cuda.get_device_properties(self.device).multi_processor_count)

    def _side_cus(self):
        if self.SIDE_EVAL_CUS >= 0:
            return self.SIDE_EVAL_CUS
        n = self._n_cus()
        return (n // 4) // 8 * 8 if n >= 64 else 0       # whole CUs-per-XCD multiples; tiny parts are not partitioned

    _MASKED = {}     # (device index, lo, hi) -> ExternalStream: one pair of hardware queues per process, shared by its engines
    #                  and never destroyed (torch's allocator and events keep referring to a stream it has seen)

    def _masked_stream(self, lo, hi):
        key = (self.device.index or 0, int(lo), int(hi))
        st = HipEngine._MASKED.get(key)
        if st is None:
            h = ctypes.c_void_p()
            check(self.lib.sml_stream_create_cu_range(ctypes.byref(h), key[0], key[1], key[2]), "sml_stream_create_cu_range")
            st = HipEngine._MASKED[key] = torch.cuda.ExternalStream(h.value, device=self.device)
        return st

    def training_stream(self):
        """The stream over the CUs the side stream does NOT own (None when the chip is not partitioned)."""
        if self._side_cus() <= 0:
            return None
        if getattr(self, "_train", None) is None:
            self._train = self._masked_stream(0, self._n_cus() - self._side_cus())
        return self._train

    def partition(self):
        """Context manager: run the enclosed training loop on the training partition (no-op when the chip is not
        partitioned or SML_TRAIN_PARTITION=0).  The stream is ordered after the current stream on entry and the
        current stream after it on exit."""
        import contextlib, os

        @contextlib.contextmanager
        def scope():
            ts = self.training_stream() if os.environ.get("SML_TRAIN_PARTITION", "1") != "0" else None
            if ts is None:
                yield
                return
            outer = torch.cuda.current_stream(self.device)
            ts.wait_stream(outer)
            with torch.cuda.stream(ts):
                yield
            outer.wait_stream(ts)
        return scope()

    def _side_eval_cap(self):
        if self.SIDE_EVAL_WORKGROUPS > 0:
            return self.SIDE_EVAL_WORKGROUPS
        return 4 * self._side_cus() if self._side_cus() > 0 else 256

    def _side_stream(self):
        if getattr(self, "_side", None) is None:
            if self._side_cus() > 0:
                n = self._n_cus()
                self._side = self._masked_stream(n - self._side_cus(), n)
                return self._side
            prio = 0
            try:
                lo, _hi = torch.cuda.Stream.priority_range()
                prio = int(lo)                      # least urgent
            except Exception:
                prio = 0
            try:
                self._side = torch.cuda.Stream(device=self.device, priority=prio)
            except Exception:
                self._side = torch.cuda.Stream(device=self.device)
        return self._side

    def copy_tables(self, pairs):
        """dst.copy_(src) for up to four (dst, src) pairs of same-shaped contiguous fp32 device tensors, in ONE launch
        on the current stream (save_MF_weight, model/transfer.py:911-943; evaluation snapshots).  The copies of one
        call run concurrently: no dst may be another pair's src."""
        pairs = list(pairs)
        for k in range(0, len(pairs), 4):
            chunk = pairs[k:k + 4]
            n = len(chunk)
            dst, src, nbytes = (ctypes.c_void_p * n)(), (ctypes.c_void_p * n)(), (ctypes.c_int64 * n)()
            for q, (d, s) in enumerate(chunk):
                if d.shape != s.shape or d.dtype != s.dtype or not (d.is_contiguous() and s.is_contiguous()) \
                        or (d.numel() * d.element_size()) % 16 or d.data_ptr() % 16 or s.data_ptr() % 16 or d.numel() == 0:
                    d.copy_(s)                    # odd shapes, misaligned views, empty tensors: torch's copy
                    dst[q], src[q], nbytes[q] = None, None, 0      # (a zero-byte job: the library skips it)
                    continue
                dst[q], src[q], nbytes[q] = d.data_ptr(), s.data_ptr(), d.numel() * d.element_size()
            check(self.lib.sml_copy_tables(n, dst, src, nbytes, self._stream()), "sml_copy_tables")

    def _snap_slot(self, cur):
        """Next snapshot slot of the ring; the current stream waits (on the device) for the evaluation that last read it."""
        ring = self.__dict__.setdefault("_snap", [])
        k = self.__dict__.get("_snap_next", 0) % self.SNAPSHOTS
        self._snap_next = k + 1
        while len(ring) <= k:
            ring.append(dict(ev=None))
        slot = ring[k]
        if slot["ev"] is not None:
            cur.wait_event(slot["ev"])          # the evaluation that last read this snapshot is done (device-side wait)
        return slot

    @staticmethod
    def _snap_buf(slot, key, like):
        if slot.get(key) is None or slot[key].shape != like.shape:
            slot[key] = torch.empty_like(like)
        return slot[key]

    def _side_order(self, cur, side):
        """The side stream may start on a snapshot once the copies queued on `cur` so far are done.  Device-side ordering
        (a flag kernel behind the copies, a polling kernel ahead of the side stream's work) instead of an event: a
        cross-queue barrier packet cost the training stream 2-4 ms per period (31 of them), the two tiny kernels cost
        nothing measurable."""
        mode = __import__("os").environ.get("SML_SIDE_SYNC", "flag")
        if mode == "event":
            check(self.lib.sml_stream_wait_stream(ctypes.c_void_p(side.cuda_stream), ctypes.c_void_p(cur.cuda_stream)),
                  "sml_stream_wait_stream")
            return
        if getattr(self, "_sync_flag", None) is None:
            self._sync_flag = torch.zeros(2, device=self.device, dtype=torch.int32)   # [sequence, time-outs]
            self._sync_seq = 0
            self._sync_seen = 0           # time-outs already reported
        self._sync_seq += 1
        check(self.lib.sml_flag_set(_ptr(self._sync_flag), self._sync_seq, ctypes.c_void_p(cur.cuda_stream)), "sml_flag_set")
        # The waiter starts polling as soon as the side stream is free, while its signal sits behind everything the
        # host has already queued on the training stream (the driver runs a whole stage ahead): the time-out is a
        # HANG GUARD, minutes not seconds (SML_FLAG_TIMEOUT_S).  A waiter that does give up is counted in flag[1]
        # (side_sync_check); the sequence word is untouched, so later evaluations stay ordered.
        check(self.lib.sml_flag_wait(_ptr(self._sync_flag), self._sync_seq, self._flag_timeout(),
                                     ctypes.c_void_p(side.cuda_stream)), "sml_flag_wait")

    def eval_submit(self, user_tab, item_tab, rows):
        """Queue the ranks of `rows` under the tables AS THEY ARE NOW (at this point of the current stream)
        and return a handle; the caller may modify the tables right away."""
        wu, wi = self._table(user_tab), self._table(item_tab)
        rows = self._dev(rows, torch.int64)
        side = self._side_stream()
        cur = torch.cuda.current_stream(self.device)
        slot = self._snap_slot(cur)
        su, si = self._snap_buf(slot, "u", wu), self._snap_buf(slot, "i", wi)
        self.copy_tables([(su, wu), (si, wi)])
        self._side_order(cur, side)
        with torch.cuda.stream(side):
            ranks = self.eval_ranks(su, si, rows, max_workgroups=self._side_eval_cap())
            ev = torch.cuda.Event()
            ev.record(side)
        slot["ev"] = ev
        rows.record_stream(side)
        return dict(ranks=ranks, event=ev, n=rows.shape[0])

    # A table-sized forward whose ONLY reader is an evaluation does not belong on the training stream either.  The
    # reference calls updata() ahead of every validation (model/transfer.py:737-739, 829-833); the one ahead of the "before
    # train transfer" test is overwritten by the next updata before any training kernel reads the tables: 10 of a period's
    # 21 table-sized forwards, 5 ms of its 110.  eval_submit_transferred snapshots what that forward READS (the four tables
    # and theta, one copy launch) and queues the forward itself, into the snapshot slot's own tables, ahead of the rank pass on
    # the evaluation stream's CUs.  Same kernels, same rows, same bits -- the tables the caller holds are not written.
    TRANSFERRED_MAX_BYTES = int(__import__("os").environ.get("SML_TRANSFERRED_MAX_BYTES", str(24 << 30)))   # ring budget

    def can_submit_transferred(self, user_tab, item_tab):
        """Whether the ring's six table copies per slot fit the budget (config-4-sized tables keep the in-place order)."""
        if __import__("os").environ.get("SML_EVAL_TRANSFERRED", "1") == "0":
            return False
        per_slot = 3 * (user_tab.numel() + item_tab.numel()) * 4
        return per_slot * self.SNAPSHOTS <= self.TRANSFERRED_MAX_BYTES

    def _side_ctx(self):
        """A second library context for forwards on the side stream: its own MFMA operand images (the training context's
        are being stepped in place by the Adam kernels of the stream this does not wait for)."""
        if getattr(self, "_side_ctx_h", None) is None:
            h = ctypes.c_void_p()
            check(self.lib.sml_ctx_create(ctypes.byref(h), self.device.index or 0, self.d, 1024), "sml_ctx_create")
            self._side_ctx_h, self._side_variant = h, -1
        if self._side_variant != getattr(self, "_variant", 0):
            # (bit 1: an evaluation-stream context -- its table-sized forwards carry their own kernel name and timing class)
            check(self.lib.sml_ctx_set_variant(self._side_ctx_h, getattr(self, "_variant", 0) | 2), "sml_ctx_set_variant")
            self._side_variant = getattr(self, "_variant", 0)
        return self._side_ctx_h

    def eval_submit_transferred(self, transfer, last_user, hat_user, last_item, hat_item, rows):
        """Queue the ranks of `rows` under the tables updata(transfer, last_*, hat_*) WOULD produce now, without producing
        them on the current stream.  Returns a handle like eval_submit; the caller may modify all five inputs right away."""
        theta = self._select(transfer)
        tabs = [self._table(t) for t in (last_user, hat_user, last_item, hat_item)]
        rows = self._dev(rows, torch.int64)
        side = self._side_stream()
        cur = torch.cuda.current_stream(self.device)
        slot = self._snap_slot(cur)
        snaps = [self._snap_buf(slot, k, t) for k, t in zip(("lu", "hu", "li", "hi"), tabs)]
        su, si = self._snap_buf(slot, "u", tabs[0]), self._snap_buf(slot, "i", tabs[2])
        sth = self._snap_buf(slot, "theta", theta)
        self.copy_tables(list(zip(snaps, tabs)))
        sth.copy_(theta)
        self._side_order(cur, side)
        ctx = self._side_ctx()
        with torch.cuda.stream(side):
            for net, (xt, xh, out) in enumerate(((snaps[0], snaps[1], su), (snaps[2], snaps[3], si))):
                check(self.lib.sml_transfer_forward(ctx, _ptr(sth), net, _ptr(xt), _ptr(xh), _ptr(out), xt.shape[0],
                                                    self._stream()), "sml_transfer_forward")
            ranks = self.eval_ranks(su, si, rows, max_workgroups=self._side_eval_cap())
            ev = torch.cuda.Event()
            ev.record(side)
        slot["ev"] = ev
        rows.record_stream(side)
        return dict(ranks=ranks, event=ev, n=rows.shape[0])

    def eval_metrics_submit(self, handle, topk):
        """(hits, ndcg_sum) of a submitted evaluation at `topk`, on the side stream; returns (out, event)."""
        side = self._side_stream()
        with torch.cuda.stream(side):
            out = torch.empty(2, device=self.device, dtype=torch.float32)
            check(self.lib.sml_eval_metrics(self._ctx, _ptr(handle["ranks"]), handle["n"], int(topk), _ptr(out),
                                            self._stream()), "sml_eval_metrics")
            ev = torch.cuda.Event()
            ev.record(side)
        return (out, ev)

    @staticmethod
    def _flag_timeout():
        import os
        return float(os.environ.get("SML_FLAG_TIMEOUT_S", "300"))

    def side_sync_check(self, block=True):
        """Raise if a side-stream evaluation gave up waiting for its table snapshot since the last check (its ranks
        were then computed over a snapshot that may have been incomplete).  Called wherever evaluation results are
        collected OR dropped (run_period without a record, bench.py).  block=True synchronises with the side stream;
        block=False never waits: it looks at the last asynchronous read-back of the counter that has completed and
        queues the next one (a period that drops its results checks the previous period's this way, and the caller
        ends the run with a blocking check).  An incident is reported once: the engine is re-armed, and later
        evaluations were ordered all along (the sequence word is never poisoned)."""
        flag = getattr(self, "_sync_flag", None)
        if flag is None:
            return
        side = self._side_stream()
        if block:
            with torch.cuda.stream(side):          # (read on the side stream: the training stream's backlog is not waited for)
                n = int(flag[1].item())
        else:
            n = self._sync_seen
            pend = getattr(self, "_sync_pending", None)
            if pend is not None and pend[1].query():
                n = int(pend[0][0])
                self._sync_pending = pend = None
            if pend is None:
                host = getattr(self, "_sync_host", None)
                if host is None:
                    host = self._sync_host = torch.zeros(1, dtype=torch.int32).pin_memory()
                with torch.cuda.stream(side):
                    host.copy_(flag[1:2], non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(side)
                self._sync_pending = (host, ev)
        if n > self._sync_seen:
            new, self._sync_seen = n - self._sync_seen, n
            raise RuntimeError("%d side-stream evaluation(s) gave up waiting for their table snapshot (sml_flag_wait "
                               "timeout, SML_FLAG_TIMEOUT_S=%g): their results are not trustworthy" % (new, self._flag_timeout()))

    def eval_result(self, pending, check=True):
        """Wait (host) for an eval_metrics_submit result: (hits, ndcg_sum).  check=False: the caller collects many
        results and calls side_sync_check() once itself (each check is a device read-back)."""
        out, ev = pending[0], pending[1]
        ev.synchronize()
        if check:
            self.side_sync_check()
        h = out.cpu()
        return float(h[0]), float(h[1])

    def eval_metrics_device(self, ranks, topk):
        """(hits, ndcg_sum) as a 2-float DEVICE tensor: no synchronisation."""
        out = torch.empty(2, device=self.device, dtype=torch.float32)
        check(self.lib.sml_eval_metrics(self._ctx, _ptr(ranks), ranks.shape[0], int(topk), _ptr(out), self._stream()),
              "sml_eval_metrics")
        return out

    def eval_metrics(self, ranks, topk):
        h = self.eval_metrics_device(ranks, topk).cpu()
        return float(h[0]), float(h[1])

    # ------------------------------------------------------------------ native RCCL exchange
    def comm_init(self, dist, group=None):
        """Create the library's own RCCL communicator over the ranks of `group` (the unique id travels
        through torch.distributed) and verify it with a small all-reduce and all-gather.
        Returns True when the native exchange is usable on this rank."""
        import os
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        if not os.path.exists(path):
            path = "librccl.so"
        loaded = self.lib.sml_comm_load(path.encode()) == 0
        # (a vote BEFORE the first collective of this function: a rank that cannot bind RCCL must not leave the others
        # inside the broadcast below)
        flag = torch.tensor([1.0 if loaded else 0.0], device=self.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        if float(flag.item()) < 0.5:
            return False
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        buf = ctypes.create_string_buffer(128)
        if rank == 0:
            check(self.lib.sml_comm_unique_id(buf), "sml_comm_unique_id")
        box = [bytes(buf.raw)]
        dist.broadcast_object_list(box, src=0, group=group)
        idb = ctypes.create_string_buffer(box[0], 128)
        check(self.lib.sml_comm_init(self._ctx, world, rank, idb), "sml_comm_init")
        t = torch.full((8,), float(rank + 1), device=self.device)
        check(self.lib.sml_comm_allreduce(self._ctx, _ptr(t), 8, self._stream()), "sml_comm_allreduce")
        src = torch.tensor([float(rank), rank + 0.5], device=self.device)
        dst = torch.empty(2 * world, device=self.device)
        check(self.lib.sml_comm_allgather(self._ctx, _ptr(src), _ptr(dst), 2, self._stream()), "sml_comm_allgather")
        want = torch.tensor([v for r in range(world) for v in (float(r), r + 0.5)], device=self.device)
        return bool(torch.all(t == world * (world + 1) / 2.0)) and bool(torch.equal(dst, want))

    def comm_destroy(self):
        self.lib.sml_comm_destroy(self._ctx)

    # ------------------------------------------------------------------ one-shot exchange over peer mappings
    def peer_region_bytes(self, world, rows_cap):
        a, b = ctypes.c_int64(0), ctypes.c_int64(0)
        check(self.lib.sml_peer_region_bytes(self._ctx, int(world), int(rows_cap), ctypes.byref(a), ctypes.byref(b)),
              "sml_peer_region_bytes")
        return a.value, b.value

    def peer_alloc(self, nbytes):
        """Zeroed device memory for an inbox / flags region (uncached or fine-grained: see sml_peer_alloc); an int address."""
        p = ctypes.c_void_p()
        check(self.lib.sml_peer_alloc(self.device.index, int(nbytes), ctypes.byref(p)), "sml_peer_alloc")
        self.__dict__.setdefault("_peer_owned", []).append(p.value)
        return p.value

    def peer_mem_kind(self, ptr):
        """0 uncached, 1 fine-grained, 2 plain device memory, -1 not a peer_alloc allocation."""
        return int(self.lib.sml_peer_mem_kind(ctypes.c_void_p(int(ptr))))

    def peer_free(self, ptr):
        """Free a peer_alloc region (only once every rank has detached and unmapped it)."""
        check(self.lib.sml_peer_free(self.device.index, ctypes.c_void_p(int(ptr))), "sml_peer_free")
        owned = self.__dict__.get("_peer_owned", [])
        if ptr in owned:
            owned.remove(ptr)

    def peer_close(self, ptr):
        check(self.lib.sml_peer_close(self.device.index, ctypes.c_void_p(int(ptr))), "sml_peer_close")
        opened = self.__dict__.get("_peer_opened", [])
        if ptr in opened:
            opened.remove(ptr)

    def peer_read(self, ptr, n_floats):
        """n_floats fp32 values at device address `ptr` (possibly another rank's memory), read with system-scope loads."""
        out = torch.empty(int(n_floats), device=self.device, dtype=torch.float32)
        check(self.lib.sml_peer_read(self.device.index, ctypes.c_void_p(int(ptr)), _ptr(out), int(n_floats) * 4, self._stream()),
              "sml_peer_read")
        return out

    def peer_tensor(self, shape, dtype=torch.float32):
        """A zeroed torch tensor over memory of its OWN device allocation (plain device memory, sml_peer_alloc): unlike a
        tensor from torch's caching allocator -- a slice of a larger segment -- its address can be exported whole with
        peer_export and opened by another process (an item-table shard other ranks read over the peer mapping)."""
        import os
        shape = tuple(int(v) for v in shape)
        n = 1
        for v in shape:
            n *= v
        nbytes = max(n * torch.empty((), dtype=dtype).element_size(), 16)
        old = os.environ.get("SML_PEER_MEM")
        os.environ["SML_PEER_MEM"] = "plain"
        try:
            ptr = self.peer_alloc(nbytes)
        finally:
            if old is None:
                del os.environ["SML_PEER_MEM"]
            else:
                os.environ["SML_PEER_MEM"] = old
        typestr = {torch.float32: "<f4", torch.float16: "<f2", torch.int32: "<i4", torch.int64: "<i8"}[dtype]

        class _Raw(object):
            __cuda_array_interface__ = {"shape": shape, "typestr": typestr, "data": (ptr, False), "version": 2, "strides": None}
        t = torch.as_tensor(_Raw(), device=self.device)
        t._sml_peer_ptr = ptr
        return t

    def peer_tensor_free(self, t):
        """Free a peer_tensor's allocation (the tensor must not be used afterwards; every rank must have unmapped it)."""
        ptr = getattr(t, "_sml_peer_ptr", None)
        if ptr is not None:
            torch.cuda.synchronize(self.device)
            t._sml_peer_ptr = None
            self.peer_free(ptr)

    def peer_export(self, ptr):
        buf = ctypes.create_string_buffer(64)
        check(self.lib.sml_peer_export(ctypes.c_void_p(ptr), buf), "sml_peer_export")
        return bytes(buf.raw)

    def peer_open(self, handle):
        p = ctypes.c_void_p()
        check(self.lib.sml_peer_open(self.device.index, ctypes.create_string_buffer(handle, 64), ctypes.byref(p)), "sml_peer_open")
        self.__dict__.setdefault("_peer_opened", []).append(p.value)
        return p.value

    def peer_attach(self, world, rank, inbox, flags, rows_cap, timeout_s=60.0):
        """inbox / flags: every rank's regions as addresses valid on THIS device (own allocations, sml_peer_open results,
        or -- several ranks in one process -- the other ranks' allocations)."""
        n = int(world)
        a, b = (ctypes.c_void_p * n)(*[int(x) for x in inbox]), (ctypes.c_void_p * n)(*[int(x) for x in flags])
        check(self.lib.sml_peer_attach(self._ctx, n, int(rank), a, b, int(rows_cap), float(timeout_s)), "sml_peer_attach")
        self.peer_world, self.peer_rows_cap = n, int(rows_cap)

    def peer_detach(self):
        self.lib.sml_peer_detach(self._ctx)
        self.peer_world = 0

    def peer_status(self):
        """Consumers that gave up waiting for a peer's push since attach (synchronous read)."""
        n = ctypes.c_int(0)
        check(self.lib.sml_peer_status(self._ctx, ctypes.byref(n)), "sml_peer_status")
        return n.value

    def peer_allreduce_check(self, src, timeout_s=0.0):
        dst = torch.empty_like(src)
        check(self.lib.sml_peer_allreduce_check(self._ctx, _ptr(src), _ptr(dst), src.numel(), float(timeout_s), self._stream()),
              "sml_peer_allreduce_check")
        return dst

    # ------------------------------------------------------------------ measurement
    def _prof_ctxs(self):
        """The library contexts whose launches are bracketed: the engine's own and, once it exists, the side stream's
        (eval_submit_transferred runs its table-sized forwards through it, on the evaluation stream)."""
        side = getattr(self, "_side_ctx_h", None)
        return [self._ctx] + ([side] if side else [])

    def profile(self, on):
        if on and hasattr(self, "_side_stream") and getattr(self, "_side_ctx_h", None) is None and self.device.type == "cuda":
            self._side_ctx()                       # (so that its launches are bracketed from their first one)
        for c in self._prof_ctxs():
            check(self.lib.sml_prof_enable(c, int(bool(on))), "sml_prof_enable")
            if on:
                check(self.lib.sml_prof_reset(c), "sml_prof_reset")

    def profile_pair_overhead(self, n=512):
        """Microseconds an EMPTY HIP-event pair reads on the current stream (subtract it from short kernels' averages)."""
        us = ctypes.c_double(0.0)
        check(self.lib.sml_prof_pair_overhead(self._ctx, int(n), self._stream(), ctypes.byref(us)), "sml_prof_pair_overhead")
        return us.value

    def profile_read(self, which="all"):
        """{kernel class: (launches, total ms)} measured with HIP events since profile(True).  which: "all" (both contexts,
        summed), "main" (the engine's own: the training stream) or "side" (the evaluation stream's forwards)."""
        out = {}
        ctxs = self._prof_ctxs()
        ctxs = ctxs[:1] if which == "main" else ctxs[1:] if which == "side" else ctxs
        for ctx in ctxs:
            for c in range(self.lib.sml_prof_classes()):
                cnt, ms = ctypes.c_int64(0), ctypes.c_double(0.0)
                check(self.lib.sml_prof_get(ctx, c, ctypes.byref(cnt), ctypes.byref(ms)), "sml_prof_get")
                if cnt.value:
                    name = self.lib.sml_prof_name(c).decode()
                    have = out.get(name, (0, 0.0))
                    out[name] = (have[0] + cnt.value, have[1] + ms.value)
        return out

    def selftest(self):
        check(self.lib.sml_selftest(self.device.index), "sml_selftest")


_engines = {}


def get_engine(device, d, max_batch=4096):
    """Process-wide engine for (device, d)."""
    dev = torch.device(device)
    if dev.type == "cuda" and dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    key = (str(dev), int(d))
    eng = _engines.get(key)
    if eng is None or eng.max_batch < max_batch:
        eng_new = HipEngine(dev, d, max(max_batch, eng.max_batch if eng else 0))
        if eng is not None:   # keep optimiser state and adopted thetas across a scratch resize
            for k in ("_flat", "mf_state", "mf_step", "tr_state", "tr_step", "grad_hook"):
                setattr(eng_new, k, getattr(eng, k))
            if hasattr(eng, "_mf_lr"):
                eng_new._mf_lr = eng._mf_lr
        _engines[key] = eng = eng_new
    return eng
