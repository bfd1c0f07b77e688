"""Transfer net: the module surface of the reference's model/conv_transfer.py.

`one_transfer` and `ConvTransfer_com` keep the reference's constructor signatures,
sub-module names (conv1, conv2, fc1, fc2; user_transfer then item_transfer -- the
creation order fixes the RNG draw sequence and state_dict keys) and parameter
shapes, so state_dicts are interchangeable.  The arithmetic runs in libsml_hip.so:
the engine re-points the parameters into one flat device buffer (HipEngine.adopt)
and the kernels read it directly.
"""
import torch
import torch.nn as nn

from .engine import get_engine


def Gelu(x):
    """x * sigmoid(1.702 x) -- reference model/conv_transfer.py:9-10 (not erf GELU)."""
    return x * torch.sigmoid(1.702 * x)


class _RunMF(torch.autograd.Function):
    """run_MF as ONE differentiable op on the HIP engine: forward = the loss of the batch, backward = the hand-derived
    gradients the kernels compute (d loss / d x_hat for the three row blocks and d loss / d theta), scaled by the
    incoming gradient.  x_t (the W_{t-1} rows) gets no gradient: every caller in the reference passes plain tensors
    there (model/transfer.py:469-471, 707-709), and x_com is built from a detached x_hat (model/conv_transfer.py:93-99)."""

    @staticmethod
    def forward(ctx, module, norm, bce, ul, uh, il, ih, nl, nh, *params):
        eng = module._engine(uh)
        need_rows = uh.requires_grad or ih.requires_grad or nh.requires_grad
        need_theta = any(p.requires_grad for p in params)
        B = uh.shape[0]
        f32 = lambda t: t.detach().to(torch.float32).contiguous()
        loss, du, di, gt = eng.run_mf_grad(module, f32(ul), f32(uh), torch.cat([f32(il), f32(nl)]), torch.cat([f32(ih), f32(nh)]),
                                           bce=bce, norm=norm, want_rows=need_rows, want_theta=need_theta)
        ctx.B, ctx.du, ctx.di, ctx.gt = B, du, di, gt
        ctx.views = eng.theta_views(module) if need_theta else None
        ctx.n_params = len(params)
        ctx.param_ids = [id(p) for p in params]
        return loss

    @staticmethod
    def backward(ctx, g):
        B = ctx.B
        g_uh = g * ctx.du if ctx.du is not None else None
        g_ih = g * ctx.di[:B] if ctx.di is not None else None
        g_nh = g * ctx.di[B:] if ctx.di is not None else None
        gp = [None] * ctx.n_params
        if ctx.gt is not None:
            by_id = {id(p): (off, n, p) for p, off, n in ctx.views}
            for k, pid in enumerate(ctx.param_ids):
                ent = by_id.get(pid)
                if ent is None:
                    continue
                off, n, p = ent
                if p.dim() == 4 and p.shape[2] == 2 and p.shape[0] == 10:      # ConvTransfer's (2,1) conv1 kernel: see HipEngine.adopt
                    gp[k] = g * ctx.gt[off:off + 30].view(10, 3)[:, :2].reshape(p.shape)
                else:
                    gp[k] = g * ctx.gt[off:off + n].view(p.shape)
        return (None, None, None, None, g_uh, None, g_ih, None, g_nh) + tuple(gp)


def _needs_graph(module, *tensors):
    return torch.is_grad_enabled() and (any(t.requires_grad for t in tensors) or any(p.requires_grad for p in module.parameters()))


class one_transfer(nn.Module):
    """conv1 (1->10, kernel (k,1)) -> Gelu -> conv2 (10->5, 1x1) -> flatten -> Gelu ->
    fc1 (5d->512) -> Gelu -> fc2 (512->out).  reference model/conv_transfer.py:18-50.
    A parameter container: the forward pass is ConvTransfer_com's, on the HIP engine."""

    def __init__(self, input_dim, out_dim, kernel=2):
        super(one_transfer, self).__init__()
        self.hidden_dim = input_dim
        self.out_channel = 10
        self.conv1 = nn.Conv2d(1, self.out_channel, (kernel, 1), stride=1)
        self.out_channel2 = 5
        self.conv2 = nn.Conv2d(self.out_channel, self.out_channel2, (1, 1), stride=1)
        self.fc1 = nn.Linear(input_dim * self.out_channel2, 512)
        self.fc2 = nn.Linear(512, out_dim)
        self.kernel = kernel
        print("kernel:", kernel)

    def forward(self, x):
        raise RuntimeError("one_transfer is evaluated through ConvTransfer_com (HIP engine), not stand-alone")


class ConvTransfer_com(nn.Module):
    """reference model/conv_transfer.py:87-135."""

    def __init__(self, in_dim, out_dim):
        super(ConvTransfer_com, self).__init__()
        if in_dim != out_dim:
            raise ValueError("ConvTransfer_com: in_dim must equal out_dim (the reference always passes the same)")
        self.user_transfer = one_transfer(in_dim, out_dim, kernel=3)
        self.item_transfer = one_transfer(in_dim, out_dim, kernel=3)
        self.dim = in_dim

    def _engine(self, like):
        if like.device.type != "cuda":
            raise RuntimeError("ConvTransfer_com runs on the HIP engine only: inputs must be on a GPU; "
                               "there is no CPU path")
        eng = getattr(self, "_sml_engine", None)
        return eng if eng is not None else get_engine(like.device, self.dim)

    def forward(self, x_t, x_hat, type):
        """x_com = x_t * detach(x_hat) / ||x_t||; stack (x_t, x_hat, x_com); run the user or
        item net.  Inference only (no autograd graph)."""
        if type not in ("user", "item"):
            raise TypeError("convtransfer has not this type")
        return self._engine(x_t).transfer_forward(self, x_t, x_hat, type)

    def run_MF(self, user_weight_last, user_weight_hat, item_weight_last, item_weight_hat, negitem_weight_last,
               negitem_weight_hat, norm=False, adpative=False, BCE=True):
        """Loss of one batch (reference model/conv_transfer.py:113-135): BCE by default, BPR when BCE=False (optionally
        with score / ||u'||).  When gradients are enabled and any x_hat block or any parameter of this module
        requires grad, the result carries a graph: loss.backward() fills x_hat.grad (through nn.Embedding, if the rows
        came from one) and the parameters' .grad, exactly as the reference's loops expect (model/transfer.py:476-502,
        714-723: zero_grad -> run_MF -> backward -> optimizer.step()).  The whole-epoch engine calls
        (HipEngine.*_stage_epoch) remain the fast path; this is the drop-in one."""
        # adpative: the reference's branch is `pass` (model/conv_transfer.py:131-132) -- accepted, changes nothing
        if _needs_graph(self, user_weight_hat, item_weight_hat, negitem_weight_hat):
            return _RunMF.apply(self, bool(norm), bool(BCE), user_weight_last, user_weight_hat, item_weight_last, item_weight_hat,
                                negitem_weight_last, negitem_weight_hat, *list(self.parameters()))
        un = self.forward(user_weight_last, user_weight_hat, "user")
        im = self.forward(item_weight_last, item_weight_hat, "item")
        nn_ = self.forward(negitem_weight_last, negitem_weight_hat, "item")
        s_pos = (un * im).sum(-1)
        s_neg = (un * nn_).sum(-1)
        if BCE:
            return -torch.mean(torch.log(torch.sigmoid(s_pos) + 1e-15)) \
                   - torch.mean(torch.log(1 - torch.sigmoid(s_neg) + 1e-15))
        score = s_pos - s_neg
        if norm:
            score = score / (un ** 2).sum(-1).sqrt()
        return -torch.sum(torch.nn.functional.logsigmoid(score))


class ConvTransfer(ConvTransfer_com):
    """The reference's first convolutional transfer (model/conv_transfer.py:52-85; `--transfer_type conv`):
    nets with a (2,1) first kernel over the stack (x_t, x_hat) -- no x_com row -- the USER output divided by
    its detached norm, and a BPR sum loss.  Same engine, same kernels: the x_com row is fed as zeros against
    a zero third kernel column (whose gradient is then exactly zero), the normalisation sits in the pair-loss
    stage of the backward kernel and in the epilogue of the table-sized forward."""

    def __init__(self, in_dim, out_dim):
        nn.Module.__init__(self)
        if in_dim != out_dim:
            raise ValueError("ConvTransfer: in_dim must equal out_dim (the reference always passes the same)")
        self.user_transfer = one_transfer(in_dim, out_dim, kernel=2)
        self.item_transfer = one_transfer(in_dim, out_dim, kernel=2)
        self.dim = in_dim

    def run_MF(self, user_weight_last, user_weight_hat, item_weight_last, item_weight_hat, negitem_weight_last,
               negitem_weight_hat, norm=False):
        """BPR loss of one batch (model/conv_transfer.py:71-85); differentiable like ConvTransfer_com.run_MF."""
        if _needs_graph(self, user_weight_hat, item_weight_hat, negitem_weight_hat):
            return _RunMF.apply(self, bool(norm), False, user_weight_last, user_weight_hat, item_weight_last, item_weight_hat,
                                negitem_weight_last, negitem_weight_hat, *list(self.parameters()))
        un = self.forward(user_weight_last, user_weight_hat, "user")        # already unit norm
        im = self.forward(item_weight_last, item_weight_hat, "item")
        nn_ = self.forward(negitem_weight_last, negitem_weight_hat, "item")
        score = (un * im).sum(-1) - (un * nn_).sum(-1)
        if norm:
            score = score / (un ** 2).sum(-1).sqrt()
        return -torch.sum(torch.nn.functional.logsigmoid(score))


def _out_of_scope(name):
    class _Stub(nn.Module):
        def __init__(self, *a, **k):
            raise NotImplementedError(
                "%s is one of the reference's unused transfer variants (model/conv_transfer.py header: "
                "'we only use ConvTransfer_com and one_transfer'); it is outside this build's scope" % name)
    _Stub.__name__ = name
    return _Stub


ConvTransfer_com2 = _out_of_scope("ConvTransfer_com2")
ConvTransfer_com3 = _out_of_scope("ConvTransfer_com3")
one_transfer_com = _out_of_scope("one_transfer_com")
