"""MF base model: the module surface of the reference's model/MF.py on the HIP engine.

Same class names, constructor signature, attribute names (misspellings included:
user_bais, item_bais, user_laten, item_laten) and creation order as the reference
(model/MF.py:18-27) -- the order fixes state_dict keys and the RNG draw sequence of
the initialisers, and the class path `model.MF.MFbasemode` is the on-disk contract
of the --pre_model checkpoint (a whole pickled module, model/transfer.py:322-325).

forward / test run on libsml_hip.so (no autograd through them: SML trains the
tables through HipEngine.mf_stage_epoch, not through this module's graph).
"""
import torch
import torch.nn as nn

from .engine import get_engine


def _engine_for(module):
    w = module.user_laten.weight
    if w.device.type != "cuda":
        raise RuntimeError("%s runs on the HIP engine only: move it to a GPU (.cuda()); there is no CPU path"
                           % type(module).__name__)
    eng = getattr(module, "_sml_engine", None)
    return eng if eng is not None else get_engine(w.device, module.hidden_dim)


class MFbasemode(nn.Module):
    def __init__(self, num_user=0, num_item=0, laten_factor=10):
        super(MFbasemode, self).__init__()
        # creation order matters (reference model/MF.py:21-24)
        self.user_bais = nn.Embedding(num_user, 1)
        self.item_bais = nn.Embedding(num_item, 1)
        self.user_laten = nn.Embedding(num_user, laten_factor)
        self.item_laten = nn.Embedding(num_item, laten_factor)
        self.user_num = num_user
        self.item_num = num_item
        self.hidden_dim = laten_factor

    def reset_parameters(self):
        # same order as the reference (model/MF.py:28-32)
        for emb in (self.user_bais, self.user_laten, self.item_bais, self.item_laten):
            emb.reset_parameters()

    def forward(self, user, item, norm=False):
        """(user rows, item rows, dot products [/ ||u|| if norm]).  model/MF.py:34-43."""
        eng = _engine_for(self)
        return eng.mf_forward(self.user_laten.weight.data, self.item_laten.weight.data, user, item, norm)

    def _ranks(self, inputs_data):
        eng = _engine_for(self)
        return eng, eng.eval_ranks(self.user_laten.weight.data, self.item_laten.weight.data, inputs_data)

    def test(self, inputs_data, topK=20):
        """inputs_data [n, 2+neg]: user, positive, negatives.  Returns (hits, ndcg_sum,
        indices of the rows that hit).  model/MF.py:45-80.  The positive's rank is the
        count of candidates scoring strictly above it, which is the position torch.topk
        gives it when scores are tie-free."""
        eng, ranks = self._ranks(inputs_data)
        hits, ndcg = eng.eval_metrics(ranks, topK)
        hit_rows = (ranks < topK).nonzero()[:, 0]
        batch_ndcg = torch.tensor(ndcg) if hits > 0 else 0
        return hits * 1.0, batch_ndcg, hit_rows

    def test2(self, inputs_data, topK=20):
        """model/MF.py:82-106: (hit rows, rank, hits, ndcg_sum); `rank` here is the
        positive's rank per row (the reference returns the top-k index matrix)."""
        eng, ranks = self._ranks(inputs_data)
        hits, ndcg = eng.eval_metrics(ranks, topK)
        hit_rows = (ranks < topK).nonzero()[:, 0]
        return hit_rows, ranks, hits * 1.0, (torch.tensor(ndcg) if hits > 0 else 0)

    def set_parameters(self, user_weight, item_weight):
        # last column is the bias (model/MF.py:108-112)
        self.user_laten.weight.data.copy_(user_weight[:, 0:-1])
        self.user_bais.weight.data.copy_(user_weight[:, -1].unsqueeze(-1))
        self.item_laten.weight.data.copy_(item_weight[:, 0:-1])
        self.item_bais.weight.data.copy_(item_weight[:, -1].unsqueeze(-1))


class _MF2Train(torch.autograd.Function):
    """MF2.forward's training branch (reference model/MF.py:129-147) as ONE differentiable op: the row gathers and the two
    dot products run on the HIP engine (k_mf_forward), the bias terms, -sum(logsigmoid) and the reference's "l2" (row NORMS,
    the negatives' as one Frobenius norm) are a handful of elementwise device ops, and backward scatters the hand-derived
    gradient rows into dense table gradients -- what autograd hands nn.Embedding(sparse=False)."""

    @staticmethod
    def forward(ctx, module, user, item, neg, wu, wi, bu, bi):
        eng = _engine_for(module)
        dev = wu.device
        user, item, neg = (x.to(dev).long() for x in (user, item, neg))
        ue, ie, sp = eng.mf_forward(wu.detach(), wi.detach(), user, item)
        _, ne, sn = eng.mf_forward(wu.detach(), wi.detach(), user, neg)
        ub, ib, nb = bu.detach()[user, 0], bi.detach()[item, 0], bi.detach()[neg, 0]
        score = (ub + ib + sp) - (ub + nb + sn)             # result_pos - result_neg, formed as the reference forms it
        bpr = -torch.sum(torch.nn.functional.logsigmoid(score))
        nu, ni, nn_ = ue.norm(dim=-1), ie.norm(dim=-1), ne.norm()
        l2 = nu.sum() + ni.sum() + nn_.sum()
        ctx.save_for_backward(user, item, neg, ue, ie, ne, score, nu, ni, nn_)
        ctx.shapes = (wu.shape, wi.shape, bu.shape, bi.shape)
        return bpr, l2

    @staticmethod
    def backward(ctx, g_bpr, g_l2):
        user, item, neg, ue, ie, ne, score, nu, ni, nn_ = ctx.saved_tensors
        ds = (-g_bpr * torch.sigmoid(-score)).unsqueeze(-1)               # d bpr / d score
        du = ds * (ie - ne) + g_l2 * ue / nu.unsqueeze(-1)
        di = ds * ue + g_l2 * ie / ni.unsqueeze(-1)
        dn = -ds * ue + g_l2 * ne / nn_
        su, si, sbu, sbi = ctx.shapes
        gwu = torch.zeros(su, device=ue.device, dtype=ue.dtype).index_add_(0, user, du)
        gwi = torch.zeros(si, device=ue.device, dtype=ue.dtype).index_add_(0, item, di).index_add_(0, neg, dn)
        gbu = torch.zeros(sbu, device=ue.device, dtype=ue.dtype)          # the user bias cancels in result_pos - result_neg
        gbi = torch.zeros(sbi, device=ue.device, dtype=ue.dtype).index_add_(0, item, ds).index_add_(0, neg, -ds)
        return None, None, None, None, gwu, gwi, gbu, gbi


class MF2(MFbasemode):
    """model/MF.py:118-156.  Test branch: the dot product plus both biases.  Training branch (`neg_item` given): the
    reference's (bpr_loss, l2loss) pair, differentiable w.r.t. the four embedding tables (`_MF2Train`).  Nothing in the
    reference calls this class; the bare BPR step at table scale is HipEngine.bare_epoch(bce=False)."""

    def forward(self, user, item, neg_item=None):
        if neg_item is not None:
            return _MF2Train.apply(self, user, item, neg_item, self.user_laten.weight, self.item_laten.weight,
                                   self.user_bais.weight, self.item_bais.weight)
        ue, ie, s = MFbasemode.forward(self, user, item)
        user = user.to(s.device).long()
        item = item.to(s.device).long()
        s = s + self.user_bais.weight.data[user, 0] + self.item_bais.weight.data[item, 0]
        return ue, ie, s


# checkpoints written by this build name the reference's module path, and vice versa
MFbasemode.__module__ = "model.MF"
MF2.__module__ = "model.MF"
