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


class MF2(MFbasemode):
    """model/MF.py:118-156.  Test-time forward adds the biases to the dot product; the
    BPR training branch of the reference is served by HipEngine.bare_epoch (loss
    kind BPR) -- this module does not build an autograd graph."""

    def forward(self, user, item, neg_item=None):
        if neg_item is not None:
            raise RuntimeError("MF2 training runs through HipEngine.bare_epoch(bce=False); "
                               "the module's forward is inference-only")
        ue, ie, s = MFbasemode.forward(self, user, item)
        user = user.to(s.device).long()
        item = item.to(s.device).long()
        s = s + self.user_bais.weight.data[user, 0] + self.item_bais.weight.data[item, 0]
        return ue, ie, s


# checkpoints written by this build name the reference's module path, and vice versa
MFbasemode.__module__ = "model.MF"
MF2.__module__ = "model.MF"
