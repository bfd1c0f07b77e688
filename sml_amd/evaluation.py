"""recall@K / ndcg@K over (user, positive, negatives) rows -- reference
evalution/evaluation2.py:8-26 on the HIP rank kernel."""
import numpy as np
import torch

from .datasets import loader_base_seed_draw


class DeviceRows(object):
    """A whole test set resident on the GPU, standing in for the reference's
    DataLoader(testDataset(rows), batch_size=1024): one kernel launch ranks every row.
    Iterating it makes the one global-RNG draw a DataLoader iteration makes, so a run's
    random stream stays aligned with the reference's."""

    def __init__(self, rows, device, stream=None):
        if isinstance(rows, np.ndarray):
            rows = torch.from_numpy(np.ascontiguousarray(rows))
        self.ready = None
        if stream is None:
            self.rows = rows.to(device=device, dtype=torch.int64).contiguous()
        else:
            # upload on a stream of its own (a copy on the default stream would first wait for every queued kernel);
            # the first consumer orders itself behind `ready` (wait_ready)
            with torch.cuda.stream(stream):
                self.rows = rows.to(device=device, dtype=torch.int64).contiguous()
                self.ready = torch.cuda.Event()
                self.ready.record(stream)

    def wait_ready(self):
        """Order the current stream behind the upload (once)."""
        if self.ready is not None:
            torch.cuda.current_stream(self.rows.device).wait_event(self.ready)
            self.ready = None

    def __len__(self):
        return 1

    def __iter__(self):
        loader_base_seed_draw()
        yield self.rows


def test_model(model, test_set, old_user=None, old_item=None, topK=10, need_pbar=False):
    """-> (recall@topK, ndcg@topK) = (hits, sum 1/log2(rank+2)) / number of rows."""
    model.eval()
    device = model.user_laten.weight.device
    num_test = 0
    hits = 0.0
    ndcg = 0.0
    for datas in test_set:
        datas = torch.as_tensor(datas).long().to(device)
        batch_hit, batch_ndcg, _ = model.test(datas, topK=topK)
        hits += batch_hit
        ndcg += float(batch_ndcg)
        num_test += datas.shape[0]
    return hits / num_test, torch.tensor(np.float32(ndcg / num_test))
