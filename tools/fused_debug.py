"""A/B of the MF stage's fused row update (SML_MF_FUSED_UPDATE=0/1) after ONE epoch of nb batches: which rows differ, and are they
rows that occur once or several times in their batch."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import make_mf, make_transfer  # noqa: E402
from sml_amd.engine import HipEngine  # noqa: E402

DEV = torch.device("cuda:0")
d, U, I, B = 32, 6000, 4000, 1024
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1
rng = np.random.RandomState(3)
n = nb * B
tri = np.stack([rng.randint(0, U, n), rng.randint(0, I, n), rng.randint(0, I, n)], 1)
wu0, wi0 = rng.randn(U, d).astype(np.float32) * 0.3, rng.randn(I, d).astype(np.float32) * 0.3
outs = []
for fused in ("0", "1"):
    os.environ["SML_MF_FUSED_UPDATE"] = fused
    eng = HipEngine(DEV, d, B)
    mf = make_mf(U, I, d, wu0, wi0, device=DEV)
    torch.manual_seed(11)
    net = make_transfer(d, device=DEV)
    lu, li = torch.from_numpy(wu0 * 0.9).to(DEV), torch.from_numpy(wi0 * 0.9).to(DEV)
    l = eng.mf_stage_epoch(mf, net, lu, li, torch.from_numpy(tri), B, 0.01, 1e-6).cpu().numpy()
    torch.cuda.synchronize()
    st = {k: v.cpu().numpy() for k, v in eng.mf_state.items()}
    outs.append((l, mf.user_laten.weight.detach().cpu().numpy(), mf.item_laten.weight.detach().cpu().numpy(), st))
    eng.close()
a, b = outs
print("losses", a[0], b[0])
last = tri[-B:]
for name, x, y, rows in (("user", a[1], b[1], last[:, 0]), ("item", a[2], b[2], np.concatenate([last[:, 1], last[:, 2]]))):
    bad = np.flatnonzero((x != y).any(1))
    cnt = np.bincount(rows, minlength=x.shape[0])
    print(name, "rows differing", bad.size, "of which occur in the last batch once / several times / never:",
          int((cnt[bad] == 1).sum()), int((cnt[bad] > 1).sum()), int((cnt[bad] == 0).sum()))
    if bad.size:
        r = bad[0]
        print("  first bad row", r, "count", cnt[r], "max abs diff", np.abs(x[r] - y[r]).max(), x[r][:4], y[r][:4])
for k in a[3]:
    print(k, "differs in", int((a[3][k] != b[3][k]).sum()), "entries")
