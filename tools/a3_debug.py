#!/usr/bin/env python3
"""Localise a3 (fused bare step) mismatches against the oracle: which table, which rows, what run lengths."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import sml_oracle as O
from sml_amd.engine import HipEngine
DEV = "cuda:0"
torch.manual_seed(33)
d, U, I, B, nb = 32, 200, 150, 96, int(os.environ.get("NB", "3"))
wu, wi = torch.randn(U, d) * 0.3, torch.randn(I, d) * 0.3
u = torch.randint(0, U, (B * nb,)); u[:20] = 7
i = torch.randint(0, I, (B * nb,)); j = torch.randint(0, I, (B * nb,))
i[5] = j[5]; j[30:40] = i[0]
tri = torch.stack([u, i, j], 1)
n = B * nb - 11
eng = HipEngine(DEV, d, 4096)
for rep in range(int(os.environ.get("REPS", "2"))):
    gu, gi = wu.clone().to(DEV), wi.clone().to(DEV)
    losses = eng.bare_epoch(gu, gi, tri[:n], B, 0.05, 1e-3, 2e-3, bce=True).cpu().numpy()
    ou, oi = wu.clone(), wi.clone()
    want = []
    for b0 in range(0, n, B):
        t = tri[b0:min(b0 + B, n)]
        want.append(O.bare_step(ou, oi, t[:, 0], t[:, 1], t[:, 2], 0.05, 1e-3, 2e-3, bce=True))
    print("rep", rep, "loss rel err", np.abs(losses - np.array(want)).max() / np.abs(want).max())
    for name, g, o, col in (("user", gu.cpu(), ou, [0]), ("item", gi.cpu(), oi, [1, 2])):
        err = (g - o).abs().max(1).values
        bad = (err > 1e-5).nonzero()[:, 0].tolist()
        print(name, "bad rows:", len(bad))
        for r in bad[:12]:
            occ = [(b, int(((tri[b * B:min((b + 1) * B, n)][:, col] == r).sum()))) for b in range(nb)]
            print("   row", r, "err %.3e" % float(err[r]), "occurrences per batch", occ)
