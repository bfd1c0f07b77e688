#!/usr/bin/env python3
"""Wall time per a3 batch with the index lists prepared beforehand (10M x 1M, d=32, batch 262,144)."""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sml_amd.engine import HipEngine
dev = torch.device("cuda", 0)
d, U, I, B, nb = 32, 10_000_000, 1_000_000, 262144, 16
eng = HipEngine(dev, d, B)
g = torch.Generator(device=dev); g.manual_seed(1)
wu = torch.randn(U, d, device=dev) * 0.1; wi = torch.randn(I, d, device=dev) * 0.1
n = B * nb
tri = torch.stack([torch.randint(0, U, (n,), device=dev, generator=g), torch.randint(0, I, (n,), device=dev, generator=g),
                   torch.randint(0, I, (n,), device=dev, generator=g)], 1).contiguous()
out = []
for rep in range(4):
    h = eng.bare_prepare(tri, B, U, I)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); eng.bare_epoch(wu, wi, tri, B, 0.01, 1e-4, 1e-4, prepared=h); e1.record()
    torch.cuda.synchronize()
    out.append(round(e0.elapsed_time(e1) * 1000 / nb, 2))
print(json.dumps({"us_per_batch": out, "frac_of_8TBps": round(792 * B / (min(out[1:]) * 1e-6) / 8e12, 4)}))
