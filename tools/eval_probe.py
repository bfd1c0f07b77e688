#!/usr/bin/env python3
"""Time the three evaluation kernels on the Yelp-shaped validation set (75k rows x 1001 candidates, 123k items, d=32)."""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sml_amd.engine import HipEngine
from sml_amd import synth
dev = torch.device("cuda", 0)
d = int(os.environ.get("D", "32")); U, I, n, neg = 60000, 123000, 75000, 999
eng = HipEngine(dev, d, 1024)
rng = np.random.RandomState(1)
_, test = synth.sample_period(rng, n, U, I, neg=neg)
rows = torch.from_numpy(test).to(dev)
wu, wi = torch.randn(U, d, device=dev) * 0.1, torch.randn(I, d, device=dev) * 0.1
out = {}
ref = None
for mode in (False, True):
    r = eng.eval_ranks(wu, wi, rows, blocked=mode)
    torch.cuda.synchronize()
    if ref is None: ref = r
    assert torch.equal(ref, r), mode
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): eng.eval_ranks(wu, wi, rows, blocked=mode)
    e1.record(); torch.cuda.synchronize()
    out[str(mode)] = round(e0.elapsed_time(e1) / 10 * 1000, 1)
for cap in (64, 128, 256, 512, 1024, 2048):
    eng.eval_ranks(wu, wi, rows, blocked=True, max_workgroups=cap)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): r = eng.eval_ranks(wu, wi, rows, blocked=True, max_workgroups=cap)
    e1.record(); torch.cuda.synchronize()
    assert torch.equal(ref, r), cap
    out["cap%d" % cap] = round(e0.elapsed_time(e1) / 5 * 1000, 1)
print(json.dumps({"d": d, "us_per_eval": out}))
