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
    r = eng.eval_ranks(wu, wi, rows, blocked=mode, sliced=False)
    torch.cuda.synchronize()
    if ref is None: ref = r
    assert torch.equal(ref, r), mode
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): eng.eval_ranks(wu, wi, rows, blocked=mode)
    e1.record(); torch.cuda.synchronize()
    out[str(mode)] = round(e0.elapsed_time(e1) / 10 * 1000, 1)
def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): r = fn()
    e1.record(); torch.cuda.synchronize()
    return r, round(e0.elapsed_time(e1) / reps * 1000, 1)
# LDS-sliced form: preparation (once per test set), whole chip, then grid sizes; ranks must equal the plain kernel's
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); eng._sliced_rows(rows, I); e1.record(); torch.cuda.synchronize()
out["sliced_prepare"] = round(e0.elapsed_time(e1) * 1000, 1)
for cap in (0, 128, 256, 512, 1024):
    r, us = timed(lambda: eng.eval_ranks(wu, wi, rows, sliced=True, max_workgroups=cap))
    assert torch.equal(ref, r), ("sliced", cap, int((ref != r).sum()))
    out["sliced_wg%d" % cap] = us
if os.environ.get("SIDE", "1") == "1":      # on the evaluation partition's 64 CUs (the period's setting), both forms
    side = eng._side_stream()
    with torch.cuda.stream(side):
        for name, kw in (("blocked", dict(blocked=True, max_workgroups=eng._side_eval_cap())),
                         ("sliced", dict(sliced=True, max_workgroups=eng._side_eval_cap())),
                         ("sliced_x2", dict(sliced=True, max_workgroups=2 * eng._side_eval_cap())),
                         ("sliced_x4", dict(sliced=True, max_workgroups=4 * eng._side_eval_cap()))):
            r, us = timed(lambda: eng.eval_ranks(wu, wi, rows, **kw), reps=5)
            assert torch.equal(ref, r), name
            out["side64_" + name] = us
for cap in (64, 128, 256, 512, 1024, 2048):
    eng.eval_ranks(wu, wi, rows, blocked=True, max_workgroups=cap)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): r = eng.eval_ranks(wu, wi, rows, blocked=True, max_workgroups=cap)
    e1.record(); torch.cuda.synchronize()
    assert torch.equal(ref, r), cap
    out["cap%d" % cap] = round(e0.elapsed_time(e1) / 5 * 1000, 1)
print(json.dumps({"d": d, "us_per_eval": out}))
