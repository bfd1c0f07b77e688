#!/usr/bin/env python3
"""Attribution runs of the LDS-sliced rank pass (SML_EVS_DBG=1: no user-row loads, 2: no LDS item reads, 3: neither; results wrong by construction)."""
import os, sys, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sml_amd.engine import HipEngine
from sml_amd import synth
dev = torch.device("cuda", 0)
d, U, I, n, neg = 32, 60000, 123000, 75000, 999
eng = HipEngine(dev, d, 1024)
rng = np.random.RandomState(1)
_, test = synth.sample_period(rng, n, U, I, neg=neg)
rows = torch.from_numpy(test).to(dev)
wu, wi = torch.randn(U, d, device=dev) * 0.1, torch.randn(I, d, device=dev) * 0.1
side = eng._side_stream()
out = {}
for name, st in (("chip", torch.cuda.current_stream()), ("side64", side)):
    with torch.cuda.stream(st):
        eng.eval_ranks(wu, wi, rows, sliced=True, max_workgroups=256); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(5): eng.eval_ranks(wu, wi, rows, sliced=True, max_workgroups=256)
        e1.record(st); torch.cuda.synchronize()
        out[name] = round(e0.elapsed_time(e1) / 5 * 1000, 1)
print(json.dumps({"dbg": os.environ.get("SML_EVS_DBG", "0"), "us": out}))
