#!/usr/bin/env python3
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sml_amd.engine import HipEngine
from sml_amd import synth
dev = torch.device("cuda", 0)
d = int(os.environ.get("D", "32")); U, I, n, neg = 60000, 123000, 75000, 999
mode = {"plain": False, "xcd": True}[os.environ.get("MODE", "xcd")]
eng = HipEngine(dev, d, 1024)
rng = np.random.RandomState(1)
_, test = synth.sample_period(rng, n, U, I, neg=neg)
rows = torch.from_numpy(test).to(dev)
wu, wi = torch.randn(U, d, device=dev) * 0.1, torch.randn(I, d, device=dev) * 0.1
for _ in range(6): eng.eval_ranks(wu, wi, rows, blocked=mode)
torch.cuda.synchronize()
