#!/usr/bin/env python3
"""The d = 128 fp16 a3 leg is bimodal between processes (k_bare_grad 77 or ~90 us per 262,144-triple batch on 50M x 5M tables).  Is it the
12.8 GB user table's placement?  In ONE process: allocate the tables, time the gradient pass, free, allocate again (after a spacer allocation of
varying size, so the table lands elsewhere), time again; prints the table's address and the kernel time per trial."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sml_amd import synth
from sml_amd.engine import HipEngine
dev = torch.device("cuda", 0)
U, I, d, B, n = 50000000, 5000000, 128, 262144, 1 << 21
rng = np.random.RandomState(5)
u, i, j = synth.synth_triples(rng, n, U, I, a_user=0.0, a_item=0.0)
tri = torch.from_numpy(np.stack([u, i, j], 1)).to(dev)
out = []
spacers = [0, 0, 3 << 30, 0, 1 << 30, 0]
keep = []
for trial, sp in enumerate(spacers):
    if sp:
        keep.append(torch.empty(sp, dtype=torch.uint8, device=dev))
    wu = torch.empty(U, d, dtype=torch.float16, device=dev).normal_(0, 0.1)
    wi = torch.empty(I, d, dtype=torch.float16, device=dev).normal_(0, 0.1)
    eng = HipEngine(dev, d, B)
    for _ in range(2):
        eng.bare_epoch(wu, wi, tri, B, 0.05, 1e-6, 1e-6, bce=True)
    cur = eng.bare_prepare(tri, B, U, I)
    torch.cuda.synchronize()
    eng.profile(True)
    for _ in range(3):
        eng.bare_epoch(wu, wi, tri, B, 0.05, 1e-6, 1e-6, bce=True, prepared=cur)
        cur = eng.bare_prepare(tri, B, U, I)
    torch.cuda.synchronize()
    prof = eng.profile_read()
    eng.profile(False)
    c, ms = prof["k_bare_grad"]
    out.append({"trial": trial, "spacer_bytes_before": sp, "wu_ptr": hex(wu.data_ptr()), "wu_ptr_mod_1GiB": wu.data_ptr() % (1 << 30), "wu_ptr_mod_2MiB": wu.data_ptr() % (2 << 20),
                "k_bare_grad_us": round(1000.0 * ms / c, 2)})
    eng.close()
    del wu, wi, eng
    torch.cuda.empty_cache()
print(json.dumps({"pid": os.getpid(), "trials": out}))
