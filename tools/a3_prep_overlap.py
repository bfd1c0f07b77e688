"""a3 epochs with the index preparation (a) inline on the same stream, (b) one epoch ahead on the side stream, (c) not at
all (lists reused: the step kernels alone) -- wall per epoch."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sml_amd import synth
from sml_amd.engine import HipEngine
dev = torch.device("cuda:0")
U, I, B, n, d = 10000000, 1000000, 262144, 4194304, 32
zipf = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
eng = HipEngine(dev, d, B)
g = torch.Generator(device=dev).manual_seed(4)
wi = torch.randn(I, d, device=dev, generator=g) * 0.1
wu = torch.randn(U, d, device=dev, generator=g) * 0.1
rng = np.random.RandomState(4)
u, i, j = synth.synth_triples(rng, n, U, I, a_user=0.0, a_item=zipf)
tri = torch.from_numpy(np.stack([u, i, j], 1)).to(dev)
reps = 8
def run(mode):
    cur = eng.bare_prepare(tri, B, U, I)
    nxt = eng.bare_prepare(tri, B, U, I)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        if mode == "inline":
            eng.bare_epoch(wu, wi, tri, B, 0.05, 1e-6, 1e-6)
        elif mode == "side":
            cur, nxt = nxt, eng.bare_prepare(tri, B, U, I)
            eng.bare_epoch(wu, wi, tri, B, 0.05, 1e-6, 1e-6, prepared=cur)
        elif mode == "none":
            eng.bare_epoch(wu, wi, tri, B, 0.05, 1e-6, 1e-6, prepared=cur)
        elif mode == "prep_only":
            cur, nxt = nxt, eng.bare_prepare(tri, B, U, I)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6
for mode in ("none", "prep_only", "inline", "side", "none", "side"):
    print("%-10s %8.1f us per epoch" % (mode, run(mode)), flush=True)
