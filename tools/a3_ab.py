"""A/B of two builds of the library on ONE box, in ONE process: the bare a3 step (prepared epochs) with the in-tree
libsml_hip.so against another build of it (tools/_ab/...), alternating.  Per round and build: HIP-event average of the
step kernels over one profiled epoch, and wall per epoch over `--reps` epochs with the next epoch's lists prepared on the
side stream (what bench.py's a3 legs time).
usage: python tools/a3_ab.py <other.so> [--d 32 --dtype f32 --zipf 0 --users U --items I --reps 8 --rounds 4]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("other")
    ap.add_argument("--users", type=int, default=10000000)
    ap.add_argument("--items", type=int, default=1000000)
    ap.add_argument("--d", type=int, default=32)
    ap.add_argument("--dtype", default="f32", choices=["f32", "f16"])
    ap.add_argument("--batch", type=int, default=262144)
    ap.add_argument("--triples", type=int, default=1 << 22)
    ap.add_argument("--zipf", type=float, default=0.0)
    ap.add_argument("--reps", type=int, default=8)
    ap.add_argument("--rounds", type=int, default=4)
    a = ap.parse_args()
    from sml_amd import _lib, synth
    from sml_amd.engine import HipEngine
    dev = torch.device("cuda:0")
    engines = {"tree": HipEngine(dev, a.d, a.batch), "other": HipEngine(dev, a.d, a.batch, lib=_lib.load_other(os.path.abspath(a.other)))}
    dt = torch.float32 if a.dtype == "f32" else torch.float16
    g = torch.Generator(device=dev).manual_seed(4)
    wi = (torch.randn(a.items, a.d, device=dev, generator=g) * 0.1).to(dt)
    wu = (torch.randn(a.users, a.d, device=dev, generator=g) * 0.1).to(dt)
    rng = np.random.RandomState(4)
    u, i, j = synth.synth_triples(rng, a.triples, a.users, a.items, a_user=0.0, a_item=a.zipf)
    tri = torch.from_numpy(np.stack([u, i, j], 1)).to(dev)
    res = {k: {"grad": [], "upd": [], "hot": [], "wall": []} for k in engines}
    for rnd in range(a.rounds):
        for name, eng in engines.items():
            for _ in range(2):
                eng.bare_epoch(wu, wi, tri, a.batch, 0.05, 1e-6, 1e-6, bce=True)
            nxt = eng.bare_prepare(tri, a.batch, a.users, a.items)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.reps):
                cur, nxt = nxt, eng.bare_prepare(tri, a.batch, a.users, a.items)
                eng.bare_epoch(wu, wi, tri, a.batch, 0.05, 1e-6, 1e-6, bce=True, prepared=cur)
            torch.cuda.synchronize()
            res[name]["wall"].append((time.perf_counter() - t0) / a.reps * 1e6)
            eng.profile(True)
            eng.bare_epoch(wu, wi, tri, a.batch, 0.05, 1e-6, 1e-6, bce=True, prepared=nxt)
            torch.cuda.synchronize()
            prof = eng.profile_read()
            eng.profile(False)
            for key, cls in (("grad", "k_bare_grad"), ("upd", "k_seg_update_sgd"), ("hot", "k_hot_rows")):
                c, ms = prof.get(cls, (0, 0.0))
                res[name][key].append(1000.0 * ms / c if c else 0.0)
    for name, r in res.items():
        print("%-6s d=%d %s zipf=%g: k_bare_grad %s | k_run_update %s | hot %s | wall per epoch %s" % (
            name, a.d, a.dtype, a.zipf, " ".join("%.2f" % x for x in r["grad"]), " ".join("%.2f" % x for x in r["upd"]),
            " ".join("%.2f" % x for x in r["hot"]), " ".join("%.0f" % x for x in r["wall"])), flush=True)


if __name__ == "__main__":
    main()
