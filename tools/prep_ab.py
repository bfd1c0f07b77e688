"""A/B of two builds of the library on ONE box, in ONE process: the index preparation alone (sml_embed_loss_sgd_prepare) with
the in-tree libsml_hip.so against another build of it (e.g. the previous commit's, built by hand into tools/_ab/), alternating.
usage: python tools/prep_ab.py <other.so> [--zipf A --reps R --rounds N]"""
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
    ap.add_argument("--batch", type=int, default=262144)
    ap.add_argument("--triples", type=int, default=4194304)
    ap.add_argument("--zipf", type=float, default=0.0)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=5)
    a = ap.parse_args()
    from sml_amd import _lib, synth
    from sml_amd.engine import HipEngine
    dev = torch.device("cuda:0")
    os.environ["SML_PREP"] = "hand"
    engines = {"tree": HipEngine(dev, 32, a.batch), "other": HipEngine(dev, 32, a.batch, lib=_lib.load_other(os.path.abspath(a.other)))}
    rng = np.random.RandomState(4)
    u, i, j = synth.synth_triples(rng, a.triples, a.users, a.items, a_user=0.0, a_item=a.zipf)
    tri = torch.from_numpy(np.stack([u, i, j], 1)).to(dev)
    res = {k: [] for k in engines}
    for rnd in range(a.rounds):
        for name, eng in engines.items():
            for _ in range(2):
                eng.bare_prepare(tri, a.batch, a.users, a.items)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.reps):
                eng.bare_prepare(tri, a.batch, a.users, a.items)
            torch.cuda.synchronize()
            res[name].append((time.perf_counter() - t0) / a.reps * 1e6)
    for name, v in res.items():
        print("%-6s zipf=%g: us per epoch by round %s  median %.1f" % (name, a.zipf, " ".join("%.1f" % x for x in v), float(np.median(v))), flush=True)


if __name__ == "__main__":
    main()
