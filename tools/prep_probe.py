"""Index preparation alone (sml_embed_loss_sgd_prepare) over table heights / batch sizes: timing and a crash probe.
usage: python tools/prep_probe.py [--users U --items I --batch B --triples N --zipf A --reps R]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--users", type=int, default=10000000)
    ap.add_argument("--items", type=int, default=1000000)
    ap.add_argument("--batch", type=int, default=262144)
    ap.add_argument("--triples", type=int, default=4194304)
    ap.add_argument("--zipf", type=float, default=0.0)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    from sml_amd import synth
    from sml_amd.engine import HipEngine
    dev = torch.device("cuda:0")
    engines = {"hand": HipEngine(dev, 32, a.batch)}
    # (the library-sort path is test infrastructure since round 5: tests/build_reference.py builds the library that has it)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    import build_reference
    from sml_amd import _lib
    engines["cub"] = HipEngine(dev, 32, a.batch, lib=_lib.load_other(build_reference.build()))
    rng = np.random.RandomState(4)
    u, i, j = synth.synth_triples(rng, a.triples, a.users, a.items, a_user=0.0, a_item=a.zipf)
    tri = torch.from_numpy(np.stack([u, i, j], 1)).to(dev)
    for mode in ("hand", "cub"):
        os.environ["SML_PREP"] = mode
        eng = engines[mode]
        for _ in range(2):
            eng.bare_prepare(tri, a.batch, a.users, a.items)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            eng.bare_prepare(tri, a.batch, a.users, a.items)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.reps
        print("%s: users=%d items=%d batch=%d triples=%d zipf=%g: %.1f us per epoch, %.2f us per batch" %
              (mode, a.users, a.items, a.batch, a.triples, a.zipf, dt * 1e6, dt * 1e6 / (-(-a.triples // a.batch))), flush=True)


if __name__ == "__main__":
    main()
