#!/usr/bin/env python3
"""Wall-clock of the real driver (meta_train.train_one_stage3) on a Yelp-shaped synthetic stage,
host-side batch supply and uploads included.  Usage: python tools/time_driver_stage.py [stages]"""
import contextlib
import io
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sml_amd import cli, synth            # noqa: E402
from sml_amd.driver import meta_train     # noqa: E402
from sml_amd.mf import MFbasemode         # noqa: E402


class Periods(object):
    """next_train() of the reference's transfer_data, from in-memory synthetic periods."""

    def __init__(self, n_stage, n, U, I, neg):
        self.user_number, self.item_number = U, I
        rng = np.random.RandomState(1)
        self.p = [synth.sample_period(rng, n, U, I, neg=neg) for _ in range(n_stage + 2)]
        self.n_stage = n_stage

    def reinit(self):
        pass

    def next_train(self, s):
        if s >= self.n_stage:
            return None, None, None, None
        return self.p[s][1], self.p[s + 1][0], None, self.p[s + 1][1]


def main():
    n_stage = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    profile = len(sys.argv) > 2 and sys.argv[2] == "--profile"
    U, I, n, neg, d = 60000, 123000, 75000, 999, 32
    args = cli.get_parse("yelp").parse_args(["--laten", str(d), "--numworkers", "0"])
    torch.set_num_threads(1)          # as cli.main does
    torch.manual_seed(0)
    np.random.seed(1)
    mf = MFbasemode(U, I, d)
    with torch.no_grad():
        mf.user_laten.weight.mul_(0.3)
        mf.item_laten.weight.mul_(0.3)
    torch.save(mf, "/tmp/_init.pkl")
    args.pre_model = "/tmp/_init.pkl"
    data = Periods(n_stage, n, U, I, neg)
    with contextlib.redirect_stdout(io.StringIO()):
        meta = meta_train(args, data, U, I, d)
    prof = None
    if profile:
        import cProfile
        prof = cProfile.Profile()
    meta._prefetch = os.environ.get("SML_PREFETCH", "1") != "0"      # as meta_train.run does
    for s in range(n_stage):
        if prof is not None and s == n_stage - 1:
            prof.enable()
        torch.cuda.synchronize()
        t0 = time.time()
        with contextlib.redirect_stdout(io.StringIO()):
            meta.train_one_stage3(args, s)
        torch.cuda.synchronize()
        dt = time.time() - t0
        print("stage %d: %.3f s wall; engine calls mf %.3f tr %.3f updata %.3f eval %.3f (cumulative, host view)"
              % (s, dt, meta.timing["mf"], meta.timing["tr"], meta.timing["updata"], meta.timing["eval"]))

    if prof is not None:
        import pstats
        prof.disable()
        pstats.Stats(prof).sort_stats("cumulative").print_stats(28)


if __name__ == "__main__":
    main()
