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
        # (SML_HARNESS_EXTRA_PERIODS: periods allocated and never used -- does the host's memory footprint alone change the stage time?)
        self.p = [synth.sample_period(rng, n, U, I, neg=neg) for _ in range(n_stage + 2 + int(os.environ.get("SML_HARNESS_EXTRA_PERIODS", "0")))]
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
    timeline = "--timeline" in sys.argv
    marks = []
    if timeline:
        # device-side start / end of every engine call of the stage (events on the call's own stream) beside the host's
        # clock at the call: where the device waits for the host, and where the host waits for the device
        eng = meta.engine

        def wrap(name):
            fn = getattr(eng, name)

            def inner(*a, **k):
                st = torch.cuda.current_stream()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                h0 = time.perf_counter()
                e0.record(st)
                r = fn(*a, **k)
                e1.record(st)
                marks.append((name, h0, time.perf_counter(), e0, e1))
                return r
            setattr(eng, name, inner)
        for nm in ("mf_stage_epoch", "tr_stage_epoch", "updata", "mf_flush"):
            wrap(nm)
        # host-only spans of the stage's first milliseconds
        hspans = []

        def hwrap(obj, name, label=None):
            fn = getattr(obj, name)

            def inner(*a, **k):
                h0 = time.perf_counter()
                r = fn(*a, **k)
                hspans.append((label or name, h0, time.perf_counter()))
                return r
            setattr(obj, name, inner)
        import sml_amd.datasets as DS
        import sml_amd.driver as DRV
        hwrap(meta, "save_MF_weight"); hwrap(meta, "get_next_data"); hwrap(meta, "_test"); hwrap(meta, "_flush_output"); hwrap(meta, "_prefetch_next")
        hwrap(DS.trainDataset_withPreSample, "epoch_triples", "presample.epoch_triples"); hwrap(DS.offlineDataset_withsample, "epoch_triples", "withsample.epoch_triples")
        hwrap(DS.trainDataset_withPreSample, "__init__", "presample.__init__"); hwrap(DRV, "SampleDaset", "SampleDaset()")
    for s in range(n_stage):
        if prof is not None and s == n_stage - 1:
            prof.enable()
        torch.cuda.synchronize()
        t0 = time.time()
        t0_perf = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            meta.train_one_stage3(args, s)
        torch.cuda.synchronize()
        dt = time.time() - t0
        print("stage %d: %.3f s wall; engine calls mf %.3f tr %.3f updata %.3f eval %.3f (cumulative, host view); TR passes drawn ahead and adopted: %d"
              % (s, dt, meta.timing["mf"], meta.timing["tr"], meta.timing["updata"], meta.timing["eval"], meta.timing.get("tr_spec_adopted", 0)))
        if timeline and s == n_stage - 1:
            base = marks[0][3]
            rows = [(nm, (h0 - t0_perf) * 1e3, (h1 - t0_perf) * 1e3, base.elapsed_time(e0), base.elapsed_time(e1)) for nm, h0, h1, e0, e1 in marks]
            d0 = (rows[0][1])          # device clock aligned at the first call (it cannot start before the host queues it)
            idle, prev_end = 0.0, None
            print("  call              host start   host end | device start  device end   device idle before")
            for nm, h0, h1, g0, g1 in rows:
                gap = (g0 - prev_end) if prev_end is not None else 0.0
                idle += max(gap, 0.0)
                print("  %-16s %9.2f  %9.2f | %11.2f  %10.2f  %8.2f" % (nm, h0, h1, d0 + g0, d0 + g1, gap))
                prev_end = g1
            print("  device idle between engine calls: %.2f ms; first call queued %.2f ms after the stage began; last call ends at %.2f ms"
                  % (idle, rows[0][1], d0 + rows[-1][4]))
            tp0 = t0_perf
            print("  host spans (ms from the stage's start), first 14 and last 4:")
            for nm, a0, a1 in hspans[:14] + hspans[-4:]:
                print("    %-28s %8.2f -> %8.2f  (%.2f)" % (nm, (a0 - tp0) * 1e3, (a1 - tp0) * 1e3, (a1 - a0) * 1e3))
        marks.clear()
        if timeline:
            hspans.clear()

    if prof is not None:
        import pstats
        prof.disable()
        pstats.Stats(prof).sort_stats("cumulative").print_stats(28)


if __name__ == "__main__":
    main()
