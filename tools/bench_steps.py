#!/usr/bin/env python3
"""Per-batch cost of the two SML training stages in isolation (Yelp-shaped tables, d=32 by default).

    python tools/bench_steps.py [--d 32] [--inter 75000] [--reps 5]

Prints one JSON line: wall microseconds per batch of a TR epoch (B=256) and of an MF epoch (B=1024)
without any profiling events in the stream, then the HIP-event average of every kernel class.  This is
the iteration harness for the transfer-net kernels; the headline number stays bench.py's.
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--d", type=int, default=32)
    ap.add_argument("--users", type=int, default=60000)
    ap.add_argument("--items", type=int, default=123000)
    ap.add_argument("--inter", type=int, default=75000)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--tr-batch", type=int, default=256)
    ap.add_argument("--mf-batch", type=int, default=1024)
    ap.add_argument("--out", default="", help="also write the JSON line to this file (RCCL prints its banner to stdout at exit)")
    ap.add_argument("--comm", default="none", choices=["none", "peer", "rccl", "torch"],
                    help="drive the multi-GPU exchange path on a forced 1-rank group: the per-step overhead of each carrier "
                         "(peer: one-shot push / poll into the rank's own inbox; rccl: the library's communicator; torch: hooks)")
    ap.add_argument("--start-step", type=int, default=0, help="MF optimiser starts at this step with a small random Adam state on every row "
                    "(0: a fresh optimiser; the closed-form replay is used from step 1024 on)")
    ap.add_argument("--lib", default="", help="another build of the library (same C ABI) instead of the in-tree one: A/B of two builds in one gpurun call")
    a = ap.parse_args()
    import contextlib
    import io
    from sml_amd import synth
    from sml_amd.conv_transfer import ConvTransfer_com
    from sml_amd.engine import HipEngine
    from sml_amd.mf import MFbasemode
    dev = torch.device("cuda", 0)
    if a.lib:
        from sml_amd import _lib
        eng = HipEngine(dev, a.d, max(a.tr_batch, a.mf_batch), lib=_lib.load_other(os.path.abspath(a.lib)))
    else:
        eng = HipEngine(dev, a.d, max(a.tr_batch, a.mf_batch))
    torch.manual_seed(2000)
    mf = MFbasemode(a.users, a.items, a.d)
    with torch.no_grad():
        mf.user_laten.weight.mul_(0.3)
        mf.item_laten.weight.mul_(0.3)
    with contextlib.redirect_stdout(io.StringIO()):
        net = ConvTransfer_com(a.d, a.d)
    mf, net = mf.to(dev), net.to(dev)
    eng.adopt(net)
    if a.start_step > 0:
        g = torch.Generator().manual_seed(5)
        eng.load_optimizer_state(mfbase=mf, mf_state=dict(
            m_user=torch.randn(a.users, a.d, generator=g) * 1e-4, v_user=torch.rand(a.users, a.d, generator=g) * 1e-7,
            m_item=torch.randn(a.items, a.d, generator=g) * 1e-4, v_item=torch.rand(a.items, a.d, generator=g) * 1e-7, step=a.start_step))
    if a.comm != "none":
        import socket
        import torch.distributed as dist
        from sml_amd import dist as SD
        os.environ["SML_COMM"] = a.comm
        s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]; s_.close()
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=dev)
        ctx = SD.attach(eng, None, dist, rows_cap=2 * a.mf_batch)
        assert ctx.mode == a.comm, (ctx.mode, a.comm)
    lu = (mf.user_laten.weight.detach() * 0.9).contiguous()
    li = (mf.item_laten.weight.detach() * 0.9).contiguous()
    hu, hi = mf.user_laten.weight.detach().clone(), mf.item_laten.weight.detach().clone()
    rng = np.random.RandomState(7)
    u, i, j = synth.synth_triples(rng, a.inter, a.users, a.items, a_user=1.1, a_item=1.0)
    tri = torch.from_numpy(np.stack([u, i, j], 1)).to(dev)

    def tr():
        eng.tr_stage_epoch(net, lu, li, hu, hi, tri, a.tr_batch, 1e-3, 1e-4)

    def mfe():
        eng.mf_stage_epoch(mf, net, lu, li, tri, a.mf_batch, 0.01, 1e-6)
        eng.mf_flush(mf)       # as the period does after every MF epoch (bounds the lazy-Adam replay windows)

    out = {"d": a.d, "inter": a.inter, "comm": a.comm}
    if os.environ.get("STEPS_NOGC"):
        import gc
        gc.collect(); gc.disable()
    for name, fn, B in (("tr", tr, a.tr_batch), ("mf", mfe, a.mf_batch)):
        nb = -(-a.inter // B)
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        trace = []
        for _ in range(a.reps):
            h0 = __import__("time").perf_counter()
            fn()
            trace.append(round((__import__("time").perf_counter() - h0) * 1e3, 2))
        e1.record()
        torch.cuda.synchronize()
        if os.environ.get("STEPS_TRACE"):
            out[name + "_host_ms_per_epoch"] = trace
        out[name + "_us_per_batch"] = round(1000.0 * e0.elapsed_time(e1) / a.reps / nb, 2)
        eng.profile(True)
        fn()
        torch.cuda.synchronize()
        prof = eng.profile_read()
        eng.profile(False)
        out[name + "_kernels_us"] = {k: round(1000.0 * ms / c, 2) for k, (c, ms) in prof.items() if c >= (1 if os.environ.get("STEPS_ALL") else nb // 2)}
    if a.comm == "peer":
        out["peer_timeouts"] = eng.peer_status()
    print(json.dumps(out))
    if a.out:
        with open(a.out, "w") as f:
            f.write(json.dumps(out) + "\n")
    if a.comm != "none":
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
