#!/usr/bin/env python3
"""Is the host ahead of the device?  Host time of each epoch call (the C loop that enqueues every launch of the epoch)
against the device time of the same epoch, Yelp-shaped tables, d=32."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import contextlib, io
from sml_amd import synth
from sml_amd.conv_transfer import ConvTransfer_com
from sml_amd.engine import HipEngine
from sml_amd.mf import MFbasemode
dev = torch.device("cuda", 0)
d, U, I, n = 32, 60000, 123000, 75000
eng = HipEngine(dev, d, 1024)
torch.manual_seed(2000)
mf = MFbasemode(U, I, d)
with torch.no_grad():
    mf.user_laten.weight.mul_(0.3); mf.item_laten.weight.mul_(0.3)
with contextlib.redirect_stdout(io.StringIO()):
    net = ConvTransfer_com(d, d)
mf, net = mf.to(dev), net.to(dev)
eng.adopt(net)
lu = (mf.user_laten.weight.detach() * 0.9).contiguous(); li = (mf.item_laten.weight.detach() * 0.9).contiguous()
hu, hi = mf.user_laten.weight.detach().clone(), mf.item_laten.weight.detach().clone()
rng = np.random.RandomState(7)
u, i, j = synth.synth_triples(rng, n, U, I, a_user=1.1, a_item=1.0)
tri = torch.from_numpy(np.stack([u, i, j], 1)).to(dev)
out = {}
for name, fn in (("tr_epoch", lambda: eng.tr_stage_epoch(net, lu, li, hu, hi, tri, 256, 1e-3, 1e-4)),
                 ("mf_epoch", lambda: eng.mf_stage_epoch(mf, net, lu, li, tri, 1024, 0.01, 1e-6)),
                 ("updata", lambda: eng.updata(net, lu, hu, li, hi, mf.user_laten.weight.data, mf.item_laten.weight.data))):
    fn(); torch.cuda.synchronize()
    host, devt = [], []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record(); t0 = time.perf_counter(); fn(); t1 = time.perf_counter(); e1.record()
        torch.cuda.synchronize()
        host.append(1e3 * (t1 - t0)); devt.append(e0.elapsed_time(e1))
    out[name] = {"host_ms": round(float(np.median(host)), 3), "device_ms": round(float(np.median(devt)), 3)}
print(json.dumps(out))

# ---- does a cross-stream wait block the host?  (queue an epoch, then time side.wait_stream(cur) on the host)
res = {}
for kind in ("plain", "masked", "masked_both"):
    if kind == "plain":
        side, cur_ctx = torch.cuda.Stream(device=dev), contextlib.nullcontext()
    elif kind == "masked":
        side, cur_ctx = eng._side_stream(), contextlib.nullcontext()
    else:
        side, cur_ctx = eng._side_stream(), torch.cuda.stream(eng.training_stream())
    with cur_ctx:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.tr_stage_epoch(net, lu, li, hu, hi, tri, 256, 1e-3, 1e-4)
        t1 = time.perf_counter()
        side.wait_stream(torch.cuda.current_stream())
        t2 = time.perf_counter()
        ev = torch.cuda.Event(); ev.record(torch.cuda.current_stream())
        t3 = time.perf_counter()
        torch.cuda.synchronize()
        t4 = time.perf_counter()
    res[kind] = {"enqueue_ms": round(1e3 * (t1 - t0), 3), "wait_stream_ms": round(1e3 * (t2 - t1), 3),
                 "record_ms": round(1e3 * (t3 - t2), 3), "drain_ms": round(1e3 * (t4 - t3), 3)}
print(json.dumps(res))

# ---- what does one cross-stream ordering point cost the SIGNALLING stream?
import ctypes
def timed(body, reps=8):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(reps):
        body()
    e1.record(); torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / reps, 3), round(1e3 * (time.perf_counter() - t0) / reps, 3)
side = eng._side_stream()
small = tri[:256 * 20]
def epoch():
    eng.tr_stage_epoch(net, lu, li, hu, hi, small, 256, 1e-3, 1e-4)       # 20 batches, ~0.6 ms
res2 = {}
with torch.cuda.stream(eng.training_stream()):
    cur = torch.cuda.current_stream()
    res2["epoch"] = timed(epoch)
    def a():
        epoch(); ev = torch.cuda.Event(); ev.record(cur)
    res2["epoch+record"] = timed(a)
    def b():
        epoch(); side.wait_stream(cur)
    res2["epoch+side.wait_stream"] = timed(b)
    z = torch.zeros(1024, device=dev)
    def c():
        epoch(); side.wait_stream(cur)
        with torch.cuda.stream(side):
            z.add_(1.0)
    res2["epoch+side.wait_stream+side kernel"] = timed(c)
    def dd():
        epoch(); side.wait_stream(cur)
        with torch.cuda.stream(side):
            z.add_(1.0)
        cur.wait_stream(side)
    res2["...+cur waits side"] = timed(dd)
print(json.dumps(res2))

res3 = {}
with torch.cuda.stream(eng.training_stream()):
    cur = torch.cuda.current_stream()
    def e():
        epoch(); ev = torch.cuda.Event(); ev.record(cur); ev.synchronize()
        with torch.cuda.stream(side):
            z.add_(1.0)
    res3["epoch, host waits event, then side kernel"] = timed(e)
    def f():
        epoch()
        with torch.cuda.stream(side):
            z.add_(1.0)
    res3["epoch + unordered side kernel"] = timed(f)
    big = torch.zeros(64 << 20, device=dev)
    def g():
        epoch(); side.wait_stream(cur)
        with torch.cuda.stream(side):
            big.add_(1.0); big.add_(1.0); big.add_(1.0)
    res3["epoch+wait+3 x 256MB side kernels"] = timed(g)
plain = torch.cuda.Stream(device=dev)
def h():
    epoch(); plain.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(plain):
        z.add_(1.0)
res3["default stream + plain side stream, wait + kernel"] = timed(h)
print(json.dumps(res3))
