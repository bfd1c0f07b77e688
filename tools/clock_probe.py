#!/usr/bin/env python3
"""TR / MF epoch device time with and without a busy neighbour on a few CUs of the evaluation partition
(tools/clock_keeper.hip): does the part clock the latency-bound training kernels higher when something keeps it busy?"""
import ctypes, json, os, sys, time
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
with contextlib.redirect_stdout(io.StringIO()):
    net = ConvTransfer_com(d, d)
mf, net = mf.to(dev), net.to(dev)
eng.adopt(net)
lu = (mf.user_laten.weight.detach() * 0.9).contiguous(); li = (mf.item_laten.weight.detach() * 0.9).contiguous()
hu, hi = mf.user_laten.weight.detach().clone(), mf.item_laten.weight.detach().clone()
rng = np.random.RandomState(7)
u, i, j = synth.synth_triples(rng, n, U, I, a_user=1.1, a_item=1.0)
tri = torch.from_numpy(np.stack([u, i, j], 1)).to(dev)
K = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "clock_keeper.so"))
K.keeper_start.argtypes = [ctypes.c_int] * 4 + [ctypes.c_double, ctypes.c_int]

def epochs():
    out = {}
    with eng.partition():
        for name, fn in (("tr", lambda: eng.tr_stage_epoch(net, lu, li, hu, hi, tri, 256, 1e-3, 1e-4)),
                         ("mf", lambda: (eng.mf_stage_epoch(mf, net, lu, li, tri, 1024, 0.01, 1e-6), eng.mf_flush(mf)))):
            fn(); torch.cuda.synchronize()
            ts = []
            for _ in range(4):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); fn(); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            out[name + "_ms"] = round(float(np.median(ts)), 3)
    return out

res = {"idle neighbour": epochs()}
for label, (c_lo, c_hi, blocks, threads, mfma) in (("8 CUs x 4 waves FMA spin", (248, 256, 8, 256, 0)),
                                               ("64 CUs x 4 waves FMA spin", (192, 256, 64, 256, 0)),
                                               ("64 CUs x 4 waves MFMA spin", (192, 256, 64, 256, 1))):
    # (the keeper's stream is created once: only the first mask takes effect -- run one configuration per process)
    if os.environ.get("KEEP", "8 CUs x 4 waves FMA spin") != label:
        continue
    assert K.keeper_start(c_lo, c_hi, blocks, threads, 20.0, mfma) == 0
    time.sleep(0.2)
    res[label] = epochs()
    assert K.keeper_stop() == 0
res["idle again"] = epochs()
print(json.dumps(res))
