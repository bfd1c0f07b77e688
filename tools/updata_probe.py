#!/usr/bin/env python3
"""Time of one `updata` (the transfer net over every user and item row, reference model/transfer.py:884-902) at the
Yelp shape, on the whole chip and on the training partition (192 CUs).  Geometry overrides for A/B runs:
SML_FWD_MT=3 (48-row tiles, one workgroup per CU), SML_FWD_HSEQ=1 (32-row tiles, one hidden pass);
default: 32-row tiles, two hidden passes, two workgroups per CU."""
import contextlib, io, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sml_amd.conv_transfer import ConvTransfer_com
from sml_amd.engine import HipEngine
d, U, I = int(os.environ.get("D", "32")), 60000, 123000
dev = torch.device("cuda", 0)
eng = HipEngine(dev, d, 1024)
torch.manual_seed(1)
with contextlib.redirect_stdout(io.StringIO()):
    net = ConvTransfer_com(d, d).to(dev)
eng.adopt(net)
lu, hu, li, hi = (torch.randn(n, d, device=dev) * 0.3 for n in (U, U, I, I))
ou, oi = torch.empty_like(lu), torch.empty_like(li)
out = {"d": d}
for name, scope in (("whole_chip", contextlib.nullcontext), ("training_partition", eng.partition)):
    with scope():
        for _ in range(3):
            eng.updata(net, lu, hu, li, hi, ou, oi)
        st = torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(20):
            eng.updata(net, lu, hu, li, hi, ou, oi)
        e1.record(st)
        st.synchronize()
        us = 1000.0 * e0.elapsed_time(e1) / 20
    out[name + "_us"] = round(us, 1)
    out[name + "_tflops"] = round((U + I) * 6304.0 * d / us / 1e6, 1)
print(json.dumps(out))
