#!/usr/bin/env python3
"""In-kernel timeline of the TR step: phase durations of workgroup 0 of each kernel and the gaps between consecutive
kernels, in microseconds (100 MHz wall clock: 10 ns resolution).  Needs a measurement build of the library:

    SML_EXTRA_FLAGS=-DSML_TIMELINE python -m sml_amd.build --force     (and a plain `--force` build afterwards)

fwd stamps: 2 gather done, 3 prologue done, 4 fc1 done, 5 barrier, 6 fc2 + reduce, 7 out stored;
bwd: 2 pair loss + dOut, 3 dA2 / dZ1, 4 dA1, 7 end;  wgrad: 2 MFMA loop done, 3 partials in LDS, 7 end."""
import ctypes, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import contextlib, io
from sml_amd import synth
from sml_amd.conv_transfer import ConvTransfer_com
from sml_amd.engine import HipEngine
from sml_amd.mf import MFbasemode
dev = torch.device("cuda", 0)
d, U, I, n = 32, 60000, 123000, 256 * 120
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
eng.tr_stage_epoch(net, lu, li, hu, hi, tri, 256, 1e-3, 1e-4)
torch.cuda.synchronize()
buf = torch.zeros(70000 + 8 * 1024, dtype=torch.int64, device=dev)
mf.user_laten.weight.data.mul_(1.0)
assert eng.lib.sml_debug_timeline(ctypes.c_void_p(buf.data_ptr())) == 0, "library was not built with -DSML_TIMELINE"
MODE = os.environ.get("MODE", "tr")
with eng.partition():
    if MODE == "mf":
        eng.mf_stage_epoch(mf, net, lu, li, tri, 1024, 0.01, 1e-6)
    else:
        eng.tr_stage_epoch(net, lu, li, hu, hi, tri, 256, 1e-3, 1e-4)
torch.cuda.synchronize()
eng.lib.sml_debug_timeline(ctypes.c_void_p(0))
b = buf.cpu().numpy()
nrec = int(b[0])
rec = b[16:16 + 16 * nrec].reshape(nrec, 16)
rec = rec[np.argsort(rec[:, 1])]
names = {1: "fwd", 2: "bwd", 3: "wgrad", 4: "fwd1", 5: "bwd_full"}
# group per kernel launch: consecutive records with the same kid (first + last block)
launches = []
for r in rec:
    kid, last = int(r[0]) // 2, int(r[0]) % 2
    if launches and launches[-1]["kid"] == kid and len(launches[-1]["recs"]) < 2 and launches[-1]["recs"][0][0] % 2 != r[0] % 2:
        launches[-1]["recs"].append(r)
    else:
        launches.append({"kid": kid, "recs": [r]})
out = {}
def us(x): return x / 100.0
stats = {}
prev_end = None
for L in launches[6:-6]:
    name = names.get(L["kid"], "?")
    r0 = [r for r in L["recs"] if r[0] % 2 == 0]
    rl = [r for r in L["recs"] if r[0] % 2 == 1]
    if not r0: continue
    r0 = r0[0]
    st = stats.setdefault(name, {})
    stamps = [int(x) for x in r0[1:8]]
    prev = stamps[0]
    for k, t in enumerate(stamps[1:], start=2):
        if t:
            st.setdefault("T%d-T%d" % (k - 1 if k > 2 else 1, k), []).append(us(t - prev)); prev = t
    if r0[9] and r0[3]:
        st.setdefault("T3-T9 (fc1 products incl. the wait for the first operands)", []).append(us(int(r0[9]) - int(r0[3])))
        if r0[4]: st.setdefault("T9-T4 (fc1 epilogue: bias, z1 / a2 stores, Gelu, a2 tile)", []).append(us(int(r0[4]) - int(r0[9])))
    if r0[8]:
        st.setdefault("TRUE gap: last workgroup of the previous launches out -> this launch's first stamp", []).append(us(stamps[0] - int(r0[8])))
    if rl:
        st.setdefault("entry skew last-first", []).append(us(int(rl[0][1]) - stamps[0]))
        if rl[0][7]: st.setdefault("last block end - first block end", []).append(us(int(rl[0][7]) - prev))
    ends = [int(r[7]) for r in L["recs"] if r[7]]
    if prev_end is not None:
        st.setdefault("gap from previous kernel's last stamp to entry", []).append(us(stamps[0] - prev_end))
    prev_end = max(ends) if ends else prev
for k, v in stats.items():
    print(k, {kk: (round(float(np.median(vv)), 2) if not isinstance(vv[0], tuple) else [round(float(x), 2) for x in np.median(np.array(vv), axis=0)]) for kk, vv in v.items()})

# per-workgroup picture of the LAST launch of each kernel: start / end relative to the earliest start
for kid, name in ((1, "fwd"), (2, "bwd"), (3, "wgrad"), (4, "fwd1"), (5, "bwd_full")):
    base = 70000 + kid * 1024
    end, start = b[base:base + 512], b[base + 512:base + 1024]
    live = start > 0
    if not live.any(): continue
    t0 = start[live].min()
    e = (end[live] - t0) / 100.0; st_ = (start[live] - t0) / 100.0
    idx = np.nonzero(live)[0]
    order = np.argsort(-e)[:6]
    print(name, "workgroups", int(live.sum()), "start spread %.2f us" % st_.max(), "end: median %.2f  max %.2f us" % (np.median(e), e.max()),
          "slowest blocks", [(int(idx[o]), round(float(e[o]), 2)) for o in order])
    if name == "wgrad":
        v2 = os.environ.get("SML_TR_V2", "1") != "0"
        groups = ((0, 96, "tail workgroups (dA1, tail, conv gradients)"), (96, 192, "first net's tiles"), (192, 288, "second net's tiles")) if v2 \
            else ((0, 2, "conv"), (2, 98, "first net's tiles"), (98, 194, "second net's tiles"))
        for lo, hi, what in groups:
            sel = (idx >= lo) & (idx < hi)
            if sel.any():
                print("   ", what, "start median %.2f  end median %.2f  max %.2f" % (np.median(st_[sel]), np.median(e[sel]), e[sel].max()))
