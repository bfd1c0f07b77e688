"""Debug aid: which entries of theta / m / v differ between SML_TR_DEFER=0 and 1 (see the A/B test in tests/test_hip_parity.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from conftest import make_transfer
from sml_amd.engine import HipEngine
DEV = "cuda:0"
d, B, nb = 32, 256, int(sys.argv[1]) if len(sys.argv) > 1 else 6
torch.manual_seed(3 * d + nb)
U, I, n = 400, 300, nb * B - 9
wu, wi = torch.randn(U, d) * 0.3, torch.randn(I, d) * 0.3
tri = torch.stack([torch.randint(0, U, (n,)), torch.randint(0, I, (n,)), torch.randint(0, I, (n,))], 1)
sd, res = None, []
for defer in ("0", "1"):
    os.environ["SML_TR_DEFER"] = defer
    eng = HipEngine(DEV, d, 1024)
    net = make_transfer(d, device=DEV)
    if sd is None:
        sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    else:
        net.load_state_dict(sd)
    l = eng.tr_stage_epoch(net, (wu * 0.9).to(DEV), (wi * 0.9).to(DEV), wu.to(DEV), wi.to(DEV), tri, B, 1e-3, 1e-4).cpu()
    res.append((l, eng.adopt(net).detach().cpu().clone(), eng.tr_state[0].cpu().clone(), eng.tr_state[1].cpu().clone()))
ns = eng.net_size
for name, x, y in zip(("loss", "theta", "m", "v"), res[0], res[1]):
    bad = (x != y).nonzero().flatten()
    print(name, "differs at", len(bad), "entries", [(int(i) // ns, int(i) % ns, float(x[i]), float(y[i])) for i in bad[:12]])
