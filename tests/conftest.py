import contextlib
import io
import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@contextlib.contextmanager
def quiet():
    with contextlib.redirect_stdout(io.StringIO()) as buf:
        yield buf


def make_transfer(d, z=None, prefix="theta.", device="cpu"):
    """The product's ConvTransfer_com module (a parameter container) filled from a fixture."""
    from sml_amd.conv_transfer import ConvTransfer, ConvTransfer_com
    cls = ConvTransfer_com
    if z is not None and np.asarray(z[prefix + "user_transfer.conv1.weight"]).shape[2] == 2:
        cls = ConvTransfer          # fixtures of --transfer_type conv (kernel (2,1) nets)
    with quiet():
        net = cls(d, d)
    if z is not None:
        sd = {k[len(prefix):]: torch.from_numpy(np.asarray(z[k])) for k in z.files if k.startswith(prefix)}
        net.load_state_dict(sd)
    return net.to(device)


def make_mf(U, I, d, wu=None, wi=None, device="cpu"):
    from sml_amd.mf import MFbasemode
    mf = MFbasemode(U, I, d)
    with torch.no_grad():
        if wu is not None:
            mf.user_laten.weight.copy_(torch.from_numpy(np.asarray(wu)))
        if wi is not None:
            mf.item_laten.weight.copy_(torch.from_numpy(np.asarray(wi)))
    return mf.to(device)


def T(a, device="cpu"):
    return torch.from_numpy(np.ascontiguousarray(np.asarray(a))).to(device)


needs_gpu = pytest.mark.gpu
