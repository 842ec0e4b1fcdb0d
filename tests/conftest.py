"""pytest configuration: `gpu` marker, import paths, golden-fixture loader."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "graphical-normalizing-flows_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    """-> dict of torch tensors (numeric arrays) / numpy arrays (strings)."""
    out = {}
    with np.load(os.path.join(GOLDEN, name + ".npz")) as f:
        for k in f.files:
            a = f[k]
            out[k] = torch.from_numpy(a) if a.dtype.kind in "fiub" else a
    return out


def params_of(g, prefix="p."):
    return {k[len(prefix):]: v for k, v in g.items() if k.startswith(prefix)}


def linear_layers(p, prefix):
    """[(W,b)] from state entries `<prefix>net.<2k>.weight/bias`."""
    layers, k = [], 0
    while "%snet.%d.weight" % (prefix, k) in p:
        layers.append((p["%snet.%d.weight" % (prefix, k)], p["%snet.%d.bias" % (prefix, k)]))
        k += 2
    return layers


def rel_err(a, b):
    a = a.double()
    b = b.double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def assert_close(a, b, rtol=1e-5, atol=1e-6, what=""):
    """element-wise |a-b| <= atol + rtol*|b| (the per-entry form of north_star's 1e-5 relative fp32; rel_err above is
    the infinity-norm form and leaves small entries unconstrained)."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    assert a.shape == b.shape, (what, tuple(a.shape), tuple(b.shape))
    excess = (a - b).abs() - (atol + rtol * b.abs())
    if excess.numel() and excess.max().item() > 0:
        k = int(excess.argmax())
        raise AssertionError("%s: entry %d differs: %.9g vs %.9g (|diff| %.3g > %.3g); %d of %d entries out of tolerance"
                             % (what, k, a.reshape(-1)[k].item(), b.reshape(-1)[k].item(),
                                (a - b).abs().reshape(-1)[k].item(), (atol + rtol * b.abs()).reshape(-1)[k].item(),
                                int((excess > 0).sum()), excess.numel()))
