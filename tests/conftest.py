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
