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


def assert_fwd(a, b, what="forward value"):
    """element-wise form of the forward tolerance (review of round 5, item 7): |a - b| <= 1e-6 max|b| + 1e-5 |b| for EVERY entry --
    the infinity-norm check rel_err(a, b) < 1e-5 beside it leaves the small entries of a tensor unconstrained"""
    b = torch.as_tensor(b)
    assert_close(a, b, rtol=1e-5, atol=1e-6 * float(b.detach().abs().max()) if b.numel() else 0., what=what)


# ------------------------------------------------------------------------------- fp64 arbitration of knife-edge decisions
# A ReLU gate / max-pool argmax decided on a quantity that lies within fp32 roundoff of the tie may legitimately differ
# between two correct fp32 evaluations (different summation orders).  Instead of a loose blanket tolerance on the gradients,
# the parity tests find those units with an fp64 evaluation, give them a ZERO cotangent (so they contribute to no gradient on
# either side), bound their number, and compare everything else at the normal tolerance.
EPS32 = float(torch.finfo(torch.float32).eps)


def conv_front_knife_images(e, W1, b1, W2, b2, ulps=16.):
    """[n] bool: images of the MNISTCNN conv front (MLP.py:36-41) holding a decision within `ulps` fp32 ulps OF ITS TERMS'
    MAGNITUDE of a tie in an fp64 evaluation: a conv1 pre-activation near 0, or a pool window in which some conv2 output is
    closer than that to the maximum without being exactly equal to it (exact ties -- constant image regions -- are decided by
    the first-maximum rule on both sides).  Also returns the per-image counts (relu, pool)."""
    import torch.nn.functional as F
    e, W1, b1, W2, b2 = [t.detach().cpu().double() for t in (e, W1, b1, W2, b2)]
    img = e.view(-1, 1, 28, 28)
    pre1 = F.conv2d(img, W1.view(16, 1, 3, 3), b1)
    mag1 = F.conv2d(img.abs(), W1.view(16, 1, 3, 3).abs(), b1.abs())
    relu_k = (pre1.abs() < ulps * EPS32 * mag1) & (pre1 != 0)
    a1 = torch.relu(pre1)
    c2 = F.conv2d(a1, W2.view(16, 16, 3, 3), b2)
    mag2 = F.conv2d(a1, W2.view(16, 16, 3, 3).abs(), b2.abs())
    n = e.shape[0]
    win = c2.view(n, 16, 12, 2, 12, 2).permute(0, 1, 2, 4, 3, 5).reshape(n, 2304, 4)
    wmag = mag2.view(n, 16, 12, 2, 12, 2).permute(0, 1, 2, 4, 3, 5).reshape(n, 2304, 4).amax(2)
    # ANY entry within the bound of the maximum without being equal to it (not only the runner-up: with an exact tie of two
    # entries at the top, a third one 1e-8 below is rounded onto them by one fp32 evaluation order and not by another --
    # found by tests/fuzz_sparse_grad.py, where the sparse and the dense kernels then chose differently)
    gap = win.amax(2, keepdim=True) - win
    pool_k = ((gap < ulps * EPS32 * wmag.unsqueeze(2)) & (gap != 0)).any(2)
    nr, npool = relu_k.flatten(1).sum(1), pool_k.sum(1)
    return (nr + npool) > 0, nr, npool


def integrand_knife_elements(x, h, layers, nb_steps, ulps=16.):
    """[B, d] bool: elements of a Monotonic normalizer (MonotonicNormalizer.py:12-38, 51-66) with a hidden ReLU
    pre-activation within `ulps` fp32 ulps of its terms' magnitude of zero at any quadrature node or at x itself, in an
    fp64 evaluation of the integrand net"""
    from oracle import gnf_oracle as O
    _, t = O.cc_rule(nb_steps)
    x, h = x.detach().cpu().double(), h.detach().cpu().double()
    layers = [(W.detach().cpu().double(), b.detach().cpu().double()) for W, b in layers]
    B, d = x.shape
    knife = torch.zeros(B, d, dtype=torch.bool)
    nodes = [x * (float(tk) + 1.) / 2. for tk in t] + [x]
    for xk in nodes:
        a = torch.cat((xk.reshape(B, d, 1), h), 2).reshape(B * d, -1)
        for W, b in layers[:-1]:
            pre = a @ W.t() + b
            mag = a.abs() @ W.abs().t() + b.abs()
            knife |= ((pre.abs() < ulps * EPS32 * mag) & (pre != 0)).any(1).view(B, d)
            a = torch.relu(pre)
    return knife
