"""train_image.py, the MI355X re-hosting of the reference's ImageExperiments.py (the caller of the headline
configuration).  CPU part: the data path (IDX reader, dequantisation + logit, bits per pixel), the argument surface and
the checkpoint key convention.  GPU part (-m gpu): one epoch of the headline model on synthetic digits, gradient
accumulation against the single-batch step, and the driver on two ranks."""
import gzip
import math
import os
import struct
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, assert_close

PKG = os.path.join(ROOT, "graphical-normalizing-flows_amd")
if PKG not in sys.path:
    sys.path.insert(0, PKG)


def _driver():
    import train_image
    return train_image


def _write_idx(path, arr, gz=False):
    arr = np.ascontiguousarray(arr, dtype=np.uint8)
    raw = struct.pack(">HBB", 0, 0x08, arr.ndim) + struct.pack(">" + "I" * arr.ndim, *arr.shape) + arr.tobytes()
    with (gzip.open if gz else open)(path, "wb") as f:
        f.write(raw)


def test_idx_reader_and_mnist_splits(tmp_path):
    T = _driver()
    g = np.random.default_rng(0)
    raw = tmp_path / "MNIST" / "raw"
    raw.mkdir(parents=True)
    pix, lab = g.integers(0, 256, (60000, 28, 28), dtype=np.uint8), g.integers(0, 10, 60000, dtype=np.uint8)
    tpix, tlab = g.integers(0, 256, (300, 28, 28), dtype=np.uint8), g.integers(0, 10, 300, dtype=np.uint8)
    _write_idx(raw / "train-images-idx3-ubyte", pix)
    _write_idx(raw / "train-labels-idx1-ubyte", lab)
    _write_idx(raw / "t10k-images-idx3-ubyte.gz", tpix, gz=True)           # torchvision keeps either form
    _write_idx(raw / "t10k-labels-idx1-ubyte.gz", tlab, gz=True)
    assert torch.equal(T.read_idx(str(raw / "t10k-images-idx3-ubyte.gz")), torch.from_numpy(tpix))
    trn, val, tst = T.load_mnist(str(tmp_path), "MNIST", torch.Generator().manual_seed(1))
    assert trn.shape == (50000, 784) and val.shape == (10000, 784) and tst.shape == (300, 784)     # reference :46
    assert trn.dtype == torch.uint8
    both = torch.cat([trn, val]).to(torch.int64).sum(1).sort().values                # a permutation of the file
    assert torch.equal(both, torch.from_numpy(pix.reshape(60000, -1).astype(np.int64).sum(1)).sort().values)
    trn3, val3, tst3 = T.load_mnist(str(tmp_path), "MNIST3", torch.Generator().manual_seed(1))
    n3 = int((lab == 3).sum())
    assert trn3.shape[0] == 5000 and val3.shape[0] == n3 - 5000 and tst3.shape[0] == int((tlab == 3).sum())   # :66
    with pytest.raises(FileNotFoundError, match="synthetic"):
        T.load_mnist(str(tmp_path / "nowhere"), "MNIST", torch.Generator())
    bad = tmp_path / "bad"
    bad.write_bytes(struct.pack(">HBB", 0, 0x0D, 1) + struct.pack(">I", 4) + b"abcd")
    with pytest.raises(ValueError, match="unsigned-byte"):
        T.read_idx(str(bad))


def test_dequantisation_and_bits_per_pixel():
    """lib/transform.py:5-21 and ImageExperiments.py:33-37 restated; checked against an independent fp64 evaluation"""
    T = _driver()
    g = torch.Generator().manual_seed(0)
    u8 = torch.randint(0, 256, (64, 784), generator=g).to(torch.uint8)
    u8[0] = 0
    u8[1] = 255
    alpha = 1e-6
    x = T.dequantise(u8, alpha, g)
    assert torch.isfinite(x).all()
    back = T.logit_back(x, alpha) * 256.                       # pixel + noise again
    assert ((back - u8.float()) > -1e-3).all() and ((back - u8.float()) < 1. + 1e-3).all()
    ll = torch.randn(64, generator=g) * 50 - 1500
    bpp = T.compute_bpp(ll, x, alpha)
    xd = x.double().numpy()
    s = 1. / (1. + np.exp(-xd))
    want = (-ll.double().numpy() / (784 * np.log(2.)) - np.log2(1. - 2. * alpha) + 8.
            + (np.log2(s) + np.log2(1. - s)).sum(1) / 784)
    # fp32 like the reference's own expression: `1 - sigmoid(x)` cancels for a saturated (all-255) row, 7e-5 bpp there
    assert_close(bpp, torch.from_numpy(want).float(), rtol=2e-5, atol=2e-4, what="bpp")
    assert_close(bpp[2:], torch.from_numpy(want).float()[2:], rtol=2e-5, atol=2e-5, what="bpp, unsaturated rows")
    # uniform data under the exact change of variables: a flow that assigns log p(x) = sum log(s (1-s)) / (1-2a) per
    # pixel (the logit Jacobian of a uniform density on [0,1]) costs exactly 8 bits per pixel
    ll_uniform = (torch.log(torch.sigmoid(x)) + torch.log(1 - torch.sigmoid(x))).sum(1) - 784 * math.log(1 - 2 * alpha)
    assert_close(T.compute_bpp(ll_uniform, x, alpha), torch.full((64,), 8.), rtol=1e-5, atol=1e-4, what="uniform = 8 bpp")


def test_arguments_are_the_references():
    """names and defaults of ImageExperiments.py:361-384"""
    T = _driver()
    a = T.parse([])
    want = dict(load=False, nb_steps_dual=100, l1=10., nb_epoch=10000, b_size=1, int_net=[50, 50, 50], nb_steps=20,
                f_number=None, solver="CC", nb_flow=[1], test=False, weight_decay=1e-5, learning_rate=1e-3,
                batch_per_optim_step=1, nb_gpus=1, dataset="MNIST", normalizer="Affine", no_hot_encoding=False,
                prior_A_kernel=None, conditioner="DAG", emb_net=[100, 100, 100, 10])
    for k, v in want.items():
        assert getattr(a, k) == v, k
    assert a.folder.startswith("MNIST" + os.sep)                 # <dataset>/<time stamp> when -folder is empty (:388)
    a = T.parse("-dataset MNIST1 -normalizer Monotonic -no_hot_encoding -prior_A_kernel 2 -nb_flow 1 1 1 -b_size 100 "
                "-int_net 50 50 50 -folder out".split())
    assert a.dataset == "MNIST1" and a.nb_flow == [1, 1, 1] and a.prior_A_kernel == 2 and a.folder == "out"


def test_checkpoint_keys_and_model_construction():
    T = _driver()
    from models import MonotonicNormalizer, DAGConditioner
    a = T.parse("-normalizer Monotonic -no_hot_encoding -prior_A_kernel 2".split())
    model, cond_t, norm_t = T.build(a)
    assert cond_t is DAGConditioner and norm_t is MonotonicNormalizer
    assert model.getNormalizers()[0].nb_steps == 15            # :146, overwritten per iteration by the loop
    assert model.getConditioners()[0].nb_epoch_update == 100
    sd = model.state_dict()
    w = T.wrapped_keys(sd)
    assert all(k.startswith("module.") for k in w) and set(T.plain_keys(w)) == set(sd) == set(T.plain_keys(sd))
    a = T.parse("-conditioner Coupling -emb_net 32 32 6 -normalizer Monotonic".split())
    model, _, _ = T.build(a)
    assert model.getNormalizers()[0].integrand_net.net[0].in_features == 1 + 6
    with pytest.raises(SystemExit, match="1 or 3"):
        T.build(T.parse("-nb_flow 1 1".split()))
    with pytest.raises(SystemExit, match="CIFAR10"):
        T.train(T.parse("-dataset CIFAR10 -folder /tmp/none".split()))


def test_sample_grid_file(tmp_path):
    T = _driver()
    x = torch.linspace(0, 1, 16 * 784).view(16, 784)
    T.write_pgm(str(tmp_path / "g.pgm"), x)
    raw = (tmp_path / "g.pgm").read_bytes()
    assert raw.startswith(b"P5\n112 112\n255\n") and len(raw) == len(b"P5\n112 112\n255\n") + 112 * 112


# ----------------------------------------------------------------------------- GPU
DEV = "cuda:0"


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.gpu
def test_gradient_accumulation_equals_the_single_batch_step():
    """k micro-batches at scale 1/k + one apply_step == one step on their concatenation (equal sizes; the constraint
    term is counted once either way) -- ImageExperiments.py:205-213"""
    from gnf_hip import dp
    from models import buildFCNormalizingFlow, CouplingConditioner, MonotonicNormalizer
    def make():
        torch.manual_seed(3)
        return buildFCNormalizingFlow(2, CouplingConditioner, {"in_size": 6, "hidden": [16, 16], "out_size": 4},
                                      MonotonicNormalizer, {"integrand_net": [12, 12], "cond_size": 4, "nb_steps": 10,
                                                            "solver": "CC"}).to(DEV)
    x = torch.randn(8, 6, device=DEV)
    fa, fb = make(), make()
    sa, sb = dp.FlatState(fa), dp.FlatState(fb)
    for _ in range(3):
        la = dp.train_step(fa, sa, x, graph=False)
        lb = sum(dp.accumulate(fb, x[4 * i:4 * i + 4], .5) for i in range(2))
        dp.apply_step(sb)
        assert_close(lb.detach(), la.detach(), rtol=1e-5, atol=1e-6, what="loss")
    assert_close(sb.flat, sa.flat, rtol=1e-5, atol=1e-6, what="parameters after 3 steps")
    assert sa.t == sb.t == 3


@pytest.mark.gpu
def test_one_epoch_of_the_headline_model_on_synthetic_digits(tmp_path):
    T = _driver()
    run = tmp_path / "run"
    a = T.parse(("-dataset MNIST -data_root synthetic -normalizer Monotonic -no_hot_encoding -prior_A_kernel 2 -b_size 4 "
                 "-nb_epoch 2 -max_batches 3 -nb_steps_dual 1 -l1 0.1 -folder %s" % run).split())
    model = T.train(a)
    lines = open(run / "logs").read().splitlines()
    ep = [l for l in lines if l.startswith("epoch:") and "Train loss" in l]
    assert len(ep) == 2 and all("Valid BPP" in l for l in ep)
    vals = [float(l.split("Valid BPP ")[1].split(" ")[0]) for l in ep]
    assert all(math.isfinite(v) for v in vals)
    assert any("Threshold: 0.950000" in l for l in lines)         # the sweep of epoch 0 (:258-286)
    sd = torch.load(run / "model.pt", map_location="cpu")
    assert all(k.startswith("module.") for k in sd)
    assert os.path.exists(run / "ADAM.pt") and os.path.exists(run / "model_0.pt")
    for c in model.getConditioners():
        assert c.stoch_gate and c.h_thresh == 0.                  # sweep restored the gate settings
    # once post-processing has frozen a DAG the driver samples (:322-330): 16 images at five temperatures, round trip
    with torch.no_grad():
        for c in model.getConditioners():
            c.A.copy_(torch.tril(c.A, -1))                         # the window prior is symmetric: keep one direction
            c.post_process(.1)
            c.is_invertible = True                                 # what update_dual_param() concludes one epoch later
    assert model.isInvertible()
    said = []
    T.sample_grids(model, torch.device(DEV), 1e-6, 7, str(run), said.append)
    assert len(said) == 5 and all(math.isfinite(float(l.split("| ")[-1])) for l in said), said
    # (after six optimiser steps z = 0 need not be reachable inside the bisection bracket [-20, 20]; codes of real
    # inputs are: the inverse must return them)
    with torch.no_grad():
        x = T.dequantise(torch.randint(0, 256, (4, 784), device=DEV).to(torch.uint8), 1e-6,
                         torch.Generator(device=DEV).manual_seed(0))
        assert (model.invert(model(x)[0]) - x).abs().max().item() < 2e-3
    assert os.path.getsize(run / ("images_7_%f.pgm" % .25)) == len(b"P5\n112 112\n255\n") + 112 * 112
    # resume: the reference's -load path (:170-183) reads those files back
    b = T.parse(("-load -dataset MNIST -data_root synthetic -normalizer Monotonic -no_hot_encoding -prior_A_kernel 2 "
                 "-b_size 4 -nb_epoch 1 -max_batches 1 -folder %s" % run).split())
    T.train(b)


@pytest.mark.gpu
def test_image_driver_on_two_ranks(tmp_path):
    """torch.distributed.run, two ranks (gloo transport on a one-GPU box), gradient accumulation over two batches"""
    env = dict(os.environ, GNF_DIST_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    run = tmp_path / "run"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(PKG, "train_image.py"), "-dataset", "MNIST",
           "-data_root", "synthetic", "-normalizer", "Affine", "-no_hot_encoding", "-prior_A_kernel", "2", "-b_size", "2",
           "-nb_epoch", "2", "-max_batches", "4", "-batch_per_optim_step", "2", "-nb_steps_dual", "1", "-l1", "0.1",
           "-folder", str(run)]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    lines = [l for l in open(run / "logs") if l.startswith("epoch:") and "Train loss" in l]
    assert len(lines) == 2
    sd = torch.load(run / "model.pt", map_location="cpu")
    assert float(sd["module.steps.0.conditioner.lambd"]) > 0      # the dual update ran on identical replicas
