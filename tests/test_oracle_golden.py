"""CPU: the oracle restatement vs golden vectors generated from the reference itself
(tests/golden/make_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest
import torch

from conftest import load_golden, params_of, linear_layers, rel_err
from oracle import gnf_oracle as O

TOL = 2e-6      # same torch CPU kernels, same op order -> expected ~1e-7


def req(t):
    return t.clone().requires_grad_(True)


def test_normal_log_density():
    g = load_golden("normal_log_density")
    assert rel_err(O.normal_log_density(g["z"]), g["out"]) < TOL


def test_affine_forward_backward_inverse():
    g = load_golden("affine")
    x, h = req(g["x"]), req(g["h"])
    z, jac = O.affine_forward(x, h)
    assert rel_err(z, g["z"]) < TOL and rel_err(jac, g["jac"]) < TOL
    gx, gh = torch.autograd.grad((z * g["gz"]).sum(), (x, h), retain_graph=True)
    assert rel_err(gx, g["gx_from_z"]) < TOL and rel_err(gh, g["gh_from_z"]) < TOL
    gh2, = torch.autograd.grad((torch.log(jac).sum(1) * g["gld"]).sum(), h)
    assert rel_err(gh2, g["gh_from_logdet"]) < TOL
    assert rel_err(O.affine_inverse(g["z"], g["h"]), g["x_inverse"]) < TOL


def test_made_masks_and_prior():
    g = load_golden("masks_prior")
    for tag in "abc":
        cfg = g[tag + "_cfg"].tolist()
        nin, hs, hidden = cfg[0], cfg[1], cfg[2:]
        masks = O.made_masks(nin, hidden, nin * hs)
        for li, m in enumerate(masks):
            assert np.array_equal(m, g["%s_net.%d.mask" % (tag, 2 * li)].numpy())
    for k in (1, 2):
        A = O.mnist_a_prior(28, k)
        idx = torch.nonzero(A).int()
        assert torch.equal(idx, g["A_prior_28_%d_idx" % k])
    assert torch.equal(O.mnist_a_prior(6, 1), g["A_prior_6_1"])


def test_coupling():
    g = load_golden("coupling")
    p = {k: req(v) for k, v in params_of(g).items()}
    x = req(g["x"])
    h = O.coupling_forward(x, p["constants"], linear_layers(p, "embeding_net."))
    assert rel_err(h, g["h"]) < TOL
    (h * g["gh"]).sum().backward()
    assert rel_err(x.grad, g["gx"]) < TOL
    for k, v in p.items():
        assert rel_err(v.grad, g["g." + k]) < TOL, k


def test_autoregressive():
    g = load_golden("autoregressive")
    p = params_of(g)
    pre = "masked_autoregressive_net."
    layers = [(req(W), req(b)) for W, b in linear_layers(p, pre)]
    masks = [p["%snet.%d.mask" % (pre, 2 * i)] for i in range(len(layers))]
    ours = O.made_masks(7, [16, 12, 16], 21)
    for m, mo in zip(masks, ours):
        assert np.array_equal(m.numpy(), mo)
    x = req(g["x"])
    h = O.made_forward(x, layers, masks)
    assert rel_err(h, g["h"]) < TOL
    (h * g["gh"]).sum().backward()
    assert rel_err(x.grad, g["gx"]) < TOL
    for i, (W, b) in enumerate(layers):
        assert rel_err(W.grad, g["g.%snet.%d.weight" % (pre, 2 * i)]) < TOL
        assert rel_err(b.grad, g["g.%snet.%d.bias" % (pre, 2 * i)]) < TOL


@pytest.mark.parametrize("tag", ["det", "gumbel", "gumbel_hot_T05", "hard", "hard_gumbel"])
def test_dag_conditioner(tag):
    g = load_golden("dag_" + tag)
    hot, stoch, hth, T = g["flags"].tolist()
    p = params_of(g)
    A = req(p["A"])
    layers = [(req(W), req(b)) for W, b in linear_layers(p, "embedding_net.")]
    x = req(g["x"])
    e = O.dag_masked_inputs(x, A, True, hth, bool(stoch), False, T, g["u1"], g["u2"], None, bool(hot))
    h = O.mlp(e, layers).view(x.shape[0], x.shape[1], -1)
    assert rel_err(h, g["h"]) < TOL
    loss = O.dag_loss(A, p["alpha"], int(g["exponent"]), p["lambd"], p["c"], p["dag_const"], p["l1_weight"])
    assert rel_err(loss, g["loss"]) < TOL
    assert rel_err(O.dag_power_trace(A, p["alpha"], int(g["exponent"])), g["trace"]) < TOL
    ((h * g["gh"]).sum() + loss).backward()
    assert rel_err(x.grad, g["gx"]) < 5e-6
    assert rel_err(A.grad, g["g.A"]) < 5e-6
    for i, (W, b) in enumerate(layers):
        assert rel_err(W.grad, g["g.embedding_net.net.%d.weight" % (2 * i)]) < 5e-6


def test_dag_trace84():
    g = load_golden("dag_trace84")
    assert rel_err(O.dag_power_trace(g["A"], g["alpha"], int(g["exponent"])), g["trace"]) < TOL


def test_mnistcnn():
    g = load_golden("mnistcnn")
    p = {k: req(v) for k, v in params_of(g).items()}
    e = req(g["e"])
    out = O.mnistcnn_forward(e, p)
    assert rel_err(out, g["out"]) < TOL
    (out * g["gout"]).sum().backward()
    assert rel_err(e.grad, g["ge"]) < TOL
    for k, v in p.items():
        assert rel_err(v.grad, g["g." + k]) < 5e-6, k


def test_integrand_and_monotonic_jacobian():
    g = load_golden("integrand")
    p = params_of(g)
    layers = [(req(W), req(b)) for W, b in linear_layers(p, "")]
    x, h = req(g["x"]), req(g["h"])
    jac = O.integrand(x, h, layers)
    assert rel_err(jac, g["jac"]) < TOL
    (torch.log(jac) * g["gj"]).sum().backward()
    assert rel_err(x.grad, g["gx"]) < TOL and rel_err(h.grad, g["gh"]) < TOL
    for i, (W, b) in enumerate(layers):
        assert rel_err(W.grad, g["g.net.%d.weight" % (2 * i)]) < TOL


def _flow_steps(p, n_steps, kind, u=None, T=1., hot=False):
    steps = []
    for s in range(n_steps):
        pre = "steps.%d.conditioner." % s
        if kind == "coupling":
            cond = (lambda x, pre=pre: O.coupling_forward(x, p[pre + "constants"],
                                                          linear_layers(p, pre + "embeding_net.")))
        elif kind == "made":
            ls = linear_layers(p, pre + "masked_autoregressive_net.")
            ms = [p["%smasked_autoregressive_net.net.%d.mask" % (pre, 2 * i)] for i in range(len(ls))]
            cond = (lambda x, ls=ls, ms=ms: O.made_forward(x, ls, ms))
        else:
            def cond(x, pre=pre, s=s):
                e = O.dag_masked_inputs(x, p[pre + "A"], True, 0., True, False, T, u["u1_%d" % s], u["u2_%d" % s],
                                        None, hot)
                return O.mlp(e, linear_layers(p, pre + "embedding_net.")).view(x.shape[0], x.shape[1], -1)
        steps.append((cond, O.affine_forward))
    return steps


@pytest.mark.parametrize("name,kind,n", [("flow_affine_coupling_1", "coupling", 1),
                                         ("flow_affine_coupling_3", "coupling", 3),
                                         ("flow_affine_made_1", "made", 1),
                                         ("flow_affine_dag_2", "dag", 2)])
def test_flows(name, kind, n):
    g = load_golden(name)
    p = {k: (req(v) if v.dtype.is_floating_point and not k.endswith("mask") else v) for k, v in params_of(g).items()}
    x = req(g["x"])
    z, ld = O.fc_flow_forward(x, _flow_steps(p, n, kind, g, T=.5, hot=True))
    assert rel_err(z, g["z"]) < TOL and rel_err(ld, g["logdet"]) < TOL
    closs = 0.
    if kind == "dag":
        for s in range(n):
            pre = "steps.%d.conditioner." % s
            closs = closs + O.dag_loss(p[pre + "A"], p[pre + "alpha"], 6 % 50, p[pre + "lambd"], p[pre + "c"],
                                       p[pre + "dag_const"], p[pre + "l1_weight"])
    loss = O.flow_loss(z, ld, closs)
    assert rel_err(loss, g["loss"]) < TOL
    loss.backward()
    assert rel_err(x.grad, g["gx"]) < 5e-6
    for k in g:
        if k.startswith("g."):
            assert rel_err(p[k[2:]].grad, g[k]) < 1e-5, k


@pytest.mark.parametrize("name,kind,depth", [("flow_affine_coupling_1", "coupling", 1),
                                             ("flow_affine_made_1", "made", 7)])
def test_flow_inverse_nb_flow_1(name, kind, depth):
    g = load_golden(name)
    gi = load_golden(name + "_inv")
    p = params_of(g)
    cond, _ = _flow_steps(p, 1, kind)[0]
    with torch.no_grad():
        x = O.step_invert(gi["z"], cond, O.affine_inverse, depth)
    assert rel_err(x, gi["x"]) < TOL
    assert rel_err(x, g["x"]) < 1e-4        # true round trip


def test_mnist_affine_dag_flow():
    g = load_golden("flow_mnist_affine_dag")
    p = params_of(g)
    A = req(O.mnist_a_prior(28, 2))
    cnn = {k[len("steps.0.conditioner.embedding_net."):]: v for k, v in p.items() if "embedding_net." in k}
    x = g["x"]
    torch.manual_seed(int(g["gate_seed"]))
    u1 = torch.rand(2, 784, 784)
    u2 = torch.rand(2, 784, 784)

    def cond(xx):
        e = O.dag_masked_inputs(xx, A, True, 0., True, False, 1., u1, u2, None, False)
        return O.mnistcnn_forward(e, cnn).view(2, 784, -1)
    z, ld = O.fc_flow_forward(x, [(cond, O.affine_forward)])
    assert rel_err(z, g["z"]) < TOL and rel_err(ld, g["logdet"]) < TOL
    pre = "steps.0.conditioner."
    closs = O.dag_loss(A, p[pre + "alpha"], 784 % 50, p[pre + "lambd"], p[pre + "c"], p[pre + "dag_const"],
                       p[pre + "l1_weight"])
    loss = O.flow_loss(z, ld, closs)
    assert rel_err(loss, g["loss"]) < TOL
    loss.backward()
    idx = g["gA_idx"].long()
    assert rel_err(A.grad[idx[:, 0], idx[:, 1]], g["gA_val"]) < 1e-5
    assert int((A.grad != 0).sum()) == idx.shape[0]      # zero entries of A get exactly zero gradient


def test_mnist_affine_dag_flow_frozen_gate():
    """the same flow after the DAG phase (binary frozen A, deterministic product): reference z / log-det / loss and the
    embedding-net gradients -- the case the sparse masked-image kernels cover"""
    g0, g = load_golden("flow_mnist_affine_dag"), load_golden("flow_mnist_affine_dag_frozen")
    p = params_of(g0)
    A = O.mnist_a_prior(28, 2)
    pre = "steps.0.conditioner.embedding_net."
    cnn = {k[len(pre):]: v.clone().requires_grad_(True) for k, v in p.items() if "embedding_net." in k}

    def cond(xx):
        e = O.dag_masked_inputs(xx, A, False, 0., False, False, 1., None, None, None, False)
        return O.mnistcnn_forward(e, cnn).view(2, 784, -1)
    z, ld = O.fc_flow_forward(g0["x"], [(cond, O.affine_forward)])
    assert rel_err(z, g["z"]) < TOL and rel_err(ld, g["logdet"]) < TOL
    c = "steps.0.conditioner."
    closs = O.dag_loss(A, p[c + "alpha"], 784 % 50, p[c + "lambd"], p[c + "c"], p[c + "dag_const"], p[c + "l1_weight"])
    loss = O.flow_loss(z, ld, closs)
    assert rel_err(loss, g["loss"]) < TOL
    loss.backward()
    n = 0
    for k, v in g.items():
        if k.startswith("g.") or k.startswith("g8."):
            got = cnn[k.split("embedding_net.")[1]].grad
            assert rel_err(got[:8] if k.startswith("g8.") else got, v) < 1e-4, k
            n += 1
    assert n == 8


def test_monotonic_flow_jacobian():
    g = load_golden("flow_mono_made_1")
    p = params_of(g)
    ls = linear_layers(p, "steps.0.conditioner.masked_autoregressive_net.")
    ms = [p["steps.0.conditioner.masked_autoregressive_net.net.%d.mask" % (2 * i)] for i in range(len(ls))]
    h = O.made_forward(g["x"], ls, ms)
    assert rel_err(h, g["h"]) < TOL
    jac = O.integrand(g["x"], h, linear_layers(p, "steps.0.normalizer.integrand_net."))
    assert rel_err(jac, g["jac"]) < TOL
    assert rel_err(torch.log(jac).sum(1), g["logdet"]) < TOL
