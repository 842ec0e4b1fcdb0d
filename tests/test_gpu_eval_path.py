"""The reference's EVALUATION path (-m gpu): every reported likelihood is computed with far more quadrature nodes than a
training step uses -- nb_steps = 150 (ImageExperiments.py:232-243), nb_steps + 20 (UCIExperiments.py:154-162), 250
(ImageExperimentsTest.py:192-195) -- under torch.no_grad().  The kernels take the node count at run time; these tests
pin z / jac / log|det J| / NLL against the oracle at those counts for the three integrand widths of BASELINE.json
([50]^3 peeled kernels, [100]^3 and [150]^3 pair-major LDS kernels), the bisection inverse at S = 150, and the cfg4
composite (MNIST d = 784, DAG(MNISTCNN) + Monotonic) at S = 150 with a deterministic and an injected-Gumbel gate.
Monotonic z / NLL: UMNN 1.0 parity unpinned (the oracle restates the rule)."""
import sys

import pytest
import torch

from conftest import ROOT, rel_err, assert_close, assert_fwd
from oracle import gnf_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-5


def _case(B, d, c, hidden, S, seed):
    from models import MonotonicNormalizer
    torch.manual_seed(seed)
    norm = MonotonicNormalizer(hidden, c, nb_steps=20, solver="CC")
    x = torch.randn(B, d) * 1.5
    h = torch.randn(B, d, c)
    ps = [p.detach().clone() for p in norm.integrand_net.flat_params()]
    layers = [(ps[i], ps[i + 1]) for i in range(0, len(ps), 2)]
    return norm, x, h, layers


@pytest.mark.parametrize("S", [35, 49, 150, 250])
@pytest.mark.parametrize("hidden", [[50, 50, 50], [100, 100, 100], [150, 150, 150]])
def test_monotonic_eval_node_counts_vs_oracle(hidden, S):
    B, d, c = 11, 9, 30                                 # 99 elements: ragged against the 16- and 32-element groups
    norm, x, h, layers = _case(B, d, c, hidden, S, seed=hidden[0] + S)
    with torch.no_grad():
        z0, j0 = O.monotonic_forward(x, h, layers, S)
        # what fp32 roundoff alone allows at this node count: the same rule in fp64
        z64, j64 = O.monotonic_forward(x.double(), h.double(), [(W.double(), b.double()) for W, b in layers], S)
        norm = norm.to(DEV)
        norm.nb_steps = S                               # the attribute the drivers poke (Image:235, UCI:157)
        z, jac = norm(x.to(DEV), h.to(DEV))
    assert rel_err(z.cpu(), z0) < TOL and rel_err(jac.cpu(), j0) < TOL
    assert_fwd(z, z0, what='z')
    assert_fwd(jac, j0, what='jac')
    assert_close(z, z0, atol=3e-6, what="z")
    assert_close(jac, j0, what="jac")
    ld0, ld = torch.log(j0).sum(1), torch.log(jac).sum(1)
    assert_close(ld, ld0, what="logdet")
    nll0 = -(ld0 + O.normal_log_density(z0))
    nll = -(ld.cpu() + O.normal_log_density(z.cpu()))
    assert_close(nll, nll0, what="NLL")
    # and the HIP result is as close to the fp64 rule as the fp32 oracle is (within a factor and a floor)
    e_hip = (z.cpu().double() - z64).abs().max().item()
    e_ref = (z0.double() - z64).abs().max().item()
    assert e_hip <= 4 * e_ref + 4e-6, (e_hip, e_ref)


@pytest.mark.parametrize("hidden", [[50, 50, 50], [150, 150, 150]])
def test_monotonic_inverse_at_eval_node_count(hidden):
    S = 150
    norm, x, h, layers = _case(7, 5, 30, hidden, S, seed=31 + hidden[0])
    with torch.no_grad():
        z0, _ = O.monotonic_forward(x, h, layers, S)
        x0 = O.monotonic_inverse(z0, h, layers, S)
        norm = norm.to(DEV)
        norm.nb_steps = S
        xi = norm.inverse_transform(z0.to(DEV), h.to(DEV))
        zz, _ = norm(xi, h.to(DEV))
    # the same 20-step bisection on nearly identical z(x): decisions can differ only by one last-step interval
    assert (xi.cpu() - x0).abs().max() <= 40. / 2 ** 20 + 1e-6
    assert (xi.cpu() - x).abs().max() < 1e-3
    assert (zz.cpu() - z0).abs().max() < 2e-4           # bisection resolution 1.9e-5 times the slope


def _bench():
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import bench
    return bench


@pytest.mark.parametrize("gate", ["deterministic", "gumbel"])
def test_cfg4_composite_eval_path_S150(gate):
    """bench.build_flow() (buildMNISTNormalizingFlow([1], Monotonic [50,50,50], prior kernel 2, hot_encoding=False) at
    B = 2, nb_steps = 150, no_grad: z, log|det J|, NLL and loss against the oracle chain."""
    bench = _bench()
    B, S = 2, 150
    flow = bench.build_flow()
    x = bench.pseudo_mnist(torch.Generator().manual_seed(91), B, 784)
    g = torch.Generator().manual_seed(92)
    u1, u2 = torch.rand(B, 784, 784, generator=g), torch.rand(B, 784, 784, generator=g)
    stoch = gate == "gumbel"
    pre, ipre = "steps.0.conditioner.", "steps.0.normalizer.integrand_net.net."
    sd = {k: v.detach().cpu().clone() for k, v in flow.state_dict().items()}
    cnn = {k[len(pre + "embedding_net."):]: v for k, v in sd.items() if "embedding_net." in k}
    layers, k = [], 0
    while ipre + "%d.weight" % k in sd:
        layers.append((sd[ipre + "%d.weight" % k], sd[ipre + "%d.bias" % k]))
        k += 2
    with torch.no_grad():
        e = O.dag_masked_inputs(x, sd[pre + "A"], True, 0., stoch, False, 1., u1, u2, None, False)
        h0 = O.mnistcnn_forward(e, cnn).view(B, 784, -1)
        z0, j0 = O.monotonic_forward(x, h0, layers, S)
        ld0 = torch.log(j0).sum(1)
        closs = O.dag_loss(sd[pre + "A"], sd[pre + "alpha"], 784 % 50, sd[pre + "lambd"], sd[pre + "c"],
                           sd[pre + "dag_const"], sd[pre + "l1_weight"])
        loss0 = O.flow_loss(z0, ld0, closs)

        flow = flow.to(DEV)
        for nrm in flow.getNormalizers():
            nrm.nb_steps = S
        cond = flow.steps[0].conditioner
        cond.stoch_gate = stoch
        if stoch:
            cond.gate_noise = (u1.to(DEV), u2.to(DEV))
        z, ld = flow(x.to(DEV))
        loss = flow.loss(z, ld)
    assert rel_err(z.cpu(), z0) < TOL and rel_err(ld.cpu(), ld0) < TOL and rel_err(loss.cpu(), loss0) < TOL
    assert_fwd(z, z0, what='z')
    assert_fwd(ld, ld0, what='ld')
    assert_fwd(loss, loss0, what='loss')
    assert_close(z, z0, atol=3e-6, what="z")
    assert_close(ld, ld0, what="logdet")
    assert_close(loss, loss0, what="loss")
    nll0 = -(ld0 + O.normal_log_density(z0))
    nll = -(ld + flow.z_log_density(z))
    assert_close(nll, nll0, what="NLL")
