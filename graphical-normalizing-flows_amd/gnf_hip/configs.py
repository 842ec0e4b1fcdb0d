"""The five BASELINE.json configurations as (flow, synthetic batch) builders -- what bench.py (cfg4, the headline), the
-m gpu configuration tests and the tools/ drivers all construct.  Model arguments follow the reference's drivers
(ToyExperiments.py:23,44; UCIExperimentsConfigurations.yml:1-14,347-358; ImageExperiments.py:146-153,371), inputs
SURVEY.md 8(d)."""
import torch

D = 784
INT_NET = [50, 50, 50]


def pseudo_mnist(gen, B, d):
    """logit-space pseudo-MNIST: reference transform (lib/transform.py:5-20) applied to a
    synthetic pixel law (p=0 w.p. 0.8 else U{1..255}) -- real MNIST is not available."""
    p = torch.where(torch.rand(B, d, generator=gen) < .8, torch.zeros(B, d),
                    torch.randint(1, 256, (B, d), generator=gen).float())
    y = (p + torch.rand(B, d, generator=gen)) / 256.
    y = 1e-6 + (1 - 2e-6) * y
    return torch.log(y) - torch.log(1 - y)


def build_cfg4_flow():
    from models import MonotonicNormalizer
    from models.NormalizingFlowFactories import buildMNISTNormalizingFlow
    torch.manual_seed(0)
    return buildMNISTNormalizingFlow([1], MonotonicNormalizer,
                                     {"integrand_net": INT_NET, "nb_steps": 15, "solver": "CC"}, l1=0.,
                                     nb_epoch_update=10, hot_encoding=False, prior_kernel=2)


def baseline_config(name, device="cuda:0"):
    """(flow, x) of a BASELINE.json configuration (cfg1..cfg5; cfg4det / cfg4dag: cfg4 in the states the
    DAG phase ends in) on `device`, synthetic inputs per SURVEY.md 8(d)."""
    from models import (buildFCNormalizingFlow, CouplingConditioner, AutoregressiveConditioner, DAGConditioner,
                        AffineNormalizer, MonotonicNormalizer)
    g = torch.Generator().manual_seed(1234)
    torch.manual_seed(0)
    if name == "cfg1":      # toy 8gaussians (lib/toy_data.py:81-98 restated), Affine+Coupling
        B = 512
        ang = torch.randint(0, 8, (B,), generator=g).float() * (3.141592653589793 / 4)
        x = (torch.stack((torch.cos(ang), torch.sin(ang)), 1) * 4 + torch.randn(B, 2, generator=g) * .5) / 1.414
        f = buildFCNormalizingFlow(1, CouplingConditioner, {"in_size": 2, "hidden": [150, 150], "out_size": 150},
                                   AffineNormalizer, {})
    elif name == "cfg2":    # POWER d=6 Monotonic+DAG (UCIExperimentsConfigurations.yml:1-14, UCI:83-93)
        x = torch.randn(10000, 6, generator=g)
        f = buildFCNormalizingFlow(1, DAGConditioner, {"in_size": 6, "hidden": [60, 60, 60], "out_size": 30, "l1": 0.,
                                                       "gumble_T": .5, "nb_epoch_update": 30, "hot_encoding": True},
                                   MonotonicNormalizer, {"integrand_net": [100, 100, 100], "cond_size": 30,
                                                         "nb_steps": 20, "solver": "CC"})
    elif name == "cfg3":    # MNIST d=784 Affine+Autoregressive 1024^3
        x = pseudo_mnist(g, 100, 784)
        f = buildFCNormalizingFlow(1, AutoregressiveConditioner, {"in_size": 784, "hidden": [1024] * 3, "out_size": 2},
                                   AffineNormalizer, {})
    elif name == "cfg4":
        x = pseudo_mnist(g, 100, 784)
        f = build_cfg4_flow()
    elif name == "cfg4det":  # cfg4 after the DAG phase: post_process() froze a binary A, the gate is deterministic
        x = pseudo_mnist(g, 100, 784)
        f = build_cfg4_flow()
        for c in f.getConditioners():
            with torch.no_grad():
                c.post_process(zero_threshold=.1)
    elif name == "cfg4dag":  # the state update_dual_param() ends in: an acyclic binary A (window parents that precede the
        x = pseudo_mnist(g, 100, 784)          # pixel in raster order), post-processed, dag_const = l1 = 0
        f = build_cfg4_flow()
        for c in f.getConditioners():
            with torch.no_grad():
                idx = torch.arange(784)
                c.A.mul_((idx[None, :] < idx[:, None]).float())
                c.post_process(zero_threshold=.1)
                c.dag_const = torch.tensor(0.)
                c.l1_weight = torch.tensor(0.)
                c.is_invertible = True
    elif name == "cfg5":    # BSDS300 d=63 synthetic (yml:347-358), B=50000
        x = torch.randn(50000, 63, generator=g)
        f = buildFCNormalizingFlow(1, AutoregressiveConditioner, {"in_size": 63, "hidden": [630] * 3, "out_size": 30},
                                   MonotonicNormalizer, {"integrand_net": [150, 150, 150], "cond_size": 30,
                                                         "nb_steps": 20, "solver": "CCParallel"})
    else:
        raise KeyError(name)
    return f.to(device), x.to(device)
