"""Generate golden vectors from the REFERENCE ITSELF (run in the build container only).

    python tests/golden/make_golden.py      # writes tests/golden/*.npz

Imports the reference package from /root/reference (read-only) and records inputs,
parameters, outputs and gradients of every part of the hot path that is importable
(SURVEY.md section 8c).  The third-party `UMNN` package the reference imports at
models/Normalizers/MonotonicNormalizer.py:2 is absent; an import-only placeholder
module (attributes = None, NO arithmetic) is registered so that `import models`
succeeds.  Nothing that touches the UMNN integral is recorded here: Monotonic `z`
is "UMNN parity unpinned" (see oracle/gnf_oracle.py header).

The .npz files are data (inputs + expected outputs); this script is committed next to
them.  /root/reference does not exist on the GPU box, so tests only read the .npz.
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def _import_reference():
    ph = types.ModuleType("UMNN")
    ph.NeuralIntegral = None
    ph.ParallelNeuralIntegral = None
    sys.modules["UMNN"] = ph
    sys.path.insert(0, REF)
    import models  # noqa: F401
    return models


def npy(t):
    return t.detach().cpu().numpy().copy()


def state_np(module, prefix=""):
    return {prefix + k: npy(v) for k, v in module.state_dict().items()}


def save(name, **arrays):
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrays)
    print("wrote", name, {k: getattr(v, "shape", None) for k, v in list(arrays.items())[:6]}, "...")


def logit_mnist_like(gen, B, d):
    """synthetic logit-space pseudo-MNIST (lib/transform.py:5-20 law; pixel law is ours)."""
    p = torch.where(torch.rand(B, d, generator=gen) < .8, torch.zeros(B, d),
                    torch.randint(1, 256, (B, d), generator=gen).float())
    u = torch.rand(B, d, generator=gen)
    y = (p + u) / 256.
    a = 1e-6
    y = a + (1 - 2 * a) * y
    return torch.log(y) - torch.log(1 - y)


def main():
    models = _import_reference()
    from models import (AffineNormalizer, MonotonicNormalizer, CouplingConditioner, AutoregressiveConditioner,
                        DAGConditioner, buildFCNormalizingFlow)
    from models.NormalizingFlowFactories import NormalLogDensity, MNIST_A_prior, buildMNISTNormalizingFlow
    from models.MLP import MNISTCNN
    from models.Normalizers.MonotonicNormalizer import IntegrandNet

    gen = torch.Generator().manual_seed(1234)

    # ---------------------------------------------------------------- (ii) Affine normalizer + clamp boundaries
    B, d, hs = 7, 5, 4
    x = torch.randn(B, d, generator=gen)
    h = torch.randn(B, d, hs, generator=gen) * 3
    h[0, 0, 0], h[0, 1, 0], h[0, 2, 0], h[0, 3, 0] = 5., -5., 9., -9.
    h[1, 0, 1], h[1, 1, 1], h[1, 2, 1], h[1, 3, 1] = 2., -5., 9., -9.
    x_ = x.clone().requires_grad_(True)
    h_ = h.clone().requires_grad_(True)
    norm = AffineNormalizer()
    z, jac = norm(x_, h_ * 1.)          # *1.: clamp_ needs a non-leaf
    gz = torch.randn(B, d, generator=gen)
    gj = torch.randn(B, d, generator=gen)
    (z * gz).sum().backward(retain_graph=True)
    gx_z, gh_z = x_.grad.clone(), h_.grad.clone()
    x_.grad = None
    h_.grad = None
    (torch.log(jac).sum(1) * gj[:, 0]).sum().backward()
    gh_ld = h_.grad.clone()
    xin = norm.inverse_transform(z.detach(), h.clone())
    save("affine", x=npy(x), h=npy(h), z=npy(z), jac=npy(jac), gz=npy(gz), gld=npy(gj[:, 0]), gx_from_z=npy(gx_z),
         gh_from_z=npy(gh_z), gh_from_logdet=npy(gh_ld), x_inverse=npy(xin))

    # ---------------------------------------------------------------- NormalLogDensity
    zz = torch.randn(9, 13, generator=gen) * 2
    save("normal_log_density", z=npy(zz), out=npy(NormalLogDensity()(zz)))

    # ---------------------------------------------------------------- (iv) MADE masks, A prior
    arrays = {}
    for tag, (nin, hid, hsz) in {"a": (5, [8, 8], 2), "b": (7, [16, 12, 16], 3), "c": (63, [64, 70], 30)}.items():
        torch.manual_seed(0)
        c = AutoregressiveConditioner(nin, hid, hsz)
        for k, v in c.state_dict().items():
            if k.endswith("mask"):
                arrays["%s_%s" % (tag, k.replace("masked_autoregressive_net.", ""))] = npy(v)
        arrays[tag + "_cfg"] = np.array([nin, hsz] + hid)
    arrays["A_prior_28_1_idx"] = np.argwhere(npy(MNIST_A_prior(28, 1)) != 0).astype(np.int32)
    arrays["A_prior_28_2_idx"] = np.argwhere(npy(MNIST_A_prior(28, 2)) != 0).astype(np.int32)
    arrays["A_prior_6_1"] = npy(MNIST_A_prior(6, 1))
    save("masks_prior", **arrays)

    # ---------------------------------------------------------------- (i) conditioners -> h, with grads
    # Coupling (toy-like)
    torch.manual_seed(1)
    cc = CouplingConditioner(5, [16, 16], 3)
    x = torch.randn(6, 5, generator=gen).requires_grad_(True)
    hh = cc(x)
    gh = torch.randn(hh.shape, generator=gen)
    (hh * gh).sum().backward()
    arr = state_np(cc, "p.")
    arr.update({"g." + k: npy(p.grad) for k, p in cc.named_parameters()})
    save("coupling", x=npy(x), h=npy(hh), gh=npy(gh), gx=npy(x.grad), **arr)

    # Autoregressive
    torch.manual_seed(2)
    ac = AutoregressiveConditioner(7, [16, 12, 16], 3)
    x = torch.randn(6, 7, generator=gen).requires_grad_(True)
    hh = ac(x)
    gh = torch.randn(hh.shape, generator=gen)
    (hh * gh).sum().backward()
    arr = state_np(ac, "p.")
    arr.update({"g." + k: npy(p.grad) for k, p in ac.named_parameters()})
    save("autoregressive", x=npy(x), h=npy(hh), gh=npy(gh), gx=npy(x.grad), **arr)

    # DAG (MLP embedding; hot-encoding on and off; gate on and off; hard threshold)
    for tag, hot, stoch, hth, T in [("det", False, False, 0., 1.), ("gumbel", False, True, 0., 1.),
                                    ("gumbel_hot_T05", True, True, 0., .5), ("hard", True, False, .5, 1.),
                                    ("hard_gumbel", False, True, .3, .7)]:
        torch.manual_seed(3)
        dc = DAGConditioner(6, [12, 10], 4, hot_encoding=hot, gumble_T=T, l1=.1)
        dc.stoch_gate = stoch
        dc.h_thresh = hth
        with torch.no_grad():       # make lambd / c non-trivial so loss() exercises every term
            dc.lambd.fill_(.3)
            dc.c.fill_(.7)
        x = torch.randn(5, 6, generator=gen).requires_grad_(True)
        torch.manual_seed(77)
        hh = dc(x)
        torch.manual_seed(77)       # the two torch.rand draws of stochastic_gate, g1 first (DAG:99-100)
        u1 = torch.rand(5, 6, 6)
        u2 = torch.rand(5, 6, 6)
        gh = torch.randn(hh.shape, generator=gen)
        loss_c = dc.loss()
        ((hh * gh).sum() + loss_c).backward()
        arr = state_np(dc, "p.")
        arr.update({"g." + k: npy(p.grad) for k, p in dc.named_parameters()})
        save("dag_" + tag, x=npy(x), h=npy(hh), gh=npy(gh), gx=npy(x.grad), u1=npy(u1), u2=npy(u2),
             loss=npy(loss_c), trace=npy(dc.get_power_trace()), exponent=np.array(dc.exponent),
             flags=np.array([float(hot), float(stoch), hth, T]), **arr)

    # (v) power trace for a larger A (exponent = d mod 50 = 34 at d=784 is too big to store; use d=84 -> 34)
    torch.manual_seed(4)
    dc = DAGConditioner(84, [8], 2)
    save("dag_trace84", A=npy(dc.A), trace=npy(dc.get_power_trace()), loss=npy(dc.loss()),
         exponent=np.array(dc.exponent), alpha=npy(dc.alpha))

    # ---------------------------------------------------------------- MNISTCNN as embedding net (a13)
    torch.manual_seed(5)
    cnn = MNISTCNN(out_d=30, fc_l=[2304, 128], size_img=[1, 28, 28])
    e = (torch.randn(6, 784, generator=gen) * (torch.rand(6, 784, generator=gen) < .05).float()).requires_grad_(True)
    out = cnn(e)
    go = torch.randn(out.shape, generator=gen)
    (out * go).sum().backward()
    arr = state_np(cnn, "p.")
    arr.update({"g." + k: npy(p.grad) for k, p in cnn.named_parameters()})
    save("mnistcnn", e=npy(e), out=npy(out), gout=npy(go), ge=npy(e.grad), **arr)

    # ---------------------------------------------------------------- (iii) IntegrandNet + Monotonic Jacobian
    torch.manual_seed(6)
    net = IntegrandNet([16, 16, 16], 5)
    B, d, c = 4, 3, 5
    x = torch.randn(B, d, generator=gen).requires_grad_(True)
    h3 = torch.randn(B, d, c, generator=gen).requires_grad_(True)
    hflat = h3.permute(0, 2, 1).contiguous().view(B, -1)          # MonotonicNormalizer.py:55
    jac = net(x, hflat)
    gj = torch.randn(B, d, generator=gen)
    (torch.log(jac) * gj).sum().backward()
    arr = state_np(net, "p.")
    arr.update({"g." + k: npy(p.grad) for k, p in net.named_parameters()})
    save("integrand", x=npy(x), h=npy(h3), jac=npy(jac), gj=npy(gj), gx=npy(x.grad), gh=npy(h3.grad), **arr)

    # ---------------------------------------------------------------- flows (Affine x {Coupling, Autoregressive, DAG})
    def flow_case(name, flow, x, seed_gate=None):
        x = x.clone().requires_grad_(True)
        if seed_gate is not None:
            torch.manual_seed(seed_gate)
        z, ld = flow(x)
        loss = flow.loss(z, ld)
        loss.backward()
        arr = state_np(flow, "p.")
        arr.update({"g." + k: npy(p.grad) for k, p in flow.named_parameters() if p.grad is not None})
        extra = {}
        if seed_gate is not None:
            torch.manual_seed(seed_gate)
            dd = x.shape[1]
            for si in range(len(flow.steps)):
                extra["u1_%d" % si] = npy(torch.rand(x.shape[0], dd, dd))
                extra["u2_%d" % si] = npy(torch.rand(x.shape[0], dd, dd))
        save(name, x=npy(x), z=npy(z), logdet=npy(ld), loss=npy(loss), gx=npy(x.grad),
             state_keys=np.array(list(flow.state_dict().keys())), **arr, **extra)
        return flow

    torch.manual_seed(10)
    f = buildFCNormalizingFlow(1, CouplingConditioner, {"in_size": 2, "hidden": [32, 32], "out_size": 2},
                               AffineNormalizer, {})
    xt = torch.randn(16, 2, generator=gen)
    flow_case("flow_affine_coupling_1", f, xt)
    with torch.no_grad():
        zt, _ = f(xt)
        import contextlib, io
        with contextlib.redirect_stdout(io.StringIO()):
            xr = f.invert(zt)
    save("flow_affine_coupling_1_inv", z=npy(zt), x=npy(xr))

    torch.manual_seed(11)
    f = buildFCNormalizingFlow(3, CouplingConditioner, {"in_size": 5, "hidden": [16, 16], "out_size": 2},
                               AffineNormalizer, {})
    flow_case("flow_affine_coupling_3", f, torch.randn(8, 5, generator=gen))

    torch.manual_seed(12)
    f = buildFCNormalizingFlow(1, AutoregressiveConditioner, {"in_size": 8, "hidden": [24, 24, 24], "out_size": 2},
                               AffineNormalizer, {})
    xt = logit_mnist_like(gen, 6, 8)
    flow_case("flow_affine_made_1", f, xt)
    with torch.no_grad():
        zt, _ = f(xt)
        with contextlib.redirect_stdout(io.StringIO()):
            xr = f.invert(zt)
    save("flow_affine_made_1_inv", z=npy(zt), x=npy(xr))

    torch.manual_seed(13)
    f = buildFCNormalizingFlow(2, DAGConditioner, {"in_size": 6, "hidden": [16, 16], "out_size": 2, "l1": .05,
                                                   "gumble_T": .5, "hot_encoding": True},
                               AffineNormalizer, {})
    flow_case("flow_affine_dag_2", f, torch.randn(8, 6, generator=gen), seed_gate=99)

    # MNIST factory, 1 step, Affine, kernel-2 prior, no hot encoding (cfg4 with the Affine normalizer), B=2
    torch.manual_seed(14)
    f = buildMNISTNormalizingFlow([1], AffineNormalizer, {}, l1=0., nb_epoch_update=10, hot_encoding=False,
                                  prior_kernel=2)
    xt = logit_mnist_like(gen, 2, 784)
    x = xt.clone()
    torch.manual_seed(55)
    z, ld = f(x)
    loss = f.loss(z, ld)
    loss.backward()
    cond = f.steps[0].conditioner
    gA = npy(cond.A.grad)
    nz = np.argwhere(gA != 0).astype(np.int32)
    arr = {"p." + k: npy(v) for k, v in f.state_dict().items() if not k.endswith("conditioner.A")}
    arr.update({"g." + k: npy(p.grad) for k, p in f.named_parameters() if not k.endswith("conditioner.A")})
    save("flow_mnist_affine_dag", x=npy(xt), z=npy(z), logdet=npy(ld), loss=npy(loss), gate_seed=np.array(55),
         gA_idx=nz, gA_val=gA[nz[:, 0], nz[:, 1]], trace=npy(cond.get_power_trace()),
         state_keys=np.array(list(f.state_dict().keys())), **arr)

    # Monotonic flows: only state_dict keys + Jacobian/logdet are recordable (integral needs UMNN)
    torch.manual_seed(15)
    f = buildFCNormalizingFlow(1, AutoregressiveConditioner, {"in_size": 4, "hidden": [12, 12], "out_size": 6},
                               MonotonicNormalizer, {"integrand_net": [10, 10], "cond_size": 6, "nb_steps": 20,
                                                     "solver": "CC"})
    x = torch.randn(5, 4, generator=gen)
    with torch.no_grad():
        hcond = f.steps[0].conditioner(x)
        hflat = hcond.permute(0, 2, 1).contiguous().view(5, -1)
        jac = f.steps[0].normalizer.integrand_net(x, hflat)
    save("flow_mono_made_1", x=npy(x), h=npy(hcond), jac=npy(jac), logdet=npy(torch.log(jac).sum(1)),
         state_keys=np.array(list(f.state_dict().keys())), **state_np(f, "p."))

    # 3-scale MNIST factory (CNNormalizingFlow, Factories.py:51-78), Affine, kernel-2 priors, deterministic gates, B=2.
    # invert() is not recordable here (DAGConditioner.depth needs networkx < 3).  Gradients of the large fc1 weights
    # are stored for their first 8 rows only.
    CNNormalizingFlow = models.NormalizingFlow.CNNormalizingFlow if hasattr(models, "NormalizingFlow") else None
    torch.manual_seed(16)
    f = buildMNISTNormalizingFlow([1, 1, 1], AffineNormalizer, {}, l1=0., nb_epoch_update=10, hot_encoding=False,
                                  prior_kernel=2)
    for c in f.getConditioners():
        c.stoch_gate = False
    xt = logit_mnist_like(gen, 2, 784)
    z, ld = f(xt.clone())
    loss = f.loss(z, ld)
    loss.backward()
    arr = {"p." + k: npy(v) for k, v in f.state_dict().items() if not k.endswith("conditioner.A")}
    for k, p_ in f.named_parameters():
        if k.endswith("conditioner.A"):
            gA = npy(p_.grad)
            nz = np.argwhere(gA != 0).astype(np.int32)
            arr["gAidx." + k] = nz
            arr["gAval." + k] = gA[nz[:, 0], nz[:, 1]]
        elif p_.numel() > 50000:
            arr["g8." + k] = npy(p_.grad)[:8]
        else:
            arr["g." + k] = npy(p_.grad)
    save("flow_mnist3_affine", x=npy(xt), z=npy(z), logdet=npy(ld), loss=npy(loss),
         state_keys=np.array(list(f.state_dict().keys())), **arr)


if __name__ == "__main__":
    main()
