"""Structural zeros of the DAG-gate backward (-m gpu): dL/dA[i,j] = dP/dA[i,j] * (...) and dP/dA is exactly zero wherever
A is (reference DAGConditioner.py:118-119), so with x frozen the fused masked-image front (gnf_hip.ops.DagConvFrontFn:
gate + MNISTCNN conv front as one autograd node, DAGConditioner.py:142-143,169 -> MLP.py:36-43) hands the gate backward
the cotangent of e at the plan's columns only.  Checked here: the compact path against the dense autograd nodes and
against the CPU oracle, the on-device fallback when a row holds more columns than the plan, the dense entry points when x
wants a gradient, and the two C-ABI entry points on plans that are NOT pixel windows."""
import ctypes

import pytest
import torch

from conftest import rel_err, assert_close
from oracle import gnf_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GTOL = 1e-4
KC = 32


def _flow(prior_kernel=2):
    from models import MonotonicNormalizer
    from models.NormalizingFlowFactories import buildMNISTNormalizingFlow
    torch.manual_seed(0)
    return buildMNISTNormalizingFlow([1], MonotonicNormalizer, {"integrand_net": [50, 50, 50], "nb_steps": 20,
                                                               "solver": "CC"}, l1=.05, nb_epoch_update=10,
                                     hot_encoding=False, prior_kernel=prior_kernel).to(DEV)


def _x(B, seed=5):
    from gnf_hip import configs
    return configs.pseudo_mnist(torch.Generator().manual_seed(seed), B, 784).to(DEV)


def _step(flow, x, fused):
    cond = flow.steps[0].conditioner
    cond.fused_front = fused
    cond._gate_calls = 0                       # same Philox offset in both runs
    for p in flow.parameters():
        p.grad = None
    z, ld = flow(x)
    loss = flow.loss(z, ld)
    loss.backward()
    return z.detach(), ld.detach(), loss.detach(), {k: p.grad.clone() for k, p in flow.named_parameters()}


@pytest.mark.parametrize("B", [1, 3, 7])
def test_compact_backward_equals_the_dense_nodes(B):
    """cfg4's model, Philox Gumbel gate, x frozen: the fused node (plan, compact de, skipped T planes) against DagGateFn +
    MnistConvFn on the same noise -- forward bits equal, the network's gradients equal bit for bit (the skipped products
    feed de only), dA to summation order (other sample chunks) with the same exact zeros."""
    flow = _flow()
    x = _x(B)
    z1, ld1, l1, g1 = _step(flow, x, True)
    z0, ld0, l0, g0 = _step(flow, x, False)
    assert torch.equal(z1, z0) and torch.equal(ld1, ld0) and torch.equal(l1, l0)
    kA = "steps.0.conditioner.A"
    for k in g0:
        if k != kA:
            assert torch.equal(g1[k], g0[k]), k
    assert_close(g1[kA], g0[kA], rtol=1e-5, atol=1e-6 * g0[kA].abs().max().item(), what="dA")
    A = flow.steps[0].conditioner.A.detach()
    assert int(((g1[kA] != 0) & (A == 0)).sum()) == 0
    assert int((g1[kA] != 0).sum()) > 15000          # ... and the 17 172 prior entries do receive one


def test_plan_overflow_falls_back_on_the_device():
    """MNIST_A_prior(28, 3): 48 columns per row > GNF_DAG_PLAN_KC -- both *_cols entry points must notice it on the device
    and run their dense code (the host never reads the plan)"""
    flow = _flow(prior_kernel=3)
    x = _x(2)
    z1, ld1, l1, g1 = _step(flow, x, True)
    z0, ld0, l0, g0 = _step(flow, x, False)
    assert torch.equal(z1, z0) and torch.equal(l1, l0)
    for k in g0:
        assert_close(g1[k], g0[k], rtol=1e-5, atol=1e-6 * g0[k].abs().max().item(), what=k)
    kA = "steps.0.conditioner.A"
    assert int((g1[kA] != 0).sum()) > 30000


def test_one_overflowing_row_is_enough():
    """the kernel-2 prior with ONE row filled beyond the plan: still the dense code, still the dense numbers"""
    flow = _flow()
    with torch.no_grad():
        flow.steps[0].conditioner.A[400, 100:140] = .7
    x = _x(2)
    _, _, l1, g1 = _step(flow, x, True)
    _, _, l0, g0 = _step(flow, x, False)
    assert torch.equal(l1, l0)
    kA = "steps.0.conditioner.A"
    assert_close(g1[kA], g0[kA], rtol=1e-5, atol=1e-6 * g0[kA].abs().max().item(), what="dA")
    assert int((g1[kA][400, 100:140] != 0).sum()) == 40


def _oracle(sd, x, u1, u2, S):
    pre = "steps.0.conditioner."
    sd = {k: v.detach().cpu().clone() for k, v in sd.items()}
    cnn = {k[len(pre + "embedding_net."):]: v for k, v in sd.items() if "embedding_net." in k}
    A = sd[pre + "A"].requires_grad_(True)
    layers, k = [], 0
    ipre = "steps.0.normalizer.integrand_net.net."
    while ipre + "%d.weight" % k in sd:
        layers.append((sd[ipre + "%d.weight" % k], sd[ipre + "%d.bias" % k]))
        k += 2
    B, d = x.shape
    x = x.clone().requires_grad_(True)
    e = O.dag_masked_inputs(x, A, True, 0., True, False, 1., u1, u2, None, False)
    h = O.mnistcnn_forward(e, cnn).view(B, d, -1)
    z, jac = O.monotonic_forward(x, h, layers, S)
    closs = O.dag_loss(A, sd[pre + "alpha"], d % 50, sd[pre + "lambd"], sd[pre + "c"], sd[pre + "dag_const"],
                       sd[pre + "l1_weight"])
    loss = O.flow_loss(z, torch.log(jac).sum(1), closs)
    loss.backward()
    return loss.detach(), x.grad, A.grad


def test_x_gradient_on_the_prior_sparse_flow_vs_oracle():
    """x.requires_grad = True: the fused node takes the dense entry points (de is needed everywhere: the gate leaks) and
    dx, dA match the oracle chain; the same inputs with x frozen give the same dA through the compact path"""
    B, S = 2, 20
    flow = _flow()
    g = torch.Generator().manual_seed(11)
    x = _x(B, seed=9)
    u1, u2 = torch.rand(B, 784, 784, generator=g), torch.rand(B, 784, 784, generator=g)
    l0, gx0, gA0 = _oracle(flow.state_dict(), x.cpu(), u1, u2, S)
    cond = flow.steps[0].conditioner
    cond.gate_noise = (u1.to(DEV), u2.to(DEV))
    xr = x.clone().requires_grad_(True)
    z, ld = flow(xr)
    loss = flow.loss(z, ld)
    loss.backward()
    assert rel_err(loss.detach().cpu(), l0) < 1e-5
    assert rel_err(xr.grad.cpu(), gx0) < GTOL, rel_err(xr.grad.cpu(), gx0)
    assert_close(xr.grad, gx0, rtol=1e-4, atol=1e-6 * gx0.abs().max().item(), what="dx")
    assert rel_err(cond.A.grad.cpu(), gA0) < GTOL
    gA_dense = cond.A.grad.clone()
    cond.A.grad = None
    z, ld = flow(x)                                # x frozen: plan + compact de
    flow.loss(z, ld).backward()
    assert_close(cond.A.grad, gA0, rtol=1e-4, atol=1e-6 * gA0.abs().max().item(), what="dA compact vs oracle")
    assert_close(cond.A.grad, gA_dense, rtol=1e-5, atol=1e-6 * gA0.abs().max().item(), what="dA compact vs dense")


def _plan_arrays(plan, d):
    cnt = plan[:d].cpu()
    cols = plan[d:d + d * KC // 2].cpu().view(torch.int16).view(d, KC)
    assert int(plan[d + d * KC // 2]) == int(bool((cnt > KC).any()))          # the overflow word
    return cnt, cols


@pytest.mark.parametrize("B,per_row", [(2, 32), (1, 5), (3, 0)])
def test_cols_entry_points_on_arbitrary_plans(B, per_row):
    """C ABI directly, a random A whose non-zeros are NOT pixel windows (per_row random columns per row; 0: an empty
    matrix): the plan lists exactly the columns with dP/dA != 0 in ascending order, ge_cols[:, k] is bit-equal to column
    cols[k] of the dense ge, the parameter gradients are bit-equal, and gnf_dag_gate_bwd_cols reproduces the dense gA"""
    from gnf_hip import abi
    from gnf_hip.abi import ptr, rawptr, call, stream
    lib = abi.load()
    d, n = 784, B * 784
    g = torch.Generator().manual_seed(3 + per_row)
    A = torch.zeros(d, d)
    for i in range(d):
        if per_row:
            A[i, torch.randperm(d, generator=g)[:per_row]] = torch.rand(per_row, generator=g) + .3
    A, x = A.to(DEV), torch.randn(B, d, generator=g).to(DEV)
    e = torch.empty(n, d, device=DEV)
    tab = torch.empty(lib.gnf_dag_gate_fwd_ws_bytes(d) // 4, device=DEV)
    nplan = lib.gnf_dag_gate_plan_bytes(d)
    plan = torch.empty(nplan // 4, dtype=torch.int32, device=DEV)
    call("gnf_dag_gate_fwd_plan", ptr(x), ptr(A), ptr(e), d, 1, 1, 0., 1., None, None, 1234, 7, 0, ptr(tab), rawptr(plan),
         nplan, B, d, stream())
    cnt, cols = _plan_arrays(plan, d)
    nz = (tab[d * d:2 * d * d].view(d, d) != 0).cpu()
    assert torch.equal(cnt.long(), nz.sum(1))
    for i in (0, 1, 391, 783):
        want = nz[i].nonzero().flatten()
        assert torch.equal(cols[i, :len(want)].long(), want) and bool((cols[i, len(want):] == -1).all())

    W1, b1 = torch.randn(16, 9, generator=g).to(DEV) * .3, torch.randn(16, generator=g).to(DEV) * .1
    W2, b2 = torch.randn(16, 144, generator=g).to(DEV) * .1, torch.randn(16, generator=g).to(DEV) * .1
    pooled = torch.empty(n, 2304, device=DEV)
    arg = torch.empty(n, 2304, dtype=torch.uint8, device=DEV)
    call("gnf_mnistcnn_conv_fwd", ptr(e), ptr(W1), ptr(b1), ptr(W2), ptr(b2), ptr(pooled), rawptr(arg), n, 0, stream())
    gp = torch.randn(n, 2304, generator=g).to(DEV)
    nws = lib.gnf_mnistcnn_conv_bwd_ws_bytes(n)
    ws = torch.empty(nws // 4, device=DEV)

    def conv_bwd(with_plan):
        ge = torch.full((n, d), float("nan"), device=DEV)
        gec = torch.full((n, KC), float("nan"), device=DEV)
        gs = [torch.empty_like(t) for t in (W1, b1, W2, b2)]
        call("gnf_mnistcnn_conv_bwd_cols", ptr(e), ptr(W1), ptr(b1), ptr(W2), ptr(gp), rawptr(arg), ptr(ge),
             rawptr(plan) if with_plan else None, d if with_plan else 0, ptr(gec) if with_plan else None,
             *[ptr(t) for t in gs], rawptr(ws), nws, n, stream())
        return ge, gec, gs
    ge0, _, gs0 = conv_bwd(False)
    ge1, gec, gs1 = conv_bwd(True)
    assert bool(torch.isnan(ge1).all()), "the compact call must not write the dense cotangent"
    for a, b in zip(gs0, gs1):
        assert torch.equal(a, b)
    colsd = cols.to(DEV).long()
    rows = torch.arange(n, device=DEV) % d
    for k in range(KC):
        cj = colsd[rows, k]
        on = cj >= 0
        got, want = gec[on, k], ge0[on.nonzero().flatten(), cj[on]]
        assert torch.equal(got, want), k
        assert bool(torch.isnan(gec[~on, k]).all())

    gA0, gA1 = torch.empty(d, d, device=DEV), torch.empty(d, d, device=DEV)
    ws0 = torch.empty(lib.gnf_dag_gate_bwd_ws_bytes(B, d) // 4, device=DEV)
    call("gnf_dag_gate_bwd", ptr(x), ptr(A), ptr(ge0), d, 1, 1, 0., 1., None, None, 1234, 7, ptr(tab), ptr(gA0), None,
         ptr(ws0), B, d, stream())
    ws1 = torch.empty(lib.gnf_dag_gate_bwd_cols_ws_bytes(B, d) // 4, device=DEV)
    call("gnf_dag_gate_bwd_cols", ptr(x), ptr(ge1), ptr(gec), rawptr(plan), 1, 1, 1., None, None, 1234, 7, ptr(tab),
         ptr(gA1), 0, ptr(ws1), B, d, stream())
    assert_close(gA1, gA0, rtol=1e-5, atol=1e-6 * max(gA0.abs().max().item(), 1e-30), what="gA")
    assert int(((gA1 != 0) & (A == 0)).sum()) == 0
    # accumulate: gA += into whatever the buffer holds (the acyclicity term's contribution in a training step)
    base = torch.randn(d, d, generator=g).to(DEV)
    gA2 = base.clone()
    call("gnf_dag_gate_bwd_cols", ptr(x), ptr(ge1), ptr(gec), rawptr(plan), 1, 1, 1., None, None, 1234, 7, ptr(tab),
         ptr(gA2), 1, ptr(ws1), B, d, stream())
    assert_close(gA2, base + gA1, rtol=1e-6, atol=1e-6 * max(gA0.abs().max().item(), 1.), what="gA accumulated")
    assert torch.equal(gA2[A == 0], base[A == 0])


def test_cols_entry_points_validate_their_arguments():
    from gnf_hip import abi
    lib = abi.load()
    P = ctypes.c_void_p
    t = torch.zeros(4096, device=DEV)
    p = P(t.data_ptr())
    # a plan for anything but one masked copy per pixel of the 28 x 28 image
    rc = lib.gnf_mnistcnn_conv_bwd_cols(p, p, p, p, p, p, p, p, 100, p, p, p, p, p, p, 1 << 30, 1, None)
    assert rc == -2
    # a plan without the compact slab
    rc = lib.gnf_mnistcnn_conv_bwd_cols(p, p, p, p, p, p, p, p, 784, None, p, p, p, p, p, 1 << 30, 1, None)
    assert rc == -1
    assert lib.gnf_dag_gate_bwd_cols(p, p, p, None, 1, 1, 1., None, None, 0, 0, p, p, 0, p, 1, 784, None) == -1
    assert lib.gnf_dag_gate_fwd_plan(p, p, p, 784, 1, 1, 0., 1., None, None, 0, 0, 0, p, p, 16, 1, 784, None) == -3
    assert lib.gnf_dag_gate_fwd_plan(p, p, p, 40000, 1, 1, 0., 1., None, None, 0, 0, 0, p, p, 1 << 40, 1, 40000, None) == -2
