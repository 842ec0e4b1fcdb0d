"""Empty batches (-m gpu).  In the reference a [0, d] batch is plain torch: the Coupling conditioner and both
normalizers run it; the DAG and Autoregressive conditioners (and MNISTCNN) stop at their own `.view(x.shape[0], ..., -1)`
with torch's "ambiguous" RuntimeError (DAGConditioner.py:169, AutoregressiveConditioner.py:109, MLP.py:47).  The C-ABI
entry points below them must take the empty call -- no zero-sized grid, no division by a zero tile count, NULL
batch-sized arrays accepted (torch's data_ptr of an empty tensor) -- and the backward ones must still write ZERO
parameter gradients."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _flow(cond, norm, d=6):
    import models.Conditionners as C
    from models import AffineNormalizer, MonotonicNormalizer
    from models.NormalizingFlowFactories import buildFCNormalizingFlow
    nargs = {} if norm == "Affine" else dict(integrand_net=[20, 20], cond_size=4, nb_steps=8, solver="CC")
    cargs = dict(in_size=d, hidden=[16, 16], out_size=2 if norm == "Affine" else 4)
    if cond == "DAGConditioner":
        cargs.update(soft_thresholding=True, h_thresh=0., gumble_T=.5, hot_encoding=True, l1=.1, nb_epoch_update=1)
    torch.manual_seed(0)
    return buildFCNormalizingFlow(2, getattr(C, cond), cargs,
                                  AffineNormalizer if norm == "Affine" else MonotonicNormalizer, nargs).to(DEV)


@pytest.mark.parametrize("norm", ["Affine", "Monotonic"])
def test_empty_batch_through_the_coupling_flows(norm):
    d = 6
    flow = _flow("CouplingConditioner", norm, d)
    z, ld = flow(torch.zeros(0, d, device=DEV))
    assert z.shape == (0, d) and ld.shape == (0,)
    (z.sum() + ld.sum()).backward()
    for n, p in flow.named_parameters():
        if p.grad is not None:
            assert torch.count_nonzero(p.grad).item() == 0, n
    assert flow.invert(torch.zeros(0, d, device=DEV)).shape == (0, d)


@pytest.mark.parametrize("cond", ["DAGConditioner", "AutoregressiveConditioner"])
def test_empty_batch_stops_where_the_reference_stops(cond):
    """same exception, same place (the conditioner's view), after the kernels under it took the empty call"""
    flow = _flow(cond, "Monotonic")
    with pytest.raises(RuntimeError, match="ambiguous"):
        flow(torch.zeros(0, 6, device=DEV))


def _zero(*ts):
    for t in ts:
        assert t is not None and torch.count_nonzero(t).item() == 0


def test_empty_batch_through_every_autograd_function():
    """the Functions the plug-ins are made of, forward and backward, on zero rows"""
    from gnf_hip import ops
    d, c = 5, 3
    g = torch.Generator(device=DEV).manual_seed(0)
    rnd = lambda *s: torch.randn(*s, device=DEV, generator=g).requires_grad_(True)
    x = torch.zeros(0, d, device=DEV, requires_grad=True)

    # Affine (+ inverse, + the row reductions of the loss)
    h = torch.zeros(0, d, 2, device=DEV, requires_grad=True)
    z, jac, ld, _ = ops.AffineFn.apply(x, h)
    assert z.shape == (0, d) and jac.shape == (0, d) and ld.shape == (0,)
    (z.sum() + jac.sum() + ld.sum()).backward()
    assert x.grad.shape == (0, d) and h.grad.shape == (0, d, 2)
    assert ops.affine_inverse(z.detach(), h.detach()).shape == (0, d)
    jj = torch.zeros(0, d, device=DEV, requires_grad=True)
    (ops.LogSumRowsFn.apply(jj).sum() + ops.NormalLogDensityFn.apply(jj).sum()).backward()
    assert jj.grad.shape == (0, d)

    # Monotonic: parameter gradients must come back as zeros, not as uninitialised memory
    params = [rnd(12, 1 + c), rnd(12), rnd(12, 12), rnd(12), rnd(1, 12), rnd(1)]
    hm = torch.zeros(0, d, c, device=DEV, requires_grad=True)
    x2 = torch.zeros(0, d, device=DEV, requires_grad=True)
    zm, jm = ops.MonotonicFn.apply(x2, hm, 8, *params)
    assert zm.shape == (0, d) and jm.shape == (0, d)
    (zm.sum() + jm.sum()).backward()
    _zero(*[p.grad for p in params])
    assert ops.monotonic_inverse(zm.detach(), hm.detach(), 8, params).shape == (0, d)

    # Linear / ReLU chain (masked and not): K = 0 weight-gradient GEMMs, M = 0 forward GEMMs
    W1, b1, W2, b2 = rnd(7, d), rnd(7), rnd(4, 7), rnd(4)
    mk = [(torch.rand(7, d, device=DEV, generator=g) > .5).float(), (torch.rand(4, 7, device=DEV, generator=g) > .5).float()]
    for masks in (None, mk):
        for p in (W1, b1, W2, b2):
            p.grad = None
        x3 = torch.zeros(0, d, device=DEV, requires_grad=True)
        y = ops.MLPFn.apply(x3, masks, False, None, W1, b1, W2, b2)
        assert y.shape == (0, 4)
        y.sum().backward()
        _zero(W1.grad, b1.grad, W2.grad, b2.grad)
        assert x3.grad.shape == (0, d)

    # DAG gate: dA = 0
    A = rnd(d, d)
    x4 = torch.zeros(0, d, device=DEV, requires_grad=True)
    for gate_mode in (0, 1, 2):
        A.grad = None
        e = ops.DagGateFn.apply(x4, A, 1, gate_mode, 0., .5, True, None, None, 1234, 0)
        assert e.shape == (0, 2 * d)
        e.sum().backward()
        _zero(A.grad)

    # MNISTCNN conv front
    cw = [rnd(16, 1, 3, 3), rnd(16), rnd(16, 16, 3, 3), rnd(16)]
    for exact in (False, True):
        for p in cw:
            p.grad = None
        img = torch.zeros(0, 784, device=DEV, requires_grad=True)
        pooled = ops.MnistConvFn.apply(img, *cw, exact)
        assert pooled.shape == (0, 2304)
        pooled.sum().backward()
        _zero(*[p.grad for p in cw])
        assert img.grad.shape == (0, 784)


def test_empty_batch_through_the_fused_gate_conv_front_and_the_loss():
    """round 5's autograd nodes and entry points on zero rows: gate + conv front as one node (dA and the conv gradients come
    back as zeros), the *_cols entry points WITH a plan (table, plan and its overflow word are written for an empty batch too:
    the backward kernels read them), and the one-launch loss (no rows: the flow falls back to the reference's expression)"""
    import ctypes
    from gnf_hip import ops, abi
    from gnf_hip.abi import ptr, rawptr, call, stream
    g = torch.Generator(device=DEV).manual_seed(1)
    rnd = lambda *s: torch.randn(*s, device=DEV, generator=g).requires_grad_(True)
    d = 784
    A = rnd(d, d)
    cw = [rnd(16, 1, 3, 3), rnd(16), rnd(16, 16, 3, 3), rnd(16)]
    for xg in (False, True):
        for p in [A] + cw:
            p.grad = None
        x = torch.zeros(0, d, device=DEV, requires_grad=xg)
        pooled = ops.dag_conv_front(x, A, ops.IMP_SOFT, ops.GATE_GUMBEL, 0., 1., None, None, 7, 1, *cw)
        assert pooled.shape == (0, 2304)
        pooled.sum().backward()
        _zero(A.grad, *[p.grad for p in cw])
    # C ABI with a plan on an empty batch
    lib = abi.load()
    tab = torch.empty(lib.gnf_dag_gate_fwd_ws_bytes(d) // 4, device=DEV)
    nplan = lib.gnf_dag_gate_plan_bytes(d)
    plan = torch.full((nplan // 4,), -7, dtype=torch.int32, device=DEV)
    Ad = A.detach()
    call("gnf_dag_gate_fwd_plan", None, ptr(Ad), None, d, 1, 1, 0., 1., None, None, 7, 1, 0, ptr(tab), rawptr(plan), nplan, 0, d,
         stream())
    assert int(plan[d + d * 16]) == 1 and int(plan[:d].min()) >= 0              # dense random A: every row overflows the plan
    gs = [torch.full_like(t.detach(), 5.) for t in cw]
    ws = torch.empty(lib.gnf_mnistcnn_conv_bwd_ws_bytes(0) // 4, device=DEV)
    call("gnf_mnistcnn_conv_bwd_cols", None, *[ptr(t.detach().contiguous()) for t in cw[:3]], None, None, None, rawptr(plan), d,
         None, *[ptr(t) for t in gs], rawptr(ws), ws.numel() * 4, 0, stream())
    _zero(*gs)
    gA = torch.full((d, d), 5., device=DEV)
    ws2 = torch.empty(lib.gnf_dag_gate_bwd_cols_ws_bytes(0, d) // 4, device=DEV)
    call("gnf_dag_gate_bwd_cols", None, None, None, rawptr(plan), 1, 1, 1., None, None, 7, 1, ptr(tab), ptr(gA), 0, ptr(ws2),
         0, d, stream())
    _zero(gA)
    # the loss launch refuses zero rows (the mean of nothing); FCNormalizingFlow.loss then evaluates the reference's expression
    z0, l0 = torch.zeros(0, 5, device=DEV), torch.zeros(0, device=DEV)
    assert not ops.nll_loss_fits(z0)
    assert lib.gnf_nll_loss_fwd(None, None, None, ctypes.c_void_p(gA.data_ptr()), 0, 5, None) == -1
    flow = _flow("CouplingConditioner", "Affine", 6)
    z, ld = flow(torch.zeros(0, 6, device=DEV))
    assert torch.isnan(flow.loss(z, ld)).item()                  # torch: mean of an empty tensor


def test_empty_contraction_and_empty_column_sum():
    """a zero-row batch as the CONTRACTION of a weight-gradient GEMM (K = 0: C = 0, operands NULL) and as the rows of a
    bias-gradient column sum (M = 0: out = 0) -- what torch returns for x.t() @ g and g.sum(0) on [0, n] tensors"""
    from gnf_hip import ops
    C = torch.full((5, 7), 3., device=DEV)
    e = torch.empty(0, device=DEV)
    ops.gemm(e, (1, 5), e, (7, 1), C, (7, 1), 5, 7, 0)
    assert torch.count_nonzero(C).item() == 0
    out = ops.colsum(torch.empty(0, 9, device=DEV))
    assert out.shape == (9,) and torch.count_nonzero(out).item() == 0
    # and the other way round: no output rows, nothing written, no launch
    ops.gemm(e, (3, 1), torch.ones(3, 4, device=DEV), (4, 1), e, (4, 1), 0, 4, 3)
