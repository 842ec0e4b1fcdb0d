"""Epoch-level DAG machinery: DAGConditioner.step() / update_dual_param() / post_process() driven
through every branch, against trajectories recorded from the REFERENCE's own code
(tests/golden/make_golden_dual.py -> dag_dual.npz; reference DAGConditioner.py:76-92,196-260,273-293).  Host control
logic: runs on the CPU here and, marked gpu, with the buffers and the fused loss kernels on the device."""
import numpy as np
import pytest
import torch

from conftest import load_golden

SCENARIOS = ["dual", "success", "exponent", "failure", "reopen"]
FIELDS = ("lambd", "c", "prev_trace", "dag_const", "l1_weight", "alpha")


def _snapshot(c):
    vec = [float(getattr(c, f).detach()) for f in FIELDS]
    vec += [float(c.exponent), float(c.no_update), float(c.stoch_gate), float(c.noise_gate), float(c.s_thresh),
            float(c.h_thresh), float(c.A.requires_grad), float(bool(c.is_invertible))]
    return np.array(vec, dtype=np.float64)


def _run(name, dev):
    from models import DAGConditioner
    g = load_golden("dag_dual")
    d, l1, nbu, th, off = [float(v) for v in g[name + ".cfg"]]
    torch.manual_seed(0)
    c = DAGConditioner(int(d), [8], 2, l1=l1, nb_epoch_update=int(nbu), A_prior=g[name + ".A0"].clone()).to(dev)
    A_obj, A_ptr = c.A, c.A.data_ptr()
    bufs = {f: getattr(c, f).data_ptr() for f in FIELDS}
    if th >= 0:                                        # the generator pins post_process()'s threshold the same way
        pp = c.post_process
        c.post_process = lambda zero_threshold=None: pp(th)
    if off:
        c.dag_const.fill_(0.)
        c.l1_weight.fill_(0.)
        c.stoch_gate, c.noise_gate, c.s_thresh, c.h_thresh = False, False, False, 0.
        c.A.requires_grad = False
    assert torch.equal(c.A.detach().cpu(), g[name + ".A_init"])
    np.testing.assert_allclose(_snapshot(c), g[name + ".state0"].numpy(), rtol=2e-6, atol=1e-12)
    for k, (epoch, loss_avg) in enumerate(g[name + ".calls"].tolist()):
        c.step(int(epoch), torch.tensor(loss_avg, device=dev))
        got, want = _snapshot(c), g[name + ".states"][k].numpy()
        # traces are sums of O(10^3) fp32 products: 1e-5 relative; flags / counters / exponent are exact
        np.testing.assert_allclose(got[:6], want[:6], rtol=1e-5, atol=1e-12, err_msg="%s call %d" % (name, k))
        assert (got[6:] == want[6:]).all(), (name, k, got[6:], want[6:])
        assert torch.allclose(c.A.detach().cpu(), g[name + ".A"][k], rtol=0, atol=0), (name, k)
        # in-place contract (deliberate divergence from the reference's rebinding): the Parameter object, its storage
        # and every buffer's storage survive -- flat optimiser buffers and captured graphs keep seeing them
        assert c.A is A_obj and c.A.data_ptr() == A_ptr
        assert all(getattr(c, f).data_ptr() == p for f, p in bufs.items())
        assert all(getattr(c, f).device.type == torch.device(dev).type for f in FIELDS)
    return c


@pytest.mark.parametrize("name", SCENARIOS)
def test_dual_update_trajectories_cpu(name):
    _run(name, "cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("name", SCENARIOS)
def test_dual_update_trajectories_gpu(name):
    c = _run(name, "cuda:0")
    if name == "success":                              # constraints off: loss() is the constant 0, no matrix power
        assert c._constraints_off() and float(c.loss()) == 0.


def test_A_keeps_training_after_a_failed_post_processing():
    """Deliberate divergence: the reference re-opens the gate by installing a NEW nn.Parameter
    (DAGConditioner.py:224) that its optimiser never steps, so A silently freezes; here the saved values go back into the
    same Parameter and an optimiser built before the failure keeps updating it."""
    from models import DAGConditioner
    g = load_golden("dag_dual")
    d, l1, nbu, th, off = [float(v) for v in g["failure.cfg"]]
    torch.manual_seed(0)
    c = DAGConditioner(int(d), [8], 2, l1=l1, nb_epoch_update=int(nbu), A_prior=g["failure.A0"].clone())
    opt = torch.optim.SGD(c.parameters(), lr=.1)      # built BEFORE the failed post-processing, like a driver's
    pp = c.post_process
    c.post_process = lambda zero_threshold=None: pp(th)
    for epoch, loss_avg in g["failure.calls"].tolist():
        c.step(int(epoch), torch.tensor(loss_avg))
    assert c.A.requires_grad and c.stoch_gate and c.A.grad is None
    before = c.A.detach().clone()
    opt.zero_grad()
    c.loss().backward()
    opt.step()
    assert (c.A.detach() != before).any()


def test_post_process_auto_threshold_loop():
    """post_process(None) raises the threshold from .1 in steps of .05 until the kept edges are acyclic
    (reference :76-92): a 3-cycle whose weakest edge has soft-thresholded weight ~.31 needs 5 increments."""
    from models import DAGConditioner
    d = 5
    A = torch.zeros(d, d)
    A[0, 1], A[1, 2], A[2, 0] = 1.2, 1.0, .4          # soft-thresholded: .89, .76, .159..
    A[3, 0], A[4, 3] = .9, .9
    c = DAGConditioner(d, [4], 2, A_prior=A.clone())
    soft = c.soft_thresholded_A().detach()
    weakest = float(soft[2, 0])
    with torch.no_grad():
        c.post_process()
    th = .1
    while th < weakest:
        th += .05
    want = (soft > th).float() * (1 - torch.eye(d))
    assert torch.equal(c.A.detach(), want) and want[2, 0] == 0 and want.sum() == 4
    assert not c.A.requires_grad and not c.stoch_gate and not c.s_thresh and c.h_thresh == 0.
    assert c.depth() == 4                              # kept edges 4->3->0->1->2
    assert c.levels() is not None


def test_frozen_A_is_not_touched_by_the_flat_adam_step():
    """torch.optim.Adam skips parameters without a gradient; after post_process() A has none.  The flat fused Adam must
    not decay or move it (FlatState.active_runs), and the other parameters must follow torch.optim.Adam."""
    from gnf_hip import dp
    from models import DAGConditioner

    def torch_adam(state, lr, weight_decay, grad_scale, b1=.9, b2=.999, eps=1e-8):
        for o, k in state.active_runs:
            p, g, m, v = (t[o:o + k] for t in (state.flat, state.grad, state.m, state.v))
            g = g * grad_scale + weight_decay * p
            m.mul_(b1).add_(g, alpha=1 - b1)
            v.mul_(b2).addcmul_(g, g, value=1 - b2)
            p.sub_(lr / (1 - b1 ** state.t) * m / (v.sqrt() / (1 - b2 ** state.t) ** .5 + eps))

    torch.manual_seed(3)
    d = 6
    c = DAGConditioner(d, [8], 2, A_prior=torch.tril(torch.rand(d, d) + .5, -1))
    ref = DAGConditioner(d, [8], 2, A_prior=c.A.detach().clone())
    ref.load_state_dict(c.state_dict())
    state = dp.FlatState(c)                            # A is in the flat buffer (it was trainable at construction)
    with torch.no_grad():
        c.post_process(.1)
        ref.post_process(.1)
    A_frozen = c.A.detach().clone()
    assert c.A.data_ptr() == state.flat.data_ptr()     # still the view of the flat buffer
    opt = torch.optim.Adam(ref.parameters(), lr=1e-2, weight_decay=1e-2)
    for t in range(3):
        for m, o in ((c, None), (ref, opt)):
            if o is not None:
                o.zero_grad()
            w = sum((p ** 2).sum() for p in m.embedding_net.parameters())
            w.backward()
        state.pack_grads()
        state.t += 1
        torch_adam(state, 1e-2, 1e-2, 1.)
        opt.step()
    assert torch.equal(c.A.detach(), A_frozen)
    assert len(state.active_runs) == 1 and state.active_runs[0][0] > 0       # one run, starting after A
    for (k, p), (_, q) in zip(c.named_parameters(), ref.named_parameters()):
        assert torch.allclose(p, q, rtol=1e-5, atol=1e-7), k
