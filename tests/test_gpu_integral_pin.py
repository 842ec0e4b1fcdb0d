"""GPU, oracle-INDEPENDENT pin of the Monotonic normalizer's integral (review of round 5, item 5; reference
models/Normalizers/MonotonicNormalizer.py:51-66).  The Clenshaw-Curtis arithmetic lives in the absent UMNN==1.0, so
`oracle/` restates it from memory and tests/test_oracle_math.py validates THAT restatement on the CPU.  This file checks
the HIP kernels themselves against mathematics, without importing anything from `oracle/`:

* z - h[..., 0] of the HIP forward against `scipy.integrate.quad` (fp64, adaptive) of the same integrand network
  evaluated in fp64 numpy, for the integrand nets of cfg4 / cfg2 / cfg5 ([50]^3, [100]^3, [150]^3) on >= 200 random
  elements each: at S = 20 the kernel is as close to the true integral as the TEXTBOOK Clenshaw-Curtis rule (weights
  from the closed form for even n, fp64) is -- its deviation beyond the rule's own error is fp32 roundoff --, and the
  error shrinks at S = 150 and 250;
* jac == the fp64 integrand at x (1e-5 relative);
* autograd dz/dx of the HIP op equals the HIP jac (Leibniz convention: the x-gradient is the exact integrand at the upper
  limit, not the derivative of the quadrature sum);
* z(0) = h[..., 0] exactly, and z is strictly increasing in x.
"""
import numpy as np
import pytest
import torch
from scipy import integrate

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
EPS32 = float(np.finfo(np.float32).eps)


def _net64(norm):
    """fp64 numpy copies of the integrand net's layers"""
    ps = [p.detach().double().cpu().numpy() for p in norm.integrand_net.flat_params()]
    return list(zip(ps[0::2], ps[1::2]))


def _f64(layers, t, hrow):
    """the reference's IntegrandNet (MonotonicNormalizer.py:21-38) at scalar / vector t for ONE element, fp64:
    rows (t, h) -> Linear -> ReLU ... -> Linear -> ELU + 1.05"""
    t = np.atleast_1d(np.asarray(t, dtype=np.float64))
    a = np.concatenate([t[:, None], np.broadcast_to(hrow, (t.shape[0], hrow.shape[0]))], 1)
    for k, (W, b) in enumerate(layers):
        a = a @ W.T + b
        if k + 1 < len(layers):
            a = np.maximum(a, 0.)
    a = a[:, 0]
    return np.where(a > 0, a, np.expm1(np.minimum(a, 0.))) + 1.05


def _cc_textbook(n):
    """Clenshaw-Curtis nodes / weights on [-1, 1] from the textbook closed form (even n):
    w_k = c_k / n * (1 - sum_{j=1}^{n/2} b_j / (4 j^2 - 1) cos(2 j k pi / n)), b_j = 1 for j = n/2 else 2, c_k = 1 at the ends else 2.
    Independent of the cosine-matrix construction oracle.cc_rule restates."""
    assert n % 2 == 0
    k = np.arange(n + 1)
    w = np.ones(n + 1)
    for j in range(1, n // 2 + 1):
        bj = 1. if j == n // 2 else 2.
        w -= bj / (4. * j * j - 1.) * np.cos(2. * j * k * np.pi / n)
    c = np.where((k == 0) | (k == n), 1., 2.)
    return np.cos(k * np.pi / n), c * w / n


def _rule64(layers, x, hrow, n):
    """(textbook rule's value, |what the treatment of the T_n term can move|).  Constructions of the rule differ in whether
    the LAST cosine term (j = n/2, i.e. the Chebyshev coefficient of T_n, which the n + 1 nodes cannot tell from T_0) is
    halved: the textbook form above halves it, the cosine-matrix construction of the UMNN package as recalled does not
    (DESIGN.md section 2).  The two weight vectors differ by c_k (-1)^k / (n (n^2 - 1)) -- 2.5e-4 per weight at n = 20, acting
    only on the alternating component of the integrand samples -- so the band between them is part of "the rule's own
    error" here; it is computed from the samples, not from anybody's weights."""
    t, w = _cc_textbook(n)
    f = _f64(layers, x * (t + 1.) / 2., hrow)
    k = np.arange(n + 1)
    c = np.where((k == 0) | (k == n), 1., 2.)
    band = abs(x) / 2. * abs(float(np.sum(c * (-1.) ** k * f))) / (n * (n * n - 1.))
    return x / 2. * float(np.sum(w * f)), band


CASES = [("cfg4", [50, 50, 50], 3), ("cfg2", [100, 100, 100], 4), ("cfg5", [150, 150, 150], 5)]


@pytest.mark.parametrize("tag,hidden,seed", CASES)
def test_hip_integral_vs_adaptive_quadrature(tag, hidden, seed):
    from models import MonotonicNormalizer
    torch.manual_seed(seed)
    B, d, c = 16, 13, 30                                   # 208 elements
    norm = MonotonicNormalizer(hidden, c, nb_steps=20)
    with torch.no_grad():                                  # livelier than the default init: kinks inside [0, x]
        for p in norm.integrand_net.flat_params():
            p.mul_(1.5)
    layers = _net64(norm)
    x = torch.randn(B, d) * 2.
    h = torch.randn(B, d, c)
    x[0, 0] = 0.                                           # an element with an empty interval
    xn, hn = x.double().numpy(), h.double().numpy()
    ref = np.zeros((B, d))
    for b in range(B):
        for i in range(d):
            if xn[b, i] != 0.:
                ref[b, i] = integrate.quad(lambda t: float(_f64(layers, t, hn[b, i])[0]), 0., xn[b, i],
                                           epsabs=1e-11, epsrel=1e-11, limit=400)[0]
    fx = np.array([[_f64(layers, xn[b, i], hn[b, i])[0] for i in range(d)] for b in range(B)])
    norm = norm.to(DEV)
    xg, hg = x.to(DEV), h.to(DEV)
    scale = np.maximum(np.abs(ref), 1e-2)
    worst = {}
    for S in (20, 150, 250):
        norm.nb_steps = S
        with torch.no_grad():
            z, jac = norm(xg, hg)
        integ = (z.double().cpu() - h[:, :, 0].double()).numpy()
        err_hip = np.abs(integ - ref)
        rb = np.array([[_rule64(layers, xn[b, i], hn[b, i], S) for i in range(d)] for b in range(B)])
        err_rule = np.abs(rb[:, :, 0] - ref) + rb[:, :, 1]
        # the kernel's deviation from the true integral = the rule's own error + fp32 roundoff: S + 1 weighted integrand
        # values of magnitude ~|I| summed in fp32, z0 added in fp32 (the subtraction above cancels it: |h0| enters), the
        # integrand's own fp32 evaluation (~1e-6 relative)
        allow = err_rule + 16. * EPS32 * (np.abs(ref) * np.sqrt(S + 1.) + np.abs(hn[:, :, 0]) + np.abs(xn) * fx) + 1e-7
        k = np.unravel_index(np.argmax(err_hip - allow), err_hip.shape)
        assert (err_hip <= allow).all(), (tag, S, k, err_hip[k], err_rule[k], allow[k], ref[k])
        worst[S] = float((err_hip / scale).max())
        # jac = f(x; h)
        assert np.abs(jac.double().cpu().numpy() - fx).max() <= 1e-5 * np.abs(fx).max()
        assert np.all(np.abs(jac.double().cpu().numpy() - fx) <= 1e-6 + 1e-5 * np.abs(fx))
        assert z[0, 0].item() == h[0, 0, 0].item()
    # convergence to the adaptive quadrature with the node count
    assert worst[150] < worst[20] and worst[250] < worst[20], worst
    assert worst[250] < 2e-4, worst
    print("\n[integral pin %s] max |HIP - quad| / max(|quad|, 1e-2): S=20 %.2e, S=150 %.2e, S=250 %.2e"
          % (tag, worst[20], worst[150], worst[250]))


@pytest.mark.parametrize("tag,hidden,seed", CASES)
def test_hip_leibniz_gradient_zero_and_monotone(tag, hidden, seed):
    from models import MonotonicNormalizer
    torch.manual_seed(seed + 100)
    B, d, c, S = 40, 9, 30, 20
    norm = MonotonicNormalizer(hidden, c, nb_steps=S).to(DEV)
    x = (torch.randn(B, d) * 2.).to(DEV).requires_grad_(True)
    h = torch.randn(B, d, c).to(DEV)
    z, jac = norm(x, h)
    gx, = torch.autograd.grad(z.sum(), x)
    # Leibniz: dz/dx = f(x; h) -- the backward kernel re-evaluates the integrand at x; it is the forward's jac (same
    # arithmetic on the same inputs; a few ulps where the two kernels order the last layer's sum differently)
    assert (jac > .05).all()
    diff = (gx - jac.detach()).abs()
    assert (diff <= 4. * EPS32 * jac.detach().abs()).all(), (tag, diff.max().item())
    nbit = int((gx != jac.detach()).sum())
    print("\n[leibniz %s] dz/dx vs jac: %d of %d entries differ in the last bits (max %.1f ulp)"
          % (tag, nbit, gx.numel(), (diff / (EPS32 * jac.detach().abs())).max().item()))
    # z(0) = h0 bit for bit
    with torch.no_grad():
        z0, _ = norm(torch.zeros(B, d, device=DEV), h)
    assert torch.equal(z0, h[:, :, 0])
    # strictly increasing in x (jac > 0.05 everywhere: ELU + 1.05)
    with torch.no_grad():
        xs = torch.linspace(-4., 4., 33, device=DEV)
        zs = torch.stack([norm(torch.full((B, d), float(v), device=DEV), h)[0] for v in xs])
    assert (zs[1:] > zs[:-1]).all()
