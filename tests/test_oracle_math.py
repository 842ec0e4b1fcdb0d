"""CPU: independent-mathematics checks of the oracle's restatement of the UMNN 1.0
Clenshaw-Curtis integral (UMNN source is unavailable: **UMNN parity unpinned**,
SURVEY.md 8c).  Uses the golden IntegrandNet parameters recorded from the reference."""
import numpy as np
import pytest
import torch
from scipy import integrate

from conftest import load_golden, params_of, linear_layers
from oracle import gnf_oracle as O


@pytest.mark.parametrize("S", [15, 20, 21, 24, 29, 40, 150, 250])
def test_cc_rule_polynomial_exactness(S):
    w, t = O.cc_rule(S)
    assert abs(w.sum() - 2.) < 1e-13 and (w > 0).all()
    assert np.allclose(t, np.cos(np.arange(S + 1) * np.pi / S))
    top = S if S % 2 else S - 1      # even S: last cosine term un-halved -> exact to degree S-1 (measured)
    for p in range(0, top + 1):
        exact = 0. if p % 2 else 2. / (p + 1)
        assert abs((w * t ** p).sum() - exact) < 1e-12, (S, p)


def _net64():
    g = load_golden("integrand")
    layers = [(W.double(), b.double()) for W, b in linear_layers(params_of(g), "")]
    return g, layers


def test_integral_vs_adaptive_quadrature_fp64():
    g, layers = _net64()
    x, h = g["x"].double() * 2, g["h"].double()
    errs = {}
    for S in (20, 150, 250):
        z = O.monotonic_integral(x, h, layers, S)
        ref = torch.zeros_like(z)
        for b in range(x.shape[0]):
            for i in range(x.shape[1]):
                f = lambda t: O.integrand(torch.tensor([[t]], dtype=torch.float64), h[b:b + 1, i:i + 1], layers).item()
                ref[b, i] = integrate.quad(f, 0., x[b, i].item(), epsabs=1e-12, epsrel=1e-12, limit=400)[0]
        errs[S] = ((z - ref).abs() / ref.abs().clamp_min(1e-3)).max().item()
    assert errs[20] < 5e-3 and errs[150] < 2e-4 and errs[250] < 1e-4, errs   # piecewise-linear-ish integrand
    assert errs[250] <= errs[20]


def test_monotone_zero_and_finite_difference():
    g, layers = _net64()
    h = g["h"].double()
    layers64 = layers
    xs = torch.linspace(-3, 3, 25, dtype=torch.float64)
    zs = torch.stack([O.monotonic_forward(torch.full_like(g["x"].double(), v), h, layers64, 150)[0] for v in xs])
    assert (zs[1:] > zs[:-1]).all()                                  # strictly increasing in x
    z0, _ = O.monotonic_forward(torch.zeros_like(g["x"].double()), h, layers64, 20)
    assert torch.allclose(z0, h[:, :, 0])                            # z(0) = h[...,0]
    x = g["x"].double()
    eps = 1e-4
    zp, _ = O.monotonic_forward(x + eps, h, layers64, 250)
    zm, _ = O.monotonic_forward(x - eps, h, layers64, 250)
    _, jac = O.monotonic_forward(x, h, layers64, 250)
    assert ((zp - zm) / (2 * eps) - jac).abs().max() < 5e-3          # dz/dx ~= importable Jacobian


def test_autograd_conventions():
    g, layers = _net64()
    x = g["x"].double().requires_grad_(True)
    h = g["h"].double().requires_grad_(True)
    ps = [p.clone().requires_grad_(True) for Wb in layers for p in Wb]
    lay = [(ps[i], ps[i + 1]) for i in range(0, len(ps), 2)]
    z, jac = O.monotonic_forward(x, h, lay, 20)
    gz = torch.randn_like(z)
    grads = torch.autograd.grad((z * gz).sum(), [x, h] + ps)
    # Leibniz: dz/dx = f(x;h) exactly
    assert torch.allclose(grads[0], (jac * gz).detach(), rtol=1e-12, atol=1e-14)
    # h / theta gradients = exact derivative of the quadrature sum (xT held fixed)
    hh = h.detach().clone().requires_grad_(True)
    ps2 = [p.detach().clone().requires_grad_(True) for p in ps]
    lay2 = [(ps2[i], ps2[i + 1]) for i in range(0, len(ps2), 2)]
    z2 = O.monotonic_integral(x.detach(), hh, lay2, 20) + hh[:, :, 0]
    ref = torch.autograd.grad((z2 * gz).sum(), [hh] + ps2)
    for a, b in zip(grads[1:], ref):
        assert torch.allclose(a, b, rtol=1e-9, atol=1e-12)


def test_inverse_round_trip():
    g, layers = _net64()
    x = g["x"].double() * 2
    h = g["h"].double()
    z, _ = O.monotonic_forward(x, h, layers, 40)
    xr = O.monotonic_inverse(z.detach(), h, layers, 40)
    assert (xr - x).abs().max() < 40. / 2 ** 20      # bisection resolution (Monotonic:69-83)
