import torch
import torch.nn as nn

from gnf_hip import ops
from .Normalizer import Normalizer


class ELUPlus(nn.Module):
    """ELU(x) + 1.05 (reference MonotonicNormalizer.py:12-18); evaluated inside the kernel."""

    def __init__(self):
        super().__init__()
        self.elu = nn.ELU()

    def forward(self, x):
        return self.elu(x) + 1.05


class IntegrandNet(nn.Module):
    """Parameter container with the reference's layout and state_dict keys
    (`net.<2k>.weight/bias`, MonotonicNormalizer.py:21-31).  Its arithmetic is fused into
    the quadrature kernel; calling it evaluates f(x;h) through that same kernel."""

    def __init__(self, hidden, cond_in):
        super(IntegrandNet, self).__init__()
        l1 = [1 + cond_in] + hidden
        l2 = hidden + [1]
        layers = []
        for h1, h2 in zip(l1, l2):
            layers += [nn.Linear(h1, h2), nn.ReLU()]
        layers.pop()
        layers.append(ELUPlus())
        self.net = nn.Sequential(*layers)
        self.cond_in = cond_in

    def flat_params(self):
        ps = []
        for m in self.net:
            if isinstance(m, nn.Linear):
                ps += [m.weight, m.bias]
        return ps

    def forward(self, x, h):
        """x: [B,d]; h: flattened cond-major [B, c*d] as the reference passes it (:33-38)."""
        B, d = x.shape
        h3 = h.view(B, -1, d).permute(0, 2, 1)
        _, jac = ops.MonotonicFn.apply(x, h3, 2, *self.flat_params())
        return jac


class MonotonicNormalizer(Normalizer):
    """UMNN monotonic transformer (reference MonotonicNormalizer.py:41-83): z = int_0^x f(t;h)dt
    + h[...,0] by Clenshaw-Curtis quadrature with `nb_steps`+1 nodes, jac = f(x;h); inverse by
    20-step bisection.  `nb_steps` is re-read on every call (drivers jitter it per batch)."""

    def __init__(self, integrand_net, cond_size, nb_steps=20, solver="CC"):
        super(MonotonicNormalizer, self).__init__()
        if type(integrand_net) is list:
            self.integrand_net = IntegrandNet(integrand_net, cond_size)
        else:
            self.integrand_net = integrand_net
        self.solver = solver
        self.nb_steps = nb_steps

    def _fused(self):
        """the reference's IntegrandNet architecture (list of hidden sizes) is fused into the gfx950 quadrature kernel;
        any other module passed as `integrand_net` (reference :44-48) is evaluated through PyTorch on the device"""
        return type(self.integrand_net) is IntegrandNet

    def _params(self):
        return self.integrand_net.flat_params()

    def forward(self, x, h, context=None):
        # "CC" loops over the nodes, "CCParallel" batches them: same rule, same numbers up to
        # summation order; both map to the one fused kernel.  Unknown solver -> None (:64-65).
        if self.solver not in ("CC", "CCParallel"):
            return None
        if not self._fused():
            return ops.module_monotonic(x, h, self.integrand_net, int(self.nb_steps))
        return ops.MonotonicFn.apply(x, h, int(self.nb_steps), *self._params())

    def forward_logdet(self, x, h, context=None):
        """(z, log|det J|) for NormalizingFlowStep: the integrand kernels emit z and jac per element (a row's elements are
        spread over wavefronts and workgroups), log(jac).sum(1) is one reduction pass behind them"""
        out = self.forward(x, h, context)
        if out is None:
            return None
        z, jac = out
        return z, ops.LogSumRowsFn.apply(jac)

    def inverse_transform(self, z, h, context=None):
        with torch.no_grad():
            if not self._fused():
                return ops.module_monotonic_inverse(z, h, self.integrand_net, int(self.nb_steps))
            return ops.monotonic_inverse(z, h, int(self.nb_steps), self._params(), pack=self._held_pack)

    def inverse_transform_into(self, z, h, out, cols, context=None):
        """out[:, cols] = inverse_transform(z, h).t() for a variable-major problem (z [R, B], h [R, B, c], out [B, d], cols
        int32 [R]) without the transposing copy and the index_put: the kernel writes where the values belong.  Returns
        False when the fused kernel does not apply (the caller then scatters the result of inverse_transform itself)."""
        if not (self._fused() and out.is_cuda and out.is_contiguous() and out.dtype == torch.float32):
            return False
        with torch.no_grad():
            ops.monotonic_inverse(z, h, int(self.nb_steps), self._params(), pack=self._held_pack, out=out, out_cols=cols)
        return True

    _held_pack = None

    def hold_pack(self):
        """context manager for a caller that inverts many times with unchanged parameters (the levels / fixed-point
        passes of one `invert`): the kernels' weight image is built once instead of once per call"""
        import contextlib

        @contextlib.contextmanager
        def hold():
            prev = self._held_pack
            if self._fused() and prev is None:
                ps = self._params()
                self._held_pack = ops.monotonic_pack(ps, ps[0])
            try:
                yield
            finally:
                self._held_pack = prev
        return hold()
