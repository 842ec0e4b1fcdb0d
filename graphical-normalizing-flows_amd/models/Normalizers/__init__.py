"""Normalizers of the MI355X build: (x[B,d], h[B,d,hs]) -> (z[B,d], jac[B,d]) plus `inverse_transform`.

`AffineNormalizer` is one fused streaming kernel (transform + log-determinant row sum); `MonotonicNormalizer` runs the
Clenshaw-Curtis quadrature of the UMNN integrand network on fp32 MFMA (forward, backward, fused bisection inverse).
The names exported here are the ones the reference's package exports."""
from .Normalizer import Normalizer
from .AffineNormalizer import AffineNormalizer
from .MonotonicNormalizer import MonotonicNormalizer

__all__ = ["Normalizer", "AffineNormalizer", "MonotonicNormalizer"]
