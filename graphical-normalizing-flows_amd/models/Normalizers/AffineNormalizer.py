from gnf_hip import ops
from .Normalizer import Normalizer


class AffineNormalizer(Normalizer):
    """z = x*exp(clamp(h[...,1],-5,2)) + clamp(h[...,0],-5,5), jac = sigma
    (reference models/Normalizers/AffineNormalizer.py:6-17) on the fused gfx950 kernel.

    `inplace_clamp=True` reproduces the reference's in-place `clamp_` of h (the clamped
    values are written back into the conditioner output); off by default because nothing
    on the flow path reads h afterwards."""

    def __init__(self):
        super(AffineNormalizer, self).__init__()
        self.inplace_clamp = False

    def forward(self, x, h, context=None):
        z, jac, _, _ = ops.AffineFn.apply(x, h, self.inplace_clamp)
        return z, jac

    def forward_logdet(self, x, h, context=None):
        """(z, log|det J|) with the row reduction fused into the kernel that computes z (used by NormalizingFlowStep)"""
        z, _, logdet, _ = ops.AffineFn.apply(x, h, self.inplace_clamp, False, False)
        return z, logdet

    def inverse_transform(self, z, h, context=None):
        return ops.affine_inverse(z, h)
