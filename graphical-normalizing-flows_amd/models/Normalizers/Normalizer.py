import torch.nn as nn


class Normalizer(nn.Module):
    """Plug-in protocol of the reference (models/Normalizers/Normalizer.py:4-28):
    forward(x[B,d], h[B,d,hs], context=None) -> (z[B,d], jac[B,d]) with jac the diagonal
    Jacobian (not its log); inverse_transform(z, h, context=None) -> x[B,d]."""

    def __init__(self):
        super(Normalizer, self).__init__()

    def forward(self, x, h, context=None):
        pass

    def inverse_transform(self, z, h, context=None):
        pass
