import torch.nn as nn


def linear_pairs(seq):
    """[(weight, bias)] of the nn.Linear members of an nn.Sequential, in order."""
    return [(m.weight, m.bias) for m in seq if isinstance(m, nn.Linear)]


def no_context(context, cond_in=0):
    """`context` is dead in the reference: all three conditioners fail when cond_in > 0
    (SURVEY.md 8b), so only context=None is supported."""
    if context is not None or cond_in:
        raise NotImplementedError("context conditioning (cond_in > 0) is not supported "
                                  "(it raises in the reference as well)")


class Conditioner(nn.Module):
    """Plug-in protocol of the reference (models/Conditionners/Conditioner.py:4-23):
    forward(x[B,d], context=None) -> h[B,d,hs]; depth() -> longest path of the equivalent
    Bayesian network; attribute is_invertible."""

    def __init__(self):
        super(Conditioner, self).__init__()
        self.is_invertible = True

    def forward(self, x, context=None):
        pass

    def depth(self):
        pass


def relu_stack(sizes):
    """nn.Sequential(Linear, ReLU, Linear, ..., Linear) over consecutive `sizes`: the parameter container every
    conditioner MLP of the reference uses (keys net.0, net.2, ...)."""
    mods = []
    for i, (n_in, n_out) in enumerate(zip(sizes[:-1], sizes[1:])):
        if i:
            mods.append(nn.ReLU())
        mods.append(nn.Linear(n_in, n_out))
    return nn.Sequential(*mods)

