import torch
import torch.nn as nn

from gnf_hip import ops
from .Conditioner import Conditioner, linear_pairs, no_context


class CouplingMLP(nn.Module):
    """Parameter container, same layout/keys as reference CouplingConditioner.py:6-19."""

    def __init__(self, in_size, hidden, out_size, cond_in=0):
        super(CouplingMLP, self).__init__()
        l1 = [in_size - int(in_size / 2) + cond_in] + hidden
        l2 = hidden + [out_size * int(in_size / 2)]
        layers = []
        for h1, h2 in zip(l1, l2):
            layers += [nn.Linear(h1, h2), nn.ReLU()]
        layers.pop()
        self.net = nn.Sequential(*layers)

    def forward(self, x):
        return ops.mlp(x, linear_pairs(self.net))


class CouplingConditioner(Conditioner):
    """First d-floor(d/2) dims: learned constants; last floor(d/2) dims: MLP of the first
    ones (reference CouplingConditioner.py:21-39)."""

    def __init__(self, in_size, hidden, out_size, cond_in=0):
        super(CouplingConditioner, self).__init__()
        self.in_size = in_size
        self.out_size = out_size
        self.cond_size = int(in_size / 2)
        self.indep_size = in_size - self.cond_size
        self.cond_in = cond_in
        self.embeding_net = CouplingMLP(in_size, hidden, out_size, cond_in)   # (sic) reference spelling
        self.constants = nn.Parameter(torch.randn(self.indep_size, out_size))

    def forward(self, x, context=None):
        no_context(context, self.cond_in)
        h1 = self.constants.unsqueeze(0).expand(x.shape[0], -1, -1)
        h2 = self.embeding_net(x[:, :self.indep_size]).view(x.shape[0], self.cond_size, self.out_size)
        return torch.cat((h1, h2), 1)

    def depth(self):
        return 1
