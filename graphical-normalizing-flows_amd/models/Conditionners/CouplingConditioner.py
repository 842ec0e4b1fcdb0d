"""Coupling conditioner on the MI355X path.

Public surface and `state_dict` keys follow the reference's models/Conditionners/CouplingConditioner.py
(`CouplingMLP.net.{0,2,..}`, `CouplingConditioner.constants`, `.embeding_net` with the reference's spelling):
the first d - floor(d/2) variables get learned constant embeddings, the remaining floor(d/2) variables are embedded
by an MLP of the first group (reference :21-39).  The MLP runs as one fused MFMA GEMM chain (gnf_hip.ops.mlp)."""
import torch
import torch.nn as nn

from gnf_hip import ops
from .Conditioner import Conditioner, linear_pairs, no_context, relu_stack


def _split(in_size):
    """(number of independent variables, number of conditioned variables)"""
    conditioned = in_size // 2
    return in_size - conditioned, conditioned


class CouplingMLP(nn.Module):
    """Parameter container of the embedding MLP: [indep (+cond_in)] -> hidden... -> out_size * conditioned."""

    def __init__(self, in_size, hidden, out_size, cond_in=0):
        super().__init__()
        indep, conditioned = _split(in_size)
        self.net = relu_stack([indep + cond_in] + list(hidden) + [out_size * conditioned])

    def forward(self, x):
        return ops.mlp(x, linear_pairs(self.net))


class CouplingConditioner(Conditioner):
    def __init__(self, in_size, hidden, out_size, cond_in=0):
        super().__init__()
        self.in_size, self.out_size, self.cond_in = in_size, out_size, cond_in
        self.indep_size, self.cond_size = _split(in_size)
        self.embeding_net = CouplingMLP(in_size, hidden, out_size, cond_in)
        self.constants = nn.Parameter(torch.randn(self.indep_size, out_size))

    def forward(self, x, context=None):
        no_context(context, self.cond_in)
        batch = x.shape[0]
        learned = self.embeding_net(x[:, :self.indep_size]).view(batch, self.cond_size, self.out_size)
        return torch.cat((self.constants.expand(batch, self.indep_size, self.out_size), learned), dim=1)

    def depth(self):
        # one pass fixes the independent half, the second the conditioned half (NormalizingFlowStep.invert runs depth()+1)
        return 1
