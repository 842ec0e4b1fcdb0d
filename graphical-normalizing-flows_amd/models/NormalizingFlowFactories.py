from math import pi

import torch
import torch.nn as nn

from gnf_hip import ops
from .Normalizers import *
from .Conditionners import *
from .NormalizingFlow import NormalizingFlowStep, FCNormalizingFlow, CNNormalizingFlow
from .MLP import MNISTCNN, CIFAR10CNN


class NormalLogDensity(nn.Module):
    """log N(z; 0, I) per row (reference NormalizingFlowFactories.py:10-16); the `pi` buffer is
    kept for state_dict compatibility (`z_log_density.pi`)."""

    def __init__(self):
        super(NormalLogDensity, self).__init__()
        self.register_buffer("pi", torch.tensor(pi))

    def forward(self, z):
        return ops.NormalLogDensityFn.apply(z)


def buildFCNormalizingFlow(nb_steps, conditioner_type, conditioner_args, normalizer_type, normalizer_args):
    """nb_steps x (conditioner, normalizer) -> FCNormalizingFlow (reference :19-32)."""
    flow_steps = []
    for step in range(nb_steps):
        conditioner = conditioner_type(**conditioner_args)
        normalizer = normalizer_type(**normalizer_args)
        flow_steps.append(NormalizingFlowStep(conditioner, normalizer))
    return FCNormalizingFlow(flow_steps, NormalLogDensity())


def MNIST_A_prior(in_size, kernel):
    """(2k+1)^2-window pixel adjacency minus self on an in_size x in_size grid (reference :35-46)."""
    n = in_size
    A = torch.zeros(n * n, n * n)
    r = torch.arange(n).view(-1, 1).expand(n, n).reshape(-1)
    c = torch.arange(n).view(1, -1).expand(n, n).reshape(-1)
    p = r * n + c
    for di in range(-kernel, kernel + 1):
        for dj in range(-kernel, kernel + 1):
            rr, cc = r + dj, c + di
            ok = (rr >= 0) & (rr < n) & (cc >= 0) & (cc < n)
            A[p[ok], (rr * n + cc)[ok]] = 1.
    A.fill_diagonal_(0.)
    return A


def buildMNISTNormalizingFlow(nb_inner_steps, normalizer_type, normalizer_args, l1=0., nb_epoch_update=10,
                              hot_encoding=False, prior_kernel=None):
    """MNIST DAG flows of the reference (:49-97): one 28x28 scale, or three scales 28/14/7 (CNNormalizingFlow)."""
    if len(nb_inner_steps) == 3:
        img_sizes = [[1, 28, 28], [1, 14, 14], [1, 7, 7]]
        dropping_factors = [[1, 2, 2], [1, 2, 2], [1, 1, 1]]
        fc_l = [[2304, 128], [400, 64], [16, 16]]
        outter_steps = []
        for i, fc in enumerate(fc_l):
            in_size = img_sizes[i][0] * img_sizes[i][1] * img_sizes[i][2]
            inner_steps = []
            for step in range(nb_inner_steps[i]):
                emb_s = 2 if normalizer_type is AffineNormalizer else 30
                hidden = MNISTCNN(fc_l=fc, size_img=img_sizes[i], out_d=emb_s)
                A_prior = MNIST_A_prior(img_sizes[i][1], prior_kernel) if prior_kernel is not None else None
                cond = DAGConditioner(in_size, hidden, emb_s, l1=l1, nb_epoch_update=nb_epoch_update,
                                      hot_encoding=hot_encoding, A_prior=A_prior)
                if normalizer_type is MonotonicNormalizer:
                    emb_s = 30 + in_size if hot_encoding else 30
                    norm = normalizer_type(**normalizer_args, cond_size=emb_s)
                else:
                    norm = normalizer_type(**normalizer_args)
                inner_steps.append(NormalizingFlowStep(cond, norm))
            flow = FCNormalizingFlow(inner_steps, None)
            flow.img_sizes = img_sizes[i]
            outter_steps.append(flow)
        return CNNormalizingFlow(outter_steps, NormalLogDensity(), dropping_factors)
    elif len(nb_inner_steps) == 1:
        inner_steps = []
        for step in range(nb_inner_steps[0]):
            emb_s = 2 if normalizer_type is AffineNormalizer else 30
            hidden = MNISTCNN(fc_l=[2304, 128], size_img=[1, 28, 28], out_d=emb_s)
            A_prior = MNIST_A_prior(28, prior_kernel) if prior_kernel is not None else None
            cond = DAGConditioner(1 * 28 * 28, hidden, emb_s, l1=l1, nb_epoch_update=nb_epoch_update,
                                  hot_encoding=hot_encoding, A_prior=A_prior)
            if normalizer_type is MonotonicNormalizer:
                emb_s = 30 + 28 * 28 if hot_encoding else 30
                norm = normalizer_type(**normalizer_args, cond_size=emb_s)
            else:
                norm = normalizer_type(**normalizer_args)
            inner_steps.append(NormalizingFlowStep(cond, norm))
        return FCNormalizingFlow(inner_steps, NormalLogDensity())
    else:
        return None
