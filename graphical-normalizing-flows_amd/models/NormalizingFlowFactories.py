"""Flow factories of the reference (models/NormalizingFlowFactories.py), building the MI355X-backed classes.

`NormalLogDensity`, `buildFCNormalizingFlow`, `MNIST_A_prior` and `buildMNISTNormalizingFlow` keep the reference's
names, arguments and the `state_dict` layout of what they build (`z_log_density.pi`, `steps.N....`)."""
import math

import torch
import torch.nn as nn

from gnf_hip import ops
from .Normalizers import *            # noqa: F401,F403  (the reference star-imports both packages here)
from .Conditionners import *          # noqa: F401,F403
from .NormalizingFlow import NormalizingFlowStep, FCNormalizingFlow, CNNormalizingFlow
from .MLP import MNISTCNN, CIFAR10CNN  # noqa: F401


class NormalLogDensity(nn.Module):
    """Row-wise log N(z; 0, I) = -1/2 sum_d (log 2 pi + z_d^2)  (reference :10-16), one fused reduction kernel.  The
    `pi` buffer exists only because reference checkpoints carry `z_log_density.pi`.  Always computed from the z it is
    handed; FCNormalizingFlow.loss folds this density into its own launch (gnf_hip.ops.NllLossFn) when this exact class
    is the flow's base density."""

    standard_normal = True               # FCNormalizingFlow.loss may fold this density into its own launch

    def __init__(self):
        super().__init__()
        self.register_buffer("pi", torch.tensor(math.pi))

    def forward(self, z):
        return ops.NormalLogDensityFn.apply(z)

    _std_forward = forward               # the fold applies only while type(module).forward IS this function: a subclass that
                                         # overrides forward (tempered, scaled, conditional density) is called instead


def buildFCNormalizingFlow(nb_steps, conditioner_type, conditioner_args, normalizer_type, normalizer_args):
    """`nb_steps` independent (conditioner, normalizer) pairs on a standard-normal base density (reference :19-32)."""
    return FCNormalizingFlow([NormalizingFlowStep(conditioner_type(**conditioner_args),
                                                  normalizer_type(**normalizer_args)) for _ in range(nb_steps)],
                             NormalLogDensity())


def MNIST_A_prior(in_size, kernel):
    """Adjacency prior of an in_size x in_size pixel grid: pixel p depends on every pixel of its (2 kernel + 1)^2
    window, itself excluded (reference :35-46)."""
    n = in_size
    row = torch.arange(n).repeat_interleave(n)
    col = torch.arange(n).repeat(n)
    A = torch.zeros(n * n, n * n)
    for dr in range(-kernel, kernel + 1):
        for dc in range(-kernel, kernel + 1):
            r2, c2 = row + dr, col + dc
            inside = (r2 >= 0) & (r2 < n) & (c2 >= 0) & (c2 < n)
            A[(row * n + col)[inside], (r2 * n + c2)[inside]] = 1.
    A.fill_diagonal_(0.)
    return A


# the scales of the MNIST factories: (image size, block dropped after the scale, MNISTCNN fc sizes)
_MNIST_SCALES = (([1, 28, 28], [1, 2, 2], [2304, 128]),
                 ([1, 14, 14], [1, 2, 2], [400, 64]),
                 ([1, 7, 7], [1, 1, 1], [16, 16]))


def _mnist_dag_steps(n_steps, img_size, fc, normalizer_type, normalizer_args, l1, nb_epoch_update, hot_encoding,
                     prior_kernel):
    """`n_steps` DAG-conditioner steps on one image scale, each with its own MNISTCNN embedding net."""
    pixels = img_size[0] * img_size[1] * img_size[2]
    monotonic = normalizer_type is MonotonicNormalizer
    emb_size = 30 if monotonic else 2
    out = []
    for _ in range(n_steps):
        prior = MNIST_A_prior(img_size[1], prior_kernel) if prior_kernel is not None else None
        cond = DAGConditioner(pixels, MNISTCNN(fc_l=fc, size_img=img_size, out_d=emb_size), emb_size, l1=l1,
                              nb_epoch_update=nb_epoch_update, hot_encoding=hot_encoding, A_prior=prior)
        if monotonic:   # the reference widens cond_size by the one-hot width when hot_encoding is on (:67,:87)
            norm = normalizer_type(**normalizer_args, cond_size=emb_size + pixels if hot_encoding else emb_size)
        else:
            norm = normalizer_type(**normalizer_args)
        out.append(NormalizingFlowStep(cond, norm))
    return out


def buildMNISTNormalizingFlow(nb_inner_steps, normalizer_type, normalizer_args, l1=0., nb_epoch_update=10,
                              hot_encoding=False, prior_kernel=None):
    """MNIST DAG flows of the reference (:49-97): `len(nb_inner_steps) == 1` -> one 28x28 scale (FCNormalizingFlow),
    `== 3` -> scales 28 / 14 / 7 chained by CNNormalizingFlow, anything else -> None."""
    common = (normalizer_type, normalizer_args, l1, nb_epoch_update, hot_encoding, prior_kernel)
    if len(nb_inner_steps) == 1:
        img_size, _, fc = _MNIST_SCALES[0]
        return FCNormalizingFlow(_mnist_dag_steps(nb_inner_steps[0], img_size, fc, *common), NormalLogDensity())
    if len(nb_inner_steps) == 3:
        scales = []
        for n_steps, (img_size, _, fc) in zip(nb_inner_steps, _MNIST_SCALES):
            flow = FCNormalizingFlow(_mnist_dag_steps(n_steps, img_size, fc, *common), None)
            flow.img_sizes = img_size
            scales.append(flow)
        return CNNormalizingFlow(scales, NormalLogDensity(), [drop for _, drop, _ in _MNIST_SCALES])
    return None
