"""Drop-in mirror of the reference's `models` package (models/__init__.py:1-4): same
class names, constructor arguments, attributes and state_dict keys; the arithmetic runs
in the gfx950 kernels of gnf_hip."""
from .MLP import MLP, MNISTCNN, CIFAR10CNN
from .NormalizingFlowFactories import buildFCNormalizingFlow
from .Conditionners import AutoregressiveConditioner, DAGConditioner, CouplingConditioner, Conditioner
from .Normalizers import AffineNormalizer, MonotonicNormalizer
