"""`models` package of the MI355X build.  It exports the names the reference's `models/__init__.py` exports, so
putting this directory on PYTHONPATH ahead of the reference makes its drivers (`from models import ...`) run on the
gfx950 kernels; everything else (`models.NormalizingFlow`, `models.NormalizingFlowFactories`, ...) is importable by
the same module paths as well."""
from .Conditionners import (Conditioner, AutoregressiveConditioner, CouplingConditioner, DAGConditioner)
from .Normalizers import (AffineNormalizer, MonotonicNormalizer)
from .MLP import (MLP, MNISTCNN, CIFAR10CNN)
from .NormalizingFlowFactories import buildFCNormalizingFlow

__all__ = ["MLP", "MNISTCNN", "CIFAR10CNN", "buildFCNormalizingFlow", "AutoregressiveConditioner", "DAGConditioner",
           "CouplingConditioner", "Conditioner", "AffineNormalizer", "MonotonicNormalizer"]
