from .Normalizer import Normalizer
from .AffineNormalizer import AffineNormalizer
from .MonotonicNormalizer import MonotonicNormalizer
