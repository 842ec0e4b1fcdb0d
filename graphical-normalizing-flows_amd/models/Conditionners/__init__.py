"""Conditioners: x[B,d] -> h[B,d,hs] (package name spelled as in the reference)."""
from .Conditioner import Conditioner
from .CouplingConditioner import CouplingConditioner
from .AutoregressiveConditioner import AutoregressiveConditioner
from .DAGConditioner import DAGConditioner

__all__ = ["Conditioner", "AutoregressiveConditioner", "CouplingConditioner", "DAGConditioner"]
