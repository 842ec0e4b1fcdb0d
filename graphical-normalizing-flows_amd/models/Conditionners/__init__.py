from .Conditioner import Conditioner
from .AutoregressiveConditioner import AutoregressiveConditioner
from .CouplingConditioner import CouplingConditioner
from .DAGConditioner import DAGConditioner
