"""gnf_hip: hand-written gfx950 (MI355X) kernels for the Graphical-Normalizing-Flows hot
path, reached through the C ABI of include/gnf_hip.h (libgnf_hip.so, loaded with ctypes).

The package never falls back to the CPU or to an eager PyTorch implementation: a
missing library raises ImportError on first use, a non-HIP tensor raises GnfError."""
from . import abi, ops  # noqa: F401
from .abi import GnfError, load  # noqa: F401
