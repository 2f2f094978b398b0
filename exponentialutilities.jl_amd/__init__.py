"""exponentialutilities.jl_amd -- MI355X-native Krylov exp(tA)v engine behind the
ExponentialUtilities.jl expv / phiv / arnoldi / KrylovSubspace API surface.

The directory name contains a dot, so it is loaded through ``expv_mi_loader.load()`` (repo root),
which registers it as the module ``exponentialutilities_jl_amd``.  The compute path is the HIP
library ``libexpv_mi.so`` built from ``csrc/``; importing fails loudly when it has not been built.
"""
from . import _lib
from .api import *  # noqa: F401,F403
from .api import __all__  # noqa: F401

_lib.load()      # fail at import time, not at first use, when the HIP extension is missing
__version__ = _lib.load().expv_mi_version().decode()
