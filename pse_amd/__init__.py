"""pse_amd -- MI355X-native Positively Split Ewald engine (drop-in for the hot path of HOOMD's PSEv1 plugin).

Layout: csrc/ (HIP kernels + C-ABI + C++ host classes), engine.py (ctypes owner of a handle),
integrate.py / shear_function.py / variant.py (mirror of the reference's Python UI, PSEv1/*.py).
"""
from .engine import Engine, host_lanczos_sqrt_e1, host_select_params  # noqa: F401
from ._lib import PSEError  # noqa: F401
