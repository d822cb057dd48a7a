"""pse_amd -- MI355X-native Positively Split Ewald engine (drop-in for the hot path of HOOMD's PSEv1 plugin).

Layout: csrc/ (HIP kernels + C-ABI + C++ host classes), engine.py (ctypes owner of a handle),
integrate.py / shear_function.py / variant.py (mirror of the reference's Python UI, PSEv1/*.py).
"""
try:   # torch first: its bundled HIP/rocFFT/RCCL (same SONAMEs) must be the copies libpse_amd.so and _PSEv1 bind to
    import torch as _torch  # noqa: F401
except ImportError:  # pragma: no cover - the C-ABI itself does not need torch
    _torch = None
from .engine import Engine, host_lanczos_sqrt_e1, host_select_params  # noqa: F401
from ._lib import PSEError  # noqa: F401
