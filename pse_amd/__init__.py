"""pse_amd -- MI355X-native Positively Split Ewald engine (drop-in for the hot path of HOOMD's PSEv1 plugin).

Layout: csrc/ (HIP kernels + C-ABI + C++ host classes), engine.py (ctypes owner of a handle),
integrate.py / shear_function.py / variant.py (mirror of the reference's Python UI, PSEv1/*.py).
"""
from ._lib import _ASAN_DIR   # the CPU sanitizer run (tools/asan.py; python -m pse_amd.build --asan): marker-checked and announced there
if _ASAN_DIR:                # ... its _PSEv1 module shadows the product's
    __path__.insert(0, _ASAN_DIR)
try:   # torch first: its bundled HIP/rocFFT/RCCL (same SONAMEs) must be the copies libpse_amd.so and _PSEv1 bind to
    if _ASAN_DIR:
        raise ImportError   # the sanitizer build links no device library (and torch under a preloaded ASan takes minutes to load)
    import torch as _torch  # noqa: F401
except ImportError:  # pragma: no cover - the C-ABI itself does not need torch
    _torch = None
from .engine import Engine, host_lanczos_sqrt_e1, host_select_params  # noqa: F401
from ._lib import PSEError  # noqa: F401
