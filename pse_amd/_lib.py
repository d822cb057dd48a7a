"""ctypes binding of the C-ABI in include/pse_amd.h (libpse_amd.so).

This is the only way Python reaches the engine: plain pointers and sizes.  torch is used solely as the owner of
device memory (tensor.data_ptr()).  There is no CPU fallback: if the library is missing this raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))


def asan_dir():
    """The CPU sanitizer build (python -m pse_amd.build --asan; tools/asan.py runs the CPU tests on it), or None.  PSE_ASAN_DIR
    alone is not enough: the directory must carry the marker that build writes, and the redirection is announced on stderr -- a
    stray variable must not silently swap the product for a library whose device calls all fail."""
    d = os.environ.get("PSE_ASAN_DIR")
    if not d:
        return None
    if not os.path.exists(os.path.join(d, ".pse_asan_build")):
        import sys
        print(f"pse_amd: PSE_ASAN_DIR={d} ignored: no sanitizer build there (python -m pse_amd.build --asan writes one)", file=sys.stderr)
        return None
    if not getattr(asan_dir, "_said", False):
        import sys
        print(f"pse_amd: using the CPU SANITIZER build in {d} (PSE_ASAN_DIR): device entry points are stubs", file=sys.stderr)
        asan_dir._said = True
    return d


_ASAN_DIR = asan_dir()
LIB_PATH = os.path.join(_ASAN_DIR or _HERE, "libpse_amd.so")

PSE_OK = 0


class pse_params(ctypes.Structure):
    _fields_ = [
        ("n_max", ctypes.c_uint),
        ("Lx", ctypes.c_double), ("Ly", ctypes.c_double), ("Lz", ctypes.c_double), ("xy", ctypes.c_double),
        ("xi", ctypes.c_double), ("error", ctypes.c_double), ("max_strain", ctypes.c_double),
        ("seed", ctypes.c_uint),
        ("Nx", ctypes.c_int), ("Ny", ctypes.c_int), ("Nz", ctypes.c_int),
        ("P", ctypes.c_int),
        ("rcut", ctypes.c_double),
        ("device", ctypes.c_int),
        ("n_slabs", ctypes.c_int), ("slab_rank", ctypes.c_int),
        ("local_rows", ctypes.c_int),
    ]


class pse_info(ctypes.Structure):
    _fields_ = [
        ("Nx", ctypes.c_int), ("Ny", ctypes.c_int), ("Nz", ctypes.c_int), ("P", ctypes.c_int),
        ("rcut", ctypes.c_double), ("xi", ctypes.c_double), ("eta", ctypes.c_double), ("gaussm", ctypes.c_double),
        ("lam", ctypes.c_double), ("self_mobility", ctypes.c_double),
        ("hx", ctypes.c_double), ("hy", ctypes.c_double), ("hz", ctypes.c_double),
        ("ncell_x", ctypes.c_int), ("ncell_y", ctypes.c_int), ("ncell_z", ctypes.c_int),
        ("lanczos_m", ctypes.c_int), ("lanczos_matvecs", ctypes.c_int),
        ("lanczos_stepnorm", ctypes.c_double),
        ("t_sort", ctypes.c_double), ("t_spread", ctypes.c_double), ("t_fft_fwd", ctypes.c_double),
        ("t_scale", ctypes.c_double), ("t_fft_inv", ctypes.c_double), ("t_gather", ctypes.c_double),
        ("t_real", ctypes.c_double), ("t_lanczos", ctypes.c_double), ("t_integrate", ctypes.c_double),
        ("t_comm", ctypes.c_double), ("t_total", ctypes.c_double),
        ("device_bytes", ctypes.c_ulonglong),
        ("t_matvec", ctypes.c_double),
        ("t_records", ctypes.c_double),
        ("lanczos_exchanges", ctypes.c_int),
        ("lanczos_status", ctypes.c_int),
        ("lanczos_open_calls", ctypes.c_ulonglong),
    ]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


class pse_host_xfer(ctypes.Structure):
    _fields_ = [("send", ctypes.POINTER(ctypes.c_double)), ("send_count", ctypes.c_size_t), ("send_to", ctypes.c_int),
                ("recv", ctypes.POINTER(ctypes.c_double)), ("recv_count", ctypes.c_size_t), ("recv_from", ctypes.c_int)]


PSE_DIAG_MAX = 48


class pse_team_diag(ctypes.Structure):
    _fields_ = [("n_exchanges", ctypes.c_int), ("kind", ctypes.c_int * PSE_DIAG_MAX), ("lane", ctypes.c_int * PSE_DIAG_MAX),
                ("device_us", ctypes.c_double * PSE_DIAG_MAX), ("host_us", ctypes.c_double * PSE_DIAG_MAX),
                ("bytes", ctypes.c_ulonglong * PSE_DIAG_MAX), ("main_lane_ms", ctypes.c_double), ("side_lane_ms", ctypes.c_double),
                ("critical_path_ms", ctypes.c_double)]


EXCHANGE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(pse_host_xfer))
ALLREDUCE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.c_size_t)


class pse_transport(ctypes.Structure):
    _fields_ = [("user", ctypes.c_void_p), ("exchange", EXCHANGE_FN), ("allreduce_sum", ALLREDUCE_FN)]


# every symbol include/pse_amd.h declares: name -> (restype, argtypes)
_vp, _i, _u, _d = ctypes.c_void_p, ctypes.c_int, ctypes.c_uint, ctypes.c_double
_ip = ctypes.POINTER(ctypes.c_int)
_dp = ctypes.POINTER(ctypes.c_double)
SYMBOLS = {
    "pse_create": (_i, [ctypes.POINTER(pse_params), ctypes.POINTER(_vp)]),
    "pse_destroy": (_i, [_vp]),
    "pse_set_box": (_i, [_vp, _d, _d, _d, _d]),
    "pse_set_stream": (_i, [_vp, _vp]),
    "pse_set_timing": (_i, [_vp, _i]),
    "pse_set_async": (_i, [_vp, _i]),
    "pse_set_timestep_offset": (_i, [_vp, _vp]),
    "pse_debug_last_gate": (_i, [_vp, _ip]),
    "pse_set_neighbor_skin": (_i, [_vp, _d]),
    "pse_neighbor_stats": (_i, [_vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_ulonglong), ctypes.POINTER(ctypes.c_ulonglong)]),
    "pse_get_info": (_i, [_vp, ctypes.POINTER(pse_info)]),
    "pse_last_error": (ctypes.c_char_p, []),
    "pse_mobility": (_i, [_vp, _vp, _vp, _vp, _vp, _u, _i]),
    "pse_brownian_velocity": (_i, [_vp, _vp, _vp, _vp, _vp, _u, _d, _d, _u, _ip]),
    "pse_brownian_velocity_part": (_i, [_vp, _vp, _vp, _vp, _vp, _u, _d, _d, _u, _i, _ip]),
    "pse_integrate": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _u, _d, _d]),
    "pse_step": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _u, _d, _d, _u, _d, _ip]),
    "pse_sqrt_mreal": (_i, [_vp, _vp, _vp, _vp, _vp, _u, _d, _ip]),
    "pse_random_psi": (_i, [_vp, _vp, _vp, _u, _u]),
    "pse_pair_repulsion": (_i, [_vp, _vp, _vp, _vp, _u, _d, _d, _i]),
    "pse_eval_realspace": (_i, [_vp, _dp, _i, _dp, _dp]),
    "pse_debug_copy_grid": (_i, [_vp, _i, _dp]),
    "pse_debug_spread": (_i, [_vp, _vp, _vp, _vp, _u]),
    "pse_debug_kvector": (_i, [_vp, _i, _ip, _dp]),
    "pse_debug_grid_placement": (_i, [_vp, _ip, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]),
    "pse_debug_vq_roundtrip": (_i, [_i, _dp, _dp]),
    "pse_debug_matvec_ms": (_i, [_vp, _i, ctypes.POINTER(ctypes.c_float)]),
    "pse_team_unique_id": (_i, [_vp]),
    "pse_team_create": (_i, [ctypes.POINTER(_vp), _i, _vp, ctypes.POINTER(_vp)]),
    "pse_team_create_transport": (_i, [_vp, ctypes.POINTER(pse_transport), ctypes.POINTER(_vp)]),
    "pse_team_destroy": (_i, [_vp]),
    "pse_team_debug_solo": (_i, [_vp, _i]),
    "pse_team_mobility": (_i, [_vp, ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_vp), _vp, _u, _i]),
    "pse_team_brownian_velocity": (_i, [_vp, ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_vp), _vp, _u, _d, _d, _u, _ip]),
    "pse_team_step": (_i, [_vp, ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_vp),
                           ctypes.POINTER(_vp), _vp, _u, _d, _d, _u, _d, _ip]),
    "pse_team_step_local": (_i, [_vp, ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_vp),
                                 ctypes.POINTER(_vp), ctypes.POINTER(_vp), ctypes.POINTER(_vp), _d, _d, _u, _d, _i, _ip]),
    "pse_local_layout": (_i, [_vp, _ip, _ip, _ip, _ip, _ip]),
    "pse_team_local_status": (_i, [_vp, _ip]),
    "pse_team_set_lanczos_extra": (_i, [_vp, _i]),
    "pse_set_lanczos_extra": (_i, [_vp, _i]),
    "pse_team_redistribute_local": (_i, [_vp] + [ctypes.POINTER(_vp)] * 7),
    "pse_team_set_diag": (_i, [_vp, _i]),
    "pse_team_get_diag": (_i, [_vp, ctypes.POINTER(pse_team_diag)]),
    "pse_host_lanczos_sqrt_e1": (_i, [_i, _dp, _dp, _dp]),
    "pse_host_select_params": (_i, [ctypes.POINTER(pse_params), ctypes.POINTER(pse_info)]),
}

_lib = None


class PSEError(RuntimeError):
    pass


def load():
    """Load libpse_amd.so (after torch, so both share one HIP runtime). Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PSEError(f"{LIB_PATH} not found: build it with `python -m pse_amd.build` "
                       "(there is no CPU fallback for the PSE hot path)")
    if not _ASAN_DIR:
        try:
            import torch  # noqa: F401  (loads libamdhip64/librocfft with the SONAMEs our library needs)
        except ImportError:
            pass
    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status):
    if status != PSE_OK:
        raise PSEError(f"pse_amd error {status}: {load().pse_last_error().decode()}")
