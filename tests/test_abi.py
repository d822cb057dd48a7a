"""CPU tests of the C-ABI boundary: the library loads, exports exactly what include/pse_amd.h declares, and its
host-only entry points (parameter rule, Lanczos tridiagonal square root) agree with the oracle. No compute calls."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from pse_amd import build, _lib
    build.build_lib()
    return _lib.load()


def test_every_declared_symbol_is_exported(lib):
    from pse_amd import _lib
    header = open(os.path.join(ROOT, "include", "pse_amd.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(pse_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for name in declared:
        assert getattr(lib, name) is not None


def test_struct_layout_matches_header():
    import ctypes
    from pse_amd._lib import pse_params, pse_info
    # 1 uint + pad, 7 doubles, uint + 3 int, int + pad, double, 3 int + pad
    assert ctypes.sizeof(pse_params) == 112
    assert pse_params.rcut.offset == 88 and pse_params.device.offset == 96
    assert ctypes.sizeof(pse_info) % 8 == 0 and pse_info.t_matvec.offset == ctypes.sizeof(pse_info) - 32
    assert pse_info.t_records.offset == pse_info.t_matvec.offset + 8 and pse_info.lanczos_exchanges.offset == pse_info.t_records.offset + 8


def test_host_parameter_rule_matches_oracle(lib, oracle):
    import pse_amd
    for box, xi, err, ms in (((64, 64, 64, 0), 0.5, 1e-3, 0.5), ((64, 64, 64, 0), 0.5, 1e-3, 0.0),
                             ((43.756, 43.756, 43.756, 0), 0.5, 1e-3, 0.5), ((30, 45, 60, 0.2), 0.7, 1e-6, 0.5),
                             ((140.02,) * 3 + (0,), 0.273, 1e-3, 0.5)):
        i = pse_amd.host_select_params(box, xi, err, ms)
        p = oracle.select_params(box, xi, err, ms)
        assert (i["Nx"], i["Ny"], i["Nz"]) == p["grid"] and i["P"] == p["P"]
        for a, b in (("rcut", "rcut"), ("eta", "eta"), ("gaussm", "gaussm"), ("lam", "lambda"), ("self_mobility", "self")):
            assert abs(i[a] - p[b]) < 1e-14 * max(1, abs(p[b])), (a, i[a], p[b])
    i = pse_amd.host_select_params((140.02,) * 3 + (0,), 0.273, 1e-3, 0.5, grid=(64, 64, 64))
    assert i["Nx"] == 64 and abs(i["eta"] - 0.722) < 1e-3          # cfg2 as an explicit override


def test_invalid_parameters_are_reported_not_fatal(lib):
    import pse_amd
    with pytest.raises(pse_amd.PSEError, match="eta"):
        pse_amd.host_select_params((140.0, 140.0, 140.0, 0), 0.65, 1e-3, 0.5, grid=(64, 64, 64))   # eta > 1
    with pytest.raises(pse_amd.PSEError):
        pse_amd.host_select_params((10, 10, 10, 0), -1.0)
    with pytest.raises(pse_amd.PSEError, match="4096"):
        pse_amd.host_select_params((1e5, 10, 10, 0), 0.5)                                               # Stokes.cc:201-214


def test_tridiagonal_sqrt_matches_numpy(lib):
    import pse_amd
    rng = np.random.default_rng(0)
    for m in (1, 2, 3, 7, 30, 100):
        a = rng.uniform(0.5, 1.5, m)
        b = np.concatenate([[0.0], rng.uniform(0.01, 0.3, m)])
        T = np.diag(a) + np.diag(b[1:m], 1) + np.diag(b[1:m], -1)
        lam, W = np.linalg.eigh(T)
        assert lam.min() > 0
        ref = W @ (np.sqrt(lam) * W[0])
        assert np.abs(pse_amd.host_lanczos_sqrt_e1(a, b) - ref).max() < 1e-14


def test_product_does_not_import_the_oracle():
    """The oracle is test infrastructure: nothing under pse_amd/ may import, link or execute it."""
    for root, _, files in os.walk(os.path.join(ROOT, "pse_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".cc")):
                text = open(os.path.join(root, f)).read()
                assert "oracle" not in text.replace("pse_oracle_golden", ""), os.path.join(root, f)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from pse_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.PSEError, match="no CPU fallback"):
        _lib.load()


def test_sanitizer_build_is_in_use():
    """Under tools/asan.py (PSE_ASAN_DIR set) the libraries this suite exercises are the -fsanitize=address,undefined builds: the
    product's host side over the device stub, and the checker.  Skipped in the ordinary run."""
    import ctypes
    d = os.environ.get("PSE_ASAN_DIR")
    if not d:
        pytest.skip("not the sanitizer run (python tools/asan.py)")
    from pse_amd import _lib, _PSEv1
    from oracle import pse_port
    assert os.path.dirname(os.path.realpath(_lib.LIB_PATH)) == os.path.realpath(d)
    assert os.path.dirname(os.path.realpath(_PSEv1.__file__)) == os.path.realpath(d)
    assert os.path.realpath(pse_port.lib()._name).startswith(os.path.realpath(d))
    lib = _lib.load()
    with open("/proc/self/maps") as f:
        maps = f.read()
    assert "libasan" in maps and "libubsan" in maps
    # a device entry point of the stub says what it is
    assert lib.pse_set_timing(None, 1) != 0 and b"sanitizer build" in lib.pse_last_error()
    assert isinstance(lib, ctypes.CDLL)


def test_sanitizer_stub_defines_every_declared_entry_point():
    """tools/asan.py runs this suite on a host-only build whose device entry points are stubs (pse_amd/csrc/asan_stub.cpp): an entry point
    added to include/pse_amd.h and not to the stub breaks that build the first time someone runs it (round 6 found eight missing)."""
    import glob
    import re
    header = open(os.path.join(ROOT, "include", "pse_amd.h")).read()
    declared = set(re.findall(r"^(?:int|const char \*|void)\s*\*?(pse_[a-z0-9_]+)\s*\(", header, flags=re.M))
    assert len(declared) > 40
    host = "".join(open(f).read() for f in glob.glob(os.path.join(ROOT, "pse_amd", "csrc", "*.cpp")))
    missing = sorted(n for n in declared if not re.search(r"\b" + n + r"\s*\(", host))
    assert not missing, missing
