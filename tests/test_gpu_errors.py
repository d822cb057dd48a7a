"""Error behaviour of the C-ABI (include/pse_amd.h), called raw through ctypes as a C host would: every misuse returns a negative
pse_status with a message in pse_last_error(), nothing exits the process (the reference calls exit(): PSEv1/Stokes.cc:203-214,
PSEv1/Brownian.cu:543-560), and the handle keeps working afterwards."""
import ctypes

import numpy as np
import pytest

from conftest import make_suspension, to4

pytestmark = pytest.mark.gpu

INVALID = -1


def _params(L, n_max=64, **kw):
    from pse_amd._lib import pse_params
    p = pse_params()
    p.n_max, p.Lx, p.Ly, p.Lz, p.xy = n_max, L, L, L, 0.0
    p.xi, p.error, p.max_strain, p.seed = 0.5, 1e-3, 0.5, 1
    p.Nx = p.Ny = p.Nz = 0
    p.P, p.rcut, p.device, p.n_slabs, p.slab_rank = 0, 0.0, -1, 1, 0
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def test_misuse_is_reported_not_fatal():
    import torch
    from pse_amd import _lib
    assert torch.cuda.is_available()
    lib = _lib.load()
    msg = lambda: lib.pse_last_error().decode()          # noqa: E731
    h = ctypes.c_void_p()

    def create(p):
        out = ctypes.c_void_p()
        return lib.pse_create(ctypes.byref(p), ctypes.byref(out)), out

    # creation: every parameter combination the rule cannot serve
    for bad, word in ((_params(20.0, n_max=0), "n_max"),
                      (_params(20.0, xi=-0.5), "xi"),
                      (_params(20.0, error=0.0), "error"),
                      (_params(20.0, error=1.5), "error"),
                      (_params(-20.0), "box"),
                      (_params(8.0), "half the box"),                       # rcut = 5.26 > L / 2
                      (_params(40.0, Nx=16, Ny=16, Nz=16), "eta"),          # grid too coarse for xi: eta >= 1
                      (_params(20.0, Nx=256, Ny=16, Nz=16, P=4), "spreading Gaussian"),   # spacings 16 x apart: NaN from the weight recurrence otherwise
                      (_params(20.0, P=-2), "P"),
                      (_params(20.0, Nx=-8), "grid"),
                      (_params(20.0, rcut=-1.0), "rcut"),
                      (_params(20.0, n_slabs=2, slab_rank=2), "slab_rank"),
                      (_params(20.0, xy=0.8), "xy")):
        rc, out = create(bad)
        assert rc == INVALID and not out.value, (word, rc)
        assert msg(), word
    assert lib.pse_create(None, ctypes.byref(h)) == INVALID and "null" in msg()
    # a good handle
    n = 64
    pos, force, box = make_suspension(n, L=20.0)
    rc, h = create(_params(20.0, n_max=n))
    assert rc == 0, msg()
    dpos, dF, vel = to4(pos), to4(force), to4(np.zeros((n, 3)))
    P = lambda t: ctypes.c_void_p(t.data_ptr())            # noqa: E731
    m = ctypes.c_int(2)
    assert lib.pse_mobility(h, P(dpos), P(dF), P(vel), None, n, 3) == 0, msg()
    good = vel.cpu().numpy().copy()
    # evaluation entry points
    assert lib.pse_mobility(None, P(dpos), P(dF), P(vel), None, n, 3) == INVALID and "null handle" in msg()
    assert lib.pse_mobility(h, None, P(dF), P(vel), None, n, 3) == INVALID and "null array" in msg()
    assert lib.pse_mobility(h, P(dpos), P(dF), P(vel), None, 0, 3) == INVALID and "n_max" in msg()
    assert lib.pse_mobility(h, P(dpos), P(dF), P(vel), None, n + 1, 3) == INVALID and "n_max" in msg()
    assert lib.pse_mobility(h, P(dpos), P(dF), P(vel), None, n, 0) == INVALID and "parts" in msg()
    assert lib.pse_brownian_velocity(h, P(dpos), P(dF), P(vel), None, n, -1.0, 1e-3, 0, ctypes.byref(m)) == INVALID and "kT" in msg()
    assert lib.pse_brownian_velocity(h, P(dpos), P(dF), P(vel), None, n, 1.0, 0.0, 0, ctypes.byref(m)) == INVALID and "dt" in msg()
    assert lib.pse_step(h, P(dpos), P(vel), None, None, P(dF), None, n, 1.0, 1e-3, 0, 0.0, ctypes.byref(m)) == INVALID and "null array" in msg()
    assert lib.pse_set_box(h, 20.0, -1.0, 20.0, 0.0) == INVALID and "positive" in msg()
    assert lib.pse_set_box(h, 20.0, 20.0, 20.0, 0.75) == INVALID and "tilt" in msg()
    assert lib.pse_set_box(h, 200.0, 200.0, 200.0, 0.0) == INVALID                     # grew beyond what creation sized
    assert lib.pse_set_neighbor_skin(h, -0.1) == INVALID
    assert lib.pse_set_neighbor_skin(h, 5.0) == INVALID and "exceeds" in msg()
    assert lib.pse_pair_repulsion(h, P(dpos), P(dF), None, n, 1.0, 50.0, 0) == INVALID and "repulsion range" in msg()
    r = np.array([1e9]); f = np.zeros(1); g = np.zeros(1)
    dp = ctypes.POINTER(ctypes.c_double)
    assert lib.pse_eval_realspace(h, r.ctypes.data_as(dp), 1, f.ctypes.data_as(dp), g.ctypes.data_as(dp)) == INVALID and "table range" in msg()
    ijk = np.array([0, 0, 10 ** 6], dtype=np.int32); out = np.zeros(5)
    assert lib.pse_debug_kvector(h, 1, ijk.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), out.ctypes.data_as(dp)) == INVALID and "outside" in msg()
    info = _lib.pse_info()
    assert lib.pse_get_info(h, None) == INVALID and lib.pse_get_info(None, ctypes.byref(info)) == INVALID
    # a slab rank refuses to be driven alone
    rc, hs = create(_params(40.0, n_max=n, n_slabs=2, slab_rank=0, Nx=48, Ny=48, Nz=48))
    if rc == 0:
        assert lib.pse_mobility(hs, P(dpos), P(dF), P(vel), None, n, 3) == INVALID and "pse_team" in msg()
        assert lib.pse_destroy(hs) == 0
    # ... and after all of that the handle still gives the same answer
    vel.zero_()
    assert lib.pse_mobility(h, P(dpos), P(dF), P(vel), None, n, 3) == 0, msg()
    assert np.abs(vel.cpu().numpy() - good).max() < 1e-13 * np.abs(good).max()      # (the kept list changes the order of the pair sums)
    assert lib.pse_destroy(h) == 0
    assert lib.pse_destroy(None) in (0, INVALID)
