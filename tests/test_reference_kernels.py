"""The oracle's port AND the device held to what the reference's KERNELS compute (tests/golden/reference_kernels.json.gz: the
text of gpu_stokes_{SetGridk,Green,BrownianGridGenerate,Spread,Mreal,step_one}_kernel read from the reference tree and
executed thread by thread by tests/golden/make_kernel_fixture.py; the fixture holds inputs and results, no reference text).

What this pins that mathematics alone cannot ("missing" 3): the support centring and wrap of the spread
(PSEv1/Mobility.cu:212-219), the sheared node position and weight (:223-246), the FFT index folding of the wave vectors and the
scale factor (PSEv1/Helper.cu:300-327), the Green projector (PSEv1/Mobility.cu:283-295), the half-space / Nyquist / conjugate
rule of the k-space noise (PSEv1/Brownian.cu:210-220,255-288,317-335), the pair formula with its table (PSEv1/Mobility.cu:
661-677) and the Euler step (PSEv1/Stokes.cu:156-190).

Two things the fixture brought to light, both now stated in DESIGN.md:
  * the reference works on the FULL complex grid and keeps the real part, so wherever an index is a Nyquist index the operator
    is the mean of the node's and its partner's (the build applied this on the planes kz = 0, Nz/2 only; fixed in round 3);
  * its noise kernel runs BOTH members of a conjugate pair on the plane kz = Nz/2 and on the line (ky = Ny/2, kz = 0), each
    adding to both nodes (PSEv1/Brownian.cu:210-215 restricts the half space only for kz = 0): those modes get twice the
    amplitude (with independent draws: twice the variance) of the fluctuation-dissipation value.  A reference defect; the build
    does not reproduce it, the test accounts for it explicitly.
"""
import gzip
import json
import math
import os

import numpy as np
import pytest

from oracle import pse_port as oracle

HERE = os.path.dirname(os.path.abspath(__file__))
with gzip.open(os.path.join(HERE, "golden", "reference_kernels.json.gz")) as f:
    FIX = json.load(f)


def rel(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / np.abs(np.asarray(b)).max()


def params(c):
    grid, box = tuple(c["grid"]), tuple(c["box"])
    return grid, box, dict(grid=grid, xi=c["xi"], eta=c["eta"], P=c.get("P", 0), h=tuple(box[a] / grid[a] for a in range(3)))


# ------------------------------------------------------------------------------------------------ port vs reference kernels (CPU)
@pytest.mark.parametrize("c", FIX["setgridk"]["cases"], ids=lambda c: "x".join(map(str, c["grid"])))
def test_wave_vectors_and_scale_factor(c):
    grid, box, p = params(c)
    kx, ky, kz, k2, w, sinc = oracle.kvectors(box, p, full=True)
    ref = np.array(c["exact_pi"])
    assert np.abs(ref[..., 0] - kx).max() < 1e-15 and np.abs(ref[..., 1] - ky).max() < 1e-15 and np.abs(ref[..., 2] - kz).max() < 1e-15
    assert rel(w, ref[..., 3]) < 1e-14
    # "as written": the same with the reference's two pi literals
    aw = np.array(c["as_written"])
    assert np.abs(aw[..., :3] - ref[..., :3] * (3.1416926536 / math.pi)).max() < 1e-14
    k2w = (aw[..., :3] ** 2).sum(-1)
    q = k2w / (4 * c["xi"] ** 2)
    with np.errstate(divide="ignore", invalid="ignore"):
        w_aw = 6.0 * 3.1415926536 * (1 + q) * np.exp(-(1 - c["eta"]) * q) / k2w / np.prod(grid)
    w_aw[0, 0, 0] = 0.0
    assert rel(w_aw, aw[..., 3]) < 1e-14


@pytest.mark.parametrize("c", FIX["green"]["cases"], ids=lambda c: "x".join(map(str, c["grid"])))
def test_green_operator(c):
    grid, box, p = params(c)
    real = np.array(c["real_in"])
    u = np.fft.irfftn(oracle.wave_scale(np.fft.rfftn(real, axes=(1, 2, 3)), box, p), s=grid, axes=(1, 2, 3), norm="forward")
    assert rel(u, c["real_out"]) < 1e-13
    if any(n % 2 == 0 for n in grid):     # what the reference writes is not Hermitian on Nyquist lines: it relies on reading the real part
        assert c["max_imag_out"] > 1e-6
    else:
        assert c["max_imag_out"] < 1e-13


def doubled_modes(grid):
    """Half-spectrum nodes the reference's noise kernel writes twice (module docstring)."""
    Nx, Ny, Nz = grid
    i, j, k = np.meshgrid(np.arange(Nx), np.arange(Ny), np.arange(Nz // 2 + 1), indexing="ij")
    selfc = ((2 * i) % Nx == 0) & ((2 * j) % Ny == 0) & ((2 * k) % Nz == 0)
    m = np.ones(i.shape)
    if Nz % 2 == 0:
        m[(k == Nz // 2) & ~selfc] = 2.0
    if Ny % 2 == 0:
        m[(k == 0) & (j == Ny // 2) & ~selfc] = 2.0
    return m


@pytest.mark.parametrize("c", FIX["brownian_grid"]["cases"], ids=lambda c: "x".join(map(str, c["grid"])))
def test_kspace_noise_bookkeeping(c):
    grid, box, p = params(c)
    nk = oracle.noise_k(box, p, c["kT"], c["dt"], c["seed"], c["timestep"])
    m = doubled_modes(grid)
    u = np.fft.irfftn(nk * m, s=grid, axes=(1, 2, 3), norm="forward")
    assert rel(u, c["real_out"]) < 1e-13
    if (m > 1).any():                     # and without the reference's double count the fields differ: the defect is real
        assert rel(np.fft.irfftn(nk, s=grid, axes=(1, 2, 3), norm="forward"), c["real_out"]) > 1e-3
    # every node of the full grid but the origin is written (own thread or its partner's)
    assert c["nodes_written"] == np.prod(grid) - 1


def sparse_grid(c):
    g = np.zeros((3, int(np.prod(c["grid"]))))
    g[:, np.array(c["nodes"])] = np.array(c["values"]).T
    return g.reshape((3,) + tuple(c["grid"]))


@pytest.mark.parametrize("c", FIX["spread"]["cases"], ids=lambda c: f"P{c['P']}-" + "x".join(map(str, c["grid"])))
def test_spread_support_and_weights(c):
    grid, box, p = params(c)
    assert abs(oracle.select_params(box, c["xi"], c["error"], c["max_strain"], grid=grid, P=c["P"])["eta"] - c["eta"]) < 1e-15
    g = oracle.spread(np.array(c["pos"]), np.array(c["force"]), box, p)
    ref = sparse_grid(c)
    assert rel(g, ref) < 1e-13
    assert np.array_equal(np.abs(g).sum(0) > 0, np.abs(ref).sum(0) > 0)          # exactly the same nodes: centring and wrap


def test_pair_formula():
    c = FIX["mreal"]
    u = oracle.mobility_real(np.array(c["pos"]), np.array(c["force"]), tuple(c["box"]), c["xi"], c["rcut"])
    assert rel(u, c["vel_dr_1e-6"]) < 1e-11                                     # table spacing 1e-6: the interpolation error is gone
    assert 1e-10 < rel(u, c["vel_dr_1e-3"]) < 5e-7                              # as written (dr = 1e-3): its 4e-8 interpolation error
    assert abs(oracle.self_mobility(c["xi"]) - c["self"]) < 1e-15


def test_euler_step_and_wrap():
    c = FIX["step_one"]
    p2, im2 = oracle.integrate(np.array(c["pos"]), np.array(c["image"]), np.array(c["vel"]), tuple(c["box"]), c["dt"], c["shear_rate"])
    assert np.abs(p2 - np.array(c["pos_out"])).max() < 1e-13 and np.array_equal(im2, np.array(c["image_out"]))
    assert np.abs(np.array(c["force"]) / c["mass"] - np.array(c["accel_out"])).max() < 1e-15
    assert (np.array(c["image_out"]) != np.array(c["image"])).any()             # the inputs do cross faces
    assert all(w == 7.0 for w in c["pos_w_kept"])


# ------------------------------------------------------------------------------------------------ device vs reference kernels (GPU)
@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch


def _engine(c, n_max):
    import pse_amd
    return pse_amd.Engine(n_max, tuple(c["box"]), xi=c["xi"], error=c["error"], max_strain=c["max_strain"], grid=tuple(c["grid"]),
                          P=c.get("P", 0), rcut=2.5, seed=3)


@pytest.mark.gpu
@pytest.mark.parametrize("c", FIX["spread"]["cases"], ids=lambda c: f"P{c['P']}-" + "x".join(map(str, c["grid"])))
def test_device_spread_against_the_reference_kernel(torch_cuda, c):
    from conftest import to4
    pos, force = np.array(c["pos"]), np.array(c["force"])
    eng = _engine(c, len(pos))
    i = eng.info()
    assert abs(i["eta"] - c["eta"]) < 1e-14 and i["P"] == c["P"]
    g = eng.debug_spread(to4(pos), to4(force))
    ref = sparse_grid(c)
    assert rel(g, ref) < 1e-12
    assert np.array_equal(np.abs(g).sum(0) > 0, np.abs(ref).sum(0) > 0)


@pytest.mark.gpu
@pytest.mark.parametrize("c", FIX["setgridk_engine"]["cases"], ids=lambda c: "x".join(map(str, c["grid"])))
def test_device_wave_vectors_against_the_reference_kernel(torch_cuda, c):
    eng = _engine(c, 8)
    assert abs(eng.info()["eta"] - c["eta"]) < 1e-14
    out = eng.debug_kvector(np.array(c["nodes"]))
    ref = np.array(c["k"])
    assert np.abs(out[:, :3] - ref[:, :3]).max() < 1e-14
    kn = np.sqrt((ref[:, :3] ** 2).sum(1))
    with np.errstate(divide="ignore", invalid="ignore"):
        sinc = np.where(kn > 0, np.sin(kn) / kn, 0.0)
    assert rel(out[:, 3], ref[:, 3] * sinc * sinc) < 1e-13                      # B = w sinc^2 (PSEv1/Mobility.cu:290)
    assert rel(out[:, 4], np.sqrt(ref[:, 3]) * sinc) < 1e-13                    # sqrt(w) sinc (PSEv1/Brownian.cu:274-276)
