"""CPU tests of the oracle itself: pinned to the known-answer values of SURVEY.md 8(c) (the reference ships no
golden vectors), to an independent quadrature of the defining Fourier integral, and to the committed fixtures."""
import json
import math
import os

import numpy as np
import pytest

from conftest import make_suspension

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_kat1_hasimoto_single_sphere(oracle):
    for xi in (0.3, 0.5, 0.8):
        M = oracle.mobility_dense(np.zeros((1, 3)), (20, 20, 20, 0), xi)
        assert abs(M[0, 0] - 0.85865872480157) < 2e-13
        assert abs(M[1, 1] - M[0, 0]) < 1e-14 and abs(M[0, 1]) < 1e-15
    L = 20.0
    assert abs(M[0, 0] - (1 - 2.837297 / L + 4 * math.pi / 3 / L ** 3)) < 5e-8   # Hasimoto series


def test_kat2_3_4_pair_blocks(oracle):
    box = (20, 20, 20, 0)
    for xi in (0.4, 0.5):
        B = oracle.mobility_dense(np.array([[0, 0, 0], [3, 1, 0.5]], float), box, xi)[3:6, 0:3]
        ref = np.array([[0.279076722785218, 0.053435549043855, 0.026722650338542],
                        [0.053435549043855, 0.128141730481001, 0.008928425757363],
                        [0.026722650338542, 0.008928425757363, 0.113989288350582]])
        assert np.abs(B - ref).max() < 5e-15
        B = oracle.mobility_dense(np.array([[0, 0, 0], [1.2, 0.3, 0]], float), box, xi)[3:6, 0:3]
        assert np.abs(np.diag(B) - [0.6207687795007937, 0.5177902235372976, 0.5109249998759693]).max() < 5e-15
        assert abs(B[0, 1] - 0.02707348137451859) < 5e-15
        B = oracle.mobility_dense(np.array([[0, 0, 0], [2, 0, 0]], float), box, xi)[3:6, 0:3]
        assert abs(B[0, 0] - 0.4860111219717241) < 5e-15 and abs(B[1, 1] - 0.2965766796953576) < 5e-15


def test_kat5_self(oracle):
    assert abs(oracle.self_mobility(0.5) - 0.3356171244690146) < 1e-16


def test_kat6_parameter_rule(oracle):
    p = oracle.select_params((64, 64, 64, 0), 0.5, 1e-3, 0.5)
    assert abs(p["lambda"] - 1.6403882) < 1e-7 and abs(p["gaussm"] - 4.22) < 1e-12 and p["P"] == 6
    assert abs(p["rcut"] - 5.2565) < 1e-4 and p["grid"] == (64, 64, 64) and abs(p["eta"] - 0.5054) < 1e-4
    assert p["ewald_n"] == 5255
    p0 = oracle.select_params((64, 64, 64, 0), 0.5, 1e-3, 0.0)
    assert abs(p0["gaussm"] - 3.30) < 1e-12 and p0["P"] == 4


def test_closed_form_equals_fourier_quadrature(oracle):
    for xi in (0.27, 0.5, 0.8):
        for r in (0.26, 0.7, 1.2, 1.999, 2.0, 2.3, 3.3, 5.0, 7.1, 9.0):
            f1, g1 = oracle.fg_wave(r, xi)
            f2, g2 = oracle.fg_wave(r, xi, quad=True)
            assert abs(f1 - f2) < 3e-15 and abs(g1 - g2) < 3e-15, (xi, r, f1 - f2, g1 - g2)


def test_self_is_r_to_zero_limit(oracle):
    f, g = oracle.fg_real(np.array([1e-6]), 0.5)
    assert abs(f[0] - oracle.self_mobility(0.5)) < 1e-6 and abs(g[0] - oracle.self_mobility(0.5)) < 1e-6


def test_xi_independence_and_spd(oracle):
    pos, _, box = make_suspension(30, L=14.0, xy=0.3)
    M1 = oracle.mobility_dense(pos, box, 0.45)
    M2 = oracle.mobility_dense(pos, box, 0.7)
    assert np.abs(M1 - M2).max() < 1e-13
    assert np.abs(M1 - M1.T).max() < 1e-14
    assert np.linalg.eigvalsh(M1).min() > 0
    # the "positive split": both halves SPD on their own
    for parts in (1, 2):
        Mp = oracle.mobility_dense(pos, box, 0.5, parts=parts)
        assert np.linalg.eigvalsh(0.5 * (Mp + Mp.T)).min() > -1e-13


def test_translation_and_image_invariance(oracle):
    pos, force, box = make_suspension(40, L=15.0, xy=0.2)
    u = oracle.mobility_direct(pos, force, box, 0.5)
    shift = pos + np.array([3.3, -1.1, 0.7])
    assert np.abs(oracle.mobility_direct(shift, force, box, 0.5) - u).max() < 1e-13
    img = pos.copy(); img[::3] += np.array([box[3] * box[1], box[1], 0.0]); img[1::3] -= np.array([box[0], 0, box[2]])
    assert np.abs(oracle.mobility_direct(img, force, box, 0.5) - u).max() < 1e-13


def test_zero_mode_is_dropped(oracle):
    """k = 0 is excluded from the wave sum (Helper.cu:321-323, Mobility.cu:287): the velocity grid has zero mean."""
    pos, force, box = make_suspension(50, L=15.0)
    p = oracle.select_params(box, 0.5, 1e-4, 0.0)
    F = force + np.array([1.0, -2.0, 0.5])                     # net force != 0
    uh = oracle.wave_scale(np.fft.rfftn(oracle.spread(pos, F, box, p), axes=(1, 2, 3)), box, p)
    ug = np.fft.irfftn(uh, s=p["grid"], axes=(1, 2, 3), norm="forward")
    assert np.abs(ug.mean(axis=(1, 2, 3))).max() < 1e-15 * np.abs(ug).max()
    assert np.all(uh[:, 0, 0, 0] == 0)


@pytest.mark.parametrize("err,bound", [(1e-3, 5e-3), (1e-6, 5e-6)])
def test_port_converges_to_direct_sum(oracle, err, bound):
    pos, force, box = make_suspension(300, L=30.0, xy=0.3)
    ref = oracle.mobility_direct(pos, force, box, 0.5)
    p = oracle.select_params(box, 0.5, err, 0.5)
    u = oracle.mobility(pos, force, box, p)
    assert np.linalg.norm(u - ref) / np.linalg.norm(ref) < bound


def test_lanczos_port_matches_dense_sqrt(oracle):
    import scipy.linalg as sl
    n = 30
    pos, _, box = make_suspension(n, L=14.0)
    rcut = 5.2565
    eye = np.eye(3 * n)
    M = np.stack([oracle.mobility_real(pos, eye[c].reshape(n, 3), box, 0.5, rcut).ravel() for c in range(3 * n)], 1)
    psi = oracle.psi_particles(n, 11, 4)
    u, m = oracle.lanczos_sqrt(lambda v: oracle.mobility_real(pos, np.ascontiguousarray(v), box, 0.5, rcut), psi, 2, 1e-9)
    ref = sl.sqrtm(M).real @ psi.ravel()
    assert np.linalg.norm(u.ravel() - ref) / np.linalg.norm(ref) < 1e-8
    assert abs(np.dot(u.ravel(), u.ravel()) - psi.ravel() @ M @ psi.ravel()) < 1e-7 * (psi.ravel() @ M @ psi.ravel())


def test_rounded_pair_coefficients_keep_one_symmetric_operator(oracle):
    """The build's Lanczos mat-vecs read pair coefficients rounded to single-precision accuracy (pse_kernels.hip pair_coef; restated in
    oracle/pse_oracle.c pair_term).  Both directions of a pair round alike, so the operator stays exactly symmetric; it is positive
    definite, within 3e-7 of the double-precision near-field matrix, and its Lanczos square root agrees with the dense one."""
    import scipy.linalg as sl
    n = 40
    pos, _, box = make_suspension(n, L=15.0, xy=0.25)
    rcut = 5.2565
    eye = np.eye(3 * n)
    M = np.stack([oracle.mobility_real(pos, eye[c].reshape(n, 3), box, 0.5, rcut).ravel() for c in range(3 * n)], 1)
    Mf = np.stack([oracle.mobility_real(pos, eye[c].reshape(n, 3), box, 0.5, rcut, rounded=True).ravel() for c in range(3 * n)], 1)
    assert np.array_equal(Mf, Mf.T)
    assert 1e-9 < np.abs(Mf - M).max() < 3e-7 * np.abs(M).max()
    assert np.linalg.eigvalsh(Mf).min() > 0.0
    psi = oracle.psi_particles(n, 3, 8)
    u, m = oracle.lanczos_sqrt(lambda v: oracle.mobility_real(pos, np.ascontiguousarray(v), box, 0.5, rcut, rounded=True), psi, 2, 1e-9)
    ref = sl.sqrtm(Mf).real @ psi.ravel()
    assert np.linalg.norm(u.ravel() - ref) / np.linalg.norm(ref) < 1e-8


def test_kspace_noise_is_hermitian_and_divergence_free(oracle):
    box = (12.0, 12.0, 12.0, 0.2)
    p = oracle.select_params(box, 0.5, 1e-3, 0.5, grid=(12, 10, 8))
    nk = oracle.noise_k(box, p, 1.0, 1e-2, 5, 9)
    full = np.fft.irfftn(nk, s=p["grid"], axes=(1, 2, 3), norm="forward")
    back = np.fft.rfftn(full, axes=(1, 2, 3)) / np.prod(p["grid"])
    # after a round trip through the real field, interior modes are unchanged (planes get symmetrised)
    assert np.abs(back[:, :, :, 1:3] - nk[:, :, :, 1:3]).max() < 1e-9 * np.abs(nk).max()
    # divergence-free: k . u_k = 0 -- on every mode whose wave vector is unambiguous; at a Nyquist index the reference's real
    # part averages the projectors of k and of its folded partner (tests/test_reference_kernels.py), which is not a projector
    kx, ky, kz, k2, w, sinc = oracle.kvectors(box, p)
    div = kx * nk[0] + ky * nk[1] + kz * nk[2]
    Nx, Ny, Nz = p["grid"]
    div[Nx // 2, :, :] = 0.0; div[:, Ny // 2, :] = 0.0; div[:, :, Nz // 2] = 0.0
    assert np.abs(div).max() < 1e-9 * np.abs(nk).max()


def test_philox_known_answer(oracle):
    # Philox4x32-10 known-answer vectors (Random123 kat_vectors): counter/key all zero, and the pi-digits vector
    r = oracle.philox4x32(0, 0, 0, 0, 0, 0)
    assert [int(x) for x in r] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    r = oracle.philox4x32(0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff)
    assert [int(x) for x in r] == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    r = oracle.philox4x32(0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344, 0xa4093822, 0x299f31d0)
    assert [int(x) for x in r] == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_shear_functions_closed_forms(oracle):
    s = oracle.SinShear(1.0, 1.0, 0, 1e-3)                       # examples/run.py:45
    assert abs(s.shear_rate(0) - 1.0) < 1e-15 and abs(s.strain(250) - 1 / (2 * math.pi)) < 1e-15
    assert abs(s.shear_rate(250)) < 1e-15
    st = oracle.SteadyShear(0.3, 10, 1e-2)
    assert st.shear_rate(5) == 0.3 and abs(st.strain(110) - 0.3) < 1e-15
    c = oracle.ChirpShear(0.1, 1.0, 10.0, 5.0, 0, 1e-3)
    assert abs(c.strain(0)) < 1e-15 and abs(c.shear_rate(0) - 0.1) < 1e-15
    # d(strain)/dt == shear rate
    for f in (s, c):
        t = 1234
        num = (f.strain(t + 1) - f.strain(t - 1)) / (2 * 1e-3)
        assert abs(num - f.shear_rate(t)) < 2e-3 * max(1.0, abs(f.shear_rate(t)))
    w = oracle.TukeyWindow(1.0, 0.5, 0, 1e-3)
    assert w.strain(0) == 0.0 and w.strain(500) == 1.0 and w.strain(1000) == 0.0 and abs(w.strain(125) - 0.5) < 1e-12
    ww = oracle.Windowed(s, w)
    assert abs(ww.strain(125) - s.strain(125) * 0.5) < 1e-15
    # wrapped strain for the box tilt (VariantShearFunction.h:46-48)
    assert abs(oracle.variant_value(oracle.SteadyShear(1.0, 0, 1e-2), 70, 1000, -0.5, 0.5) - (-0.3)) < 1e-12
    assert oracle.variant_value(oracle.SteadyShear(1.0, 5, 1e-2), 2, 1000, -0.5, 0.5) == 0.0


def test_golden_fixture(oracle):
    """Committed vectors generated by tests/golden/make_golden.py from the oracle at the time it was pinned."""
    g = json.load(open(os.path.join(GOLD, "pse_oracle_golden.json")))
    pos = np.array(g["pos"]); force = np.array(g["force"]); box = tuple(g["box"])
    u = oracle.mobility_direct(pos, force, box, g["xi"])
    assert np.abs(u - np.array(g["u_direct"])).max() < 1e-13
    p = oracle.select_params(box, g["xi"], g["error"], 0.5)
    ub, m = oracle.brownian_velocity(pos, force, box, p, g["kT"], g["dt"], g["seed"], g["timestep"], pair_rounded=False)
    assert m == g["lanczos_m"]
    assert np.abs(ub - np.array(g["u_brownian_port"])).max() < 1e-9 * np.abs(ub).max()
    # the same step with the rounded pair coefficients of the build's Lanczos mat-vecs: same m, within 1e-7
    uf, mf = oracle.brownian_velocity(pos, force, box, p, g["kT"], g["dt"], g["seed"], g["timestep"])
    assert mf == m and 0.0 < np.abs(uf - ub).max() < 1e-7 * np.abs(ub).max()
    assert [int(x) for x in oracle.philox4x32(1, 2, 3, 4, 5, 6)] == g["philox_1_2_3_4_5_6"]
    assert oracle.hash_seed(1) == g["hash_seed_1"]


def test_min_image_property(oracle):
    """the triclinic minimum image (wrap y first, shifting x by xy*Ly, then x) recovers any separation
    shorter than the near-field cutoff from all of its periodic images, for tilts up to |xy| = 0.5."""
    rng = np.random.default_rng(8)
    for xy in (0.0, 0.2, -0.35, 0.5, -0.5):
        box = (20.0, 16.0, 24.0, xy)
        a1, a2, a3 = np.array([20.0, 0, 0]), np.array([xy * 16.0, 16.0, 0]), np.array([0, 0, 24.0])
        d = rng.normal(size=(4000, 3)); d *= (rng.uniform(0, 6.0, size=(4000, 1)) / np.linalg.norm(d, axis=1, keepdims=True))
        k = rng.integers(-3, 4, size=(4000, 3))
        shifted = d + k[:, :1] * a1 + k[:, 1:2] * a2 + k[:, 2:] * a3
        back = oracle.min_image(shifted, box)
        assert np.abs(back - d).max() < 1e-12
        if xy == 0.0:      # orthogonal cell: the wrapped vector is never longer (not true of a sequential wrap in a tilted cell)
            far = rng.uniform(-60, 60, size=(4000, 3))
            assert np.all(np.linalg.norm(oracle.min_image(far, box), axis=1) <= np.linalg.norm(far, axis=1) + 1e-12)
