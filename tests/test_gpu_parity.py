"""GPU parity tests: the HIP path (through the C-ABI) against the CPU oracle on the same seeded inputs.
Tolerances: real-space <= 1e-12 relative; wave-space against the NumPy restatement of the same
algorithm <= 1e-10; total M.F against the direct Ewald sum <= 5 x error."""
import math

import numpy as np
import pytest

from conftest import make_suspension, to4

pytestmark = pytest.mark.gpu


def rel(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch


def test_realspace_functions_match_closed_form(torch_cuda, oracle):
    import pse_amd
    for xi in (0.5, 0.273, 0.8):
        rcut = math.sqrt(-math.log(1e-3)) / xi
        eng = pse_amd.Engine(8, (4 * rcut, 4 * rcut, 4 * rcut, 0.0), xi=xi, error=1e-3, grid=(32, 32, 32), P=4)
        r = np.concatenate([np.random.default_rng(1).uniform(1e-3, rcut, 2000), [2.0, 1.999999, 2.000001, 0.125, 1e-4]])
        f, g = eng.eval_realspace(r)
        fo, go = oracle.fg_real(r, xi)
        assert np.abs(f - fo).max() < 2e-13, (xi, np.abs(f - fo).max())
        assert np.abs(g - go).max() < 2e-13, (xi, np.abs(g - go).max())
        eng.close()


@pytest.mark.parametrize("xy", [0.0, 0.35])
def test_mreal_matches_oracle(torch_cuda, oracle, xy):
    import pse_amd
    n = 2000
    pos, force, box = make_suspension(n, phi=0.1, xy=xy)
    eng = pse_amd.Engine(n, box, xi=0.5, error=1e-3)
    info = eng.info()
    u = eng.mobility(to4(pos), to4(force), parts=1).cpu().numpy()[:, :3]
    ref = oracle.mobility_real(pos, force, box, 0.5, info["rcut"])
    assert rel(u, ref) < 1e-12, rel(u, ref)


@pytest.mark.parametrize("xy,err", [(0.0, 1e-3), (0.2, 1e-3), (0.0, 1e-6), (-0.45, 1e-4)])
def test_wave_matches_port(torch_cuda, oracle, xy, err):
    import pse_amd
    n = 1500
    pos, force, box = make_suspension(n, phi=0.1, xy=xy)
    eng = pse_amd.Engine(n, box, xi=0.5, error=err)
    info = eng.info()
    p = oracle.select_params(box, 0.5, err, 0.5)
    assert (info["Nx"], info["Ny"], info["Nz"]) == p["grid"] and info["P"] == p["P"]
    assert abs(info["eta"] - p["eta"]) < 1e-14
    u = eng.mobility(to4(pos), to4(force), parts=2).cpu().numpy()[:, :3]
    ref = oracle.mobility_wave(pos, force, box, p)
    assert rel(u, ref) < 1e-10, rel(u, ref)
    # the spread itself, node by node (PSEv1/Mobility.cu:114-252), not only through the gather
    g = eng.debug_spread(to4(pos), to4(force))
    gref = np.asarray(oracle.spread(pos, force, box, p))
    assert g.shape == gref.shape and np.abs(g - gref).max() < 1e-12 * np.abs(gref).max()


@pytest.mark.parametrize("err", [1e-3, 1e-6, 1e-9])
def test_total_mobility_against_direct_ewald(torch_cuda, oracle, err):
    import pse_amd
    n = 1000
    pos, force, box = make_suspension(n, phi=0.05)          # BASELINE config 1
    eng = pse_amd.Engine(n, box, xi=0.5, error=err, max_strain=0.5)
    u = eng.mobility(to4(pos), to4(force)).cpu().numpy()[:, :3]
    ref = oracle.mobility_direct(pos, force, box, 0.5)
    e = rel(u, ref)
    # the NumPy restatement of the reference algorithm (same rcut/grid/P/eta; held to the reference's kernel text and parameter rule
    # by tests/test_reference_kernels.py and tests/test_reference_pin.py) bounds what the method itself can reach: at
    # error = 1e-9 the reference's parameter rule delivers ~1.5e-8, not 1e-9 (SURVEY.md 8c assumed 1e-8 before any restatement
    # existed); the device is held to the restatement at 1e-10 and to the direct Ewald sum at what the restatement reaches
    p = oracle.select_params(box, 0.5, err, 0.5)
    e_port = rel(oracle.mobility(pos, force, box, p), ref)
    assert rel(u, oracle.mobility(pos, force, box, p)) < 1e-10
    assert e < max(5 * err, 1.05 * e_port), (err, e, e_port)
    assert e < 3e-8 or err > 1e-9


def test_xi_independence(torch_cuda, oracle):
    import pse_amd
    n = 800
    pos, force, box = make_suspension(n, phi=0.1)
    us, ports = [], []
    for xi in (0.5, 0.75):
        eng = pse_amd.Engine(n, box, xi=xi, error=1e-7)
        us.append(eng.mobility(to4(pos), to4(force)).cpu().numpy()[:, :3])
        eng.close()
        # the same evaluation by the restatement of the reference's algorithm (held to the reference's kernel text and parameter
        # rule by tests/test_reference_kernels.py and tests/test_reference_pin.py): what the METHOD gives at this xi
        ports.append(oracle.mobility(pos, force, box, oracle.select_params(box, xi, 1e-7, 0.5)))
        assert rel(us[-1], ports[-1]) < 1e-10
    # the claim of examples/run.py:50 ("xi ... will not affect results, only speed") holds to the method's accuracy: how far the
    # reference's own parameter rule lets two xi drift apart at error = 1e-7 is asked of the restatement, not assumed
    drift = rel(ports[0], ports[1])
    exact = oracle.mobility_direct(pos, force, box, 0.5)
    assert rel(us[0], us[1]) < 1.05 * drift + 1e-10
    assert drift < 1e-5 and max(rel(ports[0], exact), rel(ports[1], exact)) < 1e-5   # and both sit that close to the direct Ewald sum


def test_group_members_and_w_preserved(torch_cuda, oracle):
    import torch
    import pse_amd
    n_total, n = 1200, 700
    pos, force, box = make_suspension(n_total, phi=0.1)
    members = np.sort(np.random.default_rng(3).choice(n_total, n, replace=False)).astype(np.int32)
    eng = pse_amd.Engine(n_total, box, xi=0.5, error=1e-3)
    vel = to4(np.zeros((n_total, 3)), w=7.5)
    vel[:, :3] = -1.0
    g = torch.tensor(members, dtype=torch.int32, device="cuda")
    eng.mobility(to4(pos), to4(force), vel=vel, group=g)
    v = vel.cpu().numpy()
    sub = eng.mobility(to4(pos[members]), to4(force[members])).cpu().numpy()
    assert np.all(v[:, 3] == 7.5)
    others = np.setdiff1d(np.arange(n_total), members)
    assert np.all(v[others, :3] == -1.0)
    assert rel(v[members, :3], sub[:, :3]) < 1e-12


def test_random_psi_matches_oracle_stream(torch_cuda, oracle):
    import pse_amd
    n = 5000
    eng = pse_amd.Engine(n, (50, 50, 50, 0), xi=0.5, error=1e-3, seed=987654321)
    for ts in (0, 17, 2 ** 31 + 5):
        psi = eng.random_psi(n, ts).cpu().numpy()[:, :3]
        ref = oracle.psi_particles(n, 987654321, ts)
        assert np.abs(psi - ref).max() < 1e-15
    assert abs(psi.var() - 1.0) < 0.05


def test_lanczos_sqrt_matches_dense(torch_cuda, oracle):
    import scipy.linalg as sl
    import pse_amd
    n = 60
    pos, _, box = make_suspension(n, L=16.0)
    eng = pse_amd.Engine(n, box, xi=0.5, error=1e-3)
    rcut = eng.info()["rcut"]
    eye = np.eye(3 * n)
    # (rounded=True: the operator of the Lanczos iteration carries the rounded pair coefficients of the per-step pair list)
    M = np.stack([oracle.mobility_real(pos, eye[c].reshape(n, 3), box, 0.5, rcut, rounded=True).ravel() for c in range(3 * n)], 1)
    psi = np.random.default_rng(5).normal(size=(n, 3))
    ref = (sl.sqrtm(M).real @ psi.ravel()).reshape(n, 3)
    for tol, bound in ((1e-3, 5e-3), (1e-8, 1e-7)):
        out, m = eng.sqrt_mreal(to4(pos), to4(psi), tol=tol)
        e = rel(out.cpu().numpy()[:, :3], ref)
        assert e < bound, (tol, m, e)
        up, mp = oracle.lanczos_sqrt(lambda v: oracle.mobility_real(pos, np.ascontiguousarray(v), box, 0.5, rcut, rounded=True), psi, 2, tol)
        assert m == mp, (m, mp)
        assert rel(out.cpu().numpy()[:, :3], up) < 1e-9


def test_pair_list_overflow_rows(torch_cuda, oracle):
    """A dense cluster: its rows do not fit the per-step pair list (capacity follows the mean density) and are recomputed
    from the cells inside every Lanczos mat-vec; the result must not change."""
    import pse_amd
    rng = np.random.default_rng(17)
    n, L = 400, 30.0
    box = (L, L, L, 0.0)
    ball = rng.normal(size=(300, 3)); ball *= (4.0 * rng.uniform(size=(300, 1)) ** (1 / 3)) / np.linalg.norm(ball, axis=1, keepdims=True)
    pos = np.concatenate([ball + np.array([L / 2 - 1.0, 0.0, -L / 2 + 0.5]), rng.uniform(-L / 2, L / 2, size=(100, 3))])
    pos = oracle.wrap(pos, np.zeros((n, 3), dtype=np.int64), box)[0]
    eng = pse_amd.Engine(n, box, xi=0.5, error=1e-3)
    rcut = eng.info()["rcut"]
    psi = rng.normal(size=(n, 3))
    matvec = lambda v: oracle.mobility_real(pos, np.ascontiguousarray(v), box, 0.5, rcut, rounded=True)
    counts = (np.linalg.norm(pos[:, None] - pos[None], axis=2) < rcut).sum(1) - 1
    assert counts.max() > 100        # far beyond the list capacity of ~30 slots
    out, m = eng.sqrt_mreal(to4(pos), to4(psi), tol=1e-3)
    up, mp = oracle.lanczos_sqrt(matvec, psi, 2, 1e-3)
    assert m == mp, (m, mp)
    assert rel(out.cpu().numpy()[:, :3], up) < 1e-9
    # a long run (m ~ 60: past the point where Lanczos vectors stay orthogonal, so m itself is rounding-dependent)
    out, m = eng.sqrt_mreal(to4(pos), to4(psi), tol=1e-7)
    up, mp = oracle.lanczos_sqrt(matvec, psi, 2, 1e-7)
    assert abs(m - mp) <= 10, (m, mp)
    assert rel(out.cpu().numpy()[:, :3], up) < 1e-5


@pytest.mark.parametrize("xy", [0.0, 0.3])
def test_brownian_velocity_matches_port(torch_cuda, oracle, xy):
    import pse_amd
    n = 1000
    pos, force, box = make_suspension(n, phi=0.1, xy=xy)
    seed, ts, kT, dt = 424242, 11, 1.0, 1e-3
    eng = pse_amd.Engine(n, box, xi=0.5, error=1e-3, seed=seed)
    p = oracle.select_params(box, 0.5, 1e-3, 0.5)
    vel, m = eng.brownian_velocity(to4(pos), to4(force), kT, dt, ts)
    ref, mref = oracle.brownian_velocity(pos, force, box, p, kT, dt, seed, ts)
    assert m == mref, (m, mref)
    assert rel(vel.cpu().numpy()[:, :3], ref) < 1e-9, rel(vel.cpu().numpy()[:, :3], ref)


def test_step_integrates_and_wraps(torch_cuda, oracle):
    import torch
    import pse_amd
    n = 1000
    pos, force, box = make_suspension(n, phi=0.1, xy=0.25)
    seed, ts, kT, dt, rate = 99, 3, 1.0, 2e-2, 0.7
    eng = pse_amd.Engine(n, box, xi=0.5, error=1e-3, seed=seed)
    p = oracle.select_params(box, 0.5, 1e-3, 0.5)
    dpos = to4(pos, w=1.0); dvel = to4(np.zeros((n, 3)), w=2.0); dF = to4(force, w=0.5)
    accel = torch.zeros((n, 3), dtype=torch.float64, device="cuda")
    image = torch.zeros((n, 3), dtype=torch.int32, device="cuda")
    eng.step(dpos, dvel, accel, image, dF, kT, dt, ts, shear_rate=rate)
    u, _ = oracle.brownian_velocity(pos, force, box, p, kT, dt, seed, ts)
    newpos, newimg = oracle.integrate(pos, np.zeros((n, 3), dtype=np.int64), u, box, dt, rate)
    got = dpos.cpu().numpy()
    assert np.abs(got[:, :3] - newpos).max() < 1e-9
    assert np.array_equal(image.cpu().numpy(), newimg)
    assert np.all(got[:, 3] == 1.0)
    assert np.abs(accel.cpu().numpy() - force / 2.0).max() < 1e-15
    assert newimg.any(), "test should exercise the wrap"


@pytest.mark.parametrize("grid,xy,P", [((32, 32, 32), 0.0, 0), ((64, 48, 40), 0.3, 0), ((16, 36, 30), -0.2, 4), ((128, 32, 32), 0.1, 5),
                                       ((60, 48, 40), 0.25, 0), ((45, 45, 45), 0.0, 0), ((36, 30, 48), -0.3, 0), ((90, 40, 36), 0.1, 0), ((120, 36, 40), -0.15, 0),
                                       ((50, 32, 36), 0.0, 5), ((256, 32, 32), 0.2, 4), ((240, 32, 32), 0.15, 4), ((225, 32, 36), 0.0, 4),
                                       ((360, 32, 32), 0.1, 4), ((270, 32, 36), -0.1, 4), ((375, 32, 32), 0.0, 4), ((500, 32, 32), 0.2, 4), ((180, 36, 32), 0.0, 4),   # compile-time radix plans (round 4)
                                       ((32, 360, 36), 0.1, 4), ((32, 270, 32), 0.0, 4), ((36, 375, 30), 0.0, 4), ((32, 500, 32), -0.2, 4),   # the own y pass (round 4)
                                       ((36, 256, 30), 0.2, 4), ((256, 256, 32), 0.0, 4),   # the register y pass at Ny = 256 (round 4)
                                       ((32, 256, 256), 0.0, 4), ((32, 512, 512), 0.0, 4), ((32, 256, 512), 0.15, 4), ((32, 512, 256), 0.2, 4), ((36, 360, 256), -0.1, 4),   # the own z pass (k_zfft_rows, Nz = 256 / 512) and the register y pass at Ny = 512 (round 5)
                                       ((32, 36, 360), 0.1, 4), ((32, 40, 270), 0.0, 4), ((36, 36, 180), -0.2, 4), ((32, 36, 240), 0.0, 4), ((32, 36, 300), 0.15, 4),
                                       ((32, 36, 320), 0.0, 4), ((32, 36, 384), 0.0, 4), ((32, 36, 400), 0.1, 4), ((32, 36, 450), 0.0, 4), ((32, 36, 480), -0.1, 4),
                                       ((32, 36, 500), 0.0, 4), ((36, 256, 360), 0.2, 4),   # the own z pass at the other sizes of the reference's rule (k_zfft_rows_g, round 6)
                                       ((512, 32, 32), -0.1, 4)])  # the last in a box twice as long in x
def test_fused_x_pass_matches_port(torch_cuda, oracle, grid, xy, P):
    _check_far_field_passes(oracle, grid, xy, P)


# the sizes at which the developer switches select other kernels (tests/test_gpu_switches.py runs these under every switch): the
# register x passes (256, 360, 512), a runtime-plan size, the own y pass (360: k_fft_cols; 256, 512: k_yfft_regs), the own z pass
@pytest.mark.parametrize("grid,xy,P", [((256, 32, 32), 0.2, 4), ((360, 32, 32), 0.1, 4), ((512, 32, 32), -0.1, 4), ((240, 32, 32), 0.15, 4),
                                       ((64, 48, 40), 0.3, 0), ((32, 360, 36), 0.1, 4), ((36, 256, 30), 0.2, 4), ((32, 512, 256), 0.2, 4),
                                       ((32, 256, 512), 0.15, 4), ((32, 36, 360), 0.1, 4), ((36, 36, 180), -0.2, 4), ((32, 40, 270), 0.0, 4)])
def test_far_field_passes_at_the_switch_sizes(torch_cuda, oracle, grid, xy, P):
    _check_far_field_passes(oracle, grid, xy, P)


def _check_far_field_passes(oracle, grid, xy, P):
    """Every Nx = 2^a 3^b 5^c (the reference's grid rule, PSEv1/Stokes.cc:147-199) takes the fused forward-x FFT + k-space scaling
    (+ noise) + inverse-x FFT kernel: radix 4/2 in LDS for powers of two up to 128, the register pass k_xfft_scale_cols at 256, 360 and 512, mixed
    radix 9/8/5/4/3/2 otherwise (60 = 5 4 3, 45 = 9 5, 36 = 9 4, 90 = 9 5 2, 120 = 8 5 3, 50 = 5 5 2; 240 = 8 5 3 2 and 225 = 9 5 5 are grids above 200 (two kz columns per workgroup); 360 = 9 8 5 is the grid of the
    reference's rule at the metric point and is timed by tools/perf.py --grid 0 --xi 0.5).  Since round 4 every Ny = 2^a 3^b 5^c that is
    not a power of two takes the own in-place y pass (k_fft_cols: 48, 36, 30, 40, 45 above; 360 = 9 8 5, 270 = 9 5 3 2, 375 = 5 5 5 3 and
    500 = 5 5 5 4 are the sizes of the reference's rule at BASELINE configs 3 and 4), rocFFT keeping the 1-D z transforms; Ny = 256
    takes k_yfft_regs (the stages of the register x pass, natural order in and out).  Since round 5 Nz = 256 and 512 take the own z
    pass (k_zfft_rows: a wavefront per row, real <-> half spectrum in one kernel) wherever the y pass is the engine's own, and Ny = 512
    takes k_yfft_regs as well: rocFFT is then off the path altogether.  Since round 6 the even sizes of the reference's rule between 180
    and 500 take the own z pass too (k_zfft_rows_g in pse_zfft.hip: NC = Nz / 2 = R0 x R1 x R2 -- 360 = 2 x 3 x 6 x 10 is the grid the
    rule picks at the metric point)."""
    import pse_amd
    n = 1200
    pos, force, box = make_suspension(n, L=24.0, xy=xy)
    if grid[0] == 512:   # keep hx / hy where the 256 case has it
        pos = np.concatenate([pos[: n // 2] - [12.0, 0, 0], pos[n // 2:] + [12.0, 0, 0]])
        box = (48.0, 24.0, 24.0, xy)
    elif grid[0] > 256:  # likewise: a box stretched along x so that hx stays at the 256 case's 24 / 256
        Lx = grid[0] * 24.0 / 256.0
        pos[:, 0] = (pos[:, 0] - xy * pos[:, 1]) * (Lx / 24.0) + xy * pos[:, 1]
        box = (Lx, 24.0, 24.0, xy)
    seed, ts, kT, dt = 31, 4, 1.0, 1e-3
    eng = pse_amd.Engine(n, box, xi=0.5, error=1e-3, seed=seed, grid=grid, P=P)
    p = oracle.select_params(box, 0.5, 1e-3, 0.5, grid=grid, P=P or None)
    u = eng.mobility(to4(pos), to4(force), parts=2).cpu().numpy()[:, :3]
    assert rel(u, oracle.mobility_wave(pos, force, box, p)) < 1e-10
    vel, m = eng.brownian_velocity(to4(pos), to4(force), kT, dt, ts)
    ref, mref = oracle.brownian_velocity(pos, force, box, p, kT, dt, seed, ts)
    assert m == mref and rel(vel.cpu().numpy()[:, :3], ref) < 1e-9


def _mobility_block(eng, pos, j, i):
    """3x3 block M_ij through the engine: velocity of particle i for unit forces on particle j."""
    n = len(pos)
    M = np.zeros((3, 3))
    for c in range(3):
        f = np.zeros((n, 3)); f[j, c] = 1.0
        M[:, c] = eng.mobility(to4(pos), to4(f)).cpu().numpy()[i, :3]
    return M


@pytest.mark.parametrize("xi", [0.5, 0.75])
def test_known_answers_of_the_survey(torch_cuda, xi):
    """SURVEY.md 8c KAT-1..KAT-4 (generated from mathematics, a = 1, units 1/(6 pi eta a)) through the C-ABI at
    error = 1e-6; tolerance 5 x error (the method's own accuracy), cubic L = 20.  One and two particles: the smallest
    systems (single cell, minimum-image near field, one occupied far-field bin)."""
    import pse_amd
    box = (20.0, 20.0, 20.0, 0.0)
    tol = 5e-6
    eng = pse_amd.Engine(2, box, xi=xi, error=1e-6)
    # KAT-1: self mobility of a lone particle in the periodic box (Hasimoto), independent of xi
    one = np.array([[1.3, -2.1, 0.4]])
    M = _mobility_block(eng, one, 0, 0)
    assert abs(M[0, 0] - 0.85865872480157) < tol and abs(M[1, 1] - M[0, 0]) < tol and abs(M[0, 1]) < tol
    # KAT-2: separated pair r = (3, 1, 0.5)
    base = np.array([-4.0, 7.5, 9.0])          # near two box faces: images are exercised
    pair = np.stack([base + np.array([3.0, 1.0, 0.5]), base])
    M = _mobility_block(eng, pair, 1, 0)
    ref = np.array([[0.279076722785218, 0.053435549043855, 0.026722650338542],
                    [0.053435549043855, 0.128141730481001, 0.008928425757363],
                    [0.026722650338542, 0.008928425757363, 0.113989288350582]])
    assert np.abs(M - ref).max() < tol, np.abs(M - ref).max()
    # KAT-3: overlapping pair r = (1.2, 0.3, 0)
    pair = np.stack([base + np.array([1.2, 0.3, 0.0]), base])
    M = _mobility_block(eng, pair, 1, 0)
    assert abs(M[0, 0] - 0.6207687795007937) < tol and abs(M[1, 1] - 0.5177902235372976) < tol
    assert abs(M[2, 2] - 0.5109249998759693) < tol and abs(M[0, 1] - 0.02707348137451859) < tol
    # KAT-4: touching pair r = (2, 0, 0)
    pair = np.stack([base + np.array([2.0, 0.0, 0.0]), base])
    M = _mobility_block(eng, pair, 1, 0)
    assert abs(M[0, 0] - 0.4860111219717241) < tol and abs(M[1, 1] - 0.2965766796953576) < tol and abs(M[2, 2] - M[1, 1]) < tol


def test_single_particle_brownian_step(torch_cuda):
    """N = 1: the Lanczos iteration breaks down after one vector (M_real is 3x3 diagonal); the reference drops that
    vector and fails, the build keeps it: u = sqrt(2 kT/dt) sqrt(M) psi has the right magnitude."""
    import pse_amd
    box = (20.0, 20.0, 20.0, 0.0)
    eng = pse_amd.Engine(1, box, xi=0.5, error=1e-3, seed=7)
    pos = to4(np.array([[0.5, 0.25, -3.0]])); zero = to4(np.zeros((1, 3)))
    kT, dt = 1.0, 1e-3
    us = []
    for ts in range(200):
        vel, m = eng.brownian_velocity(pos, zero, kT, dt, ts)
        assert 1 <= m <= 2
        us.append(vel.cpu().numpy()[0, :3].copy())
    us = np.array(us)
    assert np.all(np.isfinite(us))
    var = (us ** 2).mean()                                  # <u_x^2> = 2 kT M_xx / dt
    assert abs(var / (2 * kT * 0.85865872480157 / dt) - 1.0) < 0.25


def test_fluctuation_dissipation(torch_cuda, oracle):
    """SURVEY.md 8c property (v): <u u^T> = (2 kT/dt) M for the Brownian velocity (k-space noise + Lanczos real-space
    noise; F = 0) against the dense direct-Ewald mobility, N = 12, 8000 independent timesteps.  Statistical bounds:
    the relative Frobenius error of a sample covariance of dimension d over T samples is ~sqrt((d+1)/T) = 0.068."""
    import pse_amd
    n, L, kT, dt, T = 12, 16.0, 1.0, 1e-3, 8000
    box = (L, L, L, 0.0)
    rng = np.random.default_rng(3)
    pos = rng.uniform(-L / 2, L / 2, size=(n, 3))
    eng = pse_amd.Engine(n, box, xi=0.5, error=1e-3, seed=99)
    dpos, zero = to4(pos), to4(np.zeros((n, 3)))
    acc = torch_cuda.zeros((3 * n, 3 * n), dtype=torch_cuda.float64, device="cuda")
    mean = torch_cuda.zeros(3 * n, dtype=torch_cuda.float64, device="cuda")
    m = 2
    for ts in range(T):
        vel, m = eng.brownian_velocity(dpos, zero, kT, dt, ts, lanczos_m=m)
        u = vel[:, :3].reshape(-1)
        acc += torch_cuda.outer(u, u)
        mean += u
    C = (acc / T).cpu().numpy() * dt / (2 * kT)
    M = oracle.mobility_dense(pos, box, 0.5)
    assert np.abs((mean / T).cpu().numpy()).max() * math.sqrt(dt / (2 * kT)) < 5.0 / math.sqrt(T)     # zero mean
    assert abs(np.trace(C) / np.trace(M) - 1.0) < 0.015
    assert np.linalg.norm(C - M) / np.linalg.norm(M) < 0.1


@pytest.mark.parametrize("xy,err,L", [(0.0, 1e-3, 18.0), (0.3, 1e-3, 18.0), (0.3, 1e-6, 24.0), (-0.4, 1e-5, 24.0)])   # P = 6, 6, 13, 11 (binned far field)
def test_both_halves_symmetric_positive_definite(torch_cuda, xy, err, L):
    """The "positively split" claim (SURVEY.md 8c property ii): M_real and M_wave are separately symmetric positive
    definite.  Symmetry of the wave half is the adjointness of the spread and gather kernels (same weights) through the
    real-to-complex transforms: exact to rounding."""
    import pse_amd
    n = 30
    box = (L, L, L, xy)
    pos, _, _ = make_suspension(n, L=L, xy=xy)
    eng = pse_amd.Engine(n, box, xi=0.5, error=err)
    dpos = to4(pos)
    eye = np.eye(3 * n)
    for parts, tol in ((1, 1e-13), (2, 1e-12)):
        M = np.stack([eng.mobility(dpos, to4(eye[c].reshape(n, 3)), parts=parts).cpu().numpy()[:, :3].ravel()
                      for c in range(3 * n)], 1)
        assert np.abs(M - M.T).max() < tol * np.abs(M).max(), (parts, np.abs(M - M.T).max())
        assert np.linalg.eigvalsh(0.5 * (M + M.T)).min() > 0.0, parts


@pytest.mark.parametrize("xy", [0.0, 0.3])
def test_pair_repulsion_matches_port(torch_cuda, oracle, xy):
    """Force provider: soft repulsion from the engine's cell list against the O(N^2) restatement;
    overwrite / accumulate semantics and the w component."""
    import pse_amd
    n = 1500
    pos, force, box = make_suspension(n, phi=0.35, xy=xy)      # uniform random positions: plenty of overlaps
    eng = pse_amd.Engine(n, box, xi=0.5, error=1e-3)
    ref = oracle.pair_repulsion(pos, box, 40.0, 2.0)
    assert np.count_nonzero(np.abs(ref).sum(1)) > n // 2
    f = to4(force, 7.0)
    out = eng.pair_repulsion(to4(pos), f, 40.0, 2.0, accumulate=False).cpu().numpy()
    assert rel(out[:, :3], ref) < 1e-12 and np.all(out[:, 3] == 7.0)
    out = eng.pair_repulsion(to4(pos), to4(force, 7.0), 40.0, 2.0, accumulate=True).cpu().numpy()
    assert rel(out[:, :3], ref + force) < 1e-12
    with pytest.raises(pse_amd.PSEError):
        eng.pair_repulsion(to4(pos), f, 40.0, 2.0 * eng.info()["rcut"])


def test_the_two_halves_of_a_brownian_evaluation_add_up(torch_cuda, oracle):
    """pse_brownian_velocity_part (the functional split of a two-GPU run: one GPU the real-space half with the Lanczos noise, the other
    the wave-space half with the k-space noise) and pse_integrate (K15 alone): the halves add up to pse_brownian_velocity, the
    real-space half is the port's, and halves + integrate = pse_step."""
    import torch
    import pse_amd
    n = 1500
    pos, force, box = make_suspension(n, phi=0.1, xy=0.2)
    seed, ts, kT, dt, rate = 77, 5, 1.0, 2e-2, 0.4
    eng = pse_amd.Engine(n, box, xi=0.5, error=1e-3, seed=seed)
    p = oracle.select_params(box, 0.5, 1e-3, 0.5)
    whole, m = eng.brownian_velocity(to4(pos), to4(force), kT, dt, ts)
    a, ma = eng.brownian_velocity_part(to4(pos), to4(force), kT, dt, ts, 1, vel=to4(np.zeros((n, 3)), 2.0))
    b, mb = eng.brownian_velocity_part(to4(pos), to4(force), kT, dt, ts, 2, vel=to4(np.zeros((n, 3)), 2.0), lanczos_m=3)
    assert ma == m and mb == 3                                   # (the wave half runs no Lanczos iteration: the count comes back as given)
    assert np.all(a.cpu().numpy()[:, 3] == 2.0) and np.all(b.cpu().numpy()[:, 3] == 2.0)
    s = a.cpu().numpy()[:, :3] + b.cpu().numpy()[:, :3]
    assert rel(s, whole.cpu().numpy()[:, :3]) < 1e-13
    # the real-space half against the port: M_real.F + sqrt(2 kT / dt) M_real^{1/2} psi
    psi = oracle.psi_particles(n, seed, ts)
    ub, mp = oracle.lanczos_sqrt(lambda v: oracle.mobility_real(pos, np.ascontiguousarray(v), box, 0.5, p["rcut"], rounded=True), psi, 2, 1e-3)
    ref_a = oracle.mobility_real(pos, force, box, 0.5, p["rcut"]) + np.sqrt(2.0 * kT / dt) * ub
    assert mp == m and rel(a.cpu().numpy()[:, :3], ref_a) < 1e-9
    # kT = 0: the halves are pse_mobility's parts
    a0, _ = eng.brownian_velocity_part(to4(pos), to4(force), 0.0, dt, ts, 1)
    assert rel(a0.cpu().numpy()[:, :3], eng.mobility(to4(pos), to4(force), parts=1).cpu().numpy()[:, :3]) < 1e-14
    # halves + pse_integrate = pse_step
    dpos, dvel, dF = to4(pos, 1.0), to4(np.zeros((n, 3)), 1.5), to4(force)
    accel = torch.zeros((n, 3), dtype=torch.float64, device="cuda"); image = torch.zeros((n, 3), dtype=torch.int32, device="cuda")
    eng.step(dpos, dvel, accel, image, dF, kT, dt, ts, shear_rate=rate)
    p2, v2 = to4(pos, 1.0), to4(s, 1.5)
    acc2 = torch.zeros((n, 3), dtype=torch.float64, device="cuda"); im2 = torch.zeros((n, 3), dtype=torch.int32, device="cuda")
    eng.integrate(p2, v2, acc2, im2, dF, dt, shear_rate=rate)
    assert np.abs(p2.cpu().numpy() - dpos.cpu().numpy()).max() < 1e-12 and torch.equal(im2, image) and torch.equal(acc2, accel)
    with pytest.raises(pse_amd.PSEError):
        eng.brownian_velocity_part(to4(pos), to4(force), kT, dt, ts, 0)
