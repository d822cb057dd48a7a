"""Parity at BASELINE.json's full sizes through size-independent properties (the O(N^2) oracle cannot evaluate 10^6
particles): linearity, reciprocity F2.(M F1) = F1.(M F2), positivity, periodic-image invariance, |M^{1/2} psi|^2 =
psi.M.psi, plus a direct check of 256 random rows of the near field against the C oracle."""
import math

import numpy as np
import pytest

from conftest import make_suspension, to4

pytestmark = pytest.mark.gpu


def rel(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


def _engine(n, phi, grid, err=1e-3, xy=0.0):
    import pse_amd
    pos, force, box = make_suspension(n, phi=phi, xy=xy)
    L = box[0]
    xi = math.pi * grid / (2.0 * L * math.sqrt(-math.log(err)))     # xi from the fixed grid
    return pse_amd.Engine(n, box, xi=xi, error=err, seed=11, grid=(grid,) * 3), pos, force, box, xi


@pytest.mark.parametrize("n,phi,grid,xy", [(1_000_000, 0.1, 256, 0.0),      # the metric point
                                           (65_536, 0.1, 64, 0.0),          # BASELINE config 2
                                           (1_048_576, 0.2, 256, 0.3)])     # config 3 geometry, sheared (config 5)
def test_fullsize_properties(oracle, n, phi, grid, xy):
    import torch
    eng, pos, F1, box, xi = _engine(n, phi, grid, xy=xy)
    rcut = eng.info()["rcut"]
    F2 = np.random.default_rng(9).normal(size=(n, 3)); F2 -= F2.mean(0)
    dpos = to4(pos)
    U1 = eng.mobility(dpos, to4(F1)).cpu().numpy()[:, :3]
    U2 = eng.mobility(dpos, to4(F2)).cpu().numpy()[:, :3]
    a, b = 0.7, -1.9
    U12 = eng.mobility(dpos, to4(a * F1 + b * F2)).cpu().numpy()[:, :3]
    assert rel(U12, a * U1 + b * U2) < 1e-12                              # linearity
    s12, s21 = np.sum(F2 * U1), np.sum(F1 * U2)
    assert abs(s12 - s21) < 1e-9 * max(abs(s12), np.sqrt(np.sum(F1 * U1) * np.sum(F2 * U2)))   # reciprocity (M symmetric)
    assert np.sum(F1 * U1) > 0 and np.sum(F2 * U2) > 0                    # positive definite
    # periodic images: move a third of the particles by lattice vectors
    img = pos.copy()
    img[::3] += np.array([box[3] * box[1], box[1], 0.0]); img[1::3] -= np.array([box[0], 0.0, box[2]])
    assert rel(eng.mobility(to4(img), to4(F1)).cpu().numpy()[:, :3], U1) < 1e-9
    # near field: 256 random rows against the oracle's closed form
    rows = np.random.default_rng(1).choice(n, 256, replace=False).astype(np.int32)
    Ur = eng.mobility(dpos, to4(F1), parts=1).cpu().numpy()[:, :3]
    ref = oracle.mobility_real_rows(pos, F1, box, xi, rcut, rows)
    assert rel(Ur[rows], ref) < 1e-12
    # Brownian part: |M_real^{1/2} psi|^2 = psi . M_real psi within the Lanczos tolerance
    psi = np.random.default_rng(2).normal(size=(n, 3))
    u, m = eng.sqrt_mreal(dpos, to4(psi), tol=1e-3)
    Mpsi = eng.mobility(dpos, to4(psi), parts=1).cpu().numpy()[:, :3]
    lhs, rhs = float((u[:, :3] ** 2).sum()), float(np.sum(psi * Mpsi))
    assert abs(lhs - rhs) < 5e-3 * rhs and 2 <= m <= 40, (lhs, rhs, m)
    del eng
    torch.cuda.empty_cache()


def test_step_at_metric_point_moves_particles_and_counts_lanczos():
    import torch
    eng, pos, F, box, xi = _engine(1_000_000, 0.1, 256)
    n = len(pos)
    dpos, vel, dF = to4(pos), to4(np.zeros((n, 3)), 1.0), to4(F)
    accel = torch.zeros((n, 3), dtype=torch.float64, device="cuda"); image = torch.zeros((n, 3), dtype=torch.int32, device="cuda")
    m = 2
    for ts in range(3):
        m = eng.step(dpos, vel, accel, image, dF, 1.0, 1e-3, ts, lanczos_m=m)
    p = dpos.cpu().numpy()[:, :3]
    assert np.isfinite(p).all() and np.abs(p).max() <= box[0] / 2 * (1 + 1e-12)
    disp = np.linalg.norm(p - pos, axis=1)
    msd = float(np.mean(np.minimum(disp, box[0] - disp) ** 2))
    # three Brownian steps of dt = 1e-3 with D ~ kT * (self mobility ~ 0.9): MSD ~ 6 D 3 dt ~ 0.016
    assert 0.003 < msd < 0.05, msd
    assert 4 <= m <= 20


def test_four_rank_team_at_metric_point():
    """The slab decomposition at full size (N = 1e6, 256^3, four ranks in one process): M.F and the Brownian velocity
    against the single-GPU engine."""
    import torch
    import pse_amd
    from pse_amd.sharded import LoopbackSimulation
    n, grid, err = 1_000_000, 256, 1e-3
    pos, force, box = make_suspension(n, phi=0.1)
    xi = math.pi * grid / (2.0 * box[0] * math.sqrt(-math.log(err)))
    kw = dict(xi=xi, error=err, seed=11, grid=(grid,) * 3)
    ref = pse_amd.Engine(n, box, **kw)
    u_ref = ref.mobility(to4(pos), to4(force)).cpu().numpy()[:, :3]
    v_ref, m_ref = ref.brownian_velocity(to4(pos), to4(force), 1.0, 1e-3, 4)
    v_ref = v_ref.cpu().numpy()[:, :3]
    del ref
    torch.cuda.empty_cache()
    sim = LoopbackSimulation(n, box, 4, **kw)
    sim.load(pos, force)
    vels = sim.mobility()
    assert rel(vels[0].cpu().numpy()[:, :3], u_ref) < 1e-12 and rel(vels[3].cpu().numpy()[:, :3], u_ref) < 1e-12
    vels, m = sim.brownian_velocity(1.0, 1e-3, 4)
    assert m == m_ref
    assert rel(vels[1].cpu().numpy()[:, :3], v_ref) < 1e-10 and rel(vels[2].cpu().numpy()[:, :3], v_ref) < 1e-10
    del sim
    torch.cuda.empty_cache()
