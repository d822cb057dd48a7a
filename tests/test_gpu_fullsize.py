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


def _msd_check(p, pos, box, steps, dt):
    assert np.isfinite(p).all()
    f = (p[:, 0] - box[3] * p[:, 1]) / box[0]
    assert np.abs(f).max() <= 0.5 * (1 + 1e-12) and np.abs(p[:, 1]).max() <= box[1] / 2 * (1 + 1e-12)
    d = p - pos
    d[:, 1] -= box[1] * np.rint(d[:, 1] / box[1]); d[:, 2] -= box[2] * np.rint(d[:, 2] / box[2])
    d[:, 0] -= box[0] * np.rint(d[:, 0] / box[0])
    msd = float(np.mean(d[:, 1] ** 2 + d[:, 2] ** 2))       # y, z: free of the affine shear displacement
    # Brownian steps with D ~ kT x (self mobility ~ 0.85-0.9): MSD_yz ~ 4 D steps dt
    assert 0.5 * 4 * 0.6 * steps * dt < msd < 1.5 * 4 * 1.0 * steps * dt, msd


def test_config3_full_brownian_steps():
    """BASELINE config 3: N = 1,048,576, phi = 0.20, 256^3, unsheared -- full pse_step x 3 (M.F + k-space noise + Lanczos
    + Euler), checked through what does not need the O(N^2) oracle: wrapped positions, Brownian MSD, Lanczos count, accel = F/m,
    and the velocity of the last step against a fresh pse_brownian_velocity at the same positions (same timestep)."""
    import torch
    eng, pos, F, box, xi = _engine(1_048_576, 0.2, 256)
    n = len(pos)
    dpos, vel, dF = to4(pos), to4(np.zeros((n, 3)), 2.0), to4(F)
    accel = torch.zeros((n, 3), dtype=torch.float64, device="cuda"); image = torch.zeros((n, 3), dtype=torch.int32, device="cuda")
    m, dt = 2, 1e-3
    for ts in range(3):
        before = dpos.clone()
        m = eng.step(dpos, vel, accel, image, dF, 1.0, dt, ts, lanczos_m=m)
    assert 3 <= m <= 20
    _msd_check(dpos.cpu().numpy()[:, :3], pos, box, 3, dt)
    assert torch.allclose(accel, dF[:, :3] / 2.0, rtol=0, atol=1e-15)
    v2, m2 = eng.brownian_velocity(before, dF, 1.0, dt, 2, lanczos_m=2)
    # same positions, same timestep: the same velocity -- to round-off if the Lanczos iteration stopped at the same m,
    # otherwise to the tolerance of the iteration (the stochastic part is ~30x the deterministic one at this dt)
    assert abs(m2 - m) <= 2
    assert rel(v2.cpu().numpy()[:, :3], vel.cpu().numpy()[:, :3]) < (1e-9 if m2 == m else 5e-3)
    # the step moved the particles by exactly vel dt (unsheared), up to the wrap
    d = (dpos - before).cpu().numpy()[:, :3]
    d -= np.array(box[:3]) * np.rint(d / np.array(box[:3]))
    assert np.abs(d - vel.cpu().numpy()[:, :3] * dt).max() < 1e-12
    del eng
    torch.cuda.empty_cache()


def test_config5_oscillatory_shear_steps(oracle):
    """BASELINE config 5: config 3's geometry under SinShearFunction(max_shear_rate = 1, frequency = 1, offset = 0, dt = 1e-3):
    the box tilt follows the wrapped strain of variant.shear_variant through pse_set_box, the shear rate comes from the
    function; kT = 1.  Checked against the oracle's shear formulas, the affine displacement and the Brownian MSD."""
    import torch
    from pse_amd import shear_function, variant
    eng, pos, F, box, xi = _engine(1_048_576, 0.2, 256)
    n = len(pos)
    dt, t_start = 1e-3, 240                                   # start near the peak strain gamma_max = 1/(2 pi) = 0.159
    f = shear_function.sine(dt=dt, shear_rate=1.0, shear_freq=1.0)
    var = variant.shear_variant(f, 10_000, max_strain=0.5)
    ref = oracle.SinShear(1.0, 1.0, 0, dt)
    dpos, vel, dF = to4(pos), to4(np.zeros((n, 3)), 1.0), to4(F)
    accel = torch.zeros((n, 3), dtype=torch.float64, device="cuda"); image = torch.zeros((n, 3), dtype=torch.int32, device="cuda")
    m, cur_xy = 2, 0.0
    for ts in range(t_start, t_start + 3):
        xy = var.get_value(ts)
        assert abs(xy - oracle.variant_value(ref, ts, 10_000, -0.5, 0.5)) < 1e-15
        rate = f.cpp_function.getShearRate(ts)
        assert abs(rate - ref.shear_rate(ts)) < 1e-14
        if xy != cur_xy:                                      # re-label images for the new tilt, then tell the engine
            fl = torch.floor((dpos[:, 0] - xy * dpos[:, 1]) / box[0] + 0.5)
            dpos[:, 0] -= fl * box[0]; image[:, 0] += fl.to(torch.int32)
            eng.set_box(box[0], box[1], box[2], xy); cur_xy = xy
        before = dpos.cpu().numpy()[:, :3].copy()
        m = eng.step(dpos, vel, accel, image, dF, 1.0, dt, ts, shear_rate=rate, lanczos_m=m)
    assert abs(cur_xy) > 0.15 and 3 <= m <= 20
    p = dpos.cpu().numpy()[:, :3]
    _msd_check(p, pos, (box[0], box[1], box[2], cur_xy), 3, dt)
    # the last step: x += (u_x + rate y) dt, y += u_y dt, z += u_z dt (PSEv1/Stokes.cu:164-171), then the triclinic wrap
    v = vel.cpu().numpy()[:, :3]
    d = p - before - v * dt
    d[:, 0] -= rate * before[:, 1] * dt
    ny = np.rint(d[:, 1] / box[1]); d[:, 1] -= ny * box[1]; d[:, 0] -= ny * cur_xy * box[1]
    d[:, 0] -= np.rint(d[:, 0] / box[0]) * box[0]; d[:, 2] -= np.rint(d[:, 2] / box[2]) * box[2]
    assert np.abs(d).max() < 1e-11
    del eng
    torch.cuda.empty_cache()


def test_config4_single_gpu_512_cubed(oracle):
    """BASELINE config 4's problem (N = 4,194,304, phi = 0.30, 512^3: the reference's grid cap, PSEv1/Stokes.cc:201-214) on one
    GPU: linearity, reciprocity, positivity, image invariance, 256 near-field rows against the oracle, one Brownian step."""
    import torch
    eng, pos, F1, box, xi = _engine(4_194_304, 0.3, 512)
    n = len(pos)
    info = eng.info()
    assert (info["Nx"], info["Ny"], info["Nz"], info["P"]) == (512, 512, 512, 6)
    F2 = np.random.default_rng(9).normal(size=(n, 3)); F2 -= F2.mean(0)
    dpos = to4(pos)
    U1 = eng.mobility(dpos, to4(F1)).cpu().numpy()[:, :3]
    U2 = eng.mobility(dpos, to4(F2)).cpu().numpy()[:, :3]
    U12 = eng.mobility(dpos, to4(0.7 * F1 - 1.9 * F2)).cpu().numpy()[:, :3]
    assert rel(U12, 0.7 * U1 - 1.9 * U2) < 1e-12
    s12, s21 = np.sum(F2 * U1), np.sum(F1 * U2)
    assert abs(s12 - s21) < 1e-9 * np.sqrt(np.sum(F1 * U1) * np.sum(F2 * U2))
    assert np.sum(F1 * U1) > 0 and np.sum(F2 * U2) > 0
    img = pos.copy()
    img[::3] += np.array([0.0, box[1], 0.0]); img[1::3] -= np.array([box[0], 0.0, box[2]])
    assert rel(eng.mobility(to4(img), to4(F1)).cpu().numpy()[:, :3], U1) < 1e-9
    rows = np.random.default_rng(1).choice(n, 256, replace=False).astype(np.int32)
    Ur = eng.mobility(dpos, to4(F1), parts=1).cpu().numpy()[:, :3]
    assert rel(Ur[rows], oracle.mobility_real_rows(pos, F1, box, xi, info["rcut"], rows)) < 1e-12
    vel = to4(np.zeros((n, 3)), 1.0)
    accel = torch.zeros((n, 3), dtype=torch.float64, device="cuda"); image = torch.zeros((n, 3), dtype=torch.int32, device="cuda")
    m = eng.step(dpos, vel, accel, image, to4(F1), 1.0, 1e-3, 0, lanczos_m=2)
    assert 3 <= m <= 25
    _msd_check(dpos.cpu().numpy()[:, :3], pos, box, 1, 1e-3)
    del eng
    torch.cuda.empty_cache()


def test_config4_eight_slab_loopback_matches_single_gpu():
    """BASELINE config 4 as it is meant to run: the 512^3 far field cut into eight x slabs (all eight ranks in this process,
    copies instead of RCCL): M.F against the single-GPU engine."""
    import torch
    import pse_amd
    from pse_amd.sharded import LoopbackSimulation
    n, grid, err = 4_194_304, 512, 1e-3
    pos, force, box = make_suspension(n, phi=0.3)
    xi = math.pi * grid / (2.0 * box[0] * math.sqrt(-math.log(err)))
    kw = dict(xi=xi, error=err, seed=11, grid=(grid,) * 3)
    ref = pse_amd.Engine(n, box, **kw)
    u_ref = ref.mobility(to4(pos), to4(force)).cpu().numpy()[:, :3]
    del ref
    torch.cuda.empty_cache()
    sim = LoopbackSimulation(n, box, 8, **kw)
    sim.load(pos, force)
    vels = sim.mobility()
    assert rel(vels[0].cpu().numpy()[:, :3], u_ref) < 1e-12 and rel(vels[5].cpu().numpy()[:, :3], u_ref) < 1e-12
    del sim
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


@pytest.mark.parametrize("n,phi,grid,xy", [(1_000_000, 0.1, 256, 0.0),      # the metric point
                                           (1_048_576, 0.2, 256, 0.3),      # config 3 geometry, sheared (config 5)
                                           (4_194_304, 0.3, 512, 0.0)])     # config 4 on one GPU
def test_fullsize_far_field_accuracy(oracle, n, phi, grid, xy):
    """What the property set above cannot see: a stencil that is consistently wrong at a block, bin or
    slab edge passes linearity, reciprocity and LATTICE-vector invariance.  (i) The wave part is a function of separations
    only, so moving every particle by an off-grid vector changes it by the discretisation error alone: <= 5 x error.
    (ii) The spread grid on sub-volumes (across the periodic wrap and across interior block / bin edges) against the port's
    spread of exactly the particles that reach them (PSEv1/Mobility.cu:212-246)."""
    import torch
    eng, pos, F, box, xi = _engine(n, phi, grid, xy=xy)
    info = eng.info()
    h = np.array([box[0] / grid, box[1] / grid, box[2] / grid])
    U = eng.mobility(to4(pos), to4(F), parts=2).cpu().numpy()[:, :3]
    shift = np.array([0.37, -0.61, 0.23]) * h
    moved = pos + shift
    moved[:, 0] += box[3] * shift[1]                                     # the node lattice is sheared: (0.37, -0.61, 0.23) in lattice units
    U2 = eng.mobility(to4(moved), to4(F), parts=2).cpu().numpy()[:, :3]
    assert rel(U2, U) < 5e-3, rel(U2, U)
    # (ii) sub-volumes of the force grid
    p = oracle.select_params(box, xi, 1e-3, 0.5, grid=(grid,) * 3)
    assert p["P"] == info["P"] and abs(p["eta"] - info["eta"]) < 1e-13
    g = eng.debug_spread(to4(pos), to4(F))
    idx, _ = oracle._support(pos, box, p)
    P = p["P"]
    for lo in ([grid - 6, grid - 5, grid - 7], [61, 125, 13], [120, 8, 250]):   # across the wrap; across 8- and 16-node block edges
        ext = 14
        sel = np.ones(n, dtype=bool)
        for a in range(3):
            d = (idx[a] - lo[a]) % grid                                   # (n, P): node offsets from the sub-volume's corner
            sel &= (d < ext).any(axis=1)
        sub = np.where(sel)[0]
        assert len(sub) > 20
        w, lin = oracle._weights(pos[sub], box, p)
        ref = np.zeros((3, grid ** 3 if grid <= 256 else 1))
        nodes = (np.arange(ext)[:, None, None] + lo[0]) % grid * grid * grid + (np.arange(ext)[None, :, None] + lo[1]) % grid * grid \
            + (np.arange(ext)[None, None, :] + lo[2]) % grid
        if grid <= 256:
            for c in range(3):
                np.add.at(ref[c], lin.ravel(), (w * F[sub, c][:, None, None, None]).ravel())
            want = ref[:, nodes.ravel()]
        else:                                                             # 512^3: accumulate on the sub-volume only
            pos_of = {int(v): q for q, v in enumerate(nodes.ravel())}
            want = np.zeros((3, nodes.size))
            for q, v in enumerate(lin.ravel()):
                t = pos_of.get(int(v))
                if t is not None:
                    s_ = q // (P ** 3)
                    want[:, t] += w.ravel()[q] * F[sub[s_]]
        got = g.reshape(3, -1)[:, nodes.ravel()]
        assert np.abs(got - want).max() < 1e-12 * np.abs(want).max(), lo
    del eng, g
    torch.cuda.empty_cache()


@pytest.mark.parametrize("n,phi,grid", [(1_000_000, 0.1, 256), (4_194_304, 0.3, 512)])
def test_owned_particle_team_of_eight_at_full_size(n, phi, grid):
    """The owned-particle step (pse_team_step_local) at the metric point and at BASELINE config 4, eight ranks in this process: two
    Brownian steps against the single-GPU engine (positions 1e-9, equal Lanczos counts), every particle owned by exactly one rank
    afterwards, no device-side flag."""
    import torch
    import pse_amd
    from pse_amd.sharded import LocalLoopbackSimulation
    pos, force, box = make_suspension(n, phi=phi)
    xi = math.pi * grid / (2.0 * box[0] * math.sqrt(-math.log(1e-3)))
    kw = dict(xi=xi, error=1e-3, seed=11, grid=(grid,) * 3)
    ref = pse_amd.Engine(n, box, **kw)
    dpos, dF, vel = to4(pos), to4(force), to4(np.zeros((n, 3)), 1.0)
    accel = torch.zeros((n, 3), dtype=torch.float64, device="cuda"); image = torch.zeros((n, 3), dtype=torch.int32, device="cuda")
    kT, dt = 1.0, 1e-3
    _, m = ref.brownian_velocity(dpos, dF, kT, dt, 0, vel=to4(np.zeros((n, 3)), 1.0), lanczos_m=2)
    ms = [ref.step(dpos, vel, accel, image, dF, kT, dt, 1 + k, lanczos_m=m) for k in range(2)]
    p_ref, im_ref = dpos.cpu().numpy()[:, :3].copy(), image.cpu().numpy().copy()
    del ref, dpos, dF, vel, accel, image
    torch.cuda.empty_cache()
    sim = LocalLoopbackSimulation(n, box, 8, **kw)
    sim.load(pos, force)
    for k in range(2):
        sim.step(kT, dt, 1 + k, lanczos_m=m)
        torch.cuda.synchronize()
        infos = [e.info() for e in sim.engines]
        assert all(i["lanczos_status"] == 0 and i["lanczos_m"] == ms[k] for i in infos), (k, ms[k], [i["lanczos_m"] for i in infos])
    p, u, im, owner = sim.gather()
    assert (owner >= 0).all()
    assert sum(int(s.n_local.item()) for s in sim.s) == n
    assert np.abs(p - p_ref).max() < 1e-9 and np.array_equal(im, im_ref)
    del sim
    torch.cuda.empty_cache()
