"""Neighbour list kept across calls (the reference's HOOMD nlist with r_buff = 0.4 and a distance check every step,
PSEv1/integrate.py:60,79, Stokes.cc:433): calls that reuse it must give what a fresh build gives, and the distance check
must send the call back to the cells when it has to."""
import numpy as np
import pytest

from conftest import make_suspension, to4

pytestmark = pytest.mark.gpu


def rel(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a device"
    return torch


def _fresh(n, box, pos, force, kT, dt, ts, seed, **kw):
    import pse_amd
    e = pse_amd.Engine(n, box, xi=0.5, error=1e-3, seed=seed, **kw)
    e.set_neighbor_skin(0.0)
    if kT > 0:
        v, m = e.brownian_velocity(to4(pos), to4(force), kT, dt, ts)
    else:
        v, m = e.mobility(to4(pos), to4(force)), 0
    return v.cpu().numpy()[:, :3], m


@pytest.mark.parametrize("xy", [0.0, 0.3])
def test_reused_list_matches_fresh_build(torch_cuda, oracle, xy):
    """Second and third calls at moved positions (within r_buff / 2, some across the periodic boundary) run on the kept list:
    M.F and the Brownian velocity equal what an engine that rebuilds gives, and equal the port."""
    import pse_amd
    n, seed, kT, dt = 2000, 7, 1.0, 1e-3
    pos, force, box = make_suspension(n, phi=0.1, xy=xy)
    eng = pse_amd.Engine(n, box, xi=0.5, error=1e-3, seed=seed)
    r_buff, b0, u0 = eng.neighbor_stats()
    assert r_buff == pytest.approx(0.4) and b0 == 0 and u0 == 0
    eng.mobility(to4(pos), to4(force))
    assert eng.neighbor_stats()[1:] == (1, 0)
    rng = np.random.default_rng(3)
    d = rng.normal(size=(n, 3)); d *= 0.19 * rng.uniform(size=(n, 1)) / np.linalg.norm(d, axis=1, keepdims=True)
    moved = oracle.wrap(pos + d, np.zeros((n, 3), dtype=np.int64), box)[0]
    assert (np.abs(moved - pos).max(axis=1) > 1.0).any(), "some particles should cross the boundary"
    v = eng.mobility(to4(moved), to4(force)).cpu().numpy()[:, :3]
    assert eng.neighbor_stats()[1:] == (1, 1)
    ref, _ = _fresh(n, box, moved, force, 0.0, dt, 0, seed)
    assert rel(v, ref) < 1e-12, rel(v, ref)
    vb, m = eng.brownian_velocity(to4(moved), to4(force), kT, dt, 5)
    assert eng.neighbor_stats()[1:] == (1, 2)
    refb, mref = _fresh(n, box, moved, force, kT, dt, 5, seed)
    assert m == mref
    assert rel(vb.cpu().numpy()[:, :3], refb) < 1e-11, rel(vb.cpu().numpy()[:, :3], refb)
    p = oracle.select_params(box, 0.5, 1e-3, 0.5)
    port, mp = oracle.brownian_velocity(moved, force, box, p, kT, dt, seed, 5)
    assert m == mp and rel(vb.cpu().numpy()[:, :3], port) < 1e-9


def test_distance_check_forces_a_rebuild(torch_cuda, oracle):
    import pse_amd
    n, seed = 2000, 11
    pos, force, box = make_suspension(n, phi=0.1)
    eng = pse_amd.Engine(n, box, xi=0.5, error=1e-3, seed=seed)
    eng.mobility(to4(pos), to4(force))
    moved = pos.copy()
    moved[17] += np.array([0.15, -0.12, 0.1])      # 0.216 > r_buff / 2
    moved = oracle.wrap(moved, np.zeros((n, 3), dtype=np.int64), box)[0]
    v = eng.mobility(to4(moved), to4(force)).cpu().numpy()[:, :3]
    assert eng.neighbor_stats()[1:] == (2, 0)
    ref, _ = _fresh(n, box, moved, force, 0.0, 1e-3, 0, seed)
    assert rel(v, ref) < 1e-12
    # far move of many particles, another N, another group: always a build
    eng.mobility(to4(-moved), to4(force))
    assert eng.neighbor_stats()[1:] == (3, 0)
    eng.mobility(to4(moved[:1500]), to4(force[:1500]))
    assert eng.neighbor_stats()[1:] == (4, 0)
    # a smaller r_buff; zero switches the list off
    eng.set_neighbor_skin(0.1)
    eng.mobility(to4(moved), to4(force)); eng.mobility(to4(moved), to4(force))
    assert eng.neighbor_stats() == (0.1, 5, 1)
    eng.set_neighbor_skin(0.0)
    v0 = eng.mobility(to4(moved), to4(force)).cpu().numpy()[:, :3]
    assert eng.neighbor_stats()[1:] == (6, 1) and rel(v0, ref) < 1e-12
    with pytest.raises(Exception):
        eng.set_neighbor_skin(0.5)


@pytest.mark.parametrize("dt", [2e-4, 5e-3])
def test_trajectory_with_and_without_the_kept_list(torch_cuda, dt):
    """Thirty Brownian steps: the run that keeps the list and the run that rebuilds every step end at the same positions and
    image flags, with the same Lanczos iteration counts.  Small steps: a few builds, the rest reuse.  Large steps (every
    particle diffuses past r_buff / 2 per step): the list is never reusable and gets suspended; results are the same."""
    import torch
    import pse_amd
    n, seed, kT = 3000, 5, 1.0
    pos, force, box = make_suspension(n, phi=0.15)
    out = []
    for skin in (0.4, 0.0):
        eng = pse_amd.Engine(n, box, xi=0.5, error=1e-3, seed=seed)
        eng.set_neighbor_skin(skin)
        dpos = to4(pos, 1.0); vel = to4(np.zeros((n, 3))); dF = to4(force)
        accel = torch.zeros((n, 3), dtype=torch.float64, device="cuda")
        image = torch.zeros((n, 3), dtype=torch.int32, device="cuda")
        ms, m = [], 2
        for ts in range(30):
            m = eng.step(dpos, vel, accel, image, dF, kT, dt, ts, lanczos_m=m)
            ms.append(m)
        stats = eng.neighbor_stats()
        if skin > 0 and dt > 1e-3:
            # the steps have suspended the list for steps only: evaluations repeated at fixed positions still keep and reuse it
            for _ in range(3):
                eng.mobility(dpos, dF)
            after = eng.neighbor_stats()
            assert (after[1] - stats[1], after[2] - stats[2]) == (1, 2), (stats, after)
        out.append((dpos.cpu().numpy(), image.cpu().numpy(), ms, stats))
    (pa, ia, ma, sa), (pb, ib, mb, sb) = out
    assert sb[1:] == (30, 0)
    if dt < 1e-3:
        assert 1 < sa[1] < 15 and sa[1] + sa[2] == 30, sa
    else:
        assert sa[1:] == (30, 0), sa
    assert ma == mb
    assert np.array_equal(ia, ib)
    assert np.abs(pa - pb).max() < 1e-9, np.abs(pa - pb).max()


def test_overflow_rows_on_the_kept_list(torch_cuda, oracle):
    """Rows of a dense cluster that do not fit the per-step pair list walk the kept neighbour list inside the mat-vecs; a
    cluster too dense for the neighbour list itself switches the reuse off (every call builds)."""
    import pse_amd
    rng = np.random.default_rng(17)
    n, L = 2000, 60.0
    box = (L, L, L, 0.0)
    nb = 36
    ball = rng.normal(size=(nb, 3)); ball *= (2.0 * rng.uniform(size=(nb, 1)) ** (1 / 3)) / np.linalg.norm(ball, axis=1, keepdims=True)
    pos = np.concatenate([ball + np.array([L / 2 - 1.0, 0.0, -L / 2 + 0.5]), rng.uniform(-L / 2, L / 2, size=(n - nb, 3))])
    pos = oracle.wrap(pos, np.zeros((n, 3), dtype=np.int64), box)[0]
    psi = rng.normal(size=(n, 3))
    eng = pse_amd.Engine(n, box, xi=0.5, error=1e-3)
    rcut = eng.info()["rcut"]
    eng.sqrt_mreal(to4(pos), to4(psi), tol=1e-3)
    moved = oracle.wrap(pos + 0.05 * rng.normal(size=(n, 3)).clip(-2, 2), np.zeros((n, 3), dtype=np.int64), box)[0]
    out, m = eng.sqrt_mreal(to4(moved), to4(psi), tol=1e-3)
    st = eng.neighbor_stats()
    matvec = lambda v: oracle.mobility_real(moved, np.ascontiguousarray(v), box, 0.5, rcut, rounded=True)
    up, mp = oracle.lanczos_sqrt(matvec, psi, 2, 1e-3)
    assert m == mp, (m, mp)
    assert rel(out.cpu().numpy()[:, :3], up) < 1e-9
    d = moved[:nb, None] - moved[None]
    d -= L * np.round(d / L)
    counts = (np.linalg.norm(d, axis=2) < rcut).sum(1) - 1
    assert counts.max() >= 33               # beyond the pair list's ~28 slots per row, inside the neighbour list's ~48
    assert st[1:] == (1, 1), st
    # a cluster too dense for the neighbour list as well: the build marks the overflow and the next call builds again
    big = np.concatenate([ball, ball * 0.9 + 0.05, ball * 0.8 - 0.05]) + np.array([L / 2 - 1.0, 0.0, -L / 2 + 0.5])
    pos2 = oracle.wrap(np.concatenate([big, pos[3 * nb:]]), np.zeros((n, 3), dtype=np.int64), box)[0]
    eng.sqrt_mreal(to4(pos2), to4(psi), tol=1e-3)
    out2, m2 = eng.sqrt_mreal(to4(pos2), to4(psi), tol=1e-3)
    assert eng.neighbor_stats()[1:] == (3, 1)
    mv2 = lambda v: oracle.mobility_real(pos2, np.ascontiguousarray(v), box, 0.5, rcut, rounded=True)
    up2, mp2 = oracle.lanczos_sqrt(mv2, psi, 2, 1e-3)
    assert m2 == mp2 and rel(out2.cpu().numpy()[:, :3], up2) < 1e-9


@pytest.mark.parametrize("edges,err", [(None, 1e-3), ((1.0, 1.35, 0.8), 1e-4)])
def test_random_call_sequence_matches_fresh_engines(torch_cuda, oracle, edges, err):
    """One engine driven through a random sequence of calls -- small and large moves, M.F, Brownian velocities, near-field
    square roots, repulsion (which walks the cells itself), real-space-only and wave-only evaluations, group subsets, tilt
    changes, another N -- against an engine created fresh for every call with the list switched off.  Catches state that
    one call leaves behind for the next (order of the last sort, kept list, suspended list, cell grid in use)."""
    import torch
    import pse_amd
    rng = np.random.default_rng(2024)
    n, seed = 1500, 3
    pos, force, box = make_suspension(n, phi=0.12)
    L = box[0]
    Ls = (L, L, L) if edges is None else tuple(L * e for e in edges)      # second case: three different edges, P = 8
    if edges is not None:
        pos = pos * np.array(edges)
        box = Ls + (0.0,)
    eng = pse_amd.Engine(n, box, xi=0.5, error=err, seed=seed)
    group = torch.tensor(np.sort(rng.choice(n, size=1100, replace=False)).astype(np.int32), device="cuda")
    cur, xy = pos.copy(), 0.0
    kinds = ["mf", "brown", "mf_near", "mf_wave", "sqrt", "repulse", "group", "mf", "brown", "mf"]
    for it in range(30):
        what = kinds[it % len(kinds)] if it < 24 else "mf"
        move = rng.choice(["none", "none", "small", "small", "small", "large", "tilt", "shrink"]) if it > 0 else "none"
        if it % 5 == 0:
            eng.set_neighbor_skin(0.4)   # re-arm a list that the large moves have suspended
        nn = n
        if move == "small":
            cur = cur + 0.02 * rng.normal(size=cur.shape).clip(-3, 3)
        elif move == "large":
            cur = cur + 0.5 * rng.normal(size=cur.shape)
        elif move == "tilt":
            xy = float(rng.uniform(-0.3, 0.3))
        elif move == "shrink":
            nn = 1200
        b = Ls + (xy,)
        if eng.box != b:
            eng.set_box(*b)
        cur = oracle.wrap(cur, np.zeros(cur.shape, dtype=np.int64), b)[0]
        p, f = cur[:nn], force[:nn]
        fresh = pse_amd.Engine(n, box, xi=0.5, error=err, seed=seed)
        fresh.set_neighbor_skin(0.0)
        if fresh.box != b:
            fresh.set_box(*b)
        dp, df = to4(p, 1.0), to4(f)
        tag = (it, what, move)
        if what in ("mf", "mf_near", "mf_wave"):
            parts = {"mf": 3, "mf_near": 1, "mf_wave": 2}[what]
            a = eng.mobility(dp, df, parts=parts).cpu().numpy()[:, :3]
            r = fresh.mobility(dp, df, parts=parts).cpu().numpy()[:, :3]
            assert rel(a, r) < 1e-11, (tag, rel(a, r))
        elif what == "brown":
            a, ma = eng.brownian_velocity(dp, df, 1.0, 1e-3, it)
            r, mr = fresh.brownian_velocity(dp, df, 1.0, 1e-3, it)
            assert ma == mr and rel(a.cpu().numpy()[:, :3], r.cpu().numpy()[:, :3]) < 1e-10, tag
        elif what == "sqrt":
            a, ma = eng.sqrt_mreal(dp, df, tol=1e-3)
            r, mr = fresh.sqrt_mreal(dp, df, tol=1e-3)
            assert ma == mr and rel(a.cpu().numpy()[:, :3], r.cpu().numpy()[:, :3]) < 1e-10, tag
        elif what == "repulse":
            fa, fr = to4(np.zeros((nn, 3))), to4(np.zeros((nn, 3)))
            eng.pair_repulsion(dp, fa, 10.0, sigma=2.0, accumulate=False)
            fresh.pair_repulsion(dp, fr, 10.0, sigma=2.0, accumulate=False)
            assert np.abs(fa.cpu().numpy() - fr.cpu().numpy()).max() < 1e-11, tag
        else:   # a group subset of the full arrays
            if nn != n:
                continue
            a = eng.mobility(dp, df, group=group).cpu().numpy()
            r = fresh.mobility(dp, df, group=group).cpu().numpy()
            g = group.cpu().numpy()
            assert rel(a[g, :3], r[g, :3]) < 1e-11, tag
    st = eng.neighbor_stats()
    assert st[2] >= 3, st      # some of the calls did run on the kept list


def test_set_box_growth_is_checked_against_both_cell_grids(torch_cuda):
    """pse_set_box validates the cell grid a later call may switch to, not only the one in use: with a neighbour skin the handle works
    on cells of width rcut + r_buff, but prepare() falls back to cells of width rcut (more of them) whenever the list is suspended or
    switched off.  A box that grew past the capacity sized at creation must be refused up front (before: a silent overrun of the
    cell counters in the sort); one that still fits must work in both modes."""
    import pse_amd
    n = 3000
    pos, force, _ = make_suspension(n, L=100.0)
    box = (100.0, 100.0, 100.0, 0.0)
    eng = pse_amd.Engine(n, box, xi=0.5, error=1e-3)                              # rcut 5.26: 19 cells per axis at creation (grid 96^3)
    with pytest.raises(pse_amd.PSEError):
        eng.set_box(130.0, 130.0, 130.0, 0.0)                                     # wide grid 22^3 would fit, the narrow 24^3 does not
    eng.set_box(115.0, 115.0, 115.0, 0.0)                                         # both fit
    big = pos * 1.15
    u1 = eng.mobility(to4(big), to4(force)).cpu().numpy()[:, :3]                  # builds the kept list on the wide grid
    eng.set_neighbor_skin(0.0)                                                    # from now on: the narrow grid, every call
    u2 = eng.mobility(to4(big), to4(force)).cpu().numpy()[:, :3]
    assert rel(u2, u1) < 1e-11
    assert np.isfinite(u2).all()
