"""GPU tests of the multi-GPU decomposition, run on ONE device: all slab ranks live in one process and the collectives
are device copies (in-process loopback team), so every pack/transpose/halo/row-range index of the production RCCL path
is exercised.  Reference = the single-GPU engine (itself pinned to the oracle in test_gpu_parity.py)."""
import numpy as np
import pytest

from conftest import make_suspension, to4

pytestmark = pytest.mark.gpu


def rel(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


@pytest.mark.parametrize("world,xy,mode", [(2, 0.0, "slab"), (4, 0.3, "slab"), (3, 0.0, "slab"), (2, 0.3, None), (4, 0.0, "replicated")])
def test_loopback_team_matches_single_gpu(world, xy, mode, monkeypatch):
    """mode: far field slab-decomposed, or kept whole on every rank with only the near field sharded (None = the
    library's choice: replicated for two ranks, slabs beyond)."""
    import pse_amd
    from pse_amd.sharded import LoopbackSimulation
    if mode:
        monkeypatch.setenv("PSE_WAVE_MODE", mode)
    else:
        monkeypatch.delenv("PSE_WAVE_MODE", raising=False)
    n = 3000
    pos, force, box = make_suspension(n, phi=0.1, xy=xy)
    grid = (48, 48, 40)                      # divisible by 2, 3, 4; slabs of >= 12 planes > P
    kw = dict(xi=0.5, error=1e-3, seed=77, grid=grid)
    ref = pse_amd.Engine(n, box, **kw)
    sim = LoopbackSimulation(n, box, world, **kw)
    sim.load(pos, force)
    for parts in (2, 1, 3):
        u_ref = ref.mobility(to4(pos), to4(force), parts=parts).cpu().numpy()[:, :3]
        vels = sim.mobility(parts=parts)
        for r in range(world):
            assert rel(vels[r].cpu().numpy()[:, :3], u_ref) < 1e-12, (parts, r)
    v_ref, m_ref = ref.brownian_velocity(to4(pos), to4(force), 1.0, 1e-3, 5)
    vels, m = sim.brownian_velocity(1.0, 1e-3, 5)
    assert m == m_ref
    for r in range(world):
        assert rel(vels[r].cpu().numpy()[:, :3], v_ref.cpu().numpy()[:, :3]) < 1e-11, r
    # two Lanczos iterations per exchange: the first batch covers max(m_in, 2) = 5 iterations in 3 exchanges, any further batch
    # max(2, done / 4) = 2 iterations in one (the one-step driver needs one exchange per iteration)
    i = sim.engines[0].info()
    assert i["lanczos_m"] == m and i["lanczos_exchanges"] <= 3 + max(0, (m - 5 + 1) // 2), i


@pytest.mark.parametrize("nx,world", [(360, 4), (512, 4), (360, 3)])
def test_slab_team_with_the_register_x_pass(nx, world, monkeypatch):
    """Nx = 360 and 512 take k_xfft_scale_cols (the data of a block in registers, one component in LDS at a time); on slab ranks it
    works on the transposed layout [Nx][ny_local][Nzh] with the rank's row offset.  Box stretched along x so that hx stays 24 / 256."""
    import pse_amd
    from pse_amd.sharded import LoopbackSimulation
    monkeypatch.setenv("PSE_WAVE_MODE", "slab")
    n = 2500
    pos, force, box = make_suspension(n, L=24.0, xy=0.1)
    Lx = nx * 24.0 / 256.0
    pos[:, 0] = (pos[:, 0] - 0.1 * pos[:, 1]) * (Lx / 24.0) + 0.1 * pos[:, 1]
    box = (Lx, 24.0, 24.0, 0.1)
    kw = dict(xi=0.5, error=1e-3, seed=12, grid=(nx, 48 if world == 3 else 32, 36), P=4)
    ref = pse_amd.Engine(n, box, **kw)
    sim = LoopbackSimulation(n, box, world, **kw)
    sim.load(pos, force)
    u_ref = ref.mobility(to4(pos), to4(force)).cpu().numpy()[:, :3]
    vels = sim.mobility()
    for r in range(world):
        assert rel(vels[r].cpu().numpy()[:, :3], u_ref) < 1e-12, r
    v_ref, m_ref = ref.brownian_velocity(to4(pos), to4(force), 1.0, 1e-3, 9)
    vels, m = sim.brownian_velocity(1.0, 1e-3, 9)
    assert m == m_ref
    for r in range(world):
        assert rel(vels[r].cpu().numpy()[:, :3], v_ref.cpu().numpy()[:, :3]) < 1e-11, r


@pytest.mark.parametrize("world,m_in", [(2, 2), (3, 3), (4, 6), (4, 11)])
def test_two_step_and_one_step_lanczos_agree(world, m_in, monkeypatch):
    """The team's two-iterations-per-exchange Lanczos (default) against the one-iteration driver (PSE_TEAM_SSTEP=0) and the single
    GPU: the same m, the same velocities -- from any starting count (even, odd, beyond convergence)."""
    import pse_amd
    from pse_amd.sharded import LoopbackSimulation
    n = 3000
    pos, force, box = make_suspension(n, phi=0.1, xy=0.15)
    kw = dict(xi=0.5, error=1e-3, seed=3, grid=(48, 48, 48))
    ref = pse_amd.Engine(n, box, **kw)
    v_ref, m_ref = ref.brownian_velocity(to4(pos), to4(force), 1.0, 1e-3, 9, lanczos_m=m_in)
    out = {}
    for sstep in ("1", "0"):
        monkeypatch.setenv("PSE_TEAM_SSTEP", sstep)
        sim = LoopbackSimulation(n, box, world, **kw)
        sim.load(pos, force)
        vels, m = sim.brownian_velocity(1.0, 1e-3, 9, lanczos_m=m_in)
        i = sim.engines[0].info()
        out[sstep] = (m, i["lanczos_exchanges"], i["lanczos_matvecs"])
        assert m == m_ref, (sstep, m, m_ref)
        for r in range(world):
            assert rel(vels[r].cpu().numpy()[:, :3], v_ref.cpu().numpy()[:, :3]) < 1e-10, (sstep, r)
    assert out["0"][1] == out["0"][2]                     # one exchange per iteration
    assert out["1"][1] < out["0"][1], out                 # fewer with two per exchange


def test_loopback_step_keeps_replicas_identical():
    import torch
    import pse_amd
    from pse_amd.sharded import LoopbackSimulation
    n = 2500
    pos, force, box = make_suspension(n, phi=0.1, xy=0.2)
    kw = dict(xi=0.5, error=1e-3, seed=5, grid=(48, 48, 48))
    sim = LoopbackSimulation(n, box, 4, **kw)
    sim.load(pos, force)
    ref = pse_amd.Engine(n, box, **kw)
    rp, rv, rF = to4(pos), to4(np.zeros((n, 3)), 1.0), to4(force)
    ra = torch.zeros((n, 3), dtype=torch.float64, device="cuda"); ri = torch.zeros((n, 3), dtype=torch.int32, device="cuda")
    m = mr = 2
    for ts in range(3):
        m = sim.step(1.0, 1e-2, ts, shear_rate=0.5, lanczos_m=m)
        mr = ref.step(rp, rv, ra, ri, rF, 1.0, 1e-2, ts, shear_rate=0.5, lanczos_m=mr)
    assert m == mr
    for s in sim.s:
        assert np.abs(s.pos.cpu().numpy() - rp.cpu().numpy()).max() < 1e-8     # (three steps of two engines: a rounded pair coefficient may round the other way, conftest.py)
        assert np.array_equal(s.image.cpu().numpy(), ri.cpu().numpy())


def test_rccl_team_of_one_rank():
    """The RCCL transport with a single rank: ncclCommInitRank + the collectives degenerate to self-copies."""
    import pse_amd
    from pse_amd.engine import Team
    n = 1000
    pos, force, box = make_suspension(n, phi=0.1)
    kw = dict(xi=0.5, error=1e-3, seed=1)
    ref = pse_amd.Engine(n, box, **kw)
    eng = pse_amd.Engine(n, box, **kw)
    team = Team([eng], unique_id=Team.unique_id())      # G = 1: no communication is issued, but RCCL is initialised
    vel = [to4(np.zeros((n, 3)))]
    team.mobility([to4(pos)], [to4(force)], vel)
    assert rel(vel[0].cpu().numpy()[:, :3], ref.mobility(to4(pos), to4(force)).cpu().numpy()[:, :3]) < 1e-13


@pytest.mark.parametrize("world,P", [(2, 5), (4, 4), (4, 7)])
def test_loopback_team_clustered_particles_and_odd_support(world, P):
    """Non-uniform suspension: most particles in a quarter of the box, so some cell slabs own almost nothing (zero-size
    ghost layers, unequal row blocks); odd and even support sizes change the two-sided gather halo."""
    import pse_amd
    from pse_amd.sharded import LoopbackSimulation
    n = 2500
    pos, force, box = make_suspension(n, L=48.0, xy=0.15)
    rng = np.random.default_rng(8)
    pos[: n * 4 // 5, 0] = rng.uniform(-24.0, -10.0, n * 4 // 5) + box[3] * pos[: n * 4 // 5, 1]   # crowd the low-x slab
    pos[-3:, 0] = 23.9                                                                            # and a few at the far face
    kw = dict(xi=0.5, error=1e-3, seed=3, grid=(64, 48, 40), P=P)
    ref = pse_amd.Engine(n, box, **kw)
    sim = LoopbackSimulation(n, box, world, **kw)
    sim.load(pos, force)
    u_ref = ref.mobility(to4(pos), to4(force)).cpu().numpy()[:, :3]
    for r, v in enumerate(sim.mobility()):
        assert rel(v.cpu().numpy()[:, :3], u_ref) < 1e-12, r
    v_ref, m_ref = ref.brownian_velocity(to4(pos), to4(force), 1.0, 1e-3, 9)
    vels, m = sim.brownian_velocity(1.0, 1e-3, 9)
    assert m == m_ref
    for r in range(world):
        assert rel(vels[r].cpu().numpy()[:, :3], v_ref.cpu().numpy()[:, :3]) < 1e-11, r


def test_loopback_team_of_eight():
    """The 8-rank layout of the scaling bench (slabs of 8 planes at 64^3: one far-field bin and 2 + 3 halo planes per rank,
    eight cell slabs) against the single-GPU engine: M.F, Brownian velocity, and three full steps."""
    import torch
    import pse_amd
    from pse_amd.sharded import LoopbackSimulation
    n, world = 20000, 8
    pos, force, box = make_suspension(n, phi=0.08)
    kw = dict(xi=0.3, error=1e-3, seed=5, grid=(64, 64, 64))
    ref = pse_amd.Engine(n, box, **kw)
    assert ref.info()["ncell_x"] >= 8
    sim = LoopbackSimulation(n, box, world, **kw)
    sim.load(pos, force)
    u_ref = ref.mobility(to4(pos), to4(force)).cpu().numpy()[:, :3]
    vels = sim.mobility()
    for r in range(world):
        assert rel(vels[r].cpu().numpy()[:, :3], u_ref) < 1e-12, r
    v_ref, m_ref = ref.brownian_velocity(to4(pos), to4(force), 1.0, 1e-3, 9)
    vels, m = sim.brownian_velocity(1.0, 1e-3, 9)
    assert m == m_ref
    for r in range(world):
        assert rel(vels[r].cpu().numpy()[:, :3], v_ref.cpu().numpy()[:, :3]) < 1e-11, r
    # steps: every rank integrates all particles with the exchanged velocities -> replicas stay identical
    p_ref, v0 = to4(pos, 0.0), to4(np.zeros((n, 3)), 1.0)
    acc = torch.zeros((n, 3), dtype=torch.float64, device="cuda"); img = torch.zeros((n, 3), dtype=torch.int32, device="cuda")
    m1 = m2 = 2
    for ts in range(3):
        m1 = ref.step(p_ref, v0, acc, img, to4(force), 1.0, 1e-3, ts, lanczos_m=m1)
        m2 = sim.step(1.0, 1e-3, ts, lanczos_m=m2)
    for r in range(world):
        assert float((sim.s[r].pos - p_ref).abs().max()) < 1e-8, r


@pytest.mark.parametrize("world,mode", [(2, "replicated"), (3, "slab"), (4, "slab")])
def test_loopback_team_on_a_particle_group(world, mode, monkeypatch):
    """group_members on a team (d_group_members / group_size of the reference, PSEv1/Stokes.cc:436-470): only the listed particles
    interact and move; the others' velocities, positions and images stay as they were -- on every rank, as on a single GPU."""
    import torch
    import pse_amd
    from pse_amd.engine import Team
    monkeypatch.setenv("PSE_WAVE_MODE", mode)
    n_total, n = 3000, 1900
    pos, force, box = make_suspension(n_total, phi=0.1, xy=0.2)
    members = np.sort(np.random.default_rng(11).choice(n_total, n, replace=False)).astype(np.int32)
    kw = dict(xi=0.5, error=1e-3, seed=21, grid=(48, 48, 36))
    ref = pse_amd.Engine(n_total, box, **kw)
    g = torch.tensor(members, dtype=torch.int32, device="cuda")

    def state():
        return dict(pos=to4(pos, 1.0), force=to4(force), vel=to4(-np.ones((n_total, 3)), 2.5),
                    accel=torch.zeros((n_total, 3), dtype=torch.float64, device="cuda"),
                    image=torch.zeros((n_total, 3), dtype=torch.int32, device="cuda"))

    r = state()
    engines = [pse_amd.Engine(n_total, box, n_slabs=world, slab_rank=k, **kw) for k in range(world)]
    team = Team(engines)
    S = [state() for _ in range(world)]
    col = lambda key: [s[key] for s in S]            # noqa: E731
    others = np.setdiff1d(np.arange(n_total), members)
    # M.F
    ref.mobility(r["pos"], r["force"], vel=r["vel"], group=g)
    team.mobility(col("pos"), col("force"), col("vel"), group=g)
    for k in range(world):
        v = S[k]["vel"].cpu().numpy()
        assert rel(v[members, :3], r["vel"].cpu().numpy()[members, :3]) < 1e-12, k
        assert np.all(v[others, :3] == -1.0) and np.all(v[:, 3] == 2.5), k
    # Brownian velocity
    _, m_ref = ref.brownian_velocity(r["pos"], r["force"], 1.0, 1e-3, 4, vel=r["vel"], group=g)
    _, m = team.brownian_velocity(col("pos"), col("force"), col("vel"), 1.0, 1e-3, 4, group=g)
    assert m == m_ref
    for k in range(world):
        v = S[k]["vel"].cpu().numpy()
        assert rel(v[members, :3], r["vel"].cpu().numpy()[members, :3]) < 1e-11, k
        assert np.all(v[others, :3] == -1.0), k
    # a sheared step
    m_ref = ref.step(r["pos"], r["vel"], r["accel"], r["image"], r["force"], 1.0, 2e-2, 5, shear_rate=0.4, group=g, lanczos_m=m_ref)
    m = team.step(col("pos"), col("vel"), col("accel"), col("image"), col("force"), 1.0, 2e-2, 5, shear_rate=0.4, group=g, lanczos_m=m)
    assert m == m_ref
    for k in range(world):
        p = S[k]["pos"].cpu().numpy()
        assert np.abs(p - r["pos"].cpu().numpy()).max() < 1e-8, k
        assert np.array_equal(p[others, :3], pos[others]), k
        assert np.array_equal(S[k]["image"].cpu().numpy(), r["image"].cpu().numpy()), k
        assert np.abs(S[k]["accel"].cpu().numpy() - r["accel"].cpu().numpy()).max() < 1e-15, k
    team.close()


@pytest.mark.parametrize("world", [2, 4])
def test_loopback_team_follows_tilt_and_particle_count(world):
    """Lees-Edwards on a team: every rank's handle takes pse_set_box with the new tilt between calls (what the host class does once
    per step, PSEv1/Stokes.cu:298), and the next call may bring fewer particles than the last."""
    import pse_amd
    from pse_amd.engine import Team
    n_max = 3000
    pos, force, box = make_suspension(n_max, phi=0.1)
    L = box[0]
    kw = dict(xi=0.5, error=1e-3, seed=2, grid=(48, 48, 40))
    ref = pse_amd.Engine(n_max, box, **kw)
    engines = [pse_amd.Engine(n_max, box, n_slabs=world, slab_rank=k, **kw) for k in range(world)]
    team = Team(engines)
    for xy, n in ((0.0, n_max), (0.27, n_max), (-0.41, 2200), (0.5, 2999), (0.1, 64)):
        b = (L, L, L, xy)
        p = pos[:n].copy()
        p[:, 0] += xy * p[:, 1]                                      # the same fractional coordinates in the tilted cell
        for e in engines + [ref]:
            e.set_box(*b)
        u_ref = ref.mobility(to4(p), to4(force[:n])).cpu().numpy()[:, :3]
        vels = [to4(np.zeros((n, 3))) for _ in range(world)]
        team.mobility([to4(p) for _ in range(world)], [to4(force[:n]) for _ in range(world)], vels)
        for k in range(world):
            assert rel(vels[k].cpu().numpy()[:, :3], u_ref) < 1e-11, (xy, n, k)
        v_ref, m_ref = ref.brownian_velocity(to4(p), to4(force[:n]), 1.0, 1e-3, 9)
        _, m = team.brownian_velocity([to4(p) for _ in range(world)], [to4(force[:n]) for _ in range(world)], vels, 1.0, 1e-3, 9)
        assert m == m_ref
        for k in range(world):
            assert rel(vels[k].cpu().numpy()[:, :3], v_ref.cpu().numpy()[:, :3]) < 1e-10, (xy, n, k)
    team.close()


def test_reteamed_engines_after_team_destroy():
    """a team is destroyed and a new one is built from the same engines.  The first team's deterministic call ran the
    far-field chain on member 0's shared side stream; the second team's Brownian call keeps it on the main stream -- the rocFFT
    execution infos of members >= 1 must follow (pse_team_destroy rebinds them), or their transforms run unordered on a stale stream."""
    import pse_amd
    from pse_amd.engine import Team
    n, world = 3000, 3
    pos, force, box = make_suspension(n, phi=0.1, xy=0.1)
    kw = dict(xi=0.5, error=1e-3, seed=9, grid=(48, 48, 40))
    ref = pse_amd.Engine(n, box, **kw)
    engines = [pse_amd.Engine(n, box, n_slabs=world, slab_rank=r, **kw) for r in range(world)]
    P, F = [to4(pos) for _ in range(world)], [to4(force) for _ in range(world)]
    V = [to4(np.zeros((n, 3))) for _ in range(world)]
    u_ref = ref.mobility(to4(pos), to4(force)).cpu().numpy()[:, :3]
    v_ref, m_ref = ref.brownian_velocity(to4(pos), to4(force), 1.0, 1e-3, 3)
    for cycle in range(3):
        team = Team(engines)
        team.mobility(P, F, V)                                   # kT = 0: the wave chain forks onto the shared side stream
        for r in range(world):
            assert rel(V[r].cpu().numpy()[:, :3], u_ref) < 1e-12, (cycle, r)
        team.close()
        team = Team(engines[::-1] if cycle == 1 else engines)    # members in another order: another member 0 lends its stream
        order = engines[::-1] if cycle == 1 else engines
        _, m = team.brownian_velocity(P, F, V, 1.0, 1e-3, 3)
        assert m == m_ref
        for r in range(world):
            assert rel(V[r].cpu().numpy()[:, :3], v_ref.cpu().numpy()[:, :3]) < 1e-11, (cycle, r)
        team.close()
        del order
